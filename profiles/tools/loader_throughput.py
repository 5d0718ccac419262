#!/usr/bin/env python3
"""utterances/s the data path delivers (i-dccrn-vae_amd/dataset/dataload.py: SpeechSequencesFull over 16-bit wav
files -> torch DataLoader(num_workers = k) -> (on a GPU box) DevicePrefetcher), against the rate the GPUs consume.

  python profiles/tools/loader_throughput.py [--files 48] [--seconds 30] [--batch 32] [--workers 0 1 2 4 8] [--out x.json]

Generates `files` noisy / clean / noise triples of `seconds` s of 16 kHz 16-bit PCM in a temporary directory (DNS3-shaped:
the reference's loaders read exactly such triples, dataset/dataload_nsvae.py:160-200), indexes them with the reference's
segment rule at sequence_len = 641 (4 s segments), and times one pass per worker count.  Without a GPU the H2D stage is
skipped (it overlaps the step anyway: 3 x 256 KB per utterance)."""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
from scipy.io import wavfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=48)
    ap.add_argument("--seconds", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--workers", type=int, nargs="*", default=[0, 1, 2, 4, 8])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dl = importlib.import_module("i-dccrn-vae_amd.dataset.dataload")
    hop, seq, fs = 100, 641, 16000
    res = {"files": a.files, "seconds_per_file": a.seconds, "batch": a.batch, "segment_samples": (seq - 1) * hop,
           "host_cpus": len(os.sched_getaffinity(0)), "runs": []}
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        dirs = {k: os.path.join(tmp, k) for k in ("noisy", "clean", "noise")}
        for d in dirs.values():
            os.makedirs(d)
        rng = np.random.default_rng(0)
        files = []
        for k in range(a.files):
            for kind, d in dirs.items():
                x = (rng.standard_normal(a.seconds * fs) * 0.1 * 32767).astype(np.int16)
                name = f"mix_snr5_fileid_{k}.wav" if kind == "noisy" else f"{kind}_fileid_{k}.wav"
                wavfile.write(os.path.join(d, name), fs, x)
            files.append(os.path.join(dirs["noisy"], f"mix_snr5_fileid_{k}.wav"))
        ds = dl.SpeechSequencesFull(files, dirs["clean"], False, None, None, name="thr", sr=fs, hop=hop, sequence_len=seq,
                                    first_use=True, dataset_to="train", noise_file_dir=dirs["noise"], cache_dir=tmp)
        res["segments"] = len(ds)
        gpu = torch.cuda.is_available()
        for w in a.workers:
            loader = torch.utils.data.DataLoader(ds, batch_size=a.batch, shuffle=True, num_workers=w, drop_last=True,
                                                 persistent_workers=False, prefetch_factor=4 if w else None)
            it = dl.DevicePrefetcher(loader, "cuda") if gpu else loader
            n, t0 = 0, None
            for bi, b in enumerate(it):
                if bi == 1:                      # the first batch pays the worker start-up
                    t0 = time.perf_counter()
                    n = 0
                n += b[0].shape[0]
            if gpu:
                torch.cuda.synchronize()
            el = time.perf_counter() - t0
            n -= a.batch                          # utterances counted since the clock started, minus the batch that started it
            r = {"num_workers": w, "utt_per_s": round(n / el, 1), "utterances": n, "to_gpu": gpu}
            print(r, flush=True)
            res["runs"].append(r)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
