"""Data path feeding the hot path: fixed-length 16 kHz segments from wav files, indexed exactly as the
reference indexes them, delivered to the GPU through pinned memory on a copy stream; plus the synthetic DNS3-shaped
stream bench.py / config 5 uses.

Reference behaviour kept (dataset/dataload_supervised_dccrn.py:98-219, dataset/dataload_nsvae.py:88-200):
  * segment length in samples ``seq_length = (sequence_len - 1) * hop`` (:163), segments per file
    ``n_seq = (1 + len // hop) // sequence_len`` (:165), segment i = ``[i * seq_length, (i + 1) * seq_length)``; a file
    shorter than one segment yields none (ragged tail dropped);
  * the index ``[(wavfile, start, end), ...]`` is pickled to ``<dataset_name>_train.pkl`` / ``_val.pkl`` in the working
    directory on first use and re-read afterwards (:130-143, :177-183); optional shuffle of the index;
  * clean / noise companions are found from the file id after the last ``_``: ``clean_fileid_<id>.wav``,
    ``noise_fileid_<id>.wav`` (:201-204, dataload_nsvae.py:178-181); a sampling-rate mismatch raises ValueError (:152).
librosa / soundfile are not needed: 16-bit / 32-bit / float PCM wav files are decoded with ``scipy.io.wavfile`` to float32
in [-1, 1), which is what ``librosa.load(sr=None)`` returns for them.  The hot path itself never touches the disk.
"""
from __future__ import annotations

import os
import pickle
import random
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils import data


def read_wav(path: str, start: int = 0, end: Optional[int] = None) -> Tuple[np.ndarray, int]:
    """-> (float32 mono samples[start:end], sampling rate).  Memory-mapped: only the requested slice is decoded."""
    from scipy.io import wavfile
    fs, x = wavfile.read(path, mmap=True)
    x = x[start:end]
    if x.ndim > 1:
        x = x.mean(axis=1)
    if x.dtype == np.int16:
        y = x.astype(np.float32) / 32768.0
    elif x.dtype == np.int32:
        y = x.astype(np.float32) / 2147483648.0
    elif x.dtype == np.uint8:
        y = (x.astype(np.float32) - 128.0) / 128.0
    else:
        y = np.asarray(x, dtype=np.float32)
    return y, int(fs)


def wav_length(path: str) -> Tuple[int, int]:
    from scipy.io import wavfile
    fs, x = wavfile.read(path, mmap=True)
    return int(x.shape[0]), int(fs)


def segment_index(files: Iterable[str], hop: int, sequence_len: int, fs: int) -> List[Tuple[str, int, int]]:
    """The reference's ``compute_len`` (dataload_supervised_dccrn.py:143-175)."""
    out = []
    seq_length = (sequence_len - 1) * hop
    for wav in files:
        n, fs_x = wav_length(wav)
        if fs != fs_x:
            raise ValueError('Unexpected sampling rate')
        n_seq = (1 + int(n / hop)) // sequence_len
        for i in range(n_seq):
            out.append((wav, i * seq_length, (i + 1) * seq_length))
    return out


def companion(wavfile: str, directory: str, kind: str) -> str:
    """``<directory>/<kind>_fileid_<id>.wav`` for a noisy file ``..._<id>.wav`` (dataload_supervised_dccrn.py:201-204)."""
    fileid = wavfile.split('.')[0].split('_')[-1]
    return directory + '/' + kind + '_fileid_' + fileid + '.wav'


class SpeechSequencesFull(data.Dataset):
    """Same constructor keywords and item layout as the reference class: items are ``(noisy, clean)`` float32 segments,
    or ``(noisy, clean, noise)`` when ``noise_file_dir`` is given (the NSVAE loader)."""

    def __init__(self, noisy_file_list: Sequence[str], clean_file_dir: str, shuffle: bool = False, sample_mean=None,
                 sample_std=None, name: str = 'WSJ0', sr: int = 16000, hop: int = 100, sequence_len: int = 100,
                 first_use: bool = True, dataset_to: str = 'train', noise_file_dir: Optional[str] = None,
                 cache_dir: str = "."):
        super().__init__()
        self.noisy_file_list = list(noisy_file_list)
        self.clean_file_dir, self.noise_file_dir = clean_file_dir, noise_file_dir
        self.name, self.shuffle, self.fs, self.sequence_len, self.hop = name, shuffle, sr, sequence_len, hop
        self.sample_mean = np.loadtxt(sample_mean) if isinstance(sample_mean, str) else sample_mean
        self.sample_std = np.loadtxt(sample_std) if isinstance(sample_std, str) else sample_std
        if dataset_to not in ("train", "val"):
            raise ValueError("dataset_to must be 'train' or 'val'")
        cache = os.path.join(cache_dir, f"{self.name}_{dataset_to}.pkl")
        if first_use:
            self.valid_seq_list = segment_index(self.noisy_file_list, hop, sequence_len, sr)
            if shuffle:
                random.shuffle(self.valid_seq_list)
            with open(cache, "wb") as f:
                pickle.dump(self.valid_seq_list, f)
        else:
            with open(cache, "rb") as f:
                self.valid_seq_list = pickle.load(f)

    def __len__(self):
        return len(self.valid_seq_list)

    def __getitem__(self, index):
        wavfile, s0, s1 = self.valid_seq_list[index]
        x, _ = read_wav(wavfile, s0, s1)
        c, _ = read_wav(companion(wavfile, self.clean_file_dir, 'clean'), s0, s1)
        if self.noise_file_dir is None:
            return x, c
        n, _ = read_wav(companion(wavfile, self.noise_file_dir, 'noise'), s0, s1)
        return x, c, n


class DevicePrefetcher:
    """Iterate a DataLoader (or any iterable of tuples of CPU tensors / arrays) with the NEXT batch already on its way to
    the GPU: pinned staging buffers + a dedicated copy stream, so the H2D copy of 3 x B x 256 KB overlaps the step
    (about 0.3 ms per 64 utterances at PCIe Gen5 against a >= 30 ms step)."""

    def __init__(self, loader, device, dtype=torch.float32):
        self.loader, self.device, self.dtype = loader, torch.device(device), dtype
        self.stream = torch.cuda.Stream(device=self.device)

    def _upload(self, batch):
        items = batch if isinstance(batch, (tuple, list)) else (batch,)
        out = []
        with torch.cuda.stream(self.stream):
            for t in items:
                t = torch.as_tensor(t)
                if not t.is_pinned():
                    t = t.pin_memory()
                out.append(t.to(self.device, dtype=self.dtype, non_blocking=True))
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return tuple(out), ev

    def __iter__(self) -> Iterator[tuple]:
        it = iter(self.loader)
        try:
            nxt = self._upload(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._upload(next(it))
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for t in cur:
                t.record_stream(torch.cuda.current_stream(self.device))
            yield cur


class SyntheticMixtures:
    """Endless DNS3-shaped synthetic stream (BASELINE config 5 'streaming 16 kHz DNS3-shaped synthetic
    mixtures'): batches ``(noisy, clean, noise)`` of ``0.1 * N(0, 1)`` 16 kHz mono float32, seeds incrementing per batch,
    generated on the host like decoded audio would be (use DevicePrefetcher to move them)."""

    def __init__(self, batch: int, samples: int = 64000, seed: int = 123, length: Optional[int] = None):
        self.batch, self.samples, self.seed, self.length = batch, samples, seed, length

    def __iter__(self):
        k = 0
        while self.length is None or k < self.length:
            g = torch.Generator().manual_seed(self.seed + 2 * k)
            clean = torch.randn(self.batch, self.samples, generator=g) * 0.1
            g2 = torch.Generator().manual_seed(self.seed + 2 * k + 1)
            noise = torch.randn(self.batch, self.samples, generator=g2) * 0.1
            yield clean + noise, clean, noise
            k += 1


def build_dataloader(cfg, first_use_dataset: bool = True, nsvae: bool = False, cache_dir: str = "."):
    """``build_dataloader`` of the reference (dataload_supervised_dccrn.py:15-95) on a parsed .ini (utils.read_config.myconf)."""
    def listing(path, suffix):
        if path.endswith('.txt'):
            with open(path) as f:
                return [ln.rstrip() for ln in f if ln.rstrip().endswith('.wav')]
        found = []
        for root, _, files in os.walk(path):
            found += [os.path.join(root, fn) for fn in files if fn.lower().endswith('.' + suffix.lstrip('.').lower())]
        return sorted(found)
    u = lambda k: cfg.get('User', k)
    name = cfg.get('DataFrame', 'dataset_name')
    bs, shuffle = cfg.getint('DataFrame', 'batch_size'), cfg.getboolean('DataFrame', 'shuffle')
    workers, suffix = cfg.getint('DataFrame', 'num_workers'), cfg.get('DataFrame', 'suffix')
    seq = cfg.getint('DataFrame', 'sequence_len')
    hop, fs = cfg.getint('STFT', 'hopfrac'), cfg.getint('STFT', 'fs')
    sets = []
    for split in ("train", "val"):
        kw = dict(noisy_file_list=listing(u(f'noisy_{split}_data_dir'), suffix), clean_file_dir=u(f'clean_{split}_data_dir'),
                  shuffle=shuffle, name=name, sr=fs, hop=hop, sequence_len=seq, first_use=first_use_dataset,
                  dataset_to=split, cache_dir=cache_dir)
        if nsvae:
            kw["noise_file_dir"] = u(f'noise_{split}_data_dir')
        sets.append(SpeechSequencesFull(**kw))
    mk = lambda ds: data.DataLoader(ds, batch_size=bs, shuffle=shuffle, num_workers=workers)
    return mk(sets[0]), mk(sets[1]), len(sets[0]), len(sets[1])
