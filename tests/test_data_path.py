"""CPU: the data path (i-dccrn-vae_amd/dataset/dataload.py) indexes wav files exactly as the reference does
(dataset/dataload_supervised_dccrn.py:143-219): seq_length = (sequence_len - 1) * hop, n_seq = (1 + len // hop) // sequence_len,
ragged tails dropped, companions by file id, pickled index; and the synthetic DNS3-shaped stream."""
import importlib
import os
import pickle

import numpy as np
import pytest
import torch
from scipy.io import wavfile


@pytest.fixture()
def dl():
    return importlib.import_module("i-dccrn-vae_amd.dataset.dataload")


def _write(path, n, seed, fs=16000, dtype="int16"):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal(n) * 0.1).clip(-1, 1)
    if dtype == "int16":
        wavfile.write(path, fs, (x * 32767).astype(np.int16))
    else:
        wavfile.write(path, fs, x.astype(np.float32))
    return x.astype(np.float32)


def test_segment_indexing_matches_the_reference_formulas(dl, tmp_path):
    hop, seq = 100, 31                                  # seq_length = 3000 samples
    lens = {"a": 9100, "b": 2999, "c": 3100, "d": 48000, "e": 0}
    noisy_dir, clean_dir, noise_dir = tmp_path / "noisy", tmp_path / "clean", tmp_path / "noise"
    for d_ in (noisy_dir, clean_dir, noise_dir):
        d_.mkdir()
    files, sigs = [], {}
    for k, (name, n) in enumerate(lens.items()):
        f = str(noisy_dir / f"mix_snr5_fileid_{k}.wav")
        sigs[k] = (_write(f, n, k), _write(str(clean_dir / f"clean_fileid_{k}.wav"), n, 100 + k, dtype="float32"),
                   _write(str(noise_dir / f"noise_fileid_{k}.wav"), n, 200 + k))
        files.append(f)
    idx = dl.segment_index(files, hop, seq, 16000)
    want = []
    for f, n in zip(files, lens.values()):
        n_seq = (1 + n // hop) // seq
        want += [(f, i * 3000, (i + 1) * 3000) for i in range(n_seq)]
    assert idx == want
    assert [sum(1 for t in idx if t[0] == f) for f in files] == [2, 0, 1, 15, 0]     # ragged / short / empty files
    ds = dl.SpeechSequencesFull(files, str(clean_dir), False, None, None, name="unit", sr=16000, hop=hop, sequence_len=seq,
                                first_use=True, dataset_to="train", noise_file_dir=str(noise_dir), cache_dir=str(tmp_path))
    assert len(ds) == len(want)
    with open(tmp_path / "unit_train.pkl", "rb") as fh:
        assert pickle.load(fh) == want                    # the reference's pickle cache, same content
    x, c, n = ds[2]                                        # third segment = file 'c', samples [0, 3000)
    assert x.dtype == np.float32 and x.shape == (3000,)
    np.testing.assert_allclose(x, np.round(sigs[2][0][:3000] * 32767).astype(np.int16) / 32768.0, atol=1e-4)
    np.testing.assert_allclose(c, sigs[2][1][:3000], atol=0)
    assert n.shape == (3000,)
    x1, _, _ = ds[1]                                       # second segment of file 'a': [3000, 6000)
    np.testing.assert_allclose(x1, (sigs[0][0][3000:6000] * 32767).astype(np.int16) / 32768.0, atol=1e-4)
    ds2 = dl.SpeechSequencesFull(files, str(clean_dir), False, None, None, name="unit", sr=16000, hop=hop, sequence_len=seq,
                                 first_use=False, dataset_to="train", cache_dir=str(tmp_path))
    assert ds2.valid_seq_list == want and len(ds2[0]) == 2
    with pytest.raises(ValueError, match="sampling rate"):
        dl.segment_index(files, hop, seq, 8000)
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False)
    b = next(iter(loader))
    assert len(b) == 3 and b[0].shape == (4, 3000) and b[0].dtype == torch.float32


def test_synthetic_stream_is_seeded_and_dns3_shaped(dl):
    s = dl.SyntheticMixtures(batch=3, samples=64000, seed=123, length=2)
    a = list(s)
    b = list(dl.SyntheticMixtures(batch=3, samples=64000, seed=123, length=2))
    assert len(a) == 2 and all(t.shape == (3, 64000) and t.dtype == torch.float32 for t in a[0])
    assert torch.equal(a[1][0], b[1][0]) and not torch.equal(a[0][0], a[1][0])
    assert torch.allclose(a[0][0], a[0][1] + a[0][2])
    assert 0.09 < float(a[0][1].std()) < 0.11
