#!/usr/bin/env python3
"""Build-container cross-check of bench.py's `cpu_baseline`: the SAME workload
(DCCRN-CL forward + final_ete_loss, and forward + loss + backward; B = 2, 4 s utterances, all cores) timed on
  (a) the REAL reference modules imported from /root/reference,
  (b) oracle/stock_cpu.py (what bench.py times on the GPU box, where the reference does not exist),
  (c) the loop-based oracle (round 1's baseline).
Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/cpu_baseline_crosscheck.py     (prints a markdown table)"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("IDCCRN_REFERENCE", "/root/reference"))

from oracle import idccrn_oracle as O  # noqa: E402
from oracle import stock_cpu  # noqa: E402
import model.pvae_module as R_pm  # noqa: E402  (reference)
import model.nsvae_loss as R_nl  # noqa: E402
import model.causal_netconfig as R_cnc  # noqa: E402

NFFT, HOP, WIN, LEN = 512, 100, 400, 64000
SKIP = [0, 1, 2, 3, 4, 5]
torch.set_num_threads(os.cpu_count())
np_ = R_cnc.get_net_params()
ref = R_pm.DCCRN_(NFFT, HOP, np_, True, "cpu", WIN, SKIP, "mask", False, None, None)
sd = O.synth_state_dict({k: tuple(v.shape) for k, v in ref.state_dict().items()}, 51)
ref.load_state_dict(sd)
stock = stock_cpu.StockDCCRN(O.net_params(True, 32), NFFT, HOP, WIN).load_reference_state(sd)
g = torch.Generator().manual_seed(123)
clean = torch.randn(2, LEN, generator=g) * 0.1
noisy = clean + torch.randn(2, LEN, generator=g) * 0.1
L_ref = R_nl.ete_train_se_loss([0.0, 0.0, 1.0])


def t_ref(train):
    with torch.set_grad_enabled(train):
        est, pred = ref(noisy, train=train)
        loss = L_ref.final_ete_loss(pred, ref.stft(clean), clean, est)[0]
        if train:
            ref.zero_grad(set_to_none=True)
            loss.backward()
    return float(loss)


def t_stock(train):
    with torch.set_grad_enabled(train):
        est, pred = stock(noisy, train=train)
        loss = O.multiple_recon_loss(pred, stock.stft(clean), clean, est, [0.0, 0.0, 1.0])[0]
        if train:
            stock.zero_grad(set_to_none=True)
            loss.backward()
    return float(loss)


def t_oracle(train):
    with torch.no_grad():
        est, pred, _ = O.dccrn_forward(noisy, sd, O.net_params(True, 32), True, NFFT, HOP, WIN, SKIP, "mask", False)
        return float(O.multiple_recon_loss(pred, O.stft(clean, NFFT, HOP, WIN), clean, est, [0.0, 0.0, 1.0])[0])


def bench(fn, train, n=6):
    fn(train)
    t0 = time.perf_counter()
    for _ in range(n):
        v = fn(train)
    return 2 * n / (time.perf_counter() - t0), v


print(f"| {os.cpu_count()} threads, B = 2 | utt/s | loss |")
print("|---|---|---|")
for name, fn, train in (("reference modules, forward + loss", t_ref, False), ("oracle/stock_cpu.py, forward + loss", t_stock, False),
                        ("loop-based oracle, forward + loss", t_oracle, False),
                        ("reference modules, forward + loss + backward", t_ref, True),
                        ("oracle/stock_cpu.py, forward + loss + backward", t_stock, True)):
    r, v = bench(fn, train)
    print(f"| {name} | {r:.3f} | {v:.5f} |")
