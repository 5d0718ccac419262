"""usage (GPU box): python tests/tools/coop_stress.py [iterations] -- repeats the DCCRN-CL forward (fp32: cooperative H = 128
recurrence) and the NSVAE noisy encoder (bf16x3: persistent H = 768 recurrence) at full size on fixed inputs and checks that
every repetition is finite and bit-identical to the first: an intermittent hand-off fault would show as a difference or NaN."""
import importlib
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
from oracle import idccrn_oracle as O
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops
pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
NFFT, HOP, WIN = 512, 100, 400
np_ = O.net_params(True, 32)


def synth(m, seed):
    m.load_state_dict(O.synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed))
    return m.cuda()


g = torch.Generator().manual_seed(1)
x = (torch.randn(64, 64000, generator=g) * 0.1).cuda()
with torch.no_grad():
    for prec, build, B in (("fp32", lambda: synth(pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, [0, 1, 2, 3, 4, 5], "mask", False, None, None), 3), 64),
                           ("bf16x3", lambda: synth(pm.nsvae_pvae_dccrn_encoder_twophase(np_, True, "cuda", 128, NFFT, HOP, WIN, 2, 2), 4), 32)):
        ops.set_precision(prec)
        m = build()
        eps = tuple(torch.randn(B, 2, 641, 128, generator=g).cuda() for _ in range(4))
        run = (lambda: m(x[:B], train=False)[0]) if prec == "fp32" else (lambda: m(x[:B], train=False, eps=eps)[0])
        ref = run().clone()
        torch.cuda.synchronize()
        assert torch.isfinite(ref).all()
        t0 = time.perf_counter()
        bad = 0
        for i in range(N):
            out = run()
            if not torch.equal(out, ref):
                bad += 1
                print(f"{prec}: repetition {i} differs (finite={bool(torch.isfinite(out).all())})")
        torch.cuda.synchronize()
        print(f"{prec}: {N} repetitions, {bad} differ, {(time.perf_counter() - t0) / N * 1e3:.1f} ms each")
        ops.set_precision("fp32")
