"""usage (GPU box): python tests/tools/lstm_phase_probe.py [B] -- where one step of the persistent cooperative recurrence
(csrc/lstm_pers.hip) spends its time: per-phase core-clock cycles of every workgroup (min / mean / max), averaged over the T steps of the
second layer, for H = 384 and H = 768, next to the wall time per step of the un-instrumented kernel."""
import ctypes
import importlib
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops
L = importlib.import_module("i-dccrn-vae_amd._lib")
cp = importlib.import_module("i-dccrn-vae_amd.model.complex_progress")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, I = 641, 1280
NAMES = ("spin", "acquire", "barrier", "loads+mfma", "reduce+cell", "barrier", "release", "-")
ops.set_precision("bf16x3")
for H in (384, 768):
    g = torch.Generator().manual_seed(1)
    m = cp.ComplexLSTM(I, H, "cuda", num_layers=2)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.copy_(torch.randn(*p_.shape, generator=g) / H ** 0.5 * (0.3 if p_.dim() > 1 and p_.shape[1] == I else 1.0))
    m = m.cuda()
    x = ops.Planar.from_tensor5((torch.randn(B, I, 1, T, 2, generator=g) * 0.5).cuda())
    with torch.no_grad():
        for _ in range(2):
            m.forward_planar(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m.forward_planar(x)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 5
        prof = torch.zeros(256, 8, dtype=torch.int64, device="cuda")
        L.lib().idv_lstm_pers_set_profile.restype = None
        L.lib().idv_lstm_pers_set_profile(ctypes.c_void_p(prof.data_ptr()))
        m.forward_planar(x)
        torch.cuda.synchronize()
        L.lib().idv_lstm_pers_set_profile(ctypes.c_void_p(0))
    pc = prof.cpu().double()
    used = pc[:, :7].sum(1) > 0
    cyc = pc[used][:, :7] / T
    xcc = pc[used][:, 7].long()
    tot = float(cyc.sum(1).mean())
    print(f"H={H} B={B}: whole ComplexLSTM {wall * 1e3:.2f} ms; {int(used.sum())} workgroups; instrumented step {tot:.0f} cycles; "
          f"workgroups per XCC {torch.bincount(xcc, minlength=8).tolist()}")
    for i, n in enumerate(NAMES[:7]):
        c = cyc[:, i]
        if float(c.max()):
            print(f"   {n:12s} min {float(c.min()):7.0f}  mean {float(c.mean()):7.0f}  max {float(c.max()):7.0f} cycles")
    per = torch.stack([cyc[xcc == x].mean(0) if bool((xcc == x).any()) else torch.zeros(7, dtype=torch.float64) for x in range(8)])
    print("   per-XCC mean spin      ", [int(v) for v in per[:, 0].tolist()])
    print("   per-XCC mean loads+mfma", [int(v) for v in per[:, 3].tolist()])
