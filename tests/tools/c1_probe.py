"""Last decoder block (complex transposed conv, 64 -> 1 channels, F 129 -> 257) in exact fp32 at B utterances of 4 s: time and a
digest of the output (for comparing builds of csrc/ctconv_c1_f32.hip across processes), and the first encoder block (1 -> 32
channels, F 257 -> 129).  C1_PROBE_BW=1 first measures what a streaming kernel reaches on the box (torch.sum / torch.mul over 3 GiB).
    python tests/tools/c1_probe.py [B]      (GPU box)"""
import hashlib
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = 641
dev = "cuda"
g = torch.Generator().manual_seed(0)
slope = torch.tensor([0.25], device=dev)


def run(name, cin, cskip, cout, fin, tr, stats=False):
    x = ops.Planar.empty(cin, fin, B, T, T + 1, dev, zero=True)
    x.tensor5().normal_(generator=None)
    sk = None
    if cskip:
        sk = ops.Planar.empty(cskip, fin, B, T, T + 1, dev, zero=True)
        sk.tensor5().normal_()
    ct = cin + cskip
    shape = (ct, cout, 5, 2) if tr else (cout, ct, 5, 2)
    wr, wi = torch.randn(shape, generator=g).to(dev) * 0.05, torch.randn(shape, generator=g).to(dev) * 0.05
    br, bi = torch.randn(cout, generator=g).to(dev), torch.randn(cout, generator=g).to(dev)
    wf, bias = ops.pack_cconv(wr, wi, br, bi, None, transposed=tr)
    st = torch.zeros(cout, 5, dtype=torch.float64, device=dev) if stats else None
    for _ in range(2):
        y = ops.cconv2d(x, wf, bias, cout, transposed=tr, slope=slope, skip=sk, stats=st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        y = ops.cconv2d(x, wf, bias, cout, transposed=tr, slope=slope, skip=sk, stats=st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    gb = (x.buf.numel() + (sk.buf.numel() if sk is not None else 0) + y.buf.numel()) * 4 / 1e9
    dig = hashlib.sha256(y.tensor5().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]
    print(f"{name:5s} {ms:7.3f} ms  {gb / ms:6.2f} TB/s (in + out {gb:.2f} GB)  digest {dig}  stats={stats}")


torch.manual_seed(1)
if os.environ.get("C1_PROBE_BW"):
    big = torch.empty(768 * 1024 * 1024, dtype=torch.float32, device=dev).normal_()          # 3 GiB
    for name, fn, gb in (("sum (read 3.2 GB)", lambda: big.sum(), 3.22), ("mul (read + write 6.4 GB)", lambda: big * 2.0, 6.44)):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            r = fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"torch {name}: {gb / (e0.elapsed_time(e1) / 5):.2f} TB/s")
    del big
run("dec5", 32, 32, 1, 129, True)
run("dec5", 32, 32, 1, 129, True, stats=True)
run("dec5n", 64, 0, 1, 129, True)
run("enc0", 1, 0, 32, 257, False)
