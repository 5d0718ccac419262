"""rocprofv3 --pmc FETCH_SIZE calibration on known byte counts (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your
own access pattern"): a 1 GiB torch copy (16 B / lane loads), this library's planar -> image conversion (4 B / lane loads, 1 GiB
read) and a 2 GiB-input fp32 conv launch (cgemm_kernel: float4 staging loads + 4 B / lane weight loads)."""
import importlib
import sys

import torch

sys.path.insert(0, "/root/repo")
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops
x = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device="cuda").normal_()       # 1 GiB
for _ in range(3):
    y = x.clone()
    torch.cuda.synchronize()
# planar [2][C][F][Jp]: 64 ch x 64 rows x 32768 cols x 2 x 4 B = 1 GiB
p = ops.Planar.empty(64, 64, 32, 1023, 1024, "cuda")
p.buf.normal_()
for _ in range(3):
    img = ops.to_image(p)
    torch.cuda.synchronize()
# a transposed conv with a 2 GiB input: 512 complex channels x 8 rows x 65536 columns
g = torch.Generator().manual_seed(0)
xp = ops.Planar.empty(512, 8, 64, 1023, 1024, "cuda")
xp.buf.normal_()
wr = (torch.randn(512, 128, 5, 2, generator=g) * 0.05).cuda()
wi = (torch.randn(512, 128, 5, 2, generator=g) * 0.05).cuda()
zb = torch.zeros(128, device="cuda")
wf, bias = ops.pack_cconv(wr, wi, zb, zb, None, transposed=True)
for _ in range(3):
    out = ops.cconv2d(xp, wf, bias, 128, transposed=True, causal=True)
    torch.cuda.synchronize()
print("input bytes of the conv launch:", 2 * 512 * 8 * xp.Jp * 4, "output bytes:", 2 * 128 * 15 * out.Jp * 4)
