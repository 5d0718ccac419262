"""usage (GPU box): python tests/tools/err_probe_lstm.py -- relative error of every ComplexLSTM gradient vs the float64 oracle,
next to the float32 oracle's own error (what fp32 torch-CPU arithmetic gives), at the DCCRN-CL size."""
import importlib
import sys
import torch

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import idccrn_oracle as O
amd = importlib.import_module("i-dccrn-vae_amd")
ops = amd.ops
cp = importlib.import_module("i-dccrn-vae_amd.model.complex_progress")
H, I, T, B = 128, 1280, int(sys.argv[1]) if len(sys.argv) > 1 else 161, 2
g = torch.Generator().manual_seed(11)
m = cp.ComplexLSTM(I, H, "cuda", num_layers=2)
with torch.no_grad():
    for p_ in m.parameters():
        p_.copy_(torch.randn(*p_.shape, generator=g) / H ** 0.5 * (0.3 if p_.dim() > 1 and p_.shape[1] == I else 1.0))
m = m.cuda()
x = torch.randn(T, B, I, 2, generator=g) * 0.5
R = torch.randn(T, B, H, 2, generator=g)
xp = ops.Planar.from_tensor5(x.permute(1, 2, 0, 3).unsqueeze(2).cuda())
xp.buf.requires_grad_(True)
with torch.enable_grad():
    out = m.forward_planar(xp)
    y = out.channel_slice(0, H).permute(1, 0, 2, 3)
    (y * R.cuda()).sum().backward()


def run(dt):
    sd = {k: v.detach().cpu().to(dt).clone().requires_grad_(True) for k, v in m.state_dict().items()}
    xx = x.to(dt).clone().requires_grad_(True)
    w = O.complex_lstm(xx, sd, "", 2)
    (w * R.to(dt)).sum().backward()
    return w, xx, sd


rel = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / b.detach().cpu().double().norm())
w64, x64, sd64 = run(torch.float64)
w32, x32, sd32 = run(torch.float32)
print(f"T={T} forward: HIP {rel(y, w64):.1e}  f32 {rel(w32, w64):.1e}")
gx = ops.rewrap(xp.buf.grad, xp).tensor5()[:, :, 0].permute(2, 0, 1, 3)
print(f"dx: HIP {rel(gx, x64.grad):.1e}  f32 {rel(x32.grad, x64.grad):.1e}")
for k, p_ in m.named_parameters():
    print(f"{k}: HIP {rel(p_.grad, sd64[k].grad):.1e}  f32 {rel(sd32[k].grad, sd64[k].grad):.1e}")
