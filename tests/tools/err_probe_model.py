"""usage (GPU box): python tests/tools/err_probe_model.py -- per-parameter gradient error of the full-width DCCRN-CL train step
(HIP vs float64 oracle, float32 oracle vs float64 oracle)."""
import importlib
import sys
import torch

sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_backward as T
from oracle import idccrn_oracle as O
pm = importlib.import_module("i-dccrn-vae_amd.model.pvae_module")
nl = importlib.import_module("i-dccrn-vae_amd.model.nsvae_loss")
NFFT, HOP, WIN, SKIP = 512, 100, 400, [0, 1, 2, 3, 4, 5]
np_ = O.net_params(True, 32)
m = T.load_synth(pm.DCCRN_(NFFT, HOP, np_, True, "cuda", WIN, SKIP, "mask", False, None, None), 77)
g = torch.Generator().manual_seed(21)
x = torch.randn(2, 16000, generator=g) * 0.1
c = x + torch.randn(2, 16000, generator=g) * 0.05
w = [0.2, 0.1, 1.0]
with torch.enable_grad():
    est, pred = m(x.cuda(), train=True)
    nl.ete_train_se_loss(w).final_ete_loss(pred, m.stft(c.cuda()), c.cuda(), est)[0].backward()


def run(dt):
    sd = {k: v.detach().cpu().to(dt).clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and k[-3:] not in ("Vrr", "Vri", "Vii"))
          for k, v in m.state_dict().items()}
    e, p, _ = O.dccrn_forward(x.to(dt), sd, np_, True, NFFT, HOP, WIN, SKIP, "mask", True, O.BNState())
    O.multiple_recon_loss(p, O.stft(c.to(dt), NFFT, HOP, WIN), c.to(dt), e, w)[0].backward()
    return sd


rel = lambda a, b: float((a.detach().cpu().double() - b.detach().cpu().double()).norm() / (b.detach().cpu().double().norm() + 1e-30))
s64, s32 = run(torch.float64), run(torch.float32)
for k, p_ in m.named_parameters():
    if s64[k].grad is None or k.endswith("conv_re.bias") or k.endswith("conv_im.bias"):
        continue
    print(f"{k:55s} HIP {rel(p_.grad, s64[k].grad):.1e}  f32 {rel(s32[k].grad, s64[k].grad):.1e}  HIP-f32 {rel(p_.grad, s32[k].grad):.1e}")
