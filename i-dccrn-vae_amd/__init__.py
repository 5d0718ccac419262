"""MI355X-native I-DCCRN-VAE enhancement hot path (package directory ``i-dccrn-vae_amd``).

Import with ``importlib.import_module("i-dccrn-vae_amd")`` or through the alias module
``idccrn_vae_amd`` at the repository root.  Layout mirrors the reference:
``model/complex_progress.py``, ``model/pvae_module.py``, ``model/module.py``, ``model/net_config.py``,
``model/causal_netconfig.py``, ``model/sisnr_loss.py``, ``model/nsvae_loss.py``, ``model/pretrain_pvaes_loss.py``,
``utils/read_config.py``; ``csrc/`` holds the HIP kernels and the C ABI (``include/idccrn_hip.h``).
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401


def smoke():
    """One small DCCRN-CL enhancement + SI-SNR on cuda:0 through the HIP path, checked against the CPU oracle."""
    import importlib

    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs an MI355X: the HIP hot path has no CPU fallback")
    _lib.lib()                                   # fail loudly if the C-ABI library is missing
    pm = importlib.import_module(__name__ + ".model.pvae_module")
    nl = importlib.import_module(__name__ + ".model.nsvae_loss")
    synth = importlib.import_module(__name__ + ".utils.synth")
    from oracle import idccrn_oracle as O        # checker only

    torch.set_grad_enabled(False)
    n_fft, hop, win, skip = 512, 100, 400, [0, 1, 2, 3, 4, 5]
    np_ = O.net_params(True, 8)
    model = pm.DCCRN_(n_fft, hop, np_, True, "cuda:0", win, skip, "mask", False, None, None)
    sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 7)
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    g = torch.Generator().manual_seed(0)
    clean = torch.randn(2, 3200, generator=g) * 0.1
    noisy = clean + torch.randn(2, 3200, generator=g) * 0.1
    est, pred = model(noisy.to("cuda:0"), train=False)
    loss = nl.ete_train_se_loss([0.0, 0.0, 1.0]).final_ete_loss(pred, model.stft(clean.to("cuda:0")), clean.to("cuda:0"), est)
    torch.cuda.synchronize()
    ops.coop_check(sync=False)
    o_est, o_pred, _ = O.dccrn_forward(noisy, sd, np_, True, n_fft, hop, win, skip, "mask", False)
    o_loss = O.multiple_recon_loss(o_pred, O.stft(clean, n_fft, hop, win), clean, o_est, [0.0, 0.0, 1.0])
    rel = float((est.cpu().double() - o_est.double()).norm() / o_est.double().norm())
    dl = abs(float(loss[3]) - float(o_loss[3]))
    print(f"smoke: waveform rel err vs CPU oracle {rel:.2e}, SI-SNR loss {float(loss[3]):.4f} (oracle {float(o_loss[3]):.4f})")
    if not (rel < 1e-3 and dl < 1e-2 and est.shape == o_est.shape):
        raise RuntimeError(f"smoke parity failed: rel={rel} dloss={dl}")
