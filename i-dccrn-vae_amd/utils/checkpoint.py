"""Checkpoint compatibility with the reference's trainers: the same files, keys and run-folder
conventions, so weights trained with the reference drop into the HIP modules and vice versa.

Reference behaviour reproduced here:
  * on validation improvement the trainers write ``<model_name>_curr_best_epoch.pt`` (supervised,
    supervised_dccrn/train.py:295-303) or ``<model_name>_{clean,noise}_{encoder,decoder}_best_epoch.pt`` /
    ``_noisy_encoder_best_epoch.pt`` (train_nsvae.py:709-724; pretrained_vaes/train.py:369-414 uses
    ``_encoder_best_epoch.pt`` / ``_decoder_best_epoch.pt``): plain ``state_dict()`` files;
  * plus ``<model_name>_checkpoint.pt``: a dict with ``epoch``, ``best_val_loss``, ``cpt_patience``, ``loss_log`` and,
    per model ``X`` (``model``, ``clean_encoder``, ``noisy_encoder``, ...), ``X_state_dict``, ``X_optim_dict``,
    ``X_scheduler_dict`` (supervised_dccrn/train.py:305-324, train_nsvae.py:747-774); ``--reload`` restores from it
    (supervised_dccrn/train.py:169-200);
  * the architecture hyper-parameters are NOT in the files: they are encoded in the run-folder name
    ``<date>_<model_name>_key=value_key=value...`` and parsed back with string splits (supervised_dccrn/test.py:362-393,
    train_nsvae.py:94-120, train_second_phase_decoder.py:732-775);
  * ``ComplexBatchNormal.init_flag`` is a plain attribute, not part of the state_dict (complex_progress.py:114): a freshly
    constructed module that loads a checkpoint overwrites its running statistics on the first training batch.  That
    behaviour is kept (``keep_reference_init_flag=True``); pass False to continue training with the loaded statistics.

state_dict keys and shapes are identical on both sides (tests/test_host_cpu.py), so the files are interchangeable without
any key mapping.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Optional

import torch

_BOOL = {"true": True, "false": False}


def parse_run_folder(path: str) -> Dict[str, object]:
    """Hyper-parameters from a run-folder name, with the reference's defaults for absent keys.

    Handles the four name formats (supervised_dccrn/train.py:113, pretrained_vaes/train.py:148, train_nsvae.py:231,
    train_second_phase_decoder.py:181).  ``skipuse`` comes either as digits (``skipuse=012345``) or as a Python list
    (``skipuse=[0, 1, 2, 3, 4, 5]``); ``recon=real_imag`` arrives as ``recon=real`` + ``imag`` after the split on ``_``
    and is mapped back as the reference does; ``recontype`` is ``mask`` or anything else -> ``real_imag``."""
    name = os.path.basename(os.path.normpath(path))
    out: Dict[str, object] = {"causal": False, "skipuse": [0, 1, 2, 3, 4, 5], "datanorm": False, "resynthesis": False,
                              "spadd": False, "fcl": False, "recon_type": None, "folder": name}
    for tok in name.split("_"):
        if "=" not in tok:
            continue
        key, val = tok.split("=", 1)
        lo = val.lower()
        if key == "skipuse":
            digits = [c for c in val if c.isdigit()]
            out["skipuse"] = [int(c) for c in digits]
        elif key in ("causal", "datanorm", "resynthesis", "resyn", "spadd", "fcl", "decodeupdate", "loadde"):
            out["resynthesis" if key == "resyn" else key] = _BOOL.get(lo, val)
        elif key == "recontype":
            out["recon_type"] = "mask" if val == "mask" else "real_imag"
        elif key == "recon":
            out["recon_type"] = "real_imag" if val == "real" else val
        elif key in ("zdim", "numsamples", "latentnum"):
            out[{"numsamples": "num_samples", "latentnum": "latent_num"}.get(key, key)] = int(val)
        elif key in ("alpha", "wresi", "wkl", "wdismiu", "klw", "miw"):
            try:
                out[key] = float(val)
            except ValueError:
                out[key] = val
        else:
            out[key] = val          # skipc ('True' / 'False' strings, as the reference keeps them), nsvae, match, reconw, prior ...
    return out


def set_bn_init_flag(module: torch.nn.Module, flag: bool):
    for m in module.modules():
        if hasattr(m, "init_flag") and hasattr(m, "running_mean_real"):
            m.init_flag = flag


def load_state_file(module: torch.nn.Module, path: str, map_location="cpu", keep_reference_init_flag: bool = True):
    """``*_best_epoch.pt``: a bare state_dict (e.g. supervised_dccrn/test.py:412-414)."""
    sd = torch.load(path, map_location=map_location, weights_only=False)
    module.load_state_dict(sd, strict=True)
    if not keep_reference_init_flag:
        set_bn_init_flag(module, False)
    return module


def save_best_epoch(module: torch.nn.Module, save_dir: str, model_name: str, role: Optional[str] = None) -> str:
    """supervised: ``<model_name>_curr_best_epoch.pt``; VAE roles: ``<model_name>_<role>_best_epoch.pt``."""
    fn = f"{model_name}_curr_best_epoch.pt" if role is None else f"{model_name}_{role}_best_epoch.pt"
    path = os.path.join(save_dir, fn)
    torch.save(module.state_dict(), path)
    return path


def save_checkpoint(save_dir: str, model_name: str, epoch: int, best_val_loss: float, cpt_patience: int,
                    models: Dict[str, torch.nn.Module], optimizers: Optional[Dict[str, torch.optim.Optimizer]] = None,
                    schedulers: Optional[Dict[str, object]] = None, loss_log: Optional[dict] = None) -> str:
    """``<model_name>_checkpoint.pt`` with the reference's keys; ``models`` maps the role (``model``, ``clean_encoder``,
    ``noisy_encoder`` ...) to the module."""
    d = {"epoch": epoch, "best_val_loss": best_val_loss, "cpt_patience": cpt_patience, "loss_log": loss_log or {}}
    for role, m in models.items():
        d[f"{role}_state_dict"] = m.state_dict()
    for role, o in (optimizers or {}).items():
        d[f"{role}_optim_dict"] = o.state_dict()
    for role, s in (schedulers or {}).items():
        d[f"{role}_scheduler_dict"] = s.state_dict()
    path = os.path.join(save_dir, f"{model_name}_checkpoint.pt")
    torch.save(d, path)
    return path


def load_checkpoint(path: str, models: Dict[str, torch.nn.Module], optimizers: Optional[Dict[str, torch.optim.Optimizer]] = None,
                    schedulers: Optional[Dict[str, object]] = None, map_location="cpu",
                    keep_reference_init_flag: bool = True) -> dict:
    """Restore from ``*_checkpoint.pt`` as ``--reload`` does (supervised_dccrn/train.py:169-200); returns the bookkeeping
    entries (epoch, best_val_loss, cpt_patience, loss_log).  Optimizer / scheduler entries the file does not hold (frozen
    models, train_nsvae.py:757-768) are skipped."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    for role, m in models.items():
        m.load_state_dict(ck[f"{role}_state_dict"], strict=True)
        if not keep_reference_init_flag:
            set_bn_init_flag(m, False)
    for role, o in (optimizers or {}).items():
        if f"{role}_optim_dict" in ck:
            o.load_state_dict(ck[f"{role}_optim_dict"])
    for role, s in (schedulers or {}).items():
        if f"{role}_scheduler_dict" in ck:
            s.load_state_dict(ck[f"{role}_scheduler_dict"])
    return {k: ck[k] for k in ("epoch", "best_val_loss", "cpt_patience", "loss_log") if k in ck}


def dccrn_from_run_folder(folder: str, n_fft: int, hop: int, win: int, device, data_mean=None, data_std=None,
                          net_params: Optional[dict] = None):
    """Build the supervised ``DCCRN_`` a run folder describes and load ``*_curr_best_epoch.pt`` from it
    (supervised_dccrn/test.py:362-414)."""
    from ..model import causal_netconfig, net_config
    from ..model.pvae_module import DCCRN_
    hp = parse_run_folder(folder)
    np_ = net_params or (causal_netconfig.get_net_params() if hp["causal"] else net_config.get_net_params())
    recon = hp["recon_type"] or "mask"
    if not hp["datanorm"]:
        data_mean = data_std = None
    model = DCCRN_(n_fft, hop, np_, hp["causal"], device, win, hp["skipuse"], recon, hp["resynthesis"], data_mean, data_std)
    files = [f for f in os.listdir(folder) if f.endswith("_curr_best_epoch.pt")]
    if len(files) != 1:
        raise FileNotFoundError(f"expected one *_curr_best_epoch.pt in {folder}, found {files}")
    load_state_file(model, os.path.join(folder, files[0]))
    return model.to(device), hp
