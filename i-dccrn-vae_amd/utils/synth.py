"""Deterministic synthetic parameters: the value of one state_dict entry is a function of (name, shape,
seed) only, so the reference modules (tests/golden/make_golden.py), the CPU oracle and this package's
modules get bit-identical weights without shipping a state_dict (there are no trained weights in the
reference).  Used by tests, smoke and bench."""
from __future__ import annotations

import math
from typing import Dict, Sequence

import torch

Tensor = torch.Tensor


def synth_tensor(name: str, shape: Sequence[int], seed: int) -> Tensor:
    """Deterministic value for one state_dict entry, a function of (name, shape, seed)
    only, so the reference modules (in make_golden.py) and this repo's modules get
    bit-identical parameters without shipping a state_dict."""
    import hashlib
    import numpy as np
    h = int.from_bytes(hashlib.sha256(f"{name}|{seed}".encode()).digest()[:8], "little")
    rng = np.random.default_rng(h)
    shape = tuple(int(s) for s in shape)
    leaf = name.split(".")[-1]
    if leaf.startswith("gamma_ri"):
        a = rng.standard_normal(shape) * 0.5
    elif leaf.startswith("gamma_"):
        a = 1.0 + 0.1 * rng.standard_normal(shape)
    elif leaf.startswith("beta_"):
        a = 0.1 * rng.standard_normal(shape)
    elif leaf in ("running_mean_real", "running_mean_imag"):
        a = 0.05 * rng.standard_normal(shape)
    elif leaf in ("Vrr", "Vii"):
        a = 0.5 + rng.random(shape)
    elif leaf == "Vri":
        a = 0.1 * rng.standard_normal(shape)
    elif "prelu" in name:
        a = np.full(shape, 0.25) + 0.05 * rng.standard_normal(shape)
    elif leaf.startswith("bias"):
        a = 0.05 * rng.standard_normal(shape)
    else:  # weights: U(+-sqrt(3/fan_in)) keeps unit variance through a layer
        if "tconv" in name:
            fan_in = shape[0] * shape[2] * shape[3] / 2.0   # stride-2 transposed conv
        elif len(shape) > 1:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
        else:
            fan_in = shape[0]
        bound = math.sqrt(3.0 / max(fan_in, 1)) * 0.7
        a = rng.uniform(-bound, bound, shape)
    return torch.from_numpy(a.astype("float32"))


def synth_state_dict(shapes: Dict[str, Sequence[int]], seed: int) -> Dict[str, Tensor]:
    return {k: synth_tensor(k, v, seed) for k, v in shapes.items() if v is not None}
