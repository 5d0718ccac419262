"""Multi-tensor pieces of the train step on the HIP path (csrc/bucket.hip): the flat gradient bucket and Adam.

Reference: every trainer does ``loss.backward(); optimizer.step()`` with ``torch.optim.Adam(params, lr, weight_decay=0.001)``
(supervised_dccrn/train.py:109, 239-243; i_dccrn_vae/nsvae_dccrn/train_nsvae.py:200, 557-561;
i_dccrn_vae/nsvae_dccrn/train_second_phase_decoder.py:420-433; i_dccrn_vae/pretrained_vaes/train.py:296-301).

``Adam`` here IS a ``torch.optim.Adam`` (same constructor, ``param_groups``, ``state_dict()`` keys ``step`` / ``exp_avg`` /
``exp_avg_sq``, so the reference's ``*_optim_dict`` checkpoints load and save unchanged and ``ReduceLROnPlateau`` drives it)
whose ``step()`` is ONE kernel launch per parameter group over a pointer table instead of torch's per-tensor / foreach
kernels; the moments live in one flat buffer per group, the per-parameter state tensors are views of it.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch

from . import _lib as L
from ._lib import call, p, i, f, d, ll, stream_ptr


class TensorTable:
    """Bucket layout of a list of tensors (see include/idccrn_hip.h, "bucket layout"): offsets padded to 4 floats, and the
    device-side pointer table, re-uploaded only when a pointer changed (``.grad`` tensors are re-allocated by
    ``zero_grad(set_to_none=True)``)."""

    def __init__(self, numels: Sequence[int], device):
        self.device = device
        self.numels = [int(n) for n in numels]
        self.offsets: List[int] = []
        o = 0
        for n in self.numels:
            self.offsets.append(o)
            o += (n + 3) // 4 * 4
        self.total = max(o, 4)
        self.n = len(self.numels)
        self._ptrs: Optional[tuple] = None
        self._dev: Optional[torch.Tensor] = None

    def flat(self, zero: bool = True) -> torch.Tensor:
        return (torch.zeros if zero else torch.empty)(self.total, dtype=torch.float32, device=self.device)

    def views(self, flat: torch.Tensor, shapes) -> List[torch.Tensor]:
        return [flat[o:o + n].view(s) for o, n, s in zip(self.offsets, self.numels, shapes)]

    def table(self, tensors: Sequence[Optional[torch.Tensor]]) -> torch.Tensor:
        """Device table for these tensors (None -> absent).  Every tensor must be fp32, contiguous, on the device."""
        ptrs = []
        for t, n in zip(tensors, self.numels):
            if t is None:
                ptrs.append(0)
                continue
            if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous() or t.numel() != n:
                raise RuntimeError("TensorTable: tensors must be contiguous fp32 device tensors of the registered sizes")
            ptrs.append(t.data_ptr())
        ptrs = tuple(ptrs)
        if ptrs != self._ptrs:
            rows = [[q, o, n] for q, o, n in zip(ptrs, self.offsets, self.numels)]
            host = torch.tensor(rows, dtype=torch.int64).pin_memory()
            if self._dev is None:
                self._dev = torch.empty(self.n, 3, dtype=torch.int64, device=self.device)
            # stream-ordered upload: kernels already queued on this stream keep reading the old rows until they finish
            self._dev.copy_(host, non_blocking=True)
            self._host = host                       # pinned source stays alive until the next upload
            self._ptrs = ptrs
        return self._dev


def bucket_gather(tab: TensorTable, tensors, flat: torch.Tensor):
    call("idv_bucket_gather", p(tab.table(tensors)), i(tab.n), ll(tab.total), p(flat), stream_ptr())


def bucket_scatter(tab: TensorTable, tensors, flat: torch.Tensor, scale: float = 1.0):
    call("idv_bucket_scatter", p(tab.table(tensors)), i(tab.n), ll(tab.total), p(flat), f(scale), stream_ptr())


class Adam(torch.optim.Adam):
    """torch.optim.Adam with a one-launch ``step()`` (csrc/bucket.hip ``idv_bucket_adam``).

    ``step(grad_bucket=(table, flat, scale, present))``: take the gradients of group 0 straight from an all-reduced bucket with
    the same layout (parallel.GradAllReduce hands it over), so that the scatter back into ``.grad`` disappears; ``present[k]``
    False = parameter k had no gradient (skipped, as torch skips it)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if amsgrad or kw.get("maximize") or kw.get("capturable") or kw.get("differentiable"):
            raise NotImplementedError("i-dccrn-vae_amd.optim.Adam: plain Adam (+ L2 weight decay) only, as every reference trainer uses")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self._g = {}                                    # group index -> (params, TensorTable p, TensorTable g, m, v, step)

    # ---- flat state ---------------------------------------------------------------------------------------------------
    def _group_state(self, gi: int, group):
        params = [q for q in group["params"] if q.requires_grad]
        ent = self._g.get(gi)
        if ent is not None and ent["ids"] == tuple(id(q) for q in params):
            return ent
        if not params:
            return None
        dev = params[0].device
        for q in params:
            if not q.is_cuda or q.dtype != torch.float32 or not q.is_contiguous():
                raise RuntimeError("i-dccrn-vae_amd.optim.Adam runs on the MI355X: parameters must be contiguous fp32 device tensors")
        ptab = TensorTable([q.numel() for q in params], dev)
        gtab = TensorTable([q.numel() for q in params], dev)
        m, v = ptab.flat(), ptab.flat()
        shapes = [q.shape for q in params]
        mv, vv = ptab.views(m, shapes), ptab.views(v, shapes)
        t = 0
        for q, a, b in zip(params, mv, vv):
            st = self.state.get(q)
            if st and "exp_avg" in st:                  # state restored by load_state_dict (or an earlier layout): adopt it
                a.copy_(st["exp_avg"])
                b.copy_(st["exp_avg_sq"])
                t = max(t, int(float(st["step"])))
                st["exp_avg"], st["exp_avg_sq"] = a, b
        ent = {"ids": tuple(id(q) for q in params), "params": params, "ptab": ptab, "gtab": gtab, "m": m, "v": v, "mv": mv,
               "vv": vv, "t": t}
        self._g[gi] = ent
        return ent

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._g = {}                                    # the loaded moment tensors are adopted on the next step

    @torch.no_grad()
    def step(self, closure=None, grad_bucket=None):
        """One Adam step.  As in torch, a parameter without a gradient is skipped and keeps no state; unlike torch, the
        parameters of a group share ONE step count for the bias correction (they do in every trainer of the reference, where
        a parameter either gets a gradient on every step or on none)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            ent = self._group_state(gi, group)
            if ent is None:
                continue
            params = ent["params"]
            from_bucket = grad_bucket is not None and gi == 0
            gflat, gscale, gtable = None, 1.0, None
            if from_bucket:
                btab, gflat, gscale, has = grad_bucket
                if btab.offsets != ent["ptab"].offsets or btab.numels != ent["ptab"].numels:
                    raise RuntimeError("grad_bucket layout differs from the optimiser's parameter group")
                has = list(has)
                if not any(has):
                    continue
            else:
                grads = [q.grad for q in params]
                has = [g_ is not None for g_ in grads]
                if not any(has):
                    continue
                grads = [None if g_ is None else (g_ if g_.is_contiguous() else g_.contiguous()) for g_ in grads]
                gtable = ent["gtab"].table(grads)
            t = ent["t"] = ent["t"] + 1
            b1, b2 = group["betas"]
            # a parameter without a gradient is absent from the table (NULL row): the kernel leaves it and its moments alone
            ptable = ent["ptab"].table([q if h else None for q, h in zip(params, has)])
            call("idv_bucket_adam", p(ptable), p(gtable), p(gflat), p(ent["m"]), p(ent["v"]), i(ent["ptab"].n),
                 ll(ent["ptab"].total), f(group["lr"]), d(b1), d(b2), f(group["eps"]), f(group["weight_decay"]), f(1.0 - b1 ** t),
                 f(math.sqrt(1.0 - b2 ** t)), f(gscale), stream_ptr())
            stp = torch.tensor(float(t))
            for q, h, a, b in zip(params, has, ent["mv"], ent["vv"]):
                if h:
                    st = self.state[q]
                    st["step"] = stp
                    if st.get("exp_avg") is not a:
                        st["exp_avg"], st["exp_avg_sq"] = a, b
            _bump_versions([q for q, h in zip(params, has) if h])
        return loss


def _bump_versions(params):
    """Raw-pointer writes bypass torch's version counter, on which the packed-weight caches key (complex_progress._PackCache):
    an in-place no-op per parameter would cost a launch each, so the counters are bumped directly."""
    torch._C._autograd._unsafe_set_version_counter(tuple(params), tuple(q._version + 1 for q in params))
