"""Adam with the update of all parameter tensors of a step in one kernel launch (csrc/adam.hip).

Same rule and state as the ``torch.optim.Adam(params, lr=0.0, eps=1e-15)`` the reference builds
(scene/gaussian_model.py:1034-1058): no weight decay, no amsgrad; state per parameter = ``step`` (host tensor),
``exp_avg``, ``exp_avg_sq`` — the keys gsvc_amd/densify.py (and the reference's optimizer surgery) manipulate.
"""
from __future__ import annotations

import torch

from . import _lib


class FusedAdam(torch.optim.Adam):
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries, keep = [], []
        beta_sets = set()
        dev = None
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("FusedAdam: weight_decay / amsgrad / maximize are not used by GSVC")
            b1, b2 = group["betas"]
            beta_sets.add((float(b1), float(b2), float(group["eps"])))
            lr = float(group["lr"])
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse:
                    raise _lib.GsvcError("FusedAdam updates dense float32 CUDA parameters (csrc/adam.hip)")
                dev = p.device
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = float(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not (p.is_contiguous() and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                    raise _lib.GsvcError("FusedAdam: parameters and moments must be contiguous")
                keep.append(g)
                entries.append(_lib.AdamTensorC(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                                                p.numel(), lr, 1.0 - b1 ** t, 1.0 - b2 ** t))
        if not entries:
            return loss
        if len(beta_sets) != 1:
            raise NotImplementedError("FusedAdam: one (betas, eps) for all groups")
        b1, b2, eps = next(iter(beta_sets))
        arr = (_lib.AdamTensorC * len(entries))(*entries)
        _lib.check(_lib.lib().gsvc_adam_step(len(entries), arr, b1, b2, eps, _lib.current_stream(dev)), "gsvc_adam_step")
        return loss
