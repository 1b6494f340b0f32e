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
    def step(self, closure=None, only=None, guards=None):
        """``only``: update just these parameters (a set of tensors) and drop their gradients, so that the step's later full
        ``step()`` passes over them; ``guards``: int32 device tensors (one element each) — if any is non-zero when the kernel
        runs, nothing is written (gsvc_adam_step_guarded).  The step counters of the ``only`` tensors advance either way; a
        caller that sees a guard set afterwards takes that back with ``rewind(only)``."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # host side kept short (this runs between the end of backward and the launch, with the GPU waiting): the step
        # counters advance with one foreach call, the per-tensor table is a reused ctypes array written in place
        keep, steps = [], []
        beta_key = None
        dev = None
        n = 0
        arr = self.__dict__.get("_table")
        if arr is None:
            arr = self._table = (_lib.AdamTensorC * 64)()      # per optimizer: the table is scratch for one launch
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("FusedAdam: weight_decay / amsgrad / maximize are not used by GSVC")
            b1, b2 = group["betas"]
            key = (float(b1), float(b2), float(group["eps"]))
            if beta_key is None:
                beta_key = key
            elif key != beta_key:
                raise NotImplementedError("FusedAdam: one (betas, eps) for all groups")
            lr = float(group["lr"])
            for p in group["params"]:
                g = p.grad
                if g is None or (only is not None and not any(p is q for q in only)):
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or g.is_sparse:
                    raise _lib.GsvcError("FusedAdam updates dense float32 CUDA parameters (csrc/adam.hip)")
                dev = p.device
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not g.is_contiguous():
                    g = g.contiguous()
                if not (p.is_contiguous() and m.is_contiguous() and v.is_contiguous()):
                    raise _lib.GsvcError("FusedAdam: parameters and moments must be contiguous")
                keep.append(g)
                steps.append(st["step"])
                if n == len(arr):
                    arr = self._table = self._grow(arr)
                e = arr[n]
                e.param, e.grad, e.exp_avg, e.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                e.n, e.lr = p.numel(), lr
                n += 1
        if n == 0:
            return loss
        torch._foreach_add_(steps, 1)
        b1, b2, eps = beta_key
        for i, t in enumerate(torch.stack(steps).tolist()):     # host tensors: no device synchronisation
            arr[i].bias_correction1, arr[i].bias_correction2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        if guards:
            import ctypes
            gp = (ctypes.c_void_p * len(guards))(*[g.data_ptr() for g in guards])
            _lib.check(_lib.lib().gsvc_adam_step_guarded(n, arr, b1, b2, eps, gp, len(guards), _lib.current_stream(dev)),
                       "gsvc_adam_step_guarded")
        else:
            _lib.check(_lib.lib().gsvc_adam_step(n, arr, b1, b2, eps, _lib.current_stream(dev)), "gsvc_adam_step")
        if only is not None:
            for p in only:
                p.grad = None
        return loss

    def rewind(self, params):
        """Take back the step count of parameters whose guarded update did not happen."""
        for p in params:
            st = self.state.get(p)
            if st and "step" in st:
                st["step"] -= 1

    @staticmethod
    def _grow(arr):
        new = (_lib.AdamTensorC * (2 * len(arr)))()
        for i in range(len(arr)):
            new[i] = arr[i]
        return new
