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
        # host side kept short (this runs between the end of backward and the launch, with the GPU waiting — and at small shapes
        # the step is host-bound outright): what does not change from step to step — the parameters' and moments' addresses,
        # sizes, the step tensors — sits in a cached row per parameter, re-derived only when the parameter objects or their state
        # change (densification replaces both); per step a row costs the gradient's address, the group's rate and two powers
        rows = self._rows()
        only_ids = None if only is None else {id(q) for q in only}
        arr = self.__dict__.get("_table")
        if arr is None or len(arr) < len(rows):
            arr = self._table = (_lib.AdamTensorC * max(64, 2 * len(rows)))()      # per optimizer: scratch for one launch
        b1, b2, eps = self._beta_key
        stepped, keep = [], []
        n, dev = 0, None
        for i, row in enumerate(rows):
            p = row[0]
            g = p.grad
            if g is None or (only_ids is not None and id(p) not in only_ids):
                continue
            if g.is_sparse:
                raise _lib.GsvcError("FusedAdam updates dense float32 CUDA parameters (csrc/adam.hip)")
            if not g.is_contiguous():
                g = g.contiguous()
                keep.append(g)
            if row[8] is None:                            # first gradient of this parameter: its state appears now, as in torch.optim.Adam
                self._init_state(row)
            row[6] += 1                                   # the host's copy of state["step"] (a float tensor the optimizer API owns)
            t = row[6]
            e = arr[n]
            e.param, e.exp_avg, e.exp_avg_sq, e.n = row[1], row[2], row[3], row[4]
            e.grad, e.lr = g.data_ptr(), float(row[5]["lr"])
            e.bias_correction1, e.bias_correction2 = 1.0 - b1 ** t, 1.0 - b2 ** t
            stepped.append(i)
            dev = p.device
            n += 1
        if n == 0:
            return loss
        # state["step"] of every parameter is a 0-d view of ONE host tensor: one add for all of them (52 scalar tensors through
        # _foreach_add_ cost 0.12 ms per step)
        if n == len(rows):
            self._step_base.add_(1)
        else:
            key = tuple(stepped)
            idx = self._step_idx.get(key)
            if idx is None:
                idx = self._step_idx[key] = torch.tensor(stepped, dtype=torch.int64)
            self._step_base.index_add_(0, idx, torch.ones(len(stepped)))
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

    def _rows(self):
        """One row per parameter: [param, its address, exp_avg address, exp_avg_sq address, numel, group, step count (int),
        state["step"] tensor].  Valid while the parameter objects and their moment tensors are the ones it was built from."""
        sig = tuple(id(p) for g in self.param_groups for p in g["params"])
        cache = self.__dict__.get("_row_cache")
        if cache is not None and cache[0] == sig:
            rows = cache[1]
            # the moments are replaced together with the parameter (densify.py), or by load_state_dict (same parameter objects);
            # a storage swap under the same Parameter object (module.to(), ``p.data = ...``) or under a moment moves data_ptr():
            # the raw addresses the kernel writes through are compared too (one integer compare per row), as is
            # the set of trainable parameters (a requires_grad toggle changes which rows exist)
            if (all(p.requires_grad for p in (r[0] for r in rows))
                    and sum(1 for g in self.param_groups for p in g["params"] if p.requires_grad) == len(rows)
                    and all(r[0].data_ptr() == r[1] and
                            ((st.get("exp_avg") is r[8] and st.get("step") is r[7] and r[8].data_ptr() == r[2]
                              and st["exp_avg_sq"].data_ptr() == r[3])
                             if (st := self.state.get(r[0])) else r[8] is None) for r in rows)
                    # nobody else moved a step count: the shared host tensor still sums to the rows' own counts
                    and float(self._step_base.sum()) == float(sum(r[6] for r in rows))):
                return rows
        rows, beta_key = [], None
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("FusedAdam: weight_decay / amsgrad / maximize are not used by GSVC")
            b1, b2 = group["betas"]
            key = (float(b1), float(b2), float(group["eps"]))
            if beta_key is None:
                beta_key = key
            elif key != beta_key:
                raise NotImplementedError("FusedAdam: one (betas, eps) for all groups")
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                if not p.is_cuda or p.dtype != torch.float32:
                    raise _lib.GsvcError("FusedAdam updates dense float32 CUDA parameters (csrc/adam.hip)")
                if not p.is_contiguous():
                    raise _lib.GsvcError("FusedAdam: parameters and moments must be contiguous")
                rows.append([p, p.data_ptr(), None, None, p.numel(), group, 0, None, None, len(rows)])
        self._step_base, self._step_idx = torch.zeros(max(len(rows), 1), dtype=torch.float32), {}
        for row in rows:
            if self.state.get(row[0]):
                self._fill(row, self.state[row[0]])
        self._beta_key = beta_key if beta_key is not None else (0.9, 0.999, 1e-8)
        self._row_cache = (sig, rows)
        return rows

    def _fill(self, row, st):
        m, v = st["exp_avg"], st["exp_avg_sq"]
        if not (m.is_contiguous() and v.is_contiguous()):
            raise _lib.GsvcError("FusedAdam: parameters and moments must be contiguous")
        view = self._step_base[row[9]]                   # the parameter's step count lives in the shared host tensor from now on
        view.fill_(float(st["step"]))
        st["step"] = view
        row[2], row[3], row[6], row[7], row[8] = m.data_ptr(), v.data_ptr(), int(float(view)), view, m

    def _init_state(self, row):
        p = row[0]
        st = self.state[p]
        if len(st) == 0:
            st["step"] = 0.0
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        self._fill(row, st)

    def rewind(self, params):
        """Take back the step count of parameters whose guarded update did not happen."""
        rows = {id(r[0]): r for r in (self.__dict__.get("_row_cache") or (None, []))[1]}
        for p in params:
            st = self.state.get(p)
            if st and "step" in st:
                st["step"] -= 1
                if id(p) in rows:
                    rows[id(p)][6] -= 1
