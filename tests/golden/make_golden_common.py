"""Helpers shared by the golden-vector generators (run in the build container only)."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def npy(t):
    # a COPY: .numpy() aliases the tensor's storage, and parameters are updated in place by optimizer.step()
    return t.detach().cpu().clone().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


GRAD_ROW_STRIDE = 5          # rows of a weight-gradient matrix kept (coprime to the kernels' 16-row tiles)
ROW_STRIDE = 16
WIDE_STRIDE = 32


def grads_of(named_params, out, pre, rows=True):
    """Every parameter's gradient: strided rows of the matrices, small tensors whole, float64 sum / abs-sum of all of it."""
    for name, p in named_params:
        g = p.grad
        if g is None:
            continue
        g = g.detach()
        out[f"{pre}sum::{name}"] = np.array([float(g.double().sum()), float(g.double().abs().sum()), float(g.abs().max())])
        if not rows:
            continue
        if name.startswith("_"):                                  # per-anchor tensors: every 16th anchor
            out[f"{pre}grad::{name}"] = g[::ROW_STRIDE]
        elif name.endswith("params"):                             # hash tables: every 4th row
            out[f"{pre}grad::{name}"] = g[::4]
        elif g.dim() == 2 and g.numel() > 2048:
            out[f"{pre}grad::{name}"] = g[::GRAD_ROW_STRIDE]
        else:
            out[f"{pre}grad::{name}"] = g


class RandomTape:
    """Records every tensor produced by torch.rand_like / Tensor.uniform_ so GPU tests can replay them."""

    def __init__(self):
        self.tape = []
        self._rand_like, self._uniform = torch.rand_like, torch.Tensor.uniform_

    def __enter__(self):
        tape = self.tape
        depth = [0]   # the TorchFunctionMode re-enters the patched python functions: record the outermost call only

        def rand_like(x, *a, **k):
            depth[0] += 1
            try:
                r = self._rand_like(x, *a, **k)
            finally:
                depth[0] -= 1
            if depth[0] == 0:
                tape.append(r.detach().clone().numpy())
            return r

        def uniform_(t, *a, **k):
            depth[0] += 1
            try:
                r = self._uniform(t, *a, **k)
            finally:
                depth[0] -= 1
            if depth[0] == 0:
                tape.append(r.detach().clone().numpy())
            return r

        torch.rand_like = rand_like
        torch.Tensor.uniform_ = uniform_
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.Tensor.uniform_ = self._rand_like, self._uniform


