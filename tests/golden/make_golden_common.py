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


