"""Data / checkpoint IO of the GSVC path: frames and optical flow from files, the anchor ply, the
training checkpoint tuple and the MLP checkpoint.

Mirrors reference frame_cube/frame.py:60-190 (``FrameCubeDataset``: sorted PNG directory, ``ToTensor`` scaling, images kept
transposed ``[3, W, H]``, one optical-flow file per frame pair), scene/gaussian_model.py:556-639 (``capture`` / ``restore``),
:1156-1240 (``save_ply`` / ``load_ply_sparse_gaussian``: property names and column order) and :1505-1540
(``save_mlp_checkpoints`` / ``load_mlp_checkpoints``: dictionary keys).

What differs: the ply is written and parsed here (binary little-endian float properties — the layout ``plyfile`` produces for
the reference — no third-party package); optical-flow files may be ``.npy`` / ``.npz`` besides the reference's pickles, and a
pickle is opened with an unpickler that admits NumPy arrays only; torch files are read with ``weights_only=True``.
"""
from __future__ import annotations

import io as _io
import os
import pathlib
import pickle

import numpy as np
import torch
import torch.nn as nn

from .frame import Frame, make_view_matrix


# ------------------------------------------------------------------------------------------------ frames
class _ArrayUnpickler(pickle.Unpickler):
    """Optical-flow pickles hold one NumPy array (or nested lists of numbers): nothing else is constructed."""

    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"optical-flow file names {module}.{name}: only NumPy arrays are accepted")


def load_flow(path) -> torch.Tensor:
    """One optical-flow field as a float32 tensor, from ``.npy``, ``.npz`` (first array) or an array pickle (reference
    frame_cube/frame.py:147-151)."""
    path = str(path)
    if path.endswith(".npy"):
        arr = np.load(path, allow_pickle=False)
    elif path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            arr = z[z.files[0]]
    else:
        with open(path, "rb") as f:
            arr = _ArrayUnpickler(f).load()
    return torch.tensor(np.asarray(arr), dtype=torch.float32)


def load_image(path) -> torch.Tensor:
    """RGB image file -> float32 [3, H, W] in [0, 1] (what torchvision's ``ToTensor`` returns for an 8-bit image)."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return torch.from_numpy(a.copy()).permute(2, 0, 1).to(torch.float32).div_(255)


class FrameCubeDataset:
    """A directory of frames (sorted by name) + a directory of optical-flow files, addressed like the reference's dataset:
    ``dataset[i]`` is frame i (image kept transposed, camera at ``z = (i - T/2) / scale``), ``get_optical_flow(i)`` the flow
    between frames i and i + 1.  ``device``: where prefetched tensors live (the fitting step reads two frames per step)."""

    def __init__(self, main_dir, optical_flow_dir=None, transform=None, prefetch=True, device="cpu"):
        self.main_dir = pathlib.Path(main_dir)
        self.z_frame_paths = sorted(p for p in self.main_dir.iterdir() if p.is_file())
        if not self.z_frame_paths:
            raise FileNotFoundError(f"no frames under {self.main_dir}")
        self.optical_flow_paths = sorted(pathlib.Path(optical_flow_dir).iterdir()) if optical_flow_dir is not None else []
        self.transform = transform if transform is not None else load_image
        first = self.transform(self.z_frame_paths[0])
        self.height, self.width = int(first.shape[-2]), int(first.shape[-1])
        self.scale = max(self.height, self.width, self.frame_num) / 2
        self.x_min = -self.width / 2 / self.scale
        self.y_min = -self.height / 2 / self.scale
        self.z_min = -len(self.z_frame_paths) / 2 / self.scale
        self.device = torch.device(device)
        self.prefetched_images, self.prefetched_of, self._views = [], [], {}
        if prefetch:
            self.prefetch()

    def __len__(self):
        return self.len_z_frames

    @property
    def frame_num(self):
        return self.len_z_frames

    @property
    def frame_height(self):
        return self.height

    @property
    def frame_width(self):
        return self.width

    @property
    def len_z_frames(self):
        return len(self.z_frame_paths)

    def prefetch(self):
        self.prefetched_images = [self.transform(p).permute(0, 2, 1).contiguous().to(self.device) for p in self.z_frame_paths]
        self.prefetched_of = [load_flow(p).to(self.device) for p in self.optical_flow_paths]

    def get_z_frame(self, image_id, load_image=True):
        z = (image_id - self.len_z_frames / 2) / self.scale
        if image_id not in self._views:
            self._views[image_id] = make_view_matrix(z=z, plane="xy")
        vm, vms, cam = self._views[image_id]
        img = None
        if load_image:
            img = self.prefetched_images[image_id] if self.prefetched_images else \
                self.transform(self.z_frame_paths[image_id]).permute(0, 2, 1)
        return Frame(image_id=image_id, plane="xy", image=img, x_min=self.x_min, y_min=self.y_min, z=z,
                     image_width=self.width, image_height=self.height, view_matrix=vm, view_matrix_s=vms, scale=self.scale,
                     cam_pos=cam)

    def get_dummy_frame(self, image_id):
        return self.get_z_frame(image_id, load_image=False)

    def __getitem__(self, idx):
        return self.get_z_frame(idx)

    def get_optical_flow(self, idx):
        return self.prefetched_of[idx] if self.prefetched_of else load_flow(self.optical_flow_paths[idx])


# ------------------------------------------------------------------------------------------------ ply
def construct_list_of_attributes(pc):
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_offset_{i}" for i in range(pc._offset.shape[1] * pc._offset.shape[2])]
    names += [f"f_mask_{i}" for i in range(pc._mask.shape[1] * pc._mask.shape[2])]
    names += [f"f_anchor_feat_{i}" for i in range(pc._anchor_feat.shape[1])]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(pc._scaling.shape[1])]
    names += [f"rot_{i}" for i in range(pc._rotation.shape[1])]
    return names


def write_ply(path, names, table: np.ndarray):
    """One ``vertex`` element of float32 properties, binary little-endian."""
    table = np.ascontiguousarray(table, dtype="<f4")
    assert table.ndim == 2 and table.shape[1] == len(names)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {table.shape[0]}\n" + \
        "".join(f"property float {n}\n" for n in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


_PLY_TYPES = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "int": "i4", "int32": "i4", "uint": "u4",
              "uint32": "u4", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "char": "i1", "int8": "i1",
              "uchar": "u1", "uint8": "u1"}


def read_ply(path):
    """(property names, float64 table [n, properties]) of the first element of a ply file (ascii or binary, scalar
    properties only — what the reference's anchor files contain)."""
    with open(path, "rb") as f:
        blob = f.read()
    end = blob.find(b"end_header\n")
    if not blob.startswith(b"ply") or end < 0:
        raise ValueError(f"{path}: not a ply file")
    fmt, count, props, seen_element = None, None, [], False
    for line in blob[:end].decode("ascii", "replace").splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            if seen_element:
                break                      # only the first element (vertex) is read
            seen_element, count = True, int(tok[2])
        elif tok[0] == "property":
            if tok[1] == "list":
                raise ValueError("list properties are not supported")
            props.append((tok[2], _PLY_TYPES[tok[1]]))
    body = blob[end + len(b"end_header\n"):]
    names = [n for n, _ in props]
    if fmt == "ascii":
        table = np.loadtxt(_io.BytesIO(body), max_rows=count, ndmin=2)
    else:
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        if len(body) < count * dt.itemsize:
            raise ValueError(f"{path}: truncated")
        rec = np.frombuffer(body, dtype=dt, count=count)
        table = np.stack([rec[n].astype(np.float64) for n in names], axis=1) if count else np.zeros((0, len(names)))
    return names, table


@torch.no_grad()
def save_ply(pc, path):
    """Per-anchor tensors as one ply (reference :1172-1191): offsets and masks stored slot-minor (``transpose(1, 2)``)."""
    anchor = pc._anchor.detach().cpu().numpy()
    cols = [anchor, np.zeros_like(anchor),
            pc._offset.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
            pc._mask.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy(),
            pc._anchor_feat.detach().cpu().numpy(), pc._opacity.detach().cpu().numpy(), pc._scaling.detach().cpu().numpy(),
            pc._rotation.detach().cpu().numpy()]
    write_ply(path, construct_list_of_attributes(pc), np.concatenate(cols, axis=1))


@torch.no_grad()
def load_ply_sparse_gaussian(pc, path):
    """Inverse of ``save_ply`` (reference :1193-1240); columns are found by name prefix and ordered by their numeric suffix."""
    names, table = read_ply(path)
    col = {n: i for i, n in enumerate(names)}

    def group(prefix):
        ns = sorted((n for n in names if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))
        return table[:, [col[n] for n in ns]].astype(np.float32)

    dev = pc.device
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    anchor = table[:, [col["x"], col["y"], col["z"]]].astype(np.float32)
    offsets = group("f_offset")
    offsets = offsets.reshape(offsets.shape[0], 3, -1)
    masks = group("f_mask")
    masks = masks.reshape(masks.shape[0], 1, -1)
    pc._anchor_feat = nn.Parameter(t(group("f_anchor_feat")).requires_grad_(True))
    pc._offset = nn.Parameter(t(offsets).transpose(1, 2).contiguous().requires_grad_(True))
    pc._mask = nn.Parameter(t(masks).transpose(1, 2).contiguous().requires_grad_(True))
    pc._anchor = nn.Parameter(t(anchor).requires_grad_(True))
    pc._opacity = nn.Parameter(t(table[:, [col["opacity"]]]).requires_grad_(True))
    pc._scaling = nn.Parameter(t(group("scale_")).requires_grad_(True))
    pc._rotation = nn.Parameter(t(group("rot")).requires_grad_(True))


# ------------------------------------------------------------------------------------------------ checkpoints
def _optimizer_state(optimizer):
    return optimizer.state_dict()


def _plain(obj):
    """NumPy scalars -> Python numbers (the learning-rate schedule yields ``np.float64``), so that the tuple loads with
    ``torch.load(..., weights_only=True)``."""
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def capture(pc):
    """The reference's checkpoint tuple (:556-584): (state_dict, x_bound_min, x_bound_max, max_radii2D, offset_denom,
    anchor_demon, optimizer state_dict, spatial_lr_scale)."""
    return (pc.state_dict(), pc.x_bound_min, pc.x_bound_max, pc.max_radii2D, pc.offset_denom, pc.anchor_demon,
            _plain(_optimizer_state(pc.optimizer)), _plain(pc.spatial_lr_scale))


def init_anchor_params(pc, anchor_num):
    """Empty per-anchor parameters of the right shapes, so that ``load_state_dict`` can fill them (reference
    ``init_anchor_params``)."""
    dev, K = pc.device, pc.n_offsets
    z = lambda *s: nn.Parameter(torch.zeros(*s, device=dev))  # noqa: E731
    pc._anchor, pc._offset, pc._mask = z(anchor_num, 3), z(anchor_num, K, 3), z(anchor_num, K, 1)
    pc._anchor_feat, pc._scaling = z(anchor_num, pc.feat_dim), z(anchor_num, 6)
    pc._rotation = nn.Parameter(torch.zeros(anchor_num, 4, device=dev), requires_grad=False)
    pc._opacity = nn.Parameter(torch.zeros(anchor_num, 1, device=dev), requires_grad=False)


def restore(pc, model_args, training_args):
    (state_dict, x_bound_min, x_bound_max, max_radii2D, offset_denom, anchor_demon, opt_dict, spatial_lr_scale) = model_args
    dev = pc.device
    pc.x_bound_min, pc.x_bound_max = x_bound_min.to(dev), x_bound_max.to(dev)
    pc.bound_min_host = tuple(float(v) for v in pc.x_bound_min.reshape(-1).tolist())
    pc.bound_max_host = tuple(float(v) for v in pc.x_bound_max.reshape(-1).tolist())
    pc.max_radii2D, pc.spatial_lr_scale = max_radii2D.to(dev), spatial_lr_scale
    init_anchor_params(pc, state_dict["_anchor"].shape[0])
    pc.training_setup(training_args)
    pc.offset_denom, pc.anchor_demon = offset_denom.to(dev), anchor_demon.to(dev)
    pc.load_state_dict(state_dict)
    pc.optimizer.load_state_dict(opt_dict)


_MLP_KEYS = (("opacity_mlp", "mlp_opacity"), ("cov_mlp", "mlp_cov"), ("color_mlp", "mlp_color"), ("encoding_xyz", "encoding_xyz"),
             ("deform_mlp", "mlp_deform"))


def save_mlp_checkpoints(pc, path):
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({k: getattr(pc, attr).state_dict() for k, attr in _MLP_KEYS}, path)


def load_mlp_checkpoints(pc, path):
    ck = torch.load(path, map_location=pc.device, weights_only=True)
    for k, attr in _MLP_KEYS:
        getattr(pc, attr).load_state_dict(ck[k])
