"""Frame description and the cube <-> camera conventions of GSVC.

Same quantities as reference frame_cube/frame.py: ``make_view_matrix`` (:18-43, glm.lookAt from (x,y,z) towards
-/+ the plane normal), ``Frame`` (:46-59) and ``FrameCubeDataset.get_z_frame`` (:156-190):
``scale = max(H, W, T)/2``, ``x_min = -W/2/scale``, ``y_min = -H/2/scale``, ``z = (id - T/2)/scale``.
pyglm is not needed: lookAt is written out.  As in the reference the stored ``view_matrix`` is the TRANSPOSE of
the math matrix (np.array(glm.mat4) is column-major), and the renderer passes ``view_matrix.permute(1, 0)``.
Frames from files: ``gsvc_amd.io.FrameCubeDataset``; ``SyntheticFrameCube`` provides procedural frames of the same shape.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


def _look_at(eye, center, up):
    f = center - eye
    f = f / np.linalg.norm(f)
    s = np.cross(f, up)
    s = s / np.linalg.norm(s)
    u = np.cross(s, f)
    M = np.eye(4, dtype=np.float64)
    M[0, :3], M[1, :3], M[2, :3] = s, u, -f
    M[0, 3], M[1, 3], M[2, 3] = -s.dot(eye), -u.dot(eye), f.dot(eye)
    return M


def make_view_matrix(x=0, y=0, z=0, plane="xy"):
    """Returns (view_matrix, view_matrix_s, cam_pos) as float32 tensors; the matrices are stored transposed."""
    eye = np.array([x, y, z], dtype=np.float64)
    axis = {"xy": 2, "yz": 0, "zx": 1}[plane]
    up = {"xy": [0, 1, 0], "yz": [0, 0, 1], "zx": [1, 0, 0]}[plane]
    d = np.zeros(3)
    d[axis] = 0.1
    Mf = _look_at(eye, eye - d, np.array(up, dtype=np.float64))
    Ms = _look_at(eye, eye + d, np.array(up, dtype=np.float64))
    return (torch.tensor(Mf.T.copy(), dtype=torch.float32), torch.tensor(Ms.T.copy(), dtype=torch.float32),
            torch.tensor(eye, dtype=torch.float32))


@dataclass
class Frame:
    image_id: int
    plane: str
    image: torch.Tensor      # stored [3, W, H] as in the reference (permuted back at use)
    x_min: float
    y_min: float
    z: float
    image_width: int
    image_height: int
    view_matrix: torch.Tensor
    view_matrix_s: torch.Tensor
    scale: float
    cam_pos: torch.Tensor


class SyntheticFrameCube:
    """T procedural frames (moving Gaussian blobs on a gradient) with their analytic backward flow, shaped
    and addressed like the reference's FrameCubeDataset."""

    def __init__(self, height: int, width: int, frames: int, seed: int = 1234, blobs: int = 32, device="cpu"):
        self.height, self.width, self.len = height, width, frames
        self.scale = max(height, width, frames) / 2
        self.x_min = -width / 2 / self.scale
        self.y_min = -height / 2 / self.scale
        self.z_min = -frames / 2 / self.scale
        self.device = torch.device(device)
        rng = np.random.default_rng(seed)
        self._p0 = rng.uniform([0, 0], [width, height], (blobs, 2))
        self._vel = rng.uniform(-2.0, 2.0, (blobs, 2))
        self._sig = rng.uniform(0.03, 0.12, blobs) * min(height, width)
        self._col = rng.uniform(0.1, 1.0, (blobs, 3))
        self._cache = {}
        self._views = {}
        self._cache_limit = 64

    def materialize(self):
        """Generate and keep every frame and flow field on the device (the reference's FrameCubeDataset also holds the
        whole video in memory, frame_cube/frame.py:141-152), so that fetching a frame launches no kernel."""
        self._cache_limit = 2 * self.len + 2
        for t in range(self.len):
            self._image(t)
            self.get_optical_flow(t)
        return self

    def __len__(self):
        return self.len

    @property
    def len_z_frames(self):
        return self.len

    def _image(self, t):
        if t in self._cache:
            return self._cache[t]
        H, W = self.height, self.width
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=self.device),
                                torch.arange(W, dtype=torch.float32, device=self.device), indexing="ij")
        img = torch.stack([xs / W * 0.2, ys / H * 0.2, torch.full_like(xs, 0.1)], 0)
        for k in range(self._p0.shape[0]):
            cx, cy = self._p0[k] + self._vel[k] * t
            g = torch.exp(-((xs - float(cx) % W) ** 2 + (ys - float(cy) % H) ** 2) / (2 * float(self._sig[k]) ** 2))
            img = img + g[None] * torch.tensor(self._col[k], dtype=torch.float32, device=self.device)[:, None, None] * 0.5
        img = img.clamp(0, 1)
        if len(self._cache) < self._cache_limit:
            self._cache[t] = img
        return img

    def get_z_frame(self, image_id, load_image=True):
        z = (image_id - self.len / 2) / self.scale
        if image_id not in self._views:      # the camera of a frame never changes: built once (0.15 ms of host time)
            self._views[image_id] = make_view_matrix(z=z, plane="xy")
        vm, vms, cam = self._views[image_id]
        img = self._image(image_id).permute(0, 2, 1) if load_image else None
        return Frame(image_id=image_id, plane="xy", image=img, x_min=self.x_min, y_min=self.y_min, z=z,
                     image_width=self.width, image_height=self.height, view_matrix=vm, view_matrix_s=vms,
                     scale=self.scale, cam_pos=cam)

    def __getitem__(self, idx):
        return self.get_z_frame(idx)

    def get_dummy_frame(self, image_id):
        return self.get_z_frame(image_id, load_image=False)

    def get_optical_flow(self, idx):
        """Backward flow [2, H, W] (pixels/frame): a smooth field blended from the blob velocities."""
        key = ("flow", idx)
        if key in self._cache:
            return self._cache[key]
        H, W = self.height, self.width
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=self.device),
                                torch.arange(W, dtype=torch.float32, device=self.device), indexing="ij")
        num = torch.zeros(2, H, W, device=self.device)
        den = torch.full((H, W), 1e-3, device=self.device)
        for k in range(self._p0.shape[0]):
            cx, cy = self._p0[k] + self._vel[k] * idx
            g = torch.exp(-((xs - float(cx) % W) ** 2 + (ys - float(cy) % H) ** 2) / (2 * float(self._sig[k]) ** 2))
            num += g[None] * torch.tensor(self._vel[k], dtype=torch.float32, device=self.device)[:, None, None]
            den += g
        flow = num / den
        if len(self._cache) < self._cache_limit:
            self._cache[key] = flow
        return flow


class HostResidentCube:
    """A frame cube whose pictures and flow fields stay in (pinned) HOST memory, as the reference's dataset keeps them
    (frame_cube/frame.py:141-152: CPU tensors; the step uploads its two ground-truth frames every iteration, inside its own step
    timer: pipeline/train.py:332,407-408,464).  Uploads run on a copy stream ONE STEP AHEAD: ``prefetch(idx)`` — which the fitting
    step calls as soon as it has drawn the next frame pair — queues the pair's pictures and flow into one of two device slots, and
    ``__getitem__`` hands out frames whose ``image`` is that slot's tensor (the reader queues the wait with ``ready(i)`` just before its
    first use), so the copy (2 x 24.9 MB + 16.6 MB
    at 1080p: ~1.2 ms of PCIe 5 x16) runs under the previous step's kernels instead of in front of this step's.  Without a
    prefetch (the first step, or a caller that names its own frame) the copy is issued at use."""

    def __init__(self, cube, device):
        self.cube, self.device = cube, torch.device(device)
        for name in ("height", "width", "scale", "x_min", "y_min", "z_min"):
            setattr(self, name, getattr(cube, name))
        T = cube.len_z_frames
        self._images = [cube[i].image.detach().to("cpu").contiguous().pin_memory() for i in range(T)]
        self._flows = [cube.get_optical_flow(i).detach().to("cpu").contiguous().pin_memory() for i in range(T - 1)]
        self._copy = torch.cuda.Stream(device=self.device)
        self._seq = 0
        self._pair = None
        self._slots = [self._new_slot() for _ in range(2)]
        self.uploads = 0

    def __len__(self):
        return self.cube.len_z_frames

    @property
    def len_z_frames(self):
        return self.cube.len_z_frames

    def _new_slot(self):
        img, flow = self._images[0], self._flows[0]
        return {"img": [torch.empty_like(img, device=self.device) for _ in range(2)], "flow": torch.empty_like(flow, device=self.device),
                "idx": None, "ready": None, "free": None, "busy": False, "seq": 0}

    def prefetch(self, idx):
        """Queue the upload of the frame pair (idx, idx + 1) and the flow between them.  Returns False (nothing queued) when no
        slot may be written yet: a slot handed out since the last ``step_done()`` still has readers that are not queued (the
        image losses' backward), so no event recorded now could order the copy behind them — the pair is then uploaded at use."""
        if any(s["idx"] == idx for s in self._slots):
            return True
        free = [s for s in self._slots if not s["busy"]]
        if not free:
            return False
        slot = min(free, key=lambda s: s["seq"])         # the slot written longest ago
        self._upload(slot, idx)
        return True

    def _upload(self, slot, idx):
        if slot["free"] is not None:
            self._copy.wait_event(slot["free"])          # the step that last read this slot has been queued past its readers
        with torch.cuda.stream(self._copy):
            slot["img"][0].copy_(self._images[idx], non_blocking=True)
            slot["img"][1].copy_(self._images[idx + 1], non_blocking=True)
            slot["flow"].copy_(self._flows[idx], non_blocking=True)
            slot["ready"] = self._copy.record_event()
        slot["idx"] = idx
        self._seq += 1
        slot["seq"] = self._seq
        self.uploads += 1

    def _slot_of(self, i, pair=None):
        """The slot frame i is read from: the slot of the PAIR being read (``pair`` = its first frame; a step reads idx and idx + 1
        from the one slot its prefetch filled), else the slot that starts at i, else the most recently written slot that ends at i
        ("the first slot that covers i" could be the previous step's, which the next prefetch overwrites)."""
        cands = [s for s in self._slots if s["idx"] is not None and s["idx"] <= i <= s["idx"] + 1]
        if pair is not None:
            cands = [s for s in cands if s["idx"] == pair]
        if cands:
            s = max(cands, key=lambda s: (s["idx"] == i, s["seq"]))
        else:
            idx = pair if pair is not None else min(i, self.len_z_frames - 2)
            if not self.prefetch(idx):
                # both slots are being read by the step in flight: a fresh slot (dropped again at step_done) instead of a wait that
                # could not cover readers that are not queued yet
                self._slots.append(self._new_slot())
                self._upload(self._slots[-1], idx)
            s = next(x for x in self._slots if x["idx"] == idx)
        s["busy"] = True
        return s

    def step_done(self):
        """The step that read the current slots has been queued in full: their next upload may start behind this point."""
        ev = torch.cuda.current_stream(self.device).record_event()
        for s in self._slots:
            if s["busy"]:
                s["free"], s["busy"] = ev, False
        if len(self._slots) > 2:                     # extra slots of a step that read more than two pairs: their tensors stay alive
            keep = sorted(self._slots, key=lambda s: -s["seq"])[:2]      # (caching allocator, stream-ordered) until the readers ran
            for s in self._slots:
                if s not in keep:
                    for t in (*s["img"], s["flow"]):
                        t.record_stream(torch.cuda.current_stream(self.device))
            self._slots = keep

    def _resolve(self, i):
        # a step asks for idx, then idx + 1: the second comes from the slot the first was served from (``_pair``)
        pair = self._pair if (self._pair is not None and i == self._pair + 1 and any(s["idx"] == self._pair for s in self._slots)) else None
        s = self._slot_of(i, pair)
        self._pair = s["idx"] if s["idx"] == i else None
        return s

    def __getitem__(self, i):
        """The frame with ``image`` = its device slot.  No wait is queued here — a step fetches its frames first and reads the
        pictures last (image losses, behind generation and compositing): the reader calls ``ready(i)`` just before."""
        import copy
        s = self._resolve(i)
        fr = copy.copy(self.cube.get_dummy_frame(i))
        fr.image = s["img"][i - s["idx"]]
        return fr

    def ready(self, i):
        """Make the current stream wait for the upload of frame i's picture (a no-op once it has landed); called as the frames were
        fetched (idx, then idx + 1), it resolves to the same slots and marks them as being read until ``step_done()``."""
        torch.cuda.current_stream(self.device).wait_event(self._resolve(i)["ready"])

    def get_dummy_frame(self, i):
        return self.cube.get_dummy_frame(i)

    def get_optical_flow(self, idx):
        s = self._slot_of(idx, pair=idx)             # the flow between idx and idx + 1 lives in the slot of that pair
        torch.cuda.current_stream(self.device).wait_event(s["ready"])
        return s["flow"]
