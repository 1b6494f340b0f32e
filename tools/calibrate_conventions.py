"""Fix the rasterizer's free conventions against ANY module with the API of GSVC's external extension.

GSVC's rasterizer, ``diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer``, is an un-pinned external package (reference
README.md:52); what its call sites (ortho_gaussian_renderer/renderer.py:63-98, preprocess.py:99-104) do not pin is a run-time
switch of ours (include/gsvc_hip.h GSVC_RASTER_*).  On a box that HAS the extension:

    python tools/calibrate_conventions.py --module diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer

runs the six one-scene experiments of INTEGRATION.md section 4 through the module's own ``GaussianRasterizationSettings`` /
``GaussianRasterizer`` (constructed by keyword exactly as renderer.py:63-83 does), prints what each one read, the resulting
``flags`` / ``low_pass`` for ``pipe.raster_flags`` / ``pipe.raster_low_pass``, and writes ``tests/golden/raster_calibration.npz``:
a 256 x 256 scene with the module's image, radii, ``num_rendered`` and six gradients, which ``tests/test_raster_gpu.py::
test_calibration_fixture_parity`` holds the HIP rasterizer to (integers bit-exact, pixels and gradients 1e-4) from then on — the
step from "parity unpinned" to pinned.  The module is only ever CALLED: nothing of it is read or stored but its results.

The experiments use the identity view matrix (camera at z = 0 looking down -z: frame_cube/frame.py:18-43 with z = 0), so they do
not depend on the matrix layout; experiment 0 then checks the layout on a translated camera.  ``calibrate(module)`` is importable
(tests run it against gsvc_amd.rasterizer built with each of the 64 flag combinations and against the CPU oracle).
"""
from __future__ import annotations

import argparse
import importlib
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SLAB_ONE_SIDED, PIXEL_CORNER, DEPTH_DESCENDING, MEANS2D_PIXEL_UNITS, CLAMP_STOPS_GRADIENT, NO_LOW_PASS = 1, 2, 4, 8, 16, 32
DEFAULT_LOW_PASS = 0.3          # reference arguments/__init__.py:55 (kernel_size); the settings field is commented out at renderer.py:72


class CalibrationError(RuntimeError):
    """An experiment read something none of the switches describes."""


class _Bench:
    """One tiny orthographic set-up: H x W pixels, ``scale`` pixels per world unit, slab half-width ``thr``, camera at z_cam."""

    def __init__(self, module, device, H=48, W=64, scale=100.0, thr=0.08, z_cam=0.0, bg=(0.0, 0.0, 0.0)):
        self.m, self.dev = module, torch.device(device)
        self.H, self.W, self.scale, self.thr, self.z_cam = H, W, float(scale), float(thr), float(z_cam)
        self.x_min, self.y_min = -W / (2.0 * scale), -H / (2.0 * scale)
        # row-major maths matrix of the forward view (glm.lookAt(eye, eye - 0.1 z, +y)): p_view = p - (0, 0, z_cam); what the
        # reference passes is frame.view_matrix.permute(1, 0) = this matrix
        M = torch.eye(4, dtype=torch.float32)
        M[2, 3] = -z_cam
        self.settings = module.GaussianRasterizationSettings(
            image_height=int(H), image_width=int(W), x_min=self.x_min, y_min=self.y_min, scale=self.scale, threshold=self.thr,
            bg=torch.tensor(bg, dtype=torch.float32, device=self.dev), scale_modifier=1.0, viewmatrix=M.to(self.dev), sh_degree=0,
            campos=torch.tensor([0.0, 0.0, z_cam]), prefiltered=False, debug=False)
        self.rasterizer = module.GaussianRasterizer(raster_settings=self.settings)

    def world(self, u, v, z=None):
        """World position that ``(x - x_min) * scale`` maps to (u, v)."""
        return [self.x_min + u / self.scale, self.y_min + v / self.scale, self.z_cam if z is None else z]

    def tensors(self, pos, sigma_px, opacity, color, rot=None):
        t = lambda a, w: torch.tensor(np.asarray(a, np.float32).reshape(-1, w), device=self.dev, requires_grad=True)  # noqa: E731
        n = len(pos)
        sig = np.broadcast_to(np.asarray(sigma_px, np.float32).reshape(-1, 1) if np.ndim(sigma_px) else np.float32(sigma_px), (n, 1))
        scales = np.repeat(sig / self.scale, 3, axis=1)
        rots = np.tile(np.array([[1.0, 0.0, 0.0, 0.0]], np.float32), (n, 1)) if rot is None else rot
        return dict(means3D=t(pos, 3), colors=t(color, 3), opacities=t(opacity, 1), scales=t(scales, 3), rotations=t(rots, 4))

    def render(self, d):
        means2D = torch.zeros_like(d["means3D"], requires_grad=True)
        try:
            means2D.retain_grad()
        except RuntimeError:
            pass
        image, radii, num_rendered = self.rasterizer(means3D=d["means3D"], means2D=means2D, shs=None, colors_precomp=d["colors"],
                                                     opacities=d["opacities"], scales=d["scales"], rotations=d["rotations"],
                                                     cov3D_precomp=None)
        return image, radii, num_rendered, means2D

    def visible(self, pos, sigma_px=1.5):
        d = self.tensors(pos, sigma_px, [[0.5]] * len(pos), [[1, 1, 1]] * len(pos))
        with torch.no_grad():
            r = self.rasterizer.visible_filter(means3D=d["means3D"], scales=d["scales"], rotations=d["rotations"], cov3D_precomp=None)
        return r.detach().cpu().numpy()


def _image(img, b=None):
    a = img.detach().cpu().numpy()
    if a.ndim != 3 or a.shape[0] != 3 or (b is not None and a.shape != (3, b.H, b.W)):
        raise CalibrationError(f"rendered image has shape {a.shape}, expected [3, H, W]" + (f" = [3, {b.H}, {b.W}]" if b is not None else ""))
    return a


def calibrate(module, device="cuda", log=print):
    """Run the experiments; returns dict(flags=, low_pass=, readings={...}).  ``low_pass`` = 0.0 means the default 0.3."""
    readings = {}
    flags = 0
    b = _Bench(module, device)
    thr = b.thr

    # 1. slab: two points half a threshold in front of / behind the camera plane -------------------------------------------
    r = b.visible([b.world(32.0, 24.0, -thr / 2), b.world(32.0, 24.0, +thr / 2)])
    readings["slab_radii(z_view=-thr/2, +thr/2)"] = [int(r[0]), int(r[1])]
    if r[0] > 0 and r[1] > 0:
        z_a, z_b = -thr / 2, +thr / 2               # two depths inside the slab for the depth-order experiment
    elif r[0] > 0:
        flags |= SLAB_ONE_SIDED
        z_a, z_b = -3 * thr / 4, -thr / 4
    elif r[1] > 0:
        raise CalibrationError("the extension keeps 0 <= z_view <= threshold: GSVC_RASTER_SLAB_ONE_SIDED keeps -threshold <= z_view <= 0; "
                               "mirror the sign in csrc/raster_fwd.hip preprocess before using the flags below")
    else:
        raise CalibrationError("visible_filter returned radius 0 for points half a threshold from the camera plane")
    far = b.visible([b.world(32.0, 24.0, -3 * thr), b.world(32.0, 24.0, 3 * thr)])
    readings["slab_radii(z_view=-3thr, +3thr)"] = [int(far[0]), int(far[1])]
    if far[0] > 0 or far[1] > 0:
        raise CalibrationError("a point three thresholds from the camera plane is visible: `threshold` is not the slab half-width")

    # 2. pixel position: centre of mass of one isotropic Gaussian whose world position maps to (u, v) = (20, 14) -------------
    d = b.tensors([b.world(20.0, 14.0)], 2.0, [[0.5]], [[1.0, 1.0, 1.0]])
    with torch.no_grad():
        img = _image(b.render(d)[0], b)[0]
    if not img.sum() > 0:
        raise CalibrationError("an isotropic Gaussian in the middle of the slab rendered nothing")
    ys, xs = np.mgrid[0:b.H, 0:b.W]
    cx, cy = float((img * xs).sum() / img.sum()), float((img * ys).sum() / img.sum())
    readings["centre_of_mass(u=20, v=14)"] = [round(cx, 4), round(cy, 4)]
    if abs(cx - 19.5) < 0.1 and abs(cy - 13.5) < 0.1:
        off = -0.5
    elif abs(cx - 20.0) < 0.1 and abs(cy - 14.0) < 0.1:
        flags |= PIXEL_CORNER
        off = 0.0
    else:
        raise CalibrationError(f"centre of mass ({cx:.3f}, {cy:.3f}) is neither (19.5, 13.5) nor (20, 14): x / y swapped or another pixel convention")
    on_pixel = lambda px, py, z=None: b.world(px - off, py - off, z)      # noqa: E731  world position whose centre is pixel (px, py)

    # 3. low-pass: variance of the rendered blob minus the Gaussian's own, at two sizes ---------------------------------------
    hs = []
    for var in (0.25, 1.0):
        d = b.tensors([on_pixel(20, 14)], math.sqrt(var), [[0.5]], [[1.0, 1.0, 1.0]])
        with torch.no_grad():
            img = _image(b.render(d)[0], b)[0]
        c, nb = float(img[14, 20]), [float(img[14, 21]), float(img[14, 19]), float(img[15, 20]), float(img[13, 20])]
        if not (c > 0 and min(nb) > 0):
            raise CalibrationError(f"low-pass experiment: centre {c}, neighbours {nb}")
        if max(nb) - min(nb) > 1e-3 * c:
            raise CalibrationError(f"an isotropic Gaussian centred on a pixel is not symmetric: neighbours {nb}")
        hs.append(-1.0 / (2.0 * math.log(float(np.mean(nb)) / c)) - var)
    readings["low_pass(measured at sigma^2 = 0.25, 1.0)"] = [round(h, 5) for h in hs]
    if abs(hs[0] - hs[1]) > 5e-3:
        raise CalibrationError(f"the footprint is not covariance + constant: {hs}")
    h = float(np.mean(hs))
    low_pass = 0.0
    if abs(h) < 2e-3:
        flags |= NO_LOW_PASS
    elif abs(h - DEFAULT_LOW_PASS) > 2e-3:
        low_pass = round(h, 3)

    # 4. depth order: two nearly opaque Gaussians of different colours on the same pixel ------------------------------------
    d = b.tensors([on_pixel(20, 14, z_a), on_pixel(20, 14, z_b)], 2.0, [[0.99], [0.99]], [[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    with torch.no_grad():
        px = _image(b.render(d)[0], b)[:, 14, 20]
    readings["depth_order pixel (red at smaller z_view, blue at larger)"] = [round(float(v), 4) for v in px]
    if px[0] > 0.9 and px[2] < 0.1:
        pass                                        # ascending z_view: the smaller one is composited first
    elif px[2] > 0.9 and px[0] < 0.1:
        flags |= DEPTH_DESCENDING
    else:
        raise CalibrationError(f"depth experiment: pixel {px} shows neither Gaussian in front")

    # 5. units of viewspace_points.grad: against a finite difference of the loss in u ---------------------------------------
    def loss_at(du):
        d = b.tensors([b.world(22.0 - off + du, 14.0 - off)], 2.0, [[0.5]], [[1.0, 1.0, 1.0]])
        image, _, _, m2d = b.render(d)
        return image[:, 14, 20].sum(), m2d, d
    L, m2d, _ = loss_at(0.0)
    L.backward()
    g = m2d.grad.detach().cpu().numpy()[0]
    eps = 0.05
    with torch.no_grad():
        fd = (float(loss_at(+eps)[0]) - float(loss_at(-eps)[0])) / (2 * eps)
    readings["means2D.grad[0] / finite difference d loss / d u"] = round(float(g[0] / fd), 4)
    ratio = g[0] / fd
    if abs(ratio - 0.5 * b.W) < 0.02 * 0.5 * b.W:
        pass
    elif abs(ratio - 1.0) < 0.02:
        flags |= MEANS2D_PIXEL_UNITS
    else:
        raise CalibrationError(f"viewspace gradient / finite difference = {ratio:.4f}: neither W / 2 = {0.5 * b.W} (NDC units) nor 1 (pixels)")

    # 6. the 0.99 clamp: does a clamped alpha pass a gradient to the opacity? ----------------------------------------------
    d = b.tensors([on_pixel(20, 14)], 2.0, [[0.999]], [[1.0, 1.0, 1.0]])
    image, _, _, _ = b.render(d)
    image[:, 14, 20].sum().backward()
    go = float(d["opacities"].grad.detach().cpu().numpy()[0, 0])
    readings["d centre pixel / d opacity at opacity 0.999"] = round(go, 5)
    if go == 0.0:
        flags |= CLAMP_STOPS_GRADIENT
    elif not go > 1.0:
        raise CalibrationError(f"opacity gradient under the clamp = {go}: expected 0 or about 3 (three channels)")

    # 0. layout of the view matrix, on a translated camera (checked last: it needs the slab reading) -----------------------
    bt = _Bench(module, device, z_cam=0.9)
    zin = bt.z_cam - thr / 2
    r = bt.visible([bt.world(32.0, 24.0, zin), bt.world(32.0, 24.0, zin - bt.z_cam)])
    readings["translated camera: radii(at the camera plane, at z = -thr/2)"] = [int(r[0]), int(r[1])]
    if not (r[0] > 0 and r[1] == 0):
        raise CalibrationError("with the camera at z = 0.9 the slab did not move with it: the extension reads `viewmatrix` in another layout "
                               "than frame.view_matrix.permute(1, 0) = row-major [R | t] (transpose it in gsvc_amd/rasterizer.py settings_to_c)")
    for k, v in readings.items():
        log(f"  {k}: {v}")
    names = [n for n, bit in (("SLAB_ONE_SIDED", 1), ("PIXEL_CORNER", 2), ("DEPTH_DESCENDING", 4), ("MEANS2D_PIXEL_UNITS", 8),
                              ("CLAMP_STOPS_GRADIENT", 16), ("NO_LOW_PASS", 32)) if flags & bit]
    log(f"flags = {flags} ({' | '.join('GSVC_RASTER_' + n for n in names) if names else 'the DESIGN.md spec as it is'}), "
        f"low_pass = {low_pass if low_pass else 'default (0.3)'}")
    return dict(flags=flags, low_pass=low_pass, readings=readings)


def calibration_scene(seed=7, P=3000, H=256, W=256, T=64):
    """The fixture's scene: P Gaussians in the slab of a 256 x 256 frame (sizes 0.7 .. 8 px, every rotation), camera off the origin."""
    sys.path.insert(0, ROOT)
    from gsvc_amd import synthetic
    return synthetic.raster_scene(P, H=H, W=W, T=T, seed=seed, window_frames=8, sigma_px=(0.7, 8.0))


def record_fixture(module, device="cuda", seed=7):
    """Inputs + what ``module`` makes of them: image, radii, num_rendered and the six gradients of sum(image * dL)."""
    sc = calibration_scene(seed)
    s = sc["settings"]
    dev = torch.device(device)
    bg = (0.1, 0.2, 0.3)
    rs = module.GaussianRasterizationSettings(
        image_height=s["H"], image_width=s["W"], x_min=s["x_min"], y_min=s["y_min"], scale=s["scale"], threshold=s["threshold"],
        bg=torch.tensor(bg, dtype=torch.float32, device=dev), scale_modifier=1.0,
        viewmatrix=torch.tensor(s["viewmatrix"], dtype=torch.float32, device=dev), sh_degree=0,
        campos=torch.tensor([0.0, 0.0, s["z_cam"]]), prefiltered=False, debug=False)
    r = module.GaussianRasterizer(raster_settings=rs)
    d = {k: torch.tensor(sc[k], device=dev, requires_grad=True) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    means2D = torch.zeros_like(d["means3D"], requires_grad=True)
    image, radii, num_rendered = r(means3D=d["means3D"], means2D=means2D, shs=None, colors_precomp=d["colors"],
                                   opacities=d["opacities"], scales=d["scales"], rotations=d["rotations"], cov3D_precomp=None)
    rng = np.random.default_rng(seed + 1)
    dL = rng.uniform(-1.0, 1.0, (3, s["H"], s["W"])).astype(np.float32)
    (image * torch.tensor(dL, device=dev)).sum().backward()
    with torch.no_grad():
        radii_vf = r.visible_filter(means3D=d["means3D"], scales=d["scales"], rotations=d["rotations"], cov3D_precomp=None)
    out = {f"in_{k}": sc[k] for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    out.update(settings=np.array([s["H"], s["W"], s["x_min"], s["y_min"], s["scale"], s["threshold"], s["z_cam"]], np.float64),
               viewmatrix=np.asarray(s["viewmatrix"], np.float32), bg=np.array(bg, np.float32), dL=dL,
               image=image.detach().cpu().numpy(), radii=radii.detach().cpu().numpy().astype(np.int32),
               radii_visible_filter=radii_vf.detach().cpu().numpy().astype(np.int32), num_rendered=np.int64(int(num_rendered)),
               grad_means2D=means2D.grad.detach().cpu().numpy(),
               **{f"grad_{k}": d[k].grad.detach().cpu().numpy() for k in ("means3D", "colors", "opacities", "scales", "rotations")})
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--module", required=True, help="dotted name of a module with GaussianRasterizationSettings / GaussianRasterizer "
                                                    "(the real extension: diff_gaussian_rasterization.cuda_ortho_gaussian_rasterizer)")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "raster_calibration.npz"))
    ap.add_argument("--no-fixture", action="store_true")
    a = ap.parse_args(argv)
    sys.path.insert(0, ROOT)
    module = importlib.import_module(a.module)
    print(f"calibrating against {a.module} on {a.device}")
    res = calibrate(module, a.device)
    if not a.no_fixture:
        fx = record_fixture(module, a.device)
        np.savez_compressed(a.out, flags=np.int64(res["flags"]), low_pass=np.float64(res["low_pass"]), module=np.array(a.module), **fx)
        print(f"wrote {a.out}: {fx['in_means3D'].shape[0]} Gaussians, num_rendered {int(fx['num_rendered'])} "
              f"({os.path.getsize(a.out) / 1e6:.2f} MB).  tests/test_raster_gpu.py::test_calibration_fixture_parity now pins the HIP rasterizer to it;")
        print(f"set pipe.raster_flags = {res['flags']}" + (f", pipe.raster_low_pass = {res['low_pass']}" if res["low_pass"] else "") +
              " (gsvc_amd.arguments) for the renderer.")
    return res


if __name__ == "__main__":
    main()
