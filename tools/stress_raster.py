"""Randomised forward/backward parity stress of the HIP rasterizer against the CPU oracle (GPU box).

Draws many scene configurations (image sizes incl. non-multiples of 16, Gaussian counts, footprint ranges, slab widths,
both views, non-black backgrounds, Gaussians off screen / outside the slab / with opacity <= 0) and checks what the
tests check: integers bit-exact, pixels 1e-4 outside borderline pixels, gradients 1e-4 of the tensor scale.
Usage: python tools/stress_raster.py [n_cases] [seed]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
from gsvc_amd import synthetic
from tests import test_raster_gpu as T

def run(n_cases=40, seed=123, verbose=True):
    oracle.build()
    rng = np.random.default_rng(seed)
    bad = 0
    for case in range(n_cases):
        H, W = int(rng.integers(17, 400)), int(rng.integers(17, 600))
        P = int(rng.integers(1, 30000))
        lo = float(rng.uniform(0.2, 2.0)); hi = lo * float(rng.uniform(1.5, 30.0))
        sc = synthetic.raster_scene(P, H=H, W=W, T=int(rng.integers(16, 300)), seed=int(rng.integers(1 << 30)),
                                    window_frames=float(rng.uniform(2, 40)), sigma_px=(lo, hi),
                                    opacity=(float(rng.uniform(0.0, 0.3)), float(rng.uniform(0.4, 1.0))))
        # spread some Gaussians beyond the screen / slab, and give some opacity <= 0
        sc["means3D"][:, :2] *= float(rng.uniform(0.8, 1.4))
        sc["means3D"][::7, 2] += float(rng.uniform(-0.2, 0.2))
        sc["opacities"][::11] *= -1.0
        view = "viewmatrix" if rng.random() < 0.5 else "viewmatrix_s"
        bg = tuple(float(v) for v in rng.uniform(0, 1, 3))
        try:
            r, ref, d = T._compare_forward(oracle, sc, view=view, bg=bg, max_borderline=5e-3)
            # backward on the same scene
            s = sc["settings"]
            dL = rng.standard_normal((3, H, W)).astype(np.float32)
            dL[:, ref.borderline != 0] = 0
            leaves = {k: v.clone().requires_grad_(True) for k, v in d.items()}
            m2 = torch.zeros_like(leaves["means3D"], requires_grad=True)
            img, _, _ = r(means3D=leaves["means3D"], means2D=m2, shs=None, colors_precomp=leaves["colors"], opacities=leaves["opacities"],
                          scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None)
            (img * torch.tensor(dL, device="cuda")).sum().backward()
            rb = oracle.raster_backward(T._oracle_settings(oracle, s, view, bg), sc["means3D"], sc["colors"], sc["opacities"], sc["scales"],
                                        sc["rotations"], ref, dL)
            for nm, got, want in (("means3D", leaves["means3D"].grad, rb.means3D), ("means2D", m2.grad, rb.means2D),
                                  ("colors", leaves["colors"].grad, rb.colors), ("opacities", leaves["opacities"].grad, rb.opacities),
                                  ("scales", leaves["scales"].grad, rb.scales), ("rotations", leaves["rotations"].grad, rb.rotations)):
                g = got.cpu().numpy().reshape(want.shape)
                scale = max(np.abs(want).max(), 1e-12)
                err = np.abs(g - want).max() / scale
                assert err < 2e-4, (nm, err)
            print(f"case {case}: ok  {H}x{W} P={P} sigma=({lo:.2f},{hi:.1f}) view={view} instances={ref.num_rendered}")
        except AssertionError as e:
            bad += 1
            import traceback
            tb = traceback.extract_tb(e.__traceback__)[-1]
            print(f"   at {os.path.basename(tb.filename)}:{tb.lineno}: {tb.line}")
            print(f"case {case}: FAIL {H}x{W} P={P} sigma=({lo:.2f},{hi:.1f}) view={view}: {e!r}"[:300])
    if verbose:
        print("failures:", bad, "of", n_cases)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 123) else 0)
