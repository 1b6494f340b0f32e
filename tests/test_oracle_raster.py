"""CPU tests of oracle/raster_oracle.c against the dense float64 autograd statement of the raster spec.

The rasterizer's parity is UNPINNED (source absent from the reference, no reference tests); these tests
make the oracle self-consistent: forward == dense spec, backward == autograd of the dense spec, and the
f/b view symmetry the reference's training loop relies on (pipeline/train.py:353-375).
"""
import numpy as np
import pytest
import torch

from gsvc_amd import synthetic
from tests._dense_raster import dense_render


def _tiny_scene(P=40, H=40, W=56, seed=0, sigma=(1.0, 5.0)):
    sc = synthetic.raster_scene(P, H=H, W=W, T=32, seed=seed, window_frames=8, sigma_px=sigma)
    return sc


def _settings(oracle, s, view="viewmatrix"):
    return oracle.make_settings(s["H"], s["W"], s["x_min"], s["y_min"], s["scale"], s["threshold"], s[view],
                                bg=s["bg"], scale_modifier=s["scale_modifier"], flags=s.get("flags", 0),
                                low_pass=s.get("low_pass", 0.0))


# every convention switch of include/gsvc_hip.h alone, all of them together, and a non-default low-pass
FLAG_CASES = [(0, 0.0), (1, 0.0), (2, 0.0), (4, 0.0), (8, 0.0), (16, 0.0), (32, 0.0), (0, 0.1), (1 | 2 | 4 | 8 | 16, 0.55)]


@pytest.mark.parametrize("flags,low_pass", FLAG_CASES)
def test_forward_matches_dense(oracle_lib, flags, low_pass):
    sc = _tiny_scene()
    s = sc["settings"]
    s["bg"] = (0.1, 0.2, 0.3)
    s["flags"], s["low_pass"] = flags, low_pass
    st = _settings(oracle_lib, s)
    fwd = oracle_lib.raster_forward(st, sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"])
    assert fwd.num_rendered > 0 and (fwd.radii > 0).sum() > 10
    t = {k: torch.tensor(sc[k].astype(np.float64)) for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    img = dense_render(s, t["means3D"], t["colors"], t["opacities"].view(-1), t["scales"], t["rotations"], fwd.radii)
    ok = fwd.borderline == 0
    err = np.abs(img.numpy() - fwd.image)[:, ok]
    assert err.max() < 2e-5, err.max()


@pytest.mark.parametrize("flags,low_pass", FLAG_CASES)
def test_backward_matches_autograd(oracle_lib, flags, low_pass):
    sc = _tiny_scene(P=30, seed=3)
    if flags & 16:
        sc["opacities"][::3] = 0.999          # centres above the 0.99 clamp: the switch must have something to switch
    s = sc["settings"]
    s["bg"] = (0.3, 0.1, 0.6)
    s["flags"], s["low_pass"] = flags, low_pass
    st = _settings(oracle_lib, s)
    fwd = oracle_lib.raster_forward(st, sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"])
    rng = np.random.default_rng(5)
    dL = rng.standard_normal((3, s["H"], s["W"])).astype(np.float32)
    dL[:, fwd.borderline != 0] = 0  # pixels whose threshold decisions sit on a float-rounding boundary
    assert (fwd.borderline != 0).mean() < 0.01
    bwd = oracle_lib.raster_backward(st, sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"], fwd, dL)

    t = {k: torch.tensor(sc[k].astype(np.float64), requires_grad=True)
         for k in ("means3D", "colors", "opacities", "scales", "rotations")}
    delta = torch.zeros(sc["means3D"].shape[0], 2, dtype=torch.float64, requires_grad=True)
    img = dense_render(s, t["means3D"], t["colors"], t["opacities"].view(-1), t["scales"], t["rotations"], fwd.radii,
                       uv_delta=delta)
    (img * torch.tensor(dL.astype(np.float64))).sum().backward()

    def close(a, b, name, rtol=2e-3, atol=None):
        a = np.asarray(a, dtype=np.float64)
        b = np.asarray(b, dtype=np.float64)
        atol = atol if atol is not None else 2e-4 * max(1e-12, np.abs(b).max())
        assert np.allclose(a, b, rtol=rtol, atol=atol), (name, np.abs(a - b).max(), np.abs(b).max())

    close(bwd.colors, t["colors"].grad, "colors")
    close(bwd.opacities, t["opacities"].grad, "opacities")
    close(bwd.means3D, t["means3D"].grad, "means3D")
    close(bwd.scales, t["scales"].grad, "scales")
    close(bwd.rotations, t["rotations"].grad, "rotations")
    g2 = delta.grad.numpy() * (np.array([1.0, 1.0]) if flags & 8 else np.array([0.5 * s["W"], 0.5 * s["H"]]))
    close(bwd.means2D[:, :2], g2, "means2D")
    assert np.all(bwd.means2D[:, 2] == 0)
    # no gradient along view z (orthographic: depth only orders)
    assert np.all(bwd.means3D[:, 2] == 0)


def test_opposite_view_is_mirror(oracle_lib):
    """A lone Gaussian rendered through view_matrix_s and flipped along W lands on the same pixels
    (reference pipeline/train.py:368-375 averages image_f with flip(image_b))."""
    sc = _tiny_scene(P=1, H=32, W=48, seed=11)
    s = sc["settings"]
    sc["means3D"][0] = [0.13 * -s["x_min"], -0.2 * -s["y_min"], s["z_cam"]]
    sc["opacities"][0] = 0.8
    f = oracle_lib.raster_forward(_settings(oracle_lib, s), sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"])
    b = oracle_lib.raster_forward(_settings(oracle_lib, s, "viewmatrix_s"), sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"])
    assert f.radii[0] > 0 and b.radii[0] == f.radii[0]
    assert np.abs(f.image - b.image[:, :, ::-1]).max() < 1e-5
    assert f.image.max() > 0.1


def test_two_views_reverse_depth_order(oracle_lib):
    """Through view_matrix_s the tile lists are the same Gaussians in reverse depth order."""
    sc = _tiny_scene(P=25, H=16, W=16, seed=2, sigma=(3.0, 6.0))
    s = sc["settings"]
    # all on one tile, distinct depths, centred so the mirrored tile is the same tile
    sc["means3D"][:, 0] *= 0.2
    sc["means3D"][:, 1] *= 0.2
    f = oracle_lib.raster_forward(_settings(oracle_lib, s), sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"])
    b = oracle_lib.raster_forward(_settings(oracle_lib, s, "viewmatrix_s"), sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"])
    assert f.num_rendered == b.num_rendered > 5
    assert list(f.point_list) == list(b.point_list[::-1])


def test_edge_cases(oracle_lib):
    sc = _tiny_scene(P=8, H=20, W=30, seed=4)
    s = sc["settings"]
    s["bg"] = (0.5, 0.25, 0.125)
    st = _settings(oracle_lib, s)
    # empty input
    e = oracle_lib.raster_forward(st, np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 1)), np.zeros((0, 3)), np.zeros((0, 4)))
    assert e.num_rendered == 0 and e.point_list.size == 0
    assert np.allclose(e.image[0], 0.5) and np.allclose(e.image[2], 0.125) and np.all(e.final_T == 1)
    # everything outside the slab
    m = sc["means3D"].copy()
    m[:, 2] += 10 * s["threshold"]
    c = oracle_lib.raster_forward(st, m, sc["colors"], sc["opacities"], sc["scales"], sc["rotations"])
    assert c.num_rendered == 0 and np.all(c.radii == 0)
    # off-screen and NaN are culled, on the slab boundary is kept
    m = sc["means3D"].copy()
    m[0, 0] = 50.0
    m[1, 1] = np.nan
    m[2, 2] = s["z_cam"] + np.float32(s["threshold"]) * np.float32(0.999)
    r, tiles, total = oracle_lib.raster_preprocess(st, m, sc["scales"], sc["rotations"])
    assert r[0] == 0 and r[1] == 0 and r[2] > 0
    assert total == tiles.sum()
    # ties in depth keep index order
    m = np.repeat(sc["means3D"][:1], 5, axis=0)
    f = oracle_lib.raster_forward(st, m, sc["colors"][:5], sc["opacities"][:5], sc["scales"][:5], sc["rotations"][:5])
    for t0, t1 in f.tile_ranges:
        seg = list(f.point_list[t0:t1])
        assert seg == sorted(seg)


def test_convention_switches_change_what_they_name(oracle_lib):
    """Each switch moves the result in the direction its name says (the cross-checks above would also pass if a flag were
    ignored by oracle and dense statement alike)."""
    sc = _tiny_scene(P=60, seed=8)
    s = sc["settings"]
    args = (sc["means3D"], sc["colors"], sc["opacities"], sc["scales"], sc["rotations"])

    def fwd(flags, low_pass=0.0):
        s["flags"], s["low_pass"] = flags, low_pass
        return oracle_lib.raster_forward(_settings(oracle_lib, s), *args)
    base = fwd(0)
    one = fwd(1)          # one-sided slab: only the Gaussians at or in front of the camera plane (z_view <= 0) survive
    zv = sc["means3D"][:, 2] - s["z_cam"]
    assert (one.radii > 0).sum() < (base.radii > 0).sum() and not np.any((one.radii > 0) & (zv > 0)) and np.all((one.radii > 0)[zv > 0] == 0)
    assert np.array_equal((one.radii > 0)[zv <= 0], (base.radii > 0)[zv <= 0])
    corner = fwd(2)       # without the half-pixel offset every centre sits half a pixel further right / down
    vis = (base.radii > 0) & (corner.radii > 0)
    assert np.allclose(corner.geom[vis, 0] - base.geom[vis, 0], 0.5, atol=1e-4) and np.allclose(corner.geom[vis, 1] - base.geom[vis, 1], 0.5, atol=1e-4)
    desc = fwd(4)         # reversed depth order inside every tile (distinct depths in this scene)
    assert desc.num_rendered == base.num_rendered
    for (a0, a1), (b0, b1) in zip(base.tile_ranges, desc.tile_ranges):
        assert list(base.point_list[a0:a1]) == list(desc.point_list[b0:b1][::-1])
    nolp, lp = fwd(32), fwd(0, 1.5)     # the low-pass widens every footprint
    assert np.all(nolp.radii[vis] <= base.radii[vis]) and np.all(lp.radii[vis] >= base.radii[vis]) and lp.num_rendered > nolp.num_rendered
    # backward-only switches
    dL = np.ones((3, s["H"], s["W"]), np.float32)
    s["flags"], s["low_pass"] = 0, 0.0
    b0 = oracle_lib.raster_backward(_settings(oracle_lib, s), *args, base, dL)
    s["flags"] = 8
    b8 = oracle_lib.raster_backward(_settings(oracle_lib, s), *args, base, dL)
    assert np.allclose(b0.means2D[:, 0], b8.means2D[:, 0] * 0.5 * s["W"], rtol=1e-6) and np.allclose(b0.means2D[:, 1], b8.means2D[:, 1] * 0.5 * s["H"], rtol=1e-6)
    assert np.array_equal(b0.means3D, b8.means3D)
