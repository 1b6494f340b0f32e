"""File IO of the GSVC path (gsvc_amd/io.py) on the host: frame / optical-flow loading with the reference's
conventions (frame_cube/frame.py:60-190), the anchor ply with the reference's property list (scene/gaussian_model.py:1156-1240),
the checkpoint tuple (:556-639) and the MLP checkpoint (:1505-1540)."""
import os
import pickle

import numpy as np
import pytest
import torch

from gsvc_amd import io as gio


def _tiny_model():
    from gsvc_amd.arguments import ModelParams
    from gsvc_amd.model import GaussianModel
    torch.manual_seed(5)
    pc = GaussianModel(ModelParams(), feat_dim=8, n_offsets=4, voxel_size=0.001, update_depth=3, update_init_factor=16,
                       update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=2, log2_hashmap_size=9,
                       log2_hashmap_size_2D=11, resolutions_list=(18, 24, 33), resolutions_list_2D=(130, 258), device="cpu")
    rng = np.random.default_rng(2)
    pc.create_from_points(rng.uniform(-0.4, 0.4, (300, 3)), 1.0)
    with torch.no_grad():
        for p in (pc._offset, pc._mask, pc._anchor_feat, pc._scaling):
            p.copy_(torch.randn(p.shape))
    return pc


def test_frame_cube_dataset_from_files(tmp_path):
    from PIL import Image
    from gsvc_amd.frame import SyntheticFrameCube
    H, W, T = 20, 36, 6
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, (T, H, W, 3), dtype=np.uint8)
    fdir, odir = tmp_path / "frames", tmp_path / "flow"
    fdir.mkdir(); odir.mkdir()
    for t in range(T):
        Image.fromarray(frames[t]).save(fdir / f"im{t:05d}.png")
    flows = rng.standard_normal((T - 1, 2, H, W)).astype(np.float32)
    for t in range(T - 1):
        if t % 2:
            np.save(odir / f"of{t:05d}.npy", flows[t])
        else:
            with open(odir / f"of{t:05d}.pkl", "wb") as f:
                pickle.dump(flows[t], f)
    ds = gio.FrameCubeDataset(fdir, odir)
    syn = SyntheticFrameCube(H, W, T)
    assert (len(ds), ds.height, ds.width) == (T, H, W)
    assert (ds.scale, ds.x_min, ds.y_min, ds.z_min) == (syn.scale, syn.x_min, syn.y_min, syn.z_min)
    for t in (0, 3, T - 1):
        fr, ref = ds[t], syn.get_dummy_frame(t)
        assert fr.image.shape == (3, W, H)                                   # kept transposed, as the reference does
        assert torch.equal(fr.image.permute(0, 2, 1), torch.from_numpy(frames[t]).permute(2, 0, 1).float() / 255)
        assert fr.z == ref.z and torch.equal(fr.view_matrix, ref.view_matrix) and torch.equal(fr.view_matrix_s, ref.view_matrix_s)
        assert torch.equal(fr.cam_pos, ref.cam_pos) and (fr.image_width, fr.image_height) == (W, H)
    for t in range(T - 1):
        assert torch.equal(ds.get_optical_flow(t), torch.from_numpy(flows[t]))
    assert ds.get_dummy_frame(2).image is None
    lazy = gio.FrameCubeDataset(fdir, odir, prefetch=False)
    assert torch.equal(lazy[4].image, ds[4].image) and torch.equal(lazy.get_optical_flow(1), ds.get_optical_flow(1))


def test_flow_pickle_may_only_hold_arrays(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    p = tmp_path / "evil.pkl"
    with open(p, "wb") as f:
        pickle.dump(Evil(), f)
    with pytest.raises(pickle.UnpicklingError):
        gio.load_flow(p)
    with open(p, "wb") as f:
        pickle.dump([[1.0, 2.0], [3.0, 4.0]], f)                              # nested lists of numbers are fine
    assert torch.equal(gio.load_flow(p), torch.tensor([[1.0, 2.0], [3.0, 4.0]]))


def test_ply_round_trip_and_layout(tmp_path):
    pc = _tiny_model()
    path = str(tmp_path / "pc" / "point_cloud.ply")
    gio.save_ply(pc, path)
    names, table = gio.read_ply(path)
    K, F = pc.n_offsets, pc.feat_dim
    want = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_offset_{i}" for i in range(3 * K)] + [f"f_mask_{i}" for i in range(K)] + \
        [f"f_anchor_feat_{i}" for i in range(F)] + ["opacity"] + [f"scale_{i}" for i in range(6)] + [f"rot_{i}" for i in range(4)]
    assert names == want and table.shape == (pc._anchor.shape[0], len(want))
    head = open(path, "rb").read(64)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex ")
    # offsets are stored slot-minor: column f_offset_{c*K + k} = _offset[:, k, c]
    col = {n: i for i, n in enumerate(names)}
    assert np.array_equal(table[:, col[f"f_offset_{1 * K + 2}"]].astype(np.float32), pc._offset[:, 2, 1].detach().numpy())
    assert np.all(table[:, 3:6] == 0)
    other = _tiny_model()
    with torch.no_grad():
        other._anchor_feat.zero_()
    gio.load_ply_sparse_gaussian(other, path)
    for n in ("_anchor", "_offset", "_mask", "_anchor_feat", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(other, n), getattr(pc, n)), n
    # an ascii file with the same columns loads too
    apath = str(tmp_path / "ascii.ply")
    with open(apath, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\n" % table.shape[0] + "".join(f"property float {n}\n" for n in names) + "end_header\n")
        np.savetxt(f, table, fmt="%.9g")
    n2, t2 = gio.read_ply(apath)
    assert n2 == names and np.array_equal(t2.astype(np.float32), table.astype(np.float32))
    with pytest.raises(ValueError):
        open(apath, "wb").write(open(path, "rb").read()[:-7]); gio.read_ply(apath)


def test_capture_restore_and_mlp_checkpoint(tmp_path):
    from gsvc_amd.arguments import OptimizationParams
    opt = OptimizationParams()
    pc = _tiny_model()
    pc.update_anchor_bound(-0.5, -0.5, -0.3)
    pc.training_setup(opt)
    pc.update_learning_rate(10)
    (pc._anchor_feat.sum() + pc._offset.sum() + sum(p.sum() for p in pc.mlp_cov.parameters())).backward()
    pc.optimizer.step()
    pc.offset_denom += 2
    pc.anchor_demon += 3
    args = gio.capture(pc)
    assert len(args) == 8 and "_anchor" in args[0] and "mlp_opacity.linear1.weight" in args[0]
    torch.save(args, tmp_path / "chkpnt.pth")
    args = torch.load(tmp_path / "chkpnt.pth", weights_only=True)      # the tuple holds tensors, dicts and numbers only
    new = _tiny_model()
    with torch.no_grad():
        for p in new.parameters():
            p.add_(1.0)
    gio.restore(new, args, opt)
    a, b = pc.state_dict(), new.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert torch.equal(new.offset_denom, pc.offset_denom) and torch.equal(new.anchor_demon, pc.anchor_demon)
    assert torch.equal(new.x_bound_min, pc.x_bound_min) and new.bound_max_host == pc.bound_max_host
    assert new.spatial_lr_scale == pc.spatial_lr_scale
    sa, sb = pc.optimizer.state_dict(), new.optimizer.state_dict()
    assert [g["name"] for g in sa["param_groups"]] == [g["name"] for g in sb["param_groups"]]
    for k in sa["state"]:
        for f in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(sa["state"][k][f], sb["state"][k][f])
    # MLP checkpoint: the reference's five keys
    gio.save_mlp_checkpoints(pc, str(tmp_path / "ck" / "checkpoint.pth"))
    ck = torch.load(tmp_path / "ck" / "checkpoint.pth", weights_only=True)
    assert sorted(ck) == ["color_mlp", "cov_mlp", "deform_mlp", "encoding_xyz", "opacity_mlp"]
    fresh = _tiny_model()
    with torch.no_grad():
        for p in fresh.mlp_color.parameters():
            p.zero_()
    gio.load_mlp_checkpoints(fresh, str(tmp_path / "ck" / "checkpoint.pth"))
    assert all(torch.equal(x, y) for x, y in zip(fresh.mlp_color.parameters(), pc.mlp_color.parameters()))
