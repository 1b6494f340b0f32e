"""The production-dimension model of the reference-generated fixtures (tests/golden/prod_render.npz, step_fixture.npz), rebuilt
from seeds with gsvc_amd's classes: the same state_dict keys and numbers as the reference model the generators built
(tests/golden/make_golden_prod.py, make_golden_step.py; checked through the fixtures' ``param_sum::`` entries)."""
from types import SimpleNamespace

import torch


def build(g, device="cuda"):
    from tests.golden import seeded
    from gsvc_amd.arguments import ModelParams
    from gsvc_amd.frame import SyntheticFrameCube
    from gsvc_amd.model import GaussianModel
    sc, P = seeded.SCENE, seeded.PROD
    fn = seeded.frame_numbers(sc["H"], sc["W"], sc["T"], sc["frame"])
    mp = ModelParams()
    mp.threshold = sc["threshold"]
    pc = GaussianModel(mp, feat_dim=P["feat_dim"], n_offsets=P["n_offsets"], voxel_size=0.001, update_depth=3, update_init_factor=16,
                       update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=P["n_features_per_level"],
                       log2_hashmap_size=P["log2_hashmap_size"], log2_hashmap_size_2D=P["log2_hashmap_size_2D"],
                       resolutions_list=P["resolutions_list"], resolutions_list_2D=P["resolutions_list_2D"], device=device)
    pc.update_anchor_bound(fn["x_min"], fn["y_min"], fn["z_min"])
    for name, t in seeded.anchors(sc["A"], fn, sc["threshold"], sc["seed"]).items():
        setattr(pc, name, torch.nn.Parameter(t.to(device), requires_grad=name not in ("_rotation", "_opacity")))
    seeded.fill_parameters(pc, sc["seed"])
    # the same model as the reference's: same state_dict keys, same numbers in them
    sums = {k[len("param_sum::"):]: g[k] for k in g.files if k.startswith("param_sum::")}
    mine = {k: v for k, v in pc.state_dict().items() if v.is_floating_point() and v.numel()}
    assert set(mine) == set(sums), set(mine) ^ set(sums)
    for k, v in mine.items():
        assert abs(float(v.double().sum()) - sums[k][0]) <= 1e-9 * max(1.0, sums[k][1]), k
    # the frame's numbers: SyntheticFrameCube follows the same formulas (reference frame_cube/frame.py:92-101,156-190)
    fr = SyntheticFrameCube(sc["H"], sc["W"], sc["T"]).get_dummy_frame(sc["frame"])
    assert (fr.x_min, fr.y_min, fr.scale, fr.z) == (fn["x_min"], fn["y_min"], fn["scale"], fn["z"])
    assert torch.equal(fr.view_matrix.cpu(), fn["view_matrix"]) and torch.equal(fr.view_matrix_s.cpu(), fn["view_matrix_s"])
    return pc, mp, fn


def frame(fn, view, idx=None, image=None):
    from tests.golden import seeded
    sc = seeded.SCENE
    vm, vms = (fn["view_matrix"], fn["view_matrix_s"]) if view == "f" else (fn["view_matrix_s"], fn["view_matrix"])
    return SimpleNamespace(image_id=sc["frame"] if idx is None else idx, plane="xy", image=image, x_min=fn["x_min"], y_min=fn["y_min"],
                           z=fn["z"], image_width=sc["W"], image_height=sc["H"], view_matrix=vm.clone(), view_matrix_s=vms.clone(),
                           scale=fn["scale"], cam_pos=fn["cam_pos"].clone())
