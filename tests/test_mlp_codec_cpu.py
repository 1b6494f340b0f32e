"""MLP weight stage of the stream codec (gsvc_amd/mlp_codec.py) on the host: the quantiser and the mask bits against
vectors generated with the reference's own utils/param_utils.py / utils/mask.py (tests/golden/make_golden_mlpq.py), the
Huffman code against its optimality bound and round trips, the container against corruption."""
import os
from collections import Counter

import numpy as np
import pytest
import torch

from gsvc_amd import mlp_codec as mc

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlp_quant.npz"))


@pytest.mark.parametrize("name", ["weight", "bias", "sparse", "wide"])
def test_quantize_tensor_matches_reference(name):
    t, axis = torch.from_numpy(G[f"{name}::t"]), int(G[f"{name}::axis"])
    q, m, new_t, meta = mc.quantize_tensor(t, 8, axis)
    assert np.array_equal(q.numpy(), G[f"{name}::quant"])
    assert np.array_equal(m.numpy(), G[f"{name}::mask"])
    assert np.array_equal(new_t.numpy(), G[f"{name}::new_t"])
    assert np.array_equal(np.asarray(meta["t_min"], np.float64), G[f"{name}::t_min"])
    assert np.array_equal(np.asarray(meta["scale"], np.float64), G[f"{name}::scale"])
    if axis == 0:
        assert np.array_equal(mc.dequantize_tensor(q, m, meta).numpy(), G[f"{name}::dequant"])
    assert q.max() <= 256 and q[m].min() >= 0      # 2**8 steps between min and max: symbols 0 .. 256


def test_mask_bits_match_reference():
    bits = G["mask::bits"]
    assert np.array_equal(np.frombuffer(mc.mask_to_bytes(bits), np.uint8), G["mask::packed"])
    back = mc.decode_mask(mc.encode_mask(torch.from_numpy(bits)))
    assert np.array_equal(back.numpy(), G["mask::decoded"])
    assert np.array_equal(back.numpy()[:bits.size], bits.astype(np.int64))


def _optimal_cost(counts):
    import heapq
    h = list(counts)
    heapq.heapify(h)
    cost = 0
    while len(h) > 1:
        a, b = heapq.heappop(h), heapq.heappop(h)
        cost += a + b
        heapq.heappush(h, a + b)
    return cost


@pytest.mark.parametrize("n,spread", [(1, 1), (2, 1), (5000, 3), (40000, 60), (3000, 257)])
def test_huffman_round_trip_and_optimal_length(n, spread):
    rng = np.random.default_rng(n + spread)
    sym = np.clip(np.round(rng.normal(128, spread / 4, n)), 128 - spread // 2, 128 + spread // 2).astype(np.int64) if spread > 1 \
        else np.full(n, 7, np.int64)
    code = mc.HuffmanCode.from_data(sym)
    data = code.encode(sym)
    assert np.array_equal(code.decode(data, n), sym)
    counts = Counter(sym.tolist())
    bits = sum(code.lengths[s] * c for s, c in counts.items())
    assert len(data) == (bits + 7) // 8
    if len(counts) > 1:
        assert bits == _optimal_cost(counts.values())            # a Huffman code is optimal among prefix codes
        assert sum(2.0 ** -l for l in code.lengths.values()) == 1.0   # complete (Kraft equality)
    # the decoder needs the lengths only
    assert np.array_equal(mc.HuffmanCode(dict(code.lengths)).decode(data, n), sym)


def _tiny_model():
    from gsvc_amd.arguments import ModelParams
    from gsvc_amd.model import GaussianModel
    torch.manual_seed(3)
    return GaussianModel(ModelParams(), feat_dim=8, n_offsets=4, voxel_size=0.001, update_depth=3, update_init_factor=16,
                         update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=2, log2_hashmap_size=9,
                         log2_hashmap_size_2D=11, resolutions_list=(18, 24, 33), resolutions_list_2D=(130, 258), device="cpu")


def test_model_round_trip(tmp_path):
    pc = _tiny_model()
    before = {k: v.clone() for k, v in pc.state_dict().items() if k.startswith("mlp")}
    masks, quants, metas = mc.quantize_model(pc, replace=True)
    assert [m["key"] for m in metas] == list(before)
    after = {k: v.clone() for k, v in pc.state_dict().items() if k.startswith("mlp")}
    for k in before:     # 8 bits per row / per bias: within half a step of the original
        step = (before[k].max() - before[k].min()) / 256
        assert (after[k] - before[k]).abs().max() <= 0.51 * step + 1e-9
    path = str(tmp_path / "mlp.b")
    bits = mc.encode_mlp(pc, path)
    assert bits == os.path.getsize(path) * 8
    n_weights = sum(v.numel() for v in before.values())
    # <= ~8 bits per weight + the per-row (t_min, scale) pairs, which weigh in on a model this small (40 k weights, 2 k rows)
    assert bits < 12 * n_weights
    dec = mc.decode_mlp(path)
    assert list(dec) == list(after)
    for k in after:
        assert torch.equal(dec[k], after[k]), k
    # quantising the de-quantised model again changes nothing (the values sit on their own grid)
    pc.load_state_dict({**pc.state_dict(), **dec})
    mc.quantize_model(pc, replace=True)
    for k in after:
        assert torch.allclose(pc.state_dict()[k], after[k], rtol=0, atol=float(after[k].abs().max()) * 1e-6)


def test_corrupt_container_is_rejected(tmp_path):
    pc = _tiny_model()
    mc.quantize_model(pc)
    path = str(tmp_path / "mlp.b")
    mc.encode_mlp(pc, path)
    blob = open(path, "rb").read()
    bad = tmp_path / "bad.b"
    bad.write_bytes(b"XXXXXX" + blob[6:])
    with pytest.raises(ValueError):
        mc.decode_mlp(str(bad))
    bad.write_bytes(blob[:-5])
    with pytest.raises(ValueError):
        mc.decode_mlp(str(bad))


def test_huffman_code_lengths_are_limited():
    """A histogram skewed enough for unlimited Huffman depths beyond the decoder's table (Fibonacci-like counts: depth = number of
    symbols - 1) still yields a prefix code of at most MAX_LEN bits that round-trips; an unskewed one keeps its optimal lengths."""
    import numpy as np
    from gsvc_amd.mlp_codec import HuffmanCode
    fib = [1, 1]
    while len(fib) < 30:
        fib.append(fib[-1] + fib[-2])
    data = np.concatenate([np.full(c, i, np.int64) for i, c in enumerate(fib)])
    np.random.default_rng(0).shuffle(data)
    code = HuffmanCode.from_data(data)
    assert max(code.lengths.values()) <= HuffmanCode.MAX_LEN == 16
    assert sum(2.0 ** -l for l in code.lengths.values()) <= 1.0 + 1e-12
    out = code.decode(code.encode(data), data.size)
    assert np.array_equal(out, data)
    # the limit costs little even here (Fibonacci counts are the worst case for depth): within 5 % of the entropy
    bits = sum(code.lengths[int(s)] for s in data)
    ent = -sum(c * np.log2(c / data.size) for c in fib)
    assert bits <= 1.05 * ent + 8
    with pytest.raises(ValueError):
        code.encode(np.array([99]))
    with pytest.raises(ValueError):
        HuffmanCode({0: 1, 1: 1, 2: 1})
