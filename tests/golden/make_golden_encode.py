#!/usr/bin/env python3
"""What the reference's stream encoder hands to its entropy coders: ``GaussianModel.conduct_stream_encoding``
(scene/gaussian_model.py:2313-2604) run UNMODIFIED on PyTorch-CPU on the production-dimension model of the other fixtures, with spies
in the slots of the three external packages it calls — ``gsvc_cuda_ans.ANSCoder`` (attribute streams), ``torchac`` (binary streams)
and the G-PCC executable behind ``encode_anchor`` (returns the (x, y, z) order a geometry codec hands the points back in) — and the
MLP quantisation / Huffman stage switched off (pinned separately: mlp_quant.npz).  Recorded per z-slab and attribute: the symbol
range, the integer symbols and the model (mu, sigma) the coder would code them under; the binary streams' symbols and probabilities;
slab ranges and counts — and, from the decoder's half (conduct_stream_decoding, :2625-2804, whose coder slots hand those symbols
back), the reconstructed per-anchor tensors.  Our encoder must feed ITS coder the same numbers and our decoder must rebuild the same
model (tests/test_codec_gpu.py).  Every 7th symbol + float64
sums are stored.  Build container only.  Usage: python tests/golden/make_golden_encode.py"""
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import _ref_import, seeded  # noqa: E402
from tests.golden.make_golden_common import save  # noqa: E402

STRIDE = 7


def main():
    mode_ctx = _ref_import.install()
    calls = []

    class SpyANS:
        def __init__(self, lo, hi):
            self.lo, self.hi = int(lo), int(hi)

        def encode(self, file_name, symbols, mu, sigma):
            calls.append((os.path.basename(file_name), self.lo, self.hi, symbols.clone(), mu.clone(), sigma.clone()))
            with open(file_name, "wb") as f:
                f.write(b"\0" * 8)
            return b""

        def decode(self, file_name, mu, sigma):          # the decoder's half: hands back what the encoder was given for this file
            name = os.path.basename(file_name)
            sym = next(c[3] for c in calls if c[0] == name)
            assert sym.numel() == mu.numel() == sigma.numel()
            decode_models.append((name, mu.detach().cpu().clone().reshape(-1), sigma.detach().cpu().clone().reshape(-1)))
            return sym.to(torch.float32)

    binary = {}
    decode_models = []
    decoded_binary = [0]

    def decode_float_cdf(cdf, byte_stream):            # binary streams come back in the order they are asked for: masks, then hash
        k = {0: 1, 1: 0}[decoded_binary[0]]
        decoded_binary[0] += 1
        sym = binary[k][1]
        assert sym.numel() == cdf.reshape(-1, 3).shape[0]
        return sym.clone()

    def encode_float_cdf(cdf, sym, check_input_bounds=True):
        binary[len(binary)] = (cdf.clone(), sym.clone())
        return b"\0" * 8

    sys.modules["gsvc_cuda_ans"].ANSCoder = SpyANS
    sys.modules["torchac"].encode_float_cdf = encode_float_cdf
    sys.modules["torchac"].decode_float_cdf = decode_float_cdf
    with mode_ctx:
        import arguments as A
        import scene.gaussian_model as GM
        import utils.encodings as E
        E.ANSCoder = SpyANS
        E.torchac.encode_float_cdf = encode_float_cdf
        E.torchac.decode_float_cdf = decode_float_cdf

        geometry = {}

        def encode_anchor(q_anchor, tmp_path, tmc3_path):      # the order a geometry codec returns the points in: (x, y, z)
            order = np.lexsort((q_anchor[:, 2], q_anchor[:, 1], q_anchor[:, 0]))
            geometry["decoded"] = q_anchor[order].astype(np.float32)
            return order, 8 * 1234
        GM.encode_anchor = encode_anchor
        GM.decode_anchor = lambda tmp_path, tmc3_path: geometry["decoded"]
        sc, P = seeded.SCENE, seeded.PROD
        fn = seeded.frame_numbers(sc["H"], sc["W"], sc["T"], sc["frame"])
        mp = A.ModelParams()
        mp.threshold = sc["threshold"]
        torch.manual_seed(0)
        ref = GM.GaussianModel(mp, feat_dim=P["feat_dim"], n_offsets=P["n_offsets"], voxel_size=0.001, update_depth=3, update_init_factor=16,
                               update_hierachy_factor=4, use_feat_bank=False, n_features_per_level=P["n_features_per_level"],
                               log2_hashmap_size=P["log2_hashmap_size"], log2_hashmap_size_2D=P["log2_hashmap_size_2D"],
                               resolutions_list=P["resolutions_list"], resolutions_list_2D=P["resolutions_list_2D"])
        ref.update_anchor_bound(fn["x_min"], fn["y_min"], fn["z_min"])
        for name, t in seeded.anchors_uniform(sc["A"], fn, sc["seed"]).items():
            setattr(ref, name, nn.Parameter(t, requires_grad=name not in ("_rotation", "_opacity")))
        seeded.fill_parameters(ref, sc["seed"])
        ref.quantize_model = lambda replace=True: ([], [], [])          # the 8-bit MLP stage is pinned by mlp_quant.npz
        ref.encode_mlp = lambda path: 0
        import copy
        with tempfile.TemporaryDirectory() as tmp:
            meta, prob_hash, prob_masks, bit_info = ref.conduct_stream_encoding(tmp, SimpleNamespace(tmc3_executable="tmc3"))
            # the decoder's half on a copy of the model (reference scene/gaussian_model.py:2625-2804): its coder slots hand back the
            # symbols recorded above, so what it reconstructs is what a lossless coder pair would give
            dec = copy.deepcopy(ref)
            dec.conduct_stream_decoding(tmp, meta, prob_hash, prob_masks, "tmc3")
        out = {"meta::anchor_num": np.int64(meta.anchor_num), "meta::total_anchor_num": np.int64(meta.total_anchor_num),
               "meta::prob_hash": np.float64(prob_hash), "meta::prob_masks": np.float64(prob_masks), "meta::stride": np.int64(STRIDE),
               "meta::n_calls": np.int64(len(calls))}
        for i, (name, lo, hi, sym, mu, sg) in enumerate(calls):
            pre = f"call{i}::"
            out[pre + "name"] = np.array(name)
            out[pre + "range"] = np.array([lo, hi, sym.numel()], dtype=np.int64)
            out[pre + "symbols"] = sym.numpy().astype(np.int32)[::STRIDE]
            out[pre + "mu"] = mu.numpy()[::STRIDE]
            out[pre + "sigma"] = sg.numpy()[::STRIDE]
            out[pre + "sums"] = np.array([float(sym.double().sum()), float(sym.double().abs().sum()), float(mu.double().sum()),
                                          float(sg.double().sum())])
        for i, (cdf, sym) in binary.items():
            out[f"binary{i}::p_zero"] = np.float64(cdf.reshape(-1, 3)[0, 1])
            out[f"binary{i}::n"] = np.int64(sym.numel())
            out[f"binary{i}::ones"] = np.int64(int((sym > 0).sum()))
            out[f"binary{i}::bits"] = np.packbits((sym.reshape(-1)[:4096] > 0).numpy())
        assert len(decode_models) == len(calls) and decoded_binary[0] == 2
        for nm in ("_anchor", "_anchor_feat", "_offset", "_scaling", "_mask"):
            t = getattr(dec, nm).detach()
            out["decoded::" + nm] = t[::5]
            out["decoded_sum::" + nm] = np.array([float(t.double().sum()), float(t.double().abs().sum())])
        tables = dec.get_encoding_params()
        out["decoded::hash_ones"] = np.int64(int((tables > 0).sum()))
        out["decoded::decoded_version"] = np.bool_(dec.decoded_version)
        print(len(calls), "coder calls;", meta.anchor_num, "anchors coded;", len(binary), "binary streams")
        save("stream_encode", **out)


if __name__ == "__main__":
    main()
