"""Stream encoding / decoding of a fitted model: anchors ordered by z and cut into slabs, the
three attribute streams of every slab entropy-coded with the context model, the offset masks and the binarised hash
tables coded with their Bernoulli probability.

Follows reference scene/gaussian_model.py:2313-2804 (``conduct_stream_encoding`` / ``conduct_stream_decoding``) and
utils/encodings.py:827-862 (``reorder_and_split``) in what is coded, in which order and with which model; what differs:

* the entropy coder is this library's rANS (csrc/ans.hip) — the reference's ``gsvc_cuda_ans`` / ``torchac`` are external
  packages whose sources are not in its tree — and the streams live in one in-memory ``StreamPack`` (``save`` / ``load``
  write the reference's file names: ``feat_{s}.b``, ``scaling_{s}.b``, ``offsets_{s}.b``, ``masks.b``, ``hash.b``);
* binary symbols (masks, hash tables) go through the same coder as a two-symbol alphabet whose model is a unit-range
  Gaussian placed so that P(0) = 1 - p;
* anchor geometry: the reference hands the 16-bit anchor grid to MPEG G-PCC (``tmc3``, an external executable); here an own
  lossless occupancy-octree coder (gsvc_amd/anchor_codec.py) codes it, and the decoder gets the anchors back in the
  (x, y, z)-sorted order a geometry codec returns;
* MLP weights: 8-bit quantisation + Huffman code as the reference (gsvc_amd/mlp_codec.py; ``mlp_file=`` below) — the
  quantised networks are the ones that drive the context model of the attribute streams, on both sides.

Everything runs on the device except the byte containers.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn as nn

from .codec import DeferredChecks, ans_decode, ans_decode_many, ans_encode, decoder_gaussian_many, encoder_gaussian, prepare_streams
from . import anchor_codec
from .encodings import ANCHOR_ROUND_DIGITS, Quantize_anchor, STE_multistep
from .model import calc_symbol_min_max

BASE_Q = (1.0, 0.001, 0.2)      # feature, scaling, offsets (reference scene/gaussian_model.py:2352-2354)


def _lexsort(cols):
    """Indices that sort rows by cols[0], then cols[1], ... (stable sorts from the last key to the first)."""
    order = torch.arange(cols[0].shape[0], device=cols[0].device)
    for c in reversed(cols):
        order = order[torch.sort(c[order], stable=True).indices]
    return order


def reorder_and_split(anchor, interval=0.01):
    """Order anchors by (z, x, y) and cut the order into z slabs of ``interval`` (reference utils/encodings.py:827-862).
    Returns (selection, [(start, end), ...]).

    The slab boundaries are the reference's, bit for bit: it walks ``lb += interval; ub += interval`` on float32 scalars from
    ``lb = -ceil(|z_min| / interval) * interval`` while ``ub <= ceil(|z_max| / interval) * interval``, so boundary k is a float32
    ACCUMULATION (not k * interval) — which side of a boundary an anchor with z = 0.01 or 0.02 falls on depends on it — and an anchor
    belongs to the first slab whose ``ub`` exceeds its z.  Where the reference's walk leaves anchors out (its accumulated ``ub`` can
    pass the rounded maximum one slab early, and ub_k / lb_(k+1) can differ by an ulp) it silently drops them from every slab;
    here they join the slab that follows the gap (or the last one), so the slabs always cover all anchors.  Empty slabs (the
    reference assumes there are none) are left out."""
    sel = _lexsort([anchor[:, 2], anchor[:, 0], anchor[:, 1]])
    z = anchor[sel, 2]
    z_min, z_max = z.min().float().cpu(), z.max().float().cpu()
    assert z_min < 0 and z_max > 0
    lb = -torch.ceil(z_min.abs() / interval) * interval            # float32 scalars on the host, the reference's expressions
    z_max_round = torch.ceil(z_max.abs() / interval) * interval + 1e-10
    ub = lb + interval
    ubs = []
    while ub <= z_max_round:
        ubs.append(float(ub))
        lb += interval
        ub += interval
    if not ubs:
        ubs = [float(ub)]
    ub_t = torch.tensor(ubs, dtype=torch.float32, device=z.device)
    k = torch.searchsorted(ub_t, z.float().contiguous(), right=True).clamp_(max=len(ubs) - 1)      # non-decreasing along the z order
    change = torch.ones_like(k, dtype=torch.bool)
    change[1:] = k[1:] != k[:-1]
    starts = change.nonzero(as_tuple=False).squeeze(1).tolist()
    splits = [(a, b) for a, b in zip(starts, starts[1:] + [int(z.shape[0])])]
    return sel, splits


def _bernoulli_model(n, p_one, device):
    """Unit-range Gaussian whose mass below 1/2 is P(symbol 0) = 1 - p_one (symbols {0, 1}, range [0, 1])."""
    p0 = min(max(1.0 - float(p_one), 1e-6), 1.0 - 1e-6)
    z = float(torch.special.ndtri(torch.tensor(p0, dtype=torch.float64)))
    if abs(z) < 1e-9:
        mu, sigma = 0.5, 1.0                                   # p = 1/2
    elif z > 0:
        mu, sigma = 0.0, 0.5 / z                               # Phi((1/2 - 0) / sigma) = p0
    else:
        mu, sigma = 1.0, 0.5 / (-z)                            # Phi((1/2 - 1) / sigma) = Phi(z) = p0
    return torch.full((n,), mu, device=device), torch.full((n,), sigma, device=device)


def encode_binary(x01, p_one):
    """{0,1} tensor -> stream at ~H(p_one) bits per element (reference utils/encodings.py:265-287 `encode_binary`)."""
    sym = (x01.reshape(-1) > 0.5).to(torch.int32)      # get_mask's straight-through form can hold 0.99999994 for "one"
    mu, sigma = _bernoulli_model(sym.numel(), p_one, sym.device)
    return ans_encode(sym, mu, sigma, 0, 1)


def decode_binary(stream, n, p_one, device):
    mu, sigma = _bernoulli_model(n, p_one, device)
    return ans_decode(stream, mu, sigma)


@dataclass
class StreamPack:
    n_full: int
    n: int
    anchor_interval: np.ndarray
    anchor_min: np.ndarray
    anchors_q: np.ndarray                    # uint16 [n, 3], (x, y, z)-sorted (the decoded form of anchor_stream)
    prob_masks: float
    prob_hash: float
    slabs: list = field(default_factory=list)            # [(start, end)] in z order
    feat: list = field(default_factory=list)             # one stream per slab
    scaling: list = field(default_factory=list)
    offsets: list = field(default_factory=list)
    masks: bytes = b""
    hash: bytes = b""
    bit_mlp_encoded: int = None              # size of the MLP file written beside the streams (not part of meta.json)
    anchor_stream: bytes = b""               # gsvc_amd.anchor_codec: occupancy octree + rANS of the anchor geometry
    anchors_q_dev: object = None             # the decoded geometry as a device tensor (decode_anchors_gpu): spares the decoder an upload

    def bits(self):
        """Coded size per stream in bits (same keys as BitInfo where they exist)."""
        return {**({"bit_mlp_encoded": self.bit_mlp_encoded} if self.bit_mlp_encoded is not None else {}), "bit_anchor": 8 * len(self.anchor_stream) if self.anchor_stream else self.anchors_q.size * ANCHOR_ROUND_DIGITS,
                "bit_feat": 8 * sum(map(len, self.feat)),
                "bit_scaling": 8 * sum(map(len, self.scaling)), "bit_offsets": 8 * sum(map(len, self.offsets)),
                "bit_masks": 8 * len(self.masks), "bit_hash": 8 * len(self.hash)}

    def save(self, path):
        os.makedirs(path, exist_ok=True)
        for s in range(len(self.slabs)):
            for name, streams in (("feat", self.feat), ("scaling", self.scaling), ("offsets", self.offsets)):
                with open(os.path.join(path, f"{name}_{s}.b"), "wb") as f:
                    f.write(streams[s])
        for name, blob in (("masks.b", self.masks), ("hash.b", self.hash)):
            with open(os.path.join(path, name), "wb") as f:
                f.write(blob)
        with open(os.path.join(path, "anchor.b"), "wb") as f:      # in place of the reference's anchor_compressed.drc (G-PCC)
            f.write(self.anchor_stream if self.anchor_stream else anchor_codec.encode_anchors(self.anchors_q))
        meta = {"n_full": self.n_full, "n": self.n, "prob_masks": self.prob_masks, "prob_hash": self.prob_hash,
                "slabs": [list(s) for s in self.slabs],
                # float32 values survive the trip through JSON doubles exactly
                "anchor_interval": np.asarray(self.anchor_interval, np.float32).astype(np.float64).tolist(),
                "anchor_min": np.asarray(self.anchor_min, np.float32).astype(np.float64).tolist()}
        with open(os.path.join(path, "meta.json"), "w") as f:
            json.dump(meta, f)

    @classmethod
    def load(cls, path):
        with open(os.path.join(path, "meta.json")) as f:
            meta = json.load(f)
        meta["anchor_interval"] = np.asarray(meta["anchor_interval"], np.float32)
        meta["anchor_min"] = np.asarray(meta["anchor_min"], np.float32)
        meta["slabs"] = [tuple(s) for s in meta["slabs"]]
        rd = lambda name: open(os.path.join(path, name), "rb").read()  # noqa: E731
        stream = rd("anchor.b")
        dev_q = None
        if torch.cuda.is_available():      # entropy decode + octree expansion on the GPU (csrc/anchor.hip)
            dev_q = anchor_codec.decode_anchors_gpu(stream)
            anchors_q = dev_q.cpu().numpy().astype(np.uint16)
        else:
            anchors_q = anchor_codec.decode_anchors(stream)
        pack = cls(anchors_q=anchors_q, anchor_stream=stream, anchors_q_dev=dev_q, **meta)
        for s in range(len(pack.slabs)):
            pack.feat.append(rd(f"feat_{s}.b")); pack.scaling.append(rd(f"scaling_{s}.b")); pack.offsets.append(rd(f"offsets_{s}.b"))
        pack.masks, pack.hash = rd("masks.b"), rd("hash.b")
        return pack


CONTEXT_CHUNK = 1 << 20     # anchors per evaluation of the context model (bounds its transient memory)


def _context_all(pc, anchor):
    """Context of every kept anchor (z order): per-element mean / scale / step of the three attribute groups, evaluated in
    fixed chunks.  Encoder and decoder call THIS (same chunks, hence the same kernels on the same rows: the model must agree
    bit for bit); a slab's model is a row slice.  (The reference evaluates the context slab by slab, scene/gaussian_model.py:
    2380-2450; per row it is the same function of the anchor.)"""
    parts = [[] for _ in range(9)]
    for lo in range(0, anchor.shape[0], CONTEXT_CHUNK):
        ec = pc.calc_entropy_context(anchor[lo:lo + CONTEXT_CHUNK])
        Q = [BASE_Q[0] * ec.Q_feat_adj, BASE_Q[1] * ec.Q_scaling_adj, BASE_Q[2] * ec.Q_offsets_adj]
        means = [ec.mean_feat, ec.mean_scaling, ec.mean_offsets]
        scales = [ec.scale_feat, ec.scale_scaling, ec.scale_offsets]
        for g in range(3):
            parts[3 * g].append(means[g].contiguous())
            parts[3 * g + 1].append(scales[g].contiguous())
            parts[3 * g + 2].append(Q[g].repeat(1, means[g].shape[-1]))
    cat = [torch.cat(p) if len(p) != 1 else p[0] for p in parts]
    return [(cat[0], cat[1], cat[2]), (cat[3], cat[4], cat[5]), (cat[6], cat[7], cat[8])]


@torch.no_grad()
def conduct_stream_encoding(pc, mlp_file=None) -> StreamPack:
    """``mlp_file``: quantise the MLPs to 8 bits IN PLACE and write them there first (reference :2313-2317: the attribute
    streams are then coded under the quantised entropy networks, which is what the decoder will hold)."""
    K = pc.n_offsets
    bit_mlp_encoded = None
    if mlp_file is not None:
        from . import mlp_codec
        mlp_codec.quantize_model(pc, replace=True)
        bit_mlp_encoded = mlp_codec.encode_mlp(pc, mlp_file)
    keep = pc.get_mask_anchor
    q_anchor, interval, a_min = pc.quantized_anchor
    q_anchor = q_anchor[keep]
    # the order a geometry codec returns the points in: sorted by (x, y, z) (reference utils/encodings.py:753-757)
    sel = _lexsort([q_anchor[:, 0], q_anchor[:, 1], q_anchor[:, 2]])
    anchors_q = q_anchor[sel].to(torch.int32).cpu().numpy().astype(np.uint16)
    anchor = pc.get_anchor[keep][sel]
    feat, offsets = pc._anchor_feat[keep][sel], pc._offset[keep][sel]
    scaling, mask = pc.get_scaling[keep][sel], pc.get_mask[keep][sel]
    z_order, slabs = reorder_and_split(anchor)
    anchor, feat, offsets, scaling, mask = anchor[z_order], feat[z_order], offsets[z_order], scaling[z_order], mask[z_order]
    # symbol ranges from the context of ALL kept anchors (reference :2356-2364)
    ec = pc.calc_entropy_context(anchor)
    ranges = [calc_symbol_min_max(ec.mean_feat, BASE_Q[0] * ec.Q_feat_adj), calc_symbol_min_max(ec.mean_scaling, BASE_Q[1] * ec.Q_scaling_adj),
              calc_symbol_min_max(ec.mean_offsets, BASE_Q[2] * ec.Q_offsets_adj)]
    N = anchor.shape[0]
    tables = pc.get_encoding_params()                          # {-1, +1}
    prob_hash = float((((tables + 1) / 2).sum() / tables.numel()).item())
    prob_masks = float((mask.sum() / mask.numel()).item())
    # anchor geometry: occupancy octree over the voxel lattice the anchors sit on (lattice mode), falling back to the 16-bit grid
    anchor_stream = anchor_codec.encode_anchors(anchors_q, positions=pc._anchor[keep][sel].detach().cpu().numpy(),
                                                voxel_size=float(pc.voxel_size), interval=interval.cpu().numpy(), a_min=a_min.cpu().numpy())
    # the geometry must survive its own decoder: checked with the decoder that will run (on the device: csrc/anchor.hip, 3.6 ms at
    # 3.7 M anchors; the host decoder took 0.5 s of the 4K model's encode)
    if pc._anchor.is_cuda:
        decoded_dev = anchor_codec.decode_anchors_gpu(anchor_stream, pc._anchor.device)
        same = decoded_dev.shape[0] == anchors_q.shape[0] and bool(
            (decoded_dev.to(torch.int32) == q_anchor[sel].to(torch.int32)).all())
    else:
        same = np.array_equal(anchor_codec.decode_anchors(anchor_stream), anchors_q)
    if not same:
        raise RuntimeError("stream encoding: the anchor geometry does not survive its own decoder")
    pack = StreamPack(n_full=int(pc._anchor.shape[0]), n=N, anchor_interval=interval.cpu().numpy(), anchor_min=a_min.cpu().numpy(),
                      anchors_q=anchors_q, prob_masks=prob_masks, prob_hash=prob_hash, slabs=list(slabs), anchor_stream=anchor_stream)
    model = _context_all(pc, anchor)
    for a, b in slabs:
        (mf, sf, qf), (ms, ss, qs), (mo, so, qo) = [tuple(t[a:b] for t in grp) for grp in model]
        x = STE_multistep.quantize(feat[a:b], qf, *ranges[0])
        pack.feat.append(encoder_gaussian(x, mf, sf, qf, *ranges[0])[3])
        x = STE_multistep.quantize(scaling[a:b], qs, *ranges[1])
        pack.scaling.append(encoder_gaussian(x, ms, ss, qs, *ranges[1])[3])
        m3 = mask[a:b].repeat(1, 1, 3).view(-1, 3 * K).to(torch.bool)
        x = STE_multistep.quantize(offsets[a:b].view(-1, 3 * K), qo, *ranges[2])
        pack.offsets.append(encoder_gaussian(x[m3], mo[m3], so[m3], qo[m3], *ranges[2])[3] if bool(m3.any()) else b"")
    pack.masks = encode_binary(mask, prob_masks)
    pack.hash = encode_binary((tables + 1) / 2, prob_hash)
    pack.bit_mlp_encoded = bit_mlp_encoded
    return pack


@torch.no_grad()
def conduct_stream_decoding(pc, pack: StreamPack, mlp_file=None):
    """Replace the model's anchors, attributes, masks and hash tables by the decoded ones (``decoded_version`` = True);
    the MLP weights of ``pc`` must be the encoder's — given ``mlp_file`` they are read from it first."""
    dev, K = pc._anchor.device, pc.n_offsets
    if mlp_file is not None:
        from . import mlp_codec
        sd = pc.state_dict()
        for k, v in mlp_codec.decode_mlp(mlp_file).items():
            sd[k].copy_(v.to(sd[k].device))
    if pack.anchors_q_dev is not None and pack.anchors_q_dev.device == dev:
        q = pack.anchors_q_dev.to(torch.float32)
    else:
        q = torch.from_numpy(pack.anchors_q.astype(np.float32)).to(dev)
    anchor = Quantize_anchor.dequantized(q, torch.from_numpy(pack.anchor_interval).to(dev), torch.from_numpy(pack.anchor_min).to(dev))
    z_order, slabs = reorder_and_split(anchor)
    if list(slabs) != [tuple(s) for s in pack.slabs]:
        raise RuntimeError("stream decoding: the slab split of the decoded anchors differs from the encoder's")
    anchor = anchor[z_order]
    N = pack.n
    # launch 1: the offset masks and the binarised hash tables (two streams, one launch)
    tables = pc.get_encoding_params()
    checks = DeferredChecks()
    mu_m, sg_m = _bernoulli_model(N * K, pack.prob_masks, dev)
    mu_h, sg_h = _bernoulli_model(tables.numel(), pack.prob_hash, dev)
    mask_sym, hash_sym = ans_decode_many([(pack.masks, mu_m, sg_m), (pack.hash, mu_h, sg_h)], defer=checks)
    mask = mask_sym.to(torch.float32).view(N, K, 1)
    hash_pm = (hash_sym.to(torch.float32) * 2 - 1).view(-1, tables.shape[1])
    _install_tables(pc, hash_pm)                                  # the context model reads the decoded tables
    # the coded offsets are the masked ones: index lists per slab, one read-back
    n_slab = len(slabs)
    prepared = prepare_streams(list(pack.feat) + list(pack.scaling) + list(pack.offsets), dev)
    m3_all = mask.repeat(1, 1, 3).view(N, 3 * K) > 0
    nz = m3_all.view(-1).nonzero(as_tuple=False).squeeze(1)                      # flat positions of the coded offsets
    edges = torch.tensor([a * 3 * K for a, _ in slabs] + [N * 3 * K], device=dev)
    cuts = torch.searchsorted(nz, edges).tolist()
    # launch 2: every attribute stream of every slab.  The context of all anchors is evaluated first (one pass, the encoder's
    # chunks); a decode launch lasts as long as one lane's serial segment however many streams it carries, while one launch per
    # stream ran them one after another (24 x 1.8 ms: HIP streams share hardware queues with the context kernels in between).
    model = _context_all(pc, anchor)
    jobs, off_sel = [], []
    for s_i, (a, b) in enumerate(slabs):
        (mf, sf, qf), (ms, ss, qs), (mo, so, qo) = [tuple(t[a:b] for t in grp) for grp in model]
        sel = nz[cuts[s_i]:cuts[s_i + 1]] - a * 3 * K                              # positions inside this slab's [rows, 3K] block
        jobs.append((mf, sf, qf, prepared[s_i]))
        jobs.append((ms, ss, qs, prepared[n_slab + s_i]))
        if prepared[2 * n_slab + s_i] is None and cuts[s_i + 1] != cuts[s_i]:
            raise RuntimeError("stream decoding: a slab has coded offsets but an empty offsets stream")
        jobs.append((mo.reshape(-1)[sel], so.reshape(-1)[sel], qo.reshape(-1)[sel], prepared[2 * n_slab + s_i]))
        off_sel.append((sel, mo.shape))
    vals = decoder_gaussian_many(jobs, defer=checks)
    feats, scalings, offsets = [], [], []
    for s_i in range(n_slab):
        feats.append(vals[3 * s_i])
        scalings.append(vals[3 * s_i + 1])
        sel, shape = off_sel[s_i]
        off = torch.zeros(shape, device=dev)
        off.view(-1)[sel] = vals[3 * s_i + 2]
        offsets.append(off.view(-1, K, 3))
    checks.check()
    Nf = pack.n_full

    def full(rows, *shape):
        t = torch.zeros((Nf,) + shape, device=dev)
        t[:N] = rows
        return nn.Parameter(t)

    pc._anchor_feat = full(torch.cat(feats), pc.feat_dim)
    pc._offset = full(torch.cat(offsets), K, 3)
    pc.decoded_version = True
    pc._anchor = full(anchor, 3)
    pc._scaling = full(torch.cat(scalings), 6)
    pc._mask = full(mask, K, 1)
    return pc


def _install_tables(pc, hash_pm):
    enc = pc.encoding_xyz
    if pc.use_2D:
        parts = [enc.encoding_xyz, enc.encoding_xy, enc.encoding_xz, enc.encoding_yz]
    else:
        parts = [enc]
    at = 0
    for g in parts:
        n = g.params.shape[0]
        g.params = nn.Parameter(hash_pm[at:at + n].clone())
        at += n
