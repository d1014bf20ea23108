"""Pack a ChessNet (eval mode) for the fused gfx950 forward kernel (csrc/lz_net.hip).

* BatchNorm is folded: stem_bn and every block's bn2 go into the preceding conv (scale into the
  weights, shift into a bias); bn1 / trunk_bn stay as per-channel affines applied when the fp32
  residual stream is converted to the fp16 conv input (pre-activation blocks,
  src/neural_network.py:82-95).
* 3x3 conv weights are laid out in the exact operand order of `v_mfma_f32_16x16x32_f16` with the
  weights as the A operand: for layer / tap / 32-channel K block / 16-channel output tile, lane l
  holds W[co = 16*ct + (l & 15)][ci = 32*kb + 8*(l >> 4) + j], j = 0..7 (one 16-byte load per lane).
* The two 1x1 head convs (policy conv1, value conv1, both BN-folded) are stacked into one 128-wide
  layer in the same format.

`emulate(pack, planes)` re-computes the forward pass from the *packed* buffers in plain torch; it is
the CPU check that folding + fragment order are right (tests/test_net_pack.py).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch

HEAD_CH = 64          # policy_channels == value_channels (src/neural_network.py:213-246 defaults)
MLP_CH = 128
BINS = 101
STEM_K = 32           # 11 input planes zero-padded to one 32-wide K block


def _bn_affine(bn) -> Tuple[torch.Tensor, torch.Tensor]:
    a = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    b = bn.bias.detach().double() - bn.running_mean.detach().double() * a
    return a, b


LO_SHIFT = 11        # the fp16 "low half" of a split operand is stored scaled by 2^LO_SHIFT (csrc/lz_net_x3.hip)


def _frag_conv(w: torch.Tensor, k_pad: int, lo: bool = False) -> torch.Tensor:
    """w [Co, Ci, kh, kw] (double) -> fp16 fragments [taps][kb][ct][64][8].
    `lo`: the fragments of the split-operand mode's LOW halves instead: fp16((w - fp16(w)) * 2^LO_SHIFT), same order."""
    co, ci, kh, kw = w.shape
    taps = kh * kw
    wp = torch.zeros((co, k_pad, kh, kw), dtype=torch.float64)
    wp[:, :ci] = w
    kb_n, ct_n = k_pad // 32, co // 16
    lane = torch.arange(64)
    row = lane & 15
    kbase = 8 * (lane >> 4)
    j = torch.arange(8)
    out = torch.empty((taps, kb_n, ct_n, 64, 8), dtype=torch.float64)
    for t in range(taps):
        ky, kx = divmod(t, kw)
        for kb in range(kb_n):
            for ct in range(ct_n):
                cos = (16 * ct + row).view(64, 1).expand(64, 8)
                cis = (32 * kb + kbase.view(64, 1) + j.view(1, 8))
                out[t, kb, ct] = wp[cos, cis, ky, kx]
    if lo:
        return ((out - out.to(torch.float16).to(torch.float64)) * float(1 << LO_SHIFT)).to(torch.float16)
    return out.to(torch.float16)


def _frag_conv_f32(w: torch.Tensor, k_pad: int) -> torch.Tensor:
    """w [Co, Ci, kh, kw] (double) -> fp32 fragments [taps][k_pad/4][Co/16][64] in v_mfma_f32_16x16x4_f32 A-operand order:
    lane l holds W[co = 16*ct + (l & 15)][ci = 4*kb + (l >> 4)] (csrc/lz_net_f32.hip).  Same element count per layer as
    the fp16 fragments, so the fp16 layer offsets apply."""
    co, ci, kh, kw = w.shape
    wp = torch.zeros((co, k_pad, kh, kw), dtype=torch.float64)
    wp[:, :ci] = w
    lane = torch.arange(64)
    out = torch.empty((kh * kw, k_pad // 4, co // 16, 64), dtype=torch.float64)
    for t in range(kh * kw):
        ky, kx = divmod(t, kw)
        for kb in range(k_pad // 4):
            for ct in range(co // 16):
                out[t, kb, ct] = wp[16 * ct + (lane & 15), 4 * kb + (lane >> 4), ky, kx]
    return out.to(torch.float32)


@dataclass
class NetPack:
    channels: int
    blocks: int
    wfrag: torch.Tensor            # fp16, flat
    fparams: torch.Tensor          # fp32, flat
    layer_offsets: List[int]       # offsets (in halfs) of each conv layer in wfrag; last = heads
    foff: Dict[str, int]           # offsets (in floats) into fparams
    head_offsets: List[int] = None  # offsets (in halfs) of gpool_linear / fc1 / fc2 / out-conv fragments
    wfrag_f32: torch.Tensor = None  # optional fp32 conv fragments (parity mode), same element offsets as wfrag
    wfrag_lo: torch.Tensor = None   # optional fp16 LOW halves of the conv weights (split-operand mode), same offsets as wfrag

    def to(self, device) -> "NetPack":
        return NetPack(self.channels, self.blocks, self.wfrag.to(device), self.fparams.to(device),
                       list(self.layer_offsets), dict(self.foff), list(self.head_offsets),
                       None if self.wfrag_f32 is None else self.wfrag_f32.to(device),
                       None if self.wfrag_lo is None else self.wfrag_lo.to(device))


def pack_model(model, fp32_fragments: bool = False, lo_fragments: bool = False) -> NetPack:
    """Single-threaded wrapper of `_pack_model`: the packing is a few hundred small fp64 host operators, and torch's
    intra-op pool (one thread per core: 128 on the GPU box) costs more to wake per operator than the operator takes --
    measured 0.7 - 2.5 s with the pool, 0.09 s on one thread (`scripts/micro/worker_setup.py`); it was most of the
    worker's set-up time."""
    prev = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        return _pack_model(model, fp32_fragments, lo_fragments)
    finally:
        torch.set_num_threads(prev)


def _pack_model(model, fp32_fragments: bool = False, lo_fragments: bool = False) -> NetPack:
    """`fp32_fragments`: also lay the (BN-folded) conv weights out as fp32 MFMA fragments for the parity-mode kernel.
    `lo_fragments`: also the fp16 low halves of the conv weights, w = fp16(w) + lo * 2^-LO_SHIFT to ~22 bits, for the
    split-operand mode (csrc/lz_net_x3.hip: three fp16 MFMAs per product instead of one fp32 MFMA at 1/16 of the rate).
    BN folding + MFMA fragment order, on a detached host copy: the caller's module keeps its device, its train /
    eval mode and its parameter storages (optimizer state, DDP buckets and captured graphs keep pointing at them)."""
    import copy
    m = copy.deepcopy(model).to("cpu").eval()
    C = int(m.stem_conv.weight.shape[0])
    NB = len(m.blocks)
    if C % 32 != 0:
        raise ValueError(f"fused kernel needs trunk channels % 32 == 0 (got {C})")
    ph, vh = m.policy_head, m.value_head
    if int(ph.conv1.weight.shape[0]) != HEAD_CH or int(vh.conv1.weight.shape[0]) != HEAD_CH:
        raise ValueError("fused kernel needs policy/value channels == 64")
    if int(vh.fc1.weight.shape[0]) != MLP_CH or int(vh.fc2.weight.shape[0]) != BINS:
        raise ValueError("fused kernel needs value MLP 128 and 101 bins")

    frags: List[torch.Tensor] = []
    frags32: List[torch.Tensor] = []
    frags_lo: List[torch.Tensor] = []
    fl: List[torch.Tensor] = []

    def conv_layer(w: torch.Tensor, k_pad: int) -> None:
        frags.append(_frag_conv(w, k_pad))
        if fp32_fragments:
            frags32.append(_frag_conv_f32(w, k_pad))
        if lo_fragments:
            frags_lo.append(_frag_conv(w, k_pad, lo=True))
    foff: Dict[str, int] = {}

    def put(name: str, t: torch.Tensor) -> None:
        foff[name] = sum(int(x.numel()) for x in fl)
        fl.append(t.detach().double().reshape(-1))

    a, b = _bn_affine(m.stem_bn)
    conv_layer(m.stem_conv.weight.detach().double() * a.view(-1, 1, 1, 1), STEM_K)
    put("stem_bias", b)
    for i, blk in enumerate(m.blocks):
        a1, b1 = _bn_affine(blk.bn1)
        a2, b2 = _bn_affine(blk.bn2)
        put(f"b{i}_a1", a1); put(f"b{i}_b1", b1); put(f"b{i}_bias1", b2)
        conv_layer(blk.conv1.weight.detach().double() * a2.view(-1, 1, 1, 1), C)
        conv_layer(blk.conv2.weight.detach().double(), C)
    at, bt = _bn_affine(m.trunk_bn)
    put("trunk_a", at); put("trunk_b", bt)
    pa, pb = _bn_affine(ph.bn1)
    va, vb = _bn_affine(vh.bn1)
    head_w = torch.cat([ph.conv1.weight.detach().double() * pa.view(-1, 1, 1, 1),
                        vh.conv1.weight.detach().double() * va.view(-1, 1, 1, 1)], dim=0)      # [128, C, 1, 1]
    conv_layer(head_w, C)
    put("head_bias", torch.cat([pb, vb]))
    p2a, p2b = _bn_affine(ph.bn2)
    put("p_gw", ph.gpool_linear.weight)                  # [64,192]
    put("p_a2", p2a); put("p_b2", p2b)
    put("p_out", torch.cat([ph.out_pos1.weight.view(1, -1), ph.out_pos2.weight.view(1, -1),
                            ph.out_mark.weight.view(1, -1)], dim=0))                           # [3,64]
    put("v_w1", vh.fc1.weight); put("v_b1", vh.fc1.bias)  # [128,192],[128]
    put("v_w2", vh.fc2.weight); put("v_b2", vh.fc2.bias)  # [101,128],[101]
    # transposed copies: consecutive threads (output index) read consecutive addresses in the kernel
    put("p_gwT", ph.gpool_linear.weight.detach().t().contiguous())    # [192,64]
    put("v_w1T", vh.fc1.weight.detach().t().contiguous())            # [192,128]
    put("v_w2T", vh.fc2.weight.detach().t().contiguous())            # [128,101]
    pad = (-sum(int(x.numel()) for x in fl)) % 4
    if pad:
        fl.append(torch.zeros(pad, dtype=torch.float64))

    # head dense layers as 1x1 "convs" in the same fragment format (rows padded to a multiple of 16)
    def fc_frag(w: torch.Tensor, rows_pad: int) -> torch.Tensor:
        wp = torch.zeros((rows_pad, w.shape[1]), dtype=torch.float64)
        wp[: w.shape[0]] = w.detach().double()
        return _frag_conv(wp.view(rows_pad, w.shape[1], 1, 1), int(w.shape[1]))

    head_frags = [fc_frag(ph.gpool_linear.weight, 64), fc_frag(vh.fc1.weight, 128), fc_frag(vh.fc2.weight, 112),
                  fc_frag(torch.cat([ph.out_pos1.weight.view(1, -1), ph.out_pos2.weight.view(1, -1),
                                     ph.out_mark.weight.view(1, -1)], dim=0), 16)]
    offsets, pos = [], 0
    for f in frags:
        offsets.append(pos)
        pos += int(f.numel())
    head_offsets = []
    for f in head_frags:
        head_offsets.append(pos)
        pos += int(f.numel())
    frags = frags + head_frags
    wfrag = torch.cat([f.reshape(-1) for f in frags]).contiguous()
    fparams = torch.cat(fl).to(torch.float32).contiguous()
    wf32 = torch.cat([f.reshape(-1) for f in frags32]).contiguous() if fp32_fragments else None
    wlo = torch.cat([f.reshape(-1) for f in frags_lo]).contiguous() if lo_fragments else None     # conv layers only
    return NetPack(C, NB, wfrag, fparams, offsets, foff, head_offsets, wf32, wlo)


# ------------------------------------------------------------------------------------------------
# torch emulation of the packed forward (CPU check of folding / fragment order)
# ------------------------------------------------------------------------------------------------
def _unfrag(frag: torch.Tensor, taps: int, k_pad: int, co: int) -> torch.Tensor:
    """inverse of _frag_conv -> [Co, k_pad, taps] float32"""
    f = frag.view(taps, k_pad // 32, co // 16, 64, 8).to(torch.float32)
    w = torch.zeros((co, k_pad, taps), dtype=torch.float32)
    lane = torch.arange(64)
    row = lane & 15
    kbase = 8 * (lane >> 4)
    j = torch.arange(8)
    for t in range(taps):
        for kb in range(k_pad // 32):
            for ct in range(co // 16):
                cos = (16 * ct + row).view(64, 1).expand(64, 8)
                cis = 32 * kb + kbase.view(64, 1) + j.view(1, 8)
                w[cos, cis, t] = f[t, kb, ct]
    return w


def emulate(pack: NetPack, planes: torch.Tensor, half_activations: bool = True):
    """planes f32 [N,11,6,6] -> (lp1, lp2, lpmc [N,36], value_logits [N,101], value [N])"""
    import torch.nn.functional as F
    C, NB = pack.channels, pack.blocks
    fp = pack.fparams.cpu()
    wf = pack.wfrag.cpu()

    def P(name, n):
        o = pack.foff[name]
        return fp[o:o + n]

    def q(x):   # conv inputs are rounded to fp16 in the kernel
        return x.to(torch.float16).to(torch.float32) if half_activations else x

    def conv(x, layer, k_pad, co, taps):
        n_h = taps * (k_pad // 32) * (co // 16) * 512
        w = _unfrag(wf[pack.layer_offsets[layer]:pack.layer_offsets[layer] + n_h], taps, k_pad, co)
        ci = x.shape[1]
        ks = 3 if taps == 9 else 1
        return F.conv2d(q(x), w[:, :ci].reshape(co, ci, ks, ks), padding=ks // 2)

    x = planes.to(torch.float32)
    x = torch.relu(conv(x, 0, STEM_K, C, 9) + P("stem_bias", C).view(1, C, 1, 1))
    for i in range(NB):
        t = torch.relu(x * P(f"b{i}_a1", C).view(1, C, 1, 1) + P(f"b{i}_b1", C).view(1, C, 1, 1))
        u = torch.relu(conv(t, 1 + 2 * i, C, C, 9) + P(f"b{i}_bias1", C).view(1, C, 1, 1))
        x = x + conv(u, 2 + 2 * i, C, C, 9)
    h = torch.relu(x * P("trunk_a", C).view(1, C, 1, 1) + P("trunk_b", C).view(1, C, 1, 1))
    hv = torch.relu(conv(h, 1 + 2 * NB, C, 2 * HEAD_CH, 1) + P("head_bias", 2 * HEAD_CH).view(1, -1, 1, 1))
    hv = q(hv)   # head activations are staged through LDS in fp16
    pmap, vmap = hv[:, :HEAD_CH], hv[:, HEAD_CH:]

    def gpool(z):
        f = z.flatten(2)
        return torch.cat([f.mean(2), f.amax(2), torch.sqrt(f.var(2, unbiased=False) + 1e-6)], dim=1)

    g = gpool(pmap) @ P("p_gw", HEAD_CH * 192).view(HEAD_CH, 192).t()
    p2 = torch.relu((pmap + g[:, :, None, None]) * P("p_a2", HEAD_CH).view(1, -1, 1, 1) + P("p_b2", HEAD_CH).view(1, -1, 1, 1))
    wo = P("p_out", 3 * HEAD_CH).view(3, HEAD_CH)
    logits = torch.einsum("nchw,kc->nkhw", p2, wo).flatten(2)          # [N,3,36]
    lp = torch.log_softmax(logits, dim=2)
    hid = torch.relu(gpool(vmap) @ P("v_w1", MLP_CH * 192).view(MLP_CH, 192).t() + P("v_b1", MLP_CH))
    vl = hid @ P("v_w2", BINS * MLP_CH).view(BINS, MLP_CH).t() + P("v_b2", BINS)
    pr = torch.softmax(vl, dim=1)
    val = (pr * torch.linspace(-1.0, 1.0, BINS)).sum(1)
    return lp[:, 0], lp[:, 1], lp[:, 2], vl, val
