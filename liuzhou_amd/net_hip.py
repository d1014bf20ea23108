"""Fused gfx950 network forward (csrc/lz_net.hip) behind a module-like callable.

`FusedNet(model)` packs a ChessNet once (BN folding + MFMA fragment order, net_pack.py) and then
evaluates `planes f32[N,11,6,6] -> (log_p1, log_p2, log_pmc, value_logits)` -- the same 4-tuple as
`ChessNet.forward` (src/neural_network.py:248-259) -- plus the bucket-expectation scalar in `.last_value`.
It is also usable as the `inference_engine` of V1RootMCTS (`.forward(inputs, n_valid)`), i.e. the seat the
reference gives to its TorchScript/CUDA-graph InferenceEngine (v0/src/net/inference_engine.cpp:58-202).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib as L
from .net_pack import NetPack, pack_model, BINS


MAX_LAYERS = 96                 # include/liuzhou_hip.h: LZ_NET_MAX_LAYERS (stem + 2 per block + head convs)
MAX_BLOCKS = (MAX_LAYERS - 2) // 2


class LzNetDesc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("blocks", C.c_int32), ("num_layers", C.c_int32), ("max_blocks", C.c_int32),
                ("wfrag", C.c_void_p), ("fparams", C.c_void_p), ("wfrag_bytes", C.c_int64), ("fparams_bytes", C.c_int64),
                ("layer_offsets", C.c_int32 * MAX_LAYERS),
                ("head_frag_offsets", C.c_int32 * 4)] + \
               [(n, C.c_int32) for n in ("off_stem_bias", "off_block0", "off_trunk_a", "off_trunk_b", "off_head_bias",
                                         "off_p_gwT", "off_p_a2", "off_p_b2", "off_p_out", "off_v_w1T", "off_v_b1",
                                         "off_v_w2T", "off_v_b2", "flags")] + \
               [("wfrag_f32", C.c_void_p), ("wfrag_f32_bytes", C.c_int64), ("wfrag_lo", C.c_void_p), ("wfrag_lo_bytes", C.c_int64)]


_configured = False


def fused_unsupported_reason(model) -> Optional[str]:
    """None when the fused kernel (csrc/lz_net.hip) is built for `model`'s shape, else what it is not built for.  The
    reference's ChessNet is generic in every one of these (src/neural_network.py:213-246); a worker or stage must ask
    before it constructs a `FusedNet` and otherwise evaluate with the module itself (`self_play_worker.pick_evaluator`)."""
    from .net_pack import HEAD_CH, MLP_CH
    try:
        c = int(model.stem_conv.weight.shape[0])
        nb = len(model.blocks)
        ph, vh = model.policy_head, model.value_head
        pc, vc = int(ph.conv1.weight.shape[0]), int(vh.conv1.weight.shape[0])
        mlp, bins = int(vh.fc1.weight.shape[0]), int(vh.fc2.weight.shape[0])
        planes = int(model.stem_conv.weight.shape[1])
    except AttributeError as exc:
        return f"not a ChessNet-shaped module ({exc})"
    if c not in (64, 128):
        return f"trunk channels {c} (kernels are instantiated for 64 and 128)"
    if nb > MAX_BLOCKS:
        return f"{nb} residual blocks (descriptor holds {MAX_BLOCKS})"
    if (pc, vc) != (HEAD_CH, HEAD_CH):
        return f"policy / value head channels {pc} / {vc} (kernel heads are {HEAD_CH} wide)"
    if mlp != MLP_CH or bins != BINS:
        return f"value MLP {mlp} / {bins} bins (kernel: {MLP_CH} / {BINS})"
    if planes != 11:
        return f"{planes} input planes (kernel stages 11)"
    return None


def fused_supported(model) -> bool:
    return fused_unsupported_reason(model) is None


class FusedNet:
    def __init__(self, model, device=None, max_blocks: int = 0, half_workgroups: bool = False,
                 wide_tiles: Optional[bool] = None, precision: str = "fp16") -> None:
        """`half_workgroups` (64 channels): 4-wave workgroups of 8 samples, two per CU (LzNetDesc.flags bit 0).
        `wide_tiles` (128 channels): 4-wave workgroups, one wave per SIMD with 4 channel tiles per wave -- half the LDS
        operand reads of the 8-wave shape (flags bit 1); None: env LZ_NET_WIDE (default off).
        `precision`: "fp16" = fp16 MFMA operands with fp32 accumulation (the reference's autocast inference mode);
        "fp32" = fp32 operands (csrc/lz_net_f32.hip, flags bit 2): the reference's fp32 forward within 1e-5 -- the parity
        mode, many times slower, honoured by every entry point including the captured search loops;
        "fp16x3" = split fp16 operands (round 6, flags bit 3): every operand as two fp16 numbers (22 bits), every product
        as three fp16 MFMAs -- within 1e-5 of the fp32 forward on every output like "fp32", several times faster."""
        global _configured
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("FusedNet needs a HIP device (no CPU path)")
        self.device = dev
        if precision not in ("fp16", "fp32", "fp16x3"):
            raise ValueError(f"precision must be fp16, fp32 or fp16x3, got {precision!r}")
        self.precision = precision
        why = fused_unsupported_reason(model)
        if why is not None:
            raise RuntimeError(f"fused network kernel does not cover this model: {why}")
        self.pack: NetPack = pack_model(model, fp32_fragments=precision == "fp32", lo_fragments=precision == "fp16x3").to(dev)
        d = LzNetDesc()
        d.channels, d.blocks = self.pack.channels, self.pack.blocks
        d.num_layers = len(self.pack.layer_offsets)
        d.max_blocks = int(max_blocks)
        if wide_tiles is None:
            wide_tiles = os.environ.get("LZ_NET_WIDE", "0").strip().lower() in ("1", "on", "true")
        d.flags = (1 if (half_workgroups and self.pack.channels == 64) else 0) | \
                  (2 if (wide_tiles and self.pack.channels == 128) else 0)
        if precision == "fp32":
            d.flags |= 4
            d.wfrag_f32 = self.pack.wfrag_f32.data_ptr()
            d.wfrag_f32_bytes = int(self.pack.wfrag_f32.numel()) * 4
        elif precision == "fp16x3":
            d.flags |= 8
            d.wfrag_lo = self.pack.wfrag_lo.data_ptr()
            d.wfrag_lo_bytes = int(self.pack.wfrag_lo.numel()) * 2
        d.wfrag, d.fparams = self.pack.wfrag.data_ptr(), self.pack.fparams.data_ptr()
        d.wfrag_bytes = int(self.pack.wfrag.numel()) * 2
        d.fparams_bytes = int(self.pack.fparams.numel()) * 4
        for i, o in enumerate(self.pack.layer_offsets):
            d.layer_offsets[i] = int(o)
        for i, o in enumerate(self.pack.head_offsets):
            d.head_frag_offsets[i] = int(o)
        f = self.pack.foff
        d.off_stem_bias, d.off_block0 = f["stem_bias"], f.get("b0_a1", 0)
        d.off_trunk_a, d.off_trunk_b, d.off_head_bias = f["trunk_a"], f["trunk_b"], f["head_bias"]
        d.off_p_gwT, d.off_p_a2, d.off_p_b2, d.off_p_out = f["p_gwT"], f["p_a2"], f["p_b2"], f["p_out"]
        d.off_v_w1T, d.off_v_b1, d.off_v_w2T, d.off_v_b2 = f["v_w1T"], f["v_b1"], f["v_w2T"], f["v_b2"]
        self.desc = d
        self.last_value: Optional[torch.Tensor] = None
        self.flops_per_eval = _flops(self.pack.channels, self.pack.blocks)
        if not _configured:
            with torch.cuda.device(dev):
                L.check(L.lib().lz_net_configure(), "net_configure")
            if int(L.lib().lz_net_desc_bytes()) != C.sizeof(LzNetDesc):
                raise RuntimeError(f"LzNetDesc layout mismatch: library {int(L.lib().lz_net_desc_bytes())} B, "
                                   f"binding {C.sizeof(LzNetDesc)} B (stale libliuzhou_hip.so?)")
            _configured = True

    def eval(self):
        return self

    def refresh(self, model) -> "FusedNet":
        """Re-pack `model`'s current weights INTO the existing device buffers (same architecture): descriptors, kernel
        arguments frozen in captured graphs and every `variant()` keep pointing at valid, now updated, memory.  The
        checkpoint hand-off of a training iteration (v1/train.py:978 writes `model_state_cpu.pt` for the workers)."""
        p = pack_model(model, fp32_fragments=self.precision == "fp32", lo_fragments=self.precision == "fp16x3")
        if (p.channels, p.blocks) != (self.pack.channels, self.pack.blocks) or p.wfrag.numel() != self.pack.wfrag.numel():
            raise ValueError("refresh() needs the architecture this FusedNet was built for")
        self.pack.wfrag.copy_(p.wfrag.to(self.device))
        self.pack.fparams.copy_(p.fparams.to(self.device))
        if p.wfrag_f32 is not None:
            self.pack.wfrag_f32.copy_(p.wfrag_f32.to(self.device))
        if p.wfrag_lo is not None:
            self.pack.wfrag_lo.copy_(p.wfrag_lo.to(self.device))
        return self

    def variant(self, half_workgroups: bool = False, wide_tiles: Optional[bool] = None) -> "FusedNet":
        """Same packed weights, other kernel configuration (a second descriptor over the same buffers)."""
        import copy
        other = copy.copy(self)
        d = LzNetDesc()
        C.memmove(C.byref(d), C.byref(self.desc), C.sizeof(LzNetDesc))
        wide = bool(self.desc.flags & 2) if wide_tiles is None else bool(wide_tiles)
        d.flags = (1 if (half_workgroups and self.pack.channels == 64) else 0) | \
                  (2 if (wide and self.pack.channels == 128) else 0) | (self.desc.flags & (4 | 8))
        other.desc = d
        other.last_value = None
        return other

    def forward_into(self, planes: torch.Tensor, lp1, lp2, lpm, vlogits, value) -> None:
        N = int(planes.shape[0])
        with torch.cuda.device(self.device):
            st = L.lib().lz_net_forward_f16(C.byref(self.desc), L.ptr(planes), L.i64(N), L.ptr(lp1), L.ptr(lp2),
                                            L.ptr(lpm), L.ptr(vlogits), L.ptr(value), L.stream_ptr(self.device))
        L.check(st, "net_forward_f16")

    def __call__(self, planes: torch.Tensor, want_logits: bool = True) -> Tuple[torch.Tensor, ...]:
        L.require_hip(planes, "net_forward_f16")
        x = planes if (planes.dtype == torch.float32 and planes.is_contiguous()) else planes.float().contiguous()
        N = int(x.shape[0])
        dev = x.device
        lp1 = torch.empty((N, 36), dtype=torch.float32, device=dev)
        lp2 = torch.empty((N, 36), dtype=torch.float32, device=dev)
        lpm = torch.empty((N, 36), dtype=torch.float32, device=dev)
        vl = torch.empty((N, BINS), dtype=torch.float32, device=dev) if want_logits else None
        val = torch.empty((N,), dtype=torch.float32, device=dev)
        self.forward_into(x, lp1, lp2, lpm, vl, val)
        self.last_value = val
        return lp1, lp2, lpm, vl

    def forward(self, inputs: torch.Tensor, n_valid: Optional[int] = None):
        out = self(inputs)
        return out

    def forward_packed(self, packed: torch.Tensor):
        """The same forward on 32-byte packed bitboard states int64[N,4] (lz_pack_states): the model-input encode is
        fused into the kernel's prologue.  Returns (log_p1, log_p2, log_pmc, None, value)."""
        L.require_hip(packed, "net_forward_packed_f16")
        x = packed.contiguous()
        N, dev = int(x.shape[0]), x.device
        lp1, lp2, lpm = (torch.empty((N, 36), dtype=torch.float32, device=dev) for _ in range(3))
        val = torch.empty((N,), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().lz_net_forward_packed_f16(C.byref(self.desc), L.ptr(x), L.i64(N), L.ptr(lp1), L.ptr(lp2),
                                                      L.ptr(lpm), None, L.ptr(val), L.stream_ptr(dev)),
                    "net_forward_packed_f16")
        return lp1, lp2, lpm, None, val

    def values_only(self, planes: torch.Tensor) -> torch.Tensor:
        """Scalar values only: the kernel skips the policy head and writes nothing but `value`."""
        L.require_hip(planes, "net_forward_f16")
        x = planes if (planes.dtype == torch.float32 and planes.is_contiguous()) else planes.float().contiguous()
        val = torch.empty((int(x.shape[0]),), dtype=torch.float32, device=x.device)
        self.forward_into(x, None, None, None, None, val)
        self.last_value = val
        return val


def _flops(C: int, NB: int) -> float:
    """Algorithmic FLOPs per evaluation: stem + trunk + heads."""
    stem = 2 * 36 * 9 * 11 * C
    trunk = NB * 2 * (2 * 36 * 9 * C * C)
    heads = 2 * 36 * C * 128 + 2 * (192 * 64 + 36 * 64 * 3 + 192 * 128 + 128 * 101)
    return float(stem + trunk + heads)
