"""ctypes binding of libliuzhou_hip.so (the C ABI declared in include/liuzhou_hip.h).

There is NO CPU fallback: if the library is missing or a tensor is not on a HIP device the
operators raise.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from .build import LIB

_lib: Optional[C.CDLL] = None


class LzStateSoA(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "board", "marks_black", "marks_white", "phase", "current_player",
        "pending_marks_required", "pending_marks_remaining",
        "pending_captures_required", "pending_captures_remaining",
        "forced_removals_done", "move_count", "moves_since_capture")]


SYMBOLS = (
    "lz_version", "lz_status_string", "lz_encode_actions_fast", "lz_batch_apply_moves",
    "lz_batch_apply_moves_inplace", "lz_states_to_model_input", "lz_project_policy_logits_fast",
    "lz_root_pack_rows", "lz_root_puct_allocate_visits", "lz_root_finalize_from_visits",
    "lz_self_play_step_inplace", "lz_finalize_trajectory_inplace", "lz_net_forward_f16", "lz_net_forward_packed_f16", "lz_net_configure",
    "lz_pack_states", "lz_packed_to_model_input", "lz_tree_begin", "lz_tree_select", "lz_tree_expand",
    "lz_tree_finish", "lz_tree_search", "lz_tree_advance", "lz_tree_search_continue", "lz_policy_value_loss_fwd_bwd", "lz_pack_trajectory_rows", "lz_unpack_trajectory_rows", "lz_prof_enable", "lz_prof_net_summary", "lz_prof_net_busy", "lz_net_forward_packed_counted_f16", "lz_root_prepare", "lz_root_collect", "lz_wave_record", "lz_wave_step_finish", "lz_wave_reseat", "lz_tree_wave_select", "lz_tree_wave_expand", "lz_tree_search_waves",
)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("LZ_HIP_LIB", LIB)          # kernel experiments: another build of the same HIP library
        if not os.path.exists(path):
            raise RuntimeError(
                f"liuzhou_amd: HIP extension {path} is missing. Build it with "
                "`python -m liuzhou_amd.build` (hipcc --offload-arch=gfx950); there is no CPU fallback.")
        L = C.CDLL(path)
        L.lz_version.restype = C.c_char_p
        L.lz_status_string.restype = C.c_char_p
        L.lz_status_string.argtypes = [C.c_int]
        _lib = L
    return _lib


def check(status: int, op: str) -> None:
    if status != 0:
        raise RuntimeError(f"liuzhou_amd.{op} failed: {lib().lz_status_string(int(status)).decode()} ({status})")


def require_hip(t: torch.Tensor, op: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"liuzhou_amd.{op}: tensors must live on a HIP device (got {t.device}); CUDA kernels were not built "
            "for CPU -- this build has no CPU path.")


def stream_ptr(device: torch.device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(None if t is None else t.data_ptr())


def i64(v: int) -> C.c_int64:
    return C.c_int64(int(v))


def soa(tensors: Sequence[torch.Tensor]) -> LzStateSoA:
    s = LzStateSoA()
    for (name, _), t in zip(LzStateSoA._fields_, tensors):
        setattr(s, name, t.data_ptr())
    return s
