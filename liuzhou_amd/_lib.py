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


HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "liuzhou_hip.h")

def _int_param(ctype, lo: int, hi: int):
    """Argument converter for an integer parameter of exact C width: accepts Python ints and any ctypes integer
    (call sites wrap sizes in c_int64 / c_int), rejects values the C type cannot hold instead of truncating them."""
    class _P(ctype):
        @classmethod
        def from_param(cls, v):
            v = int(getattr(v, "value", v))
            if not lo <= v <= hi:
                raise OverflowError(f"{v} does not fit {ctype.__name__}")
            return ctype(v)
    _P.__name__ = f"checked_{ctype.__name__}"
    return _P


def _float_param(ctype):
    class _P(ctype):
        @classmethod
        def from_param(cls, v):
            return ctype(float(getattr(v, "value", v)))
    _P.__name__ = f"checked_{ctype.__name__}"
    return _P


_I32 = _int_param(C.c_int32, -(1 << 31), (1 << 31) - 1)
_SCALARS = {"int": _I32, "int32_t": _I32, "int64_t": _int_param(C.c_int64, -(1 << 63), (1 << 63) - 1),
            "uint64_t": _int_param(C.c_uint64, 0, (1 << 64) - 1), "uint32_t": _int_param(C.c_uint32, 0, (1 << 32) - 1),
            "float": _float_param(C.c_float), "double": _float_param(C.c_double)}


def _parse_header(path: str):
    """`LZ_API <ret> name(args);` declarations of include/liuzhou_hip.h -> {name: (restype, [argtypes])}.  Every pointer
    is a c_void_p (callers pass tensor addresses / byref(struct)); scalars keep their exact C width, so a Python int
    of the wrong size is converted or rejected by ctypes instead of being silently truncated at the call."""
    import re
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    out = {}
    for m in re.finditer(r"LZ_API\s+(const\s+char\s*\*|int)\s+(lz_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(C.c_void_p)
                    continue
                base = [t for t in a.replace("const", " ").split() if t in _SCALARS]
                if not base:
                    raise RuntimeError(f"liuzhou_hip.h: cannot map argument '{a}' of {name}")
                types.append(_SCALARS[base[0]])
        out[name] = (C.c_char_p if "char" in ret else C.c_int, types)
    return out


DECLS = _parse_header(HEADER)
SYMBOLS = tuple(sorted(DECLS))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = os.environ.get("LZ_HIP_LIB", LIB)          # kernel experiments: another build of the same HIP library
        if not os.path.exists(path):
            raise RuntimeError(
                f"liuzhou_amd: HIP extension {path} is missing. Build it with "
                "`python -m liuzhou_amd.build` (hipcc --offload-arch=gfx950); there is no CPU fallback.")
        L = C.CDLL(path)
        for name, (restype, argtypes) in DECLS.items():
            fn = getattr(L, name)                          # AttributeError: header and library disagree
            fn.restype, fn.argtypes = restype, argtypes
        _lib = L
    return _lib


_host: Optional[C.CDLL] = None
HOST_SYMBOLS = ("lz_version", "lz_status_string", "lz_encode_actions_fast", "lz_batch_apply_moves",
                "lz_batch_apply_moves_inplace", "lz_states_to_model_input", "lz_project_policy_logits_fast",
                "lz_root_pack_rows", "lz_root_pack_plan", "lz_root_pack_fill", "lz_root_puct_allocate_visits",
                "lz_root_finalize_from_visits", "lz_self_play_step_inplace", "lz_finalize_trajectory_inplace")
_STATUS = {0: "ok", -1: "invalid argument", -2: "unsupported dimensions", -3: "kernel launch failed",
           -4: "misaligned pointer", -5: "illegal action for the state"}


def host_lib() -> C.CDLL:
    """libliuzhou_host.so: the operator subset of the same C ABI over HOST memory (csrc/lz_host.cpp, g++).  Only the
    `v0_core` operators dispatch here, for CPU tensors, like the reference extension (fast_legal_mask.cpp:453)."""
    global _host
    if _host is None:
        from .build import HOST_LIB, build_host
        path = os.environ.get("LZ_HOST_LIB") or (HOST_LIB if os.path.exists(HOST_LIB) else build_host())   # LZ_HOST_LIB: a sanitizer build
        H = C.CDLL(path)
        for name in HOST_SYMBOLS:
            fn = getattr(H, name)
            fn.restype, fn.argtypes = DECLS[name]
        _host = H
    return _host


def lib_for(t: torch.Tensor) -> C.CDLL:
    """Device dispatch of a v0_core operator: HIP tensors -> the gfx950 kernels, CPU tensors -> the host build.  There is
    no fallback in either direction: a missing library is an error."""
    return lib() if t.is_cuda else host_lib()


class device_ctx:
    """`torch.cuda.device(dev)` for HIP tensors, nothing for CPU tensors."""

    def __init__(self, device) -> None:
        self._cm = torch.cuda.device(device) if torch.device(device).type == "cuda" else None

    def __enter__(self):
        return self._cm.__enter__() if self._cm is not None else None

    def __exit__(self, *exc):
        return self._cm.__exit__(*exc) if self._cm is not None else False


def check(status: int, op: str) -> None:
    if status != 0:
        raise RuntimeError(f"liuzhou_amd.{op} failed: {_STATUS.get(int(status), 'unknown status')} ({status})")


def require_hip(t: torch.Tensor, op: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"liuzhou_amd.{op}: tensors must live on a HIP device (got {t.device}); CUDA kernels were not built "
            "for CPU -- this build has no CPU path.")


def stream_ptr(device: torch.device) -> C.c_void_p:
    if torch.device(device).type != "cuda":
        return C.c_void_p(None)                            # host build: no stream
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
