"""Build the gfx950 shared library (explicit hipcc, in-tree, no hipify, no JIT cache)."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libliuzhou_hip.so")
SOURCES = ("lz_ops.hip", "lz_engine.hip", "lz_net.hip", "lz_net_f32.hip", "lz_train.hip", "lz_search.hip")
HEADERS = ("lz_rules.h", "lz_soa.h", "lz_wave.h", "lz_rng.h", "lz_net_dev.h", "lz_tree_dev.h", os.path.join("..", "..", "include", "liuzhou_hip.h"))


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (expected /opt/rocm/bin/hipcc)")


HOST_LIB = os.path.join(PKG, "libliuzhou_host.so")
HOST_SOURCE = os.path.join(CSRC, "lz_host.cpp")
HOST_SOURCES = (HOST_SOURCE, os.path.join(CSRC, "lz_scalar.cpp"))       # the operator subset + the scalar rule surface


def build_host(force: bool = False, verbose: bool = False) -> str:
    """g++ -> liuzhou_amd/libliuzhou_host.so: the operator subset of the C ABI for CPU tensors (csrc/lz_host.cpp) and the
    scalar rule surface of include/liuzhou_scalar.h (csrc/lz_scalar.cpp)."""
    deps = list(HOST_SOURCES) + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(PKG, "..", "include", "liuzhou_scalar.h")]
    if not force and os.path.exists(HOST_LIB) and all(
            (not os.path.exists(d)) or os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in deps):
        return HOST_LIB
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no host C++ compiler (g++) found for libliuzhou_host.so")
    cmd = [cxx, "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden", "-o", HOST_LIB,
           *HOST_SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return HOST_LIB


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps += [os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> liuzhou_amd/libliuzhou_hip.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-fvisibility=hidden", "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


def build_variant(tag: str, defs, force: bool = True, verbose: bool = False) -> str:
    """An experiment build of the HIP library with extra preprocessor definitions (`-DLZ_EXP_...`, `-DLZ_NET_APF=2`):
    liuzhou_amd/_exp/liblz_<tag>.so, selected at run time with LZ_HIP_LIB.  Measurement aid (scripts/micro/*_ab.py run
    one child process per library); never loaded by default, git- ignored like every built object."""
    out = os.path.join(PKG, "_exp", f"liblz_{tag}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(LIB):
        return out
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-fvisibility=hidden", *[d if d.startswith("-") else "-D" + d for d in defs], "-o", out] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


EXT_SOURCE = os.path.join(CSRC, "v0_core_ext.cpp")
EXT_NAME = "_v0_core_native"


def ext_path() -> str:
    import sysconfig
    return os.path.join(PKG, EXT_NAME + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_ext(force: bool = False, verbose: bool = False) -> str:
    """g++ -> liuzhou_amd/_v0_core_native*.so: the compiled (PyBind11 + torch) `v0_core` operator layer over the C ABI
    (csrc/v0_core_ext.cpp) -- what the reference's v0/src/bindings/module.cpp is to its kernels.  Plain g++ against the
    torch / pybind11 / HIP headers of the image (no cmake, no JIT cache); needs no GPU to build."""
    out = ext_path()
    deps = [EXT_SOURCE, os.path.join(PKG, "..", "include", "liuzhou_hip.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    import sysconfig
    import pybind11
    from torch.utils import cpp_extension as ce
    cxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
    if not cxx:
        raise RuntimeError("no host C++ compiler (g++) found for the compiled v0_core layer")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = ["-I" + p for p in ce.include_paths()] + ["-I" + sysconfig.get_paths()["include"], "-I" + pybind11.get_include(),
                                                    "-I" + os.path.join(rocm, "include")]
    libdir = ce.library_paths()[0]
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-w", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-DTORCH_EXTENSION_NAME=" + EXT_NAME, "-D_GLIBCXX_USE_CXX11_ABI=1", *inc,
           "-o", out, EXT_SOURCE, "-L" + libdir, "-Wl,-rpath," + libdir, "-ltorch_python", "-ltorch", "-ltorch_cpu", "-lc10",
           "-lc10_hip", "-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) >= 3 and sys.argv[1] == "variant":          # python -m liuzhou_amd.build variant TAG DEF [DEF ...]
        print(build_variant(sys.argv[2], sys.argv[3:], verbose=True))
        sys.exit(0)
    print(build_hip(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
    print(build_ext(force=True, verbose=True))
