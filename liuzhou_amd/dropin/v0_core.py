"""Drop-in shim: with `liuzhou_amd/dropin` (and the repo root) on PYTHONPATH, `import v0_core`
resolves to the MI355X operator surface instead of the reference's CUDA extension."""
from liuzhou_amd.v0_core import *  # noqa: F401,F403
from liuzhou_amd.v0_core import Phase, version  # noqa: F401
from liuzhou_amd.v0_scalar import *  # noqa: F401,F403  (GameState, MoveRecord, the scalar rule functions, exported enum values)


def __getattr__(name):          # MCTSConfig / MCTSCore / InferenceEngine / EvalBatcher / TorchScriptRunner: resolved lazily by the package module
    import liuzhou_amd.v0_core as _m
    return getattr(_m, name)
