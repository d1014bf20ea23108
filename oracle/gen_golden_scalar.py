"""TEST INFRASTRUCTURE: writes tests/golden/g18_scalar_surface.npz from the REFERENCE's own compiled `v0_core` module
(oracle/_ref, built by oracle/Makefile from the sources where they lie under /root/reference; build container only).

The record is the deterministic walk of tests/scalar_walk.py through the scalar surface of the module (GameState, the
generate_* / apply_* rule functions, MoveRecord / ActionCode, TensorStateBatch: v0/src/bindings/module.cpp:877-1156):
random games to the end, crafted starts, unreachable states, every generated list and the outcome of ~30 probe calls per
state.  tests/test_scalar_surface.py replays the same walk over OUR module and compares byte for byte."""
import glob
import importlib.util
import os
import sys

import numpy as np
import torch  # noqa: F401  (the extension links libtorch: import it first)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.scalar_walk import record  # noqa: E402


def reference_module():
    found = glob.glob(os.path.join(ROOT, "oracle", "_ref", "v0_core*.so"))
    if not found:
        raise SystemExit("oracle/_ref/v0_core*.so not built (make -C oracle ref; needs /root/reference)")
    spec = importlib.util.spec_from_file_location("v0_core", found[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden", "g18_scalar_surface.npz")
    rec = record(reference_module())
    np.savez_compressed(out, **rec)
    print(out, os.path.getsize(out), "bytes;", {k: v.shape for k, v in rec.items() if v.ndim and v.shape[0] > 30})
