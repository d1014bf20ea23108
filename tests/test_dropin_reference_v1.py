"""Build container only (the reference is mounted at /root/reference; skipped elsewhere): the REFERENCE's own v1 Python
(`v1/python/self_play_gpu_runner.py::self_play_v1_gpu`, `mcts_gpu.py::V1RootMCTS`) run, unmodified, over OUR `v0_core`
drop-in module (`liuzhou_amd/dropin` first on PYTHONPATH, CPU tensors -> the host build of the C ABI) must reproduce the
trace the same code produced over the reference's own extension (tests/golden/g8_selfplay.npz) -- the zero-edit claim of
INTEGRATION.md section A, exercised end to end.  Runs in a child process so that `import v0_core` resolves freshly."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.golden_utils import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LZ_REFERENCE", "/root/reference")

CHILD = r'''
import json, os, random, sys
import numpy as np, torch
import v0_core
from liuzhou_amd import _lib
assert "liuzhou_amd" in os.path.realpath(v0_core.__file__), v0_core.__file__
from src.neural_network import ChessNet, NUM_INPUT_CHANNELS
from v1.python.self_play_gpu_runner import self_play_v1_gpu
torch.manual_seed(7)
model = ChessNet(board_size=6, num_input_channels=NUM_INPUT_CHANNELS, trunk_channels=8, num_blocks=1,
                 policy_channels=4, value_channels=4, value_mlp_channels=8).eval()
torch.manual_seed(0); np.random.seed(0); random.seed(0)
batch, stats = self_play_v1_gpu(model=model, num_games=4, mcts_simulations=32, temperature_init=1.0,
                                temperature_final=0.1, temperature_threshold=10, exploration_weight=1.0, device="cpu",
                                add_dirichlet_noise=False, soft_value_k=2.0, opening_random_moves=0, max_game_plies=512,
                                sample_moves=False, concurrent_games=4)
np.savez(sys.argv[1], state_tensors=batch.state_tensors.numpy(), legal_masks=batch.legal_masks.numpy(),
         policy_targets=batch.policy_targets.numpy(), value_targets=batch.value_targets.numpy(),
         soft_value_targets=batch.soft_value_targets.numpy(),
         outcome=np.asarray([stats.black_wins, stats.white_wins, stats.draws]),
         libs=np.asarray([int(_lib._host is not None), int(_lib._lib is not None), int(v0_core.active_binding() == "native")]))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "v1", "python")), reason="reference not mounted")
@pytest.mark.parametrize("binding", ["native", "python"])
def test_reference_v1_selfplay_runs_unmodified_over_our_v0_core(tmp_path, binding):
    """`binding`: the compiled PyBind11 layer (what `import v0_core` gives once it is built) and the ctypes layer."""
    out = tmp_path / "trace.npz"
    env = dict(os.environ)
    env["LZ_V0_CORE_NATIVE"] = "1" if binding == "native" else "0"
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "liuzhou_amd", "dropin"), ROOT, REF])
    env["PYTHONDONTWRITEBYTECODE"] = "1"
    r = subprocess.run([sys.executable, "-c", CHILD, str(out)], env=env, cwd=str(tmp_path), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got, z = np.load(out), load("g8_selfplay.npz")
    # the host build of our C ABI served the CPU tensors (ctypes layer: loaded by _lib; compiled layer: dlopen'ed by the
    # extension itself, so _lib loaded nothing); the HIP library was never loaded through _lib either way
    assert got["libs"].tolist() == ([0, 0, 1] if binding == "native" else [1, 0, 0])
    n = int(z["num_positions"])
    want_states = np.unpackbits(z["state_tensors"], axis=1)[:, :11 * 36].reshape(n, 11, 6, 6).astype(np.float32)
    assert got["state_tensors"].shape[0] == n and np.array_equal(got["state_tensors"], want_states)
    assert np.array_equal(got["legal_masks"], np.unpackbits(z["legal_masks"], axis=1)[:, :220].astype(bool))
    np.testing.assert_allclose(got["policy_targets"], z["policy_targets"], atol=1e-5, rtol=0)
    np.testing.assert_array_equal(got["value_targets"], z["value_targets"])
    np.testing.assert_allclose(got["soft_value_targets"], z["soft_value_targets"], atol=1e-6, rtol=0)
    assert got["outcome"].tolist() == [int(z["black_wins"]), int(z["white_wins"]), int(z["draws"])]
