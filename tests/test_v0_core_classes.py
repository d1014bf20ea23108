"""CPU: `TorchScriptRunner` and `EvalBatcher` of the `v0_core` class surface (liuzhou_amd/mcts_core.py; reference
v0/src/net/torchscript_runner.cpp, v0/src/mcts/eval_batcher.cpp) -- host behaviour, and the reference's own compiled
classes beside them where oracle/_ref is built."""
import glob
import importlib.util
import os
import threading

import pytest
import torch

from liuzhou_amd import v0_core
from liuzhou_amd.net import ChessNet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference_module():
    found = glob.glob(os.path.join(ROOT, "oracle", "_ref", "v0_core*.so"))
    if not found:
        return None
    spec = importlib.util.spec_from_file_location("v0_core", found[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def archive(tmp_path_factory):
    torch.manual_seed(5)
    model = ChessNet(trunk_channels=8, num_blocks=1, policy_channels=4, value_channels=4, value_mlp_channels=8).eval()
    path = str(tmp_path_factory.mktemp("ts") / "tiny.pt")
    torch.jit.trace(model, torch.zeros(2, 11, 6, 6)).save(path)
    return model, path


def test_torchscript_runner_on_the_host(archive):
    model, path = archive
    x = torch.randn(5, 11, 6, 6, generator=torch.Generator().manual_seed(1))
    r = v0_core.TorchScriptRunner(path)
    assert (r.device, r.dtype) == ("cpu", "auto")
    got = r.forward(x)
    with torch.no_grad():
        want = model(x)
    assert len(got) == 4 and all(torch.allclose(a, b, atol=1e-6) for a, b in zip(got, want))
    assert v0_core.TorchScriptRunner(path, "cpu", "fp32").dtype == "float32"
    bf = v0_core.TorchScriptRunner(path, "cpu", "bf16", False)
    assert bf.dtype == "bfloat16" and bf.forward(x)[0].dtype == torch.bfloat16
    with pytest.raises(RuntimeError, match="float16 is not supported on CPU"):
        v0_core.TorchScriptRunner(path, "cpu", "float16")
    with pytest.raises(RuntimeError, match="Unsupported dtype"):
        v0_core.TorchScriptRunner(path, "cpu", "int8")
    R = _reference_module()
    if R is not None:
        ref = R.TorchScriptRunner(path, "cpu", "auto", True)
        assert (ref.device, ref.dtype) == (r.device, r.dtype)
        for a, b in zip(got, ref.forward(x)):
            assert torch.equal(a, b)


class HostEngine:
    """What EvalBatcher needs of an InferenceEngine, on the host (the product engine is HIP-only)."""

    def __init__(self, model, batch_size):
        self.model, self.batch_size, self.device, self.calls = model, batch_size, "cpu", []

    def forward(self, input, n_valid=-1):
        self.calls.append(int(n_valid))
        assert tuple(input.shape) == (self.batch_size, 11, 6, 6) and bool((input[n_valid:] == 0).all())
        with torch.no_grad():
            return tuple(o[:n_valid] for o in self.model(input))


def test_eval_batcher_packs_callers_into_the_engine_batch(archive):
    model, path = archive
    eng = HostEngine(model, 16)
    with pytest.raises(RuntimeError, match="batch_size mismatch"):
        v0_core.EvalBatcher(eng, 32)
    with pytest.raises(RuntimeError, match="valid InferenceEngine"):
        v0_core.EvalBatcher(None, 16)
    b = v0_core.EvalBatcher(eng, 16, timeout_ms=20)
    assert (b.batch_size, b.timeout_ms) == (16, 20)
    g = torch.Generator().manual_seed(2)
    inputs = [torch.randn(int(n), 11, 6, 6, generator=g) for n in (3, 5, 1, 7, 4, 2, 6, 3)]
    out = [None] * len(inputs)

    def call(i):
        out[i] = b.forward(inputs[i])
    threads = [threading.Thread(target=call, args=(i,)) for i in range(len(inputs))]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for x, o in zip(inputs, out):
        with torch.no_grad():
            want = model(x)
        assert all(torch.allclose(a, w, atol=1e-6) for a, w in zip(o, want))
    st = b.get_eval_stats()
    assert st["eval_leaves"] == sum(int(x.shape[0]) for x in inputs) == sum(eng.calls) and st["eval_calls"] == len(eng.calls)
    assert st["eval_calls"] < len(inputs) and sum(st["hist"]) == st["eval_calls"] and len(st["hist"]) == 17
    assert all(n <= 16 for n in eng.calls)
    # n_valid: only the first rows are evaluated; per-request errors reach their caller, the batcher lives on
    two = b.forward(inputs[3], 2)
    assert two[0].shape[0] == 2
    with pytest.raises(RuntimeError, match="n_valid out of range"):
        b.forward(torch.zeros(17, 11, 6, 6))
    with pytest.raises(RuntimeError, match="must be 4D"):
        b.forward(torch.zeros(11, 6, 6))
    with pytest.raises(RuntimeError, match="shape mismatch"):
        b.forward(torch.zeros(2, 10, 6, 6))
    with pytest.raises(RuntimeError, match="smaller than n_valid"):
        b.forward(torch.zeros(2, 11, 6, 6), 3)
    before = b.get_eval_stats()
    full = b.forward(torch.zeros(16, 11, 6, 6))
    after = b.get_eval_stats()
    assert full[0].shape[0] == 16 and after["full512_calls"] == before["full512_calls"] + 1 and after["hist"][15] == before["hist"][15] + 1
    b.reset_eval_stats()
    assert b.get_eval_stats() == {"eval_calls": 0, "eval_leaves": 0, "full512_calls": 0, "hist": [0] * 17}
    b.shutdown(); b.shutdown()
    with pytest.raises(RuntimeError, match="shut down"):
        b.forward(inputs[0])


def test_eval_batcher_statistics_equal_the_reference_class(archive):
    R = _reference_module()
    if R is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    model, path = archive
    ref_engine = R.InferenceEngine(path, "cpu", "float32", 16, 11, 6, 6, 1, True)
    ref, ours = R.EvalBatcher(ref_engine, 16, 11, 6, 6, 1), v0_core.EvalBatcher(HostEngine(model, 16), 16, 11, 6, 6, 1)
    g = torch.Generator().manual_seed(3)
    for n in (1, 16, 2, 9, 16, 5, 12, 3):                          # one caller: every request is its own batch
        x = torch.randn(n, 11, 6, 6, generator=g)
        a, b = ours.forward(x), ref.forward(x)
        assert all(torch.allclose(p, q, atol=1e-5) for p, q in zip(a, b))
    for bad in (torch.zeros(17, 11, 6, 6), torch.zeros(2, 11, 6, 5)):
        for batcher in (ours, ref):
            with pytest.raises(RuntimeError):
                batcher.forward(bad)
    assert ours.get_eval_stats() == ref.get_eval_stats()
    ours.shutdown(); ref.shutdown()
