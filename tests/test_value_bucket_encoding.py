"""CPU: the 101-bucket scalar value encoding / decoding of `liuzhou_amd/net.py` -- the cases of the reference's
tests/test_value_bucket_encoding.py (src/neural_network.py:176-210), plus the same numbers from the C oracle's loss
restatement where it has them."""
import math

import pytest
import torch

from liuzhou_amd.net import bucket_logits_to_scalar, scalar_to_bucket_twohot


def test_scalar_to_bucket_twohot_endpoints():
    target = scalar_to_bucket_twohot(torch.tensor([-1.0, 0.0, 1.0]), num_bins=101)
    assert tuple(target.shape) == (3, 101)
    assert float(target[0, 0]) == pytest.approx(1.0)
    assert float(target[1, 50]) == pytest.approx(1.0)
    assert float(target[2, 100]) == pytest.approx(1.0)
    assert torch.allclose(target.sum(1), torch.ones(3))


def test_scalar_twohot_roundtrip_error_bound():
    values = torch.linspace(-1.0, 1.0, steps=401)
    target = scalar_to_bucket_twohot(values, num_bins=101)
    centers = torch.linspace(-1.0, 1.0, steps=101)
    decoded = (target * centers.view(1, -1)).sum(dim=-1)
    assert float((decoded - values).abs().max()) <= 0.02 + 1e-6
    assert int((target > 0).sum(1).max()) <= 2                       # at most two neighbouring buckets per value
    # out-of-range targets are clamped, not wrapped
    assert torch.equal(scalar_to_bucket_twohot(torch.tensor([-3.0, 7.0])), scalar_to_bucket_twohot(torch.tensor([-1.0, 1.0])))


def test_bucket_logits_to_scalar_onehot():
    logits = torch.full((3, 101), -20.0)
    logits[0, 0] = logits[1, 50] = logits[2, 100] = 20.0
    decoded = bucket_logits_to_scalar(logits, num_bins=101)
    assert float(decoded[0]) == pytest.approx(-1.0, abs=1e-4)
    assert float(decoded[1]) == pytest.approx(0.0, abs=1e-4)
    assert float(decoded[2]) == pytest.approx(1.0, abs=1e-4)


def test_search_scalar_value_from_bucket_logits():
    """tests/test_value_bucket_encoding.py:45-58 (`V1RootMCTS._to_scalar_value`): bucket 75 of 101 -> +0.5."""
    from liuzhou_amd.mcts_gpu import V1RootMCTS
    logits = torch.full((1, 101), -20.0)
    logits[0, 75] = 20.0
    out = V1RootMCTS._to_scalar_value(object.__new__(V1RootMCTS), logits)
    assert float(out.reshape(-1)[0]) == pytest.approx(-1.0 + 2.0 * 75.0 / 100.0, abs=5e-3)
    assert math.isfinite(float(out.reshape(-1)[0]))
