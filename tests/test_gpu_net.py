"""GPU: fused fp16-MFMA network forward vs the fp32 PyTorch model (and the pack emulation)."""
import numpy as np
import pytest
import torch

from tests.golden_utils import load, states, FIELDS

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _planes(n, seed=0):
    """n real positions: golden reachable states encoded with the HIP op."""
    from liuzhou_amd import v0_core
    z = load("g1_rules.npz")
    st = states(z, "s")
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, st["board"].shape[0], n)
    t = lambda k, dt: torch.from_numpy(np.ascontiguousarray(st[k][idx]).astype(dt)).to(DEV)
    return v0_core.states_to_model_input(t("board", np.int8), t("marks_black", bool), t("marks_white", bool),
                                         t("phase", np.int64), t("current_player", np.int64))


@pytest.mark.parametrize("name,n,half", [("b6c64", 16, False), ("b6c64", 1000, False), ("b6c64", 4099, False),
                                         ("b10c128", 8, False), ("b10c128", 777, False),
                                         ("b6c64", 5, True), ("b6c64", 2049, True),
                                         ("b10c128", 9, "wide"), ("b10c128", 2050, "wide")])
def test_fused_net_matches_fp32_model(name, n, half):
    """`half`: the 4-wave / 8-sample workgroup configuration used by the dual-stream search; "wide": the 4-wave shape
    of the 128-channel net with 4 channel tiles per wave (must also equal the 8-wave shape bit for bit: same MFMAs in
    the same order per output element)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, bucket_logits_to_scalar
    from liuzhou_amd.net_hip import FusedNet
    torch.manual_seed(20260314)
    m = ChessNet(**MODEL_CONFIGS[name]).eval()
    g = torch.Generator().manual_seed(5)
    for mod in m.modules():                       # non-trivial BN statistics so folding is exercised
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) * 0.5 + 0.75)
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) * 0.5 + 0.75)
            mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
    m = m.to(DEV)
    x = _planes(n, seed=n)
    fused = FusedNet(m, half_workgroups=half is True, wide_tiles=half == "wide")
    lp1, lp2, lpm, vl = fused(x)
    val = fused.last_value
    if half == "wide":
        ref8 = FusedNet(m, wide_tiles=False)
        for a, b in zip(ref8(x), (lp1, lp2, lpm, vl)):
            assert torch.equal(a, b)
        assert torch.equal(ref8.last_value, val)
    with torch.inference_mode():
        r1, r2, rm, rv = m(x)
        rval = bucket_logits_to_scalar(rv)
    # fp16 operands / fp32 accumulate: tolerance is the reduced-precision mode's, not the 1e-5 fp32 bar
    for got, want in ((lp1, r1), (lp2, r2), (lpm, rm)):
        assert torch.isfinite(got).all()
        err = (got.exp() - want.exp()).abs().max().item()
        assert err < 3e-3, err
        assert torch.allclose(got.exp().sum(1), torch.ones(n, device=DEV), atol=1e-4)
    assert (vl - rv).abs().max().item() < 3e-2
    assert (val - rval).abs().max().item() < 3e-3


def test_values_only_mode_equals_full_forward():
    """Skipping the policy head (log-prob outputs NULL) must not change the value output."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import FusedNet
    for name, n in (("b6c64", 333), ("b10c128", 100)):
        torch.manual_seed(3)
        f = FusedNet(ChessNet(**MODEL_CONFIGS[name]).eval().to(DEV))
        x = _planes(n, seed=1)
        f(x)
        full = f.last_value.clone()
        assert torch.equal(f.values_only(x), full)
