"""CPU: BN folding + MFMA fragment packing of the network (liuzhou_amd/net_pack.py) reproduce the model."""
import numpy as np
import torch

from liuzhou_amd.net import ChessNet, MODEL_CONFIGS, bucket_logits_to_scalar
from liuzhou_amd.net_pack import pack_model, emulate
from tests.golden_utils import load


def _model(name, seed):
    torch.manual_seed(seed)
    m = ChessNet(**MODEL_CONFIGS[name])
    g = torch.Generator().manual_seed(seed + 1)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) * 0.5 + 0.75)
            mod.weight.data.copy_(torch.rand(mod.weight.shape, generator=g) * 0.5 + 0.75)
            mod.bias.data.copy_(torch.randn(mod.bias.shape, generator=g) * 0.1)
    return m.eval()


def test_pack_emulation_matches_model_b6c64():
    z = load("g9_net.npz")
    x = torch.from_numpy(z["inputs"].astype(np.float32))
    m = _model("b6c64", 20260314)
    pack = pack_model(m)
    with torch.inference_mode():
        ref = m(x)
        val_ref = bucket_logits_to_scalar(ref[3])
    # exact-arithmetic check of folding / layout: keep activations fp32, weights are fp16-rounded
    out = emulate(pack, x, half_activations=False)
    for got, want in zip(out[:3], ref[:3]):
        assert float((got - want).abs().max()) < 2e-2
    assert float((out[4] - val_ref).abs().max()) < 5e-3
    # and golden outputs of the reference's ChessNet for the same seed
    np.testing.assert_allclose(out[0].numpy(), z["b6c64_lp1"], atol=2e-2, rtol=0)


def test_pack_layout_sizes():
    m = _model("b10c128", 20260314)
    pack = pack_model(m)
    C, NB = 128, 10
    want = 9 * 1 * (C // 16) * 512 + 2 * NB * 9 * (C // 32) * (C // 16) * 512 + 1 * (C // 32) * 8 * 512
    want += (6 * 4 + 6 * 8 + 4 * 7 + 2 * 1) * 512           # head dense layers (gpool_linear, fc1, fc2, out convs)
    assert int(pack.wfrag.numel()) == want
    assert pack.wfrag.dtype == torch.float16 and pack.fparams.dtype == torch.float32
    assert len(pack.layer_offsets) == 2 + 2 * NB


def test_fused_supported_names_what_the_kernel_is_not_built_for():
    """`net_hip.fused_supported`: the dispatch predicate of the worker / stage / arena (no GPU needed: shapes only)."""
    from liuzhou_amd.net import ChessNet, MODEL_CONFIGS
    from liuzhou_amd.net_hip import fused_supported, fused_unsupported_reason, MAX_BLOCKS
    for name in ("b6c64", "b10c128"):
        assert fused_supported(ChessNet(**MODEL_CONFIGS[name]))
    assert fused_supported(ChessNet(trunk_channels=64, num_blocks=20)) and MAX_BLOCKS == 47
    cases = {"trunk channels": dict(trunk_channels=32, num_blocks=1),
             "head channels": dict(trunk_channels=64, num_blocks=1, policy_channels=32),
             "value MLP": dict(trunk_channels=64, num_blocks=1, value_mlp_channels=64),
             "bins": dict(trunk_channels=64, num_blocks=1, value_bucket_bins=51),
             "residual blocks": dict(trunk_channels=64, num_blocks=48)}
    for what, arch in cases.items():
        why = fused_unsupported_reason(ChessNet(**arch))
        assert why is not None and what in why, (what, why)
    assert fused_unsupported_reason(object()) is not None
