"""Policy + bucketed-value ResNet for Liuzhou (6x6 board, 11 input planes).

Own definition of the architecture the reference calls `ChessNet` (src/neural_network.py:213-259:
stem 3x3 -> pre-activation residual blocks (:82-95) -> trunk BN/ReLU -> policy head with global-pool
bias (:97-124) and value head with global-pool MLP over 101 buckets (:126-148)).  Module and parameter
names are identical, so `state_dict()` keys/shapes match the reference's checkpoints one-to-one
(tests/golden/g9_net_keys.json).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

NUM_INPUT_CHANNELS = 11
VALUE_BUCKET_BINS = 101
BOARD_SIZE = 6


def global_pool(x: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """(N,C,H,W) -> (N,3C): mean, max, sqrt(biased var + eps)   (neural_network.py:67-80)"""
    flat = x.flatten(2)
    mean = flat.mean(dim=2)
    mx = flat.amax(dim=2)
    std = torch.sqrt(flat.var(dim=2, unbiased=False) + eps)
    return torch.cat((mean, mx, std), dim=1)


class GlobalPool(nn.Module):
    def __init__(self, eps: float = 1e-6) -> None:
        super().__init__()
        self.eps = eps

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return global_pool(x, self.eps)


class PreActResBlock(nn.Module):
    def __init__(self, channels: int) -> None:
        super().__init__()
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv1 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, 3, padding=1, bias=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.conv1(F.relu(self.bn1(x)))
        y = self.conv2(F.relu(self.bn2(y)))
        return x + y


class PolicyHead(nn.Module):
    def __init__(self, in_channels: int, policy_channels: int) -> None:
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, policy_channels, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(policy_channels)
        self.gpool = GlobalPool()
        self.gpool_linear = nn.Linear(3 * policy_channels, policy_channels, bias=False)
        self.bn2 = nn.BatchNorm2d(policy_channels)
        self.out_pos1 = nn.Conv2d(policy_channels, 1, 1, bias=False)
        self.out_pos2 = nn.Conv2d(policy_channels, 1, 1, bias=False)
        self.out_mark = nn.Conv2d(policy_channels, 1, 1, bias=False)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        p = F.relu(self.bn1(self.conv1(x)))
        p = p + self.gpool_linear(self.gpool(p))[:, :, None, None]
        p = F.relu(self.bn2(p))
        return (F.log_softmax(self.out_pos1(p).flatten(1), dim=1),
                F.log_softmax(self.out_pos2(p).flatten(1), dim=1),
                F.log_softmax(self.out_mark(p).flatten(1), dim=1))


class ValueHead(nn.Module):
    def __init__(self, in_channels: int, value_channels: int, mlp_channels: int,
                 num_value_bins: int = VALUE_BUCKET_BINS) -> None:
        super().__init__()
        if int(num_value_bins) < 2:
            raise ValueError("num_value_bins must be >= 2")
        self.conv1 = nn.Conv2d(in_channels, value_channels, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(value_channels)
        self.gpool = GlobalPool()
        self.fc1 = nn.Linear(3 * value_channels, mlp_channels)
        self.fc2 = nn.Linear(mlp_channels, int(num_value_bins))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        v = F.relu(self.bn1(self.conv1(x)))
        return self.fc2(F.relu(self.fc1(self.gpool(v))))


class ChessNet(nn.Module):
    def __init__(self, board_size: int = BOARD_SIZE, num_input_channels: int = NUM_INPUT_CHANNELS,
                 hidden_conv_channels: Optional[int] = None, trunk_channels: int = 128, num_blocks: int = 10,
                 policy_channels: int = 64, value_channels: int = 64, value_mlp_channels: int = 128,
                 value_bucket_bins: int = VALUE_BUCKET_BINS) -> None:
        super().__init__()
        if hidden_conv_channels is not None:
            trunk_channels = hidden_conv_channels
        self.board_size = board_size
        self.num_input_channels = num_input_channels
        self.stem_conv = nn.Conv2d(num_input_channels, trunk_channels, 3, padding=1, bias=False)
        self.stem_bn = nn.BatchNorm2d(trunk_channels)
        self.blocks = nn.ModuleList(PreActResBlock(trunk_channels) for _ in range(num_blocks))
        self.trunk_bn = nn.BatchNorm2d(trunk_channels)
        self.policy_head = PolicyHead(trunk_channels, policy_channels)
        self.value_head = ValueHead(trunk_channels, value_channels, value_mlp_channels, int(value_bucket_bins))

    def trunk(self, x: torch.Tensor) -> torch.Tensor:
        x = F.relu(self.stem_bn(self.stem_conv(x)))
        for blk in self.blocks:
            x = blk(x)
        return F.relu(self.trunk_bn(x))

    def forward(self, x: torch.Tensor):
        """-> (log_p1, log_p2, log_pmc [B,36] log-probabilities, value_logits [B,K])"""
        t = self.trunk(x)
        lp1, lp2, lpm = self.policy_head(t)
        return lp1, lp2, lpm, self.value_head(t)


def bucket_logits_to_scalar(logits: torch.Tensor, num_bins: int = VALUE_BUCKET_BINS) -> torch.Tensor:
    """Expectation over evenly spaced bucket centres in [-1,1] (neural_network.py:201-210)."""
    bins = int(logits.size(-1))
    probs = torch.softmax(logits, dim=-1)
    centers = torch.linspace(-1.0, 1.0, steps=bins, device=logits.device, dtype=probs.dtype)
    return (probs * centers).sum(dim=-1)


def scalar_to_bucket_twohot(value: torch.Tensor, num_bins: int = VALUE_BUCKET_BINS) -> torch.Tensor:
    """Two-hot encoding of scalars in [-1,1] (neural_network.py:176-198)."""
    bins = int(num_bins)
    if bins < 2:
        raise ValueError("num_bins must be >= 2")
    if value.dim() > 1 and value.size(-1) == 1:
        value = value.squeeze(-1)
    v = value.to(torch.float32).clamp(-1.0, 1.0)
    u = (v + 1.0) / (2.0 / float(bins - 1))
    lo = torch.floor(u).to(torch.int64).clamp(0, bins - 1)
    hi = (lo + 1).clamp(0, bins - 1)
    frac = (u - lo.to(u.dtype)).clamp(0.0, 1.0)
    frac = torch.where(hi.eq(lo), torch.zeros_like(frac), frac)
    out = torch.zeros((*v.shape, bins), dtype=torch.float32, device=v.device)
    out.scatter_add_(-1, lo.unsqueeze(-1), (1.0 - frac).unsqueeze(-1))
    out.scatter_add_(-1, hi.unsqueeze(-1), frac.unsqueeze(-1))
    return out


MODEL_CONFIGS = {
    "tiny": dict(trunk_channels=8, num_blocks=1, policy_channels=4, value_channels=4, value_mlp_channels=8),
    "b6c64": dict(trunk_channels=64, num_blocks=6),
    "b10c128": dict(trunk_channels=128, num_blocks=10),
}


def build_model(name: str = "b10c128", seed: Optional[int] = None) -> ChessNet:
    if seed is not None:
        torch.manual_seed(int(seed))
    return ChessNet(**MODEL_CONFIGS[name]).eval()


def stable_resnet_init(model: nn.Module, seed: int) -> None:
    """Bootstrap initialisation of a model without a checkpoint (v1/train.py:162-217, seed = MODEL_INIT_SEED):
    He-normal (fan_out) conv / linear weights, zero biases, unit BatchNorm except a zero gamma on every block's
    second norm (blocks start as the identity), and std-1e-3 output layers (three policy 1x1 convs, value fc2).
    Draw order follows `model.modules()`, the caller's RNG streams are left untouched."""
    if int(seed) <= 0:
        raise ValueError(f"seed must be positive for model init, got {seed}")
    import random
    cuda = torch.cuda.is_available()
    saved = (random.getstate(), torch.random.get_rng_state(), torch.cuda.get_rng_state_all() if cuda else None)
    try:
        random.seed(int(seed)); torch.manual_seed(int(seed))
        if cuda:
            torch.cuda.manual_seed_all(int(seed))
        for m in model.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight); nn.init.zeros_(m.bias)
        for block in getattr(model, "blocks", []):
            nn.init.zeros_(block.bn2.weight)
        small = [getattr(model.policy_head, n) for n in ("out_pos1", "out_pos2", "out_mark")] + [model.value_head.fc2]
        for layer in small:
            nn.init.normal_(layer.weight, mean=0.0, std=1e-3)
            if layer.bias is not None:
                nn.init.zeros_(layer.bias)
    finally:
        random.setstate(saved[0]); torch.random.set_rng_state(saved[1])
        if saved[2] is not None:
            torch.cuda.set_rng_state_all(saved[2])
