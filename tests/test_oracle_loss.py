"""CPU: the training-loss oracle reproduces the reference's own loss functions (values + autograd gradients, g11)."""
import numpy as np
import torch

from oracle.loss_oracle import policy_value_loss
from tests.golden_utils import load


def _inputs(z):
    t = lambda k: torch.from_numpy(np.array(z[k]))
    l1 = torch.log_softmax(t("raw1"), 1).requires_grad_(True)
    l2 = torch.log_softmax(t("raw2"), 1).requires_grad_(True)
    l3 = torch.log_softmax(t("raw3"), 1).requires_grad_(True)
    vl = t("value_logits").clone().requires_grad_(True)
    return l1, l2, l3, vl, t("mask").bool(), t("target"), t("value"), t("soft")


def test_loss_oracle_matches_reference_values_and_gradients():
    z = load("g11_loss.npz")
    for tag in ("a", "b"):
        alpha, anti, dw = (float(x) for x in z[f"{tag}_params"])
        l1, l2, l3, vl, mask, target, value, soft = _inputs(z)
        out = policy_value_loss(l1, l2, l3, vl, mask, target, value, soft, soft_label_alpha=alpha,
                                anti_draw_penalty=anti, policy_draw_weight=dw)
        out["loss"].backward()
        np.testing.assert_allclose(out["loss"].item(), z[f"{tag}_loss"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out["policy_loss"].item(), z[f"{tag}_policy_loss"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out["bucket_loss"].item(), z[f"{tag}_bucket_loss"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(out["wdl_aux"].detach().numpy(), z[f"{tag}_wdl_aux"], rtol=1e-5, atol=1e-6)
        for g, key in ((l1.grad, "g1"), (l2.grad, "g2"), (l3.grad, "g3"), (vl.grad, "gv")):
            np.testing.assert_allclose(g.numpy(), z[f"{tag}_{key}"], rtol=1e-5, atol=1e-7)
