"""hedit.arcface.lpips_loss.LPIPSNet (parameter container) and the oracle's restatement of lpips.LPIPS(net='vgg')
(oracle/reward_nets.py; lpips==0.1.4 is third-party and absent offline: PARITY UNPINNED): parameter inventory and
state_dict names of the package, metric properties of the restatement, and that the product has no CPU path."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, ROOT)
from oracle import reward_nets as RN  # noqa: E402
from hedit.arcface.lpips_loss import LPIPS_Loss, LPIPSNet  # noqa: E402


def test_inventory_matches_the_package():
    sd = LPIPSNet().state_dict()
    convs = [k for k in sd if k.startswith("net.") and k.endswith("weight")]
    assert len(convs) == 13 and sum(sd[k].numel() for k in sd if k.startswith("net.")) == 14714688      # VGG16 features
    for k in ("net.slice1.0.weight", "net.slice1.2.bias", "net.slice2.5.weight", "net.slice3.14.weight", "net.slice4.21.bias",
              "net.slice5.28.weight", "lin0.model.1.weight", "lin4.model.1.weight", "scaling_layer.shift", "scaling_layer.scale"):
        assert k in sd, k
    assert tuple(sd["lin2.model.1.weight"].shape) == (1, 256, 1, 1)
    assert torch.allclose(sd["scaling_layer.scale"].flatten(), torch.tensor([.458, .448, .450]))


def test_metric_properties_and_gradient():
    g = torch.Generator().manual_seed(0)
    src = torch.randn(1, 3, 32, 32, generator=g) * 0.4
    m = LPIPS_Loss(src=src, seed=0)
    assert RN.lpips_loss(m, src.clone()).item() == 0.0
    x = (torch.randn(2, 3, 32, 32, generator=g) * 0.4).requires_grad_(True)
    loss = RN.lpips_loss(m, x)
    assert loss.item() > 0
    (grad,) = torch.autograd.grad(loss, x)
    assert torch.isfinite(grad).all() and grad.abs().max() > 0
    # symmetric in its two arguments
    a, b = x[:1].detach(), x[1:].detach()
    assert abs(RN.lpips_distance(m.lpips_loss, a, b).item() - RN.lpips_distance(m.lpips_loss, b, a).item()) < 1e-7
    with pytest.raises(RuntimeError, match="HIP executor only"):
        m.get_lpips_loss(x)
    assert type(m.lpips_loss).forward is torch.nn.Module.forward
