"""-m gpu: the native identity reward (hedit_irse50_features / hedit_irse50_cos_fwd_bwd, csrc/irse.hip -- SURVEY.md
section 8 row a23) against (1) vectors produced by RUNNING the reference's IDLoss + IR-SE50 backbone
(face-swapping/arcface/arcface_model.py:11-67, facial_recognition/model_irse.py, helpers.py) with hash-seeded weights,
tests/golden/g12_idloss.npz, and (2) the oracle's fp32 restatement (oracle/reward_nets.py, pinned on the same vectors,
run on the CPU) on the same weights at batch 8.

Tolerances.  The executor keeps 16 mantissa bits per operand (three-term split-bf16 products, fp32 accumulation and
activations): features / loss agree with fp32 to ~1e-5 relative.  The IMAGE GRADIENT of this 50-layer network is far
more sensitive: perturbing the weights by 1.5e-5 relative (what 16 bits do) moves the fp64 gradient by 6e-3, inputs
likewise by 3e-3 (measured) -- so 5e-3 relative L2 is the arithmetic's floor, against 7e-2 for bf16 storage."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from helpers import gpu as G  # noqa: E402
from oracle import reward_nets as RN  # noqa: E402
from helpers.tiny import hash_normal  # noqa: E402
from test_host_arcface import G12, irse_state_dict  # noqa: E402
from hedit.arcface import Backbone, IDLoss  # noqa: E402


@pytest.fixture(scope="module")
def idl(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from PIL import Image
    g = np.load(G12)
    path = str(tmp_path_factory.mktemp("face") / "ref.png")
    Image.fromarray(g["ref_rgb"]).save(path)
    sd = irse_state_dict({k: tuple(v.shape) for k, v in Backbone().state_dict().items()})
    return IDLoss(ref_path=path, weights=sd, device=G.dev()), g


@pytest.mark.parametrize("i,b,hw", [(0, 1, 256), (1, 2, 128)])
def test_native_identity_reward_matches_reference_vectors(idl, i, b, hw):
    m, g = idl
    x = G.f32(hash_normal((b, 3, hw, hw), 40 + i) * 0.4).requires_grad_(True)
    with torch.no_grad():
        feat = m.extract_feats(x.detach())
        sim = m.get_cosine_sim(x.detach())
    G.sync()
    assert G.rel_err(feat, torch.from_numpy(g[f"feat{i}"])) < 2e-4
    assert np.allclose(sim.cpu().numpy(), g[f"sim{i}"], atol=1e-4)
    loss = m.get_cosine_loss(x)
    (grad,) = torch.autograd.grad(loss, x)
    G.sync()
    assert abs(loss.item() - g[f"loss{i}"][0]) < 1e-4
    assert G.rel_err(grad[:, :, ::4, ::4], torch.from_numpy(g[f"grad_sub{i}"])) < 5e-3
    # get_cosine_sim is differentiable like the reference's: (1 - sim).mean() gives the same gradient
    x2 = x.detach().clone().requires_grad_(True)
    (grad2,) = torch.autograd.grad((1 - m.get_cosine_sim(x2)).mean(), x2)
    G.sync()
    assert G.rel_err(grad2, grad) < 1e-5
    with pytest.raises(NotImplementedError):
        m.extract_feats(x2)


def test_native_matches_oracle_at_batch_8_and_is_batch_invariant():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    gen = torch.Generator().manual_seed(0)
    ref = torch.randn(1, 3, 256, 256, generator=gen) * 0.4
    nat = IDLoss(ref=ref, device=G.dev(), seed=1)
    tor = IDLoss(ref=ref, seed=1)                       # the same parameters on the CPU, for the oracle
    xc = torch.randn(8, 3, 256, 256, generator=gen) * 0.4
    x = G.f32(xc)
    with torch.no_grad():
        fn = nat.extract_feats(x)
        ft = torch.nn.functional.normalize(RN.idloss_extract_feats(tor, xc), dim=-1)
    assert G.rel_err(fn, ft) < 2e-4
    xs = [x.clone().requires_grad_(True), xc.clone().requires_grad_(True)]
    ln, lt = nat.get_cosine_loss(xs[0]), RN.idloss_cosine_loss(tor, xs[1])
    gn, gt = torch.autograd.grad(ln, xs[0])[0], torch.autograd.grad(lt, xs[1])[0]
    G.sync()
    assert abs(ln.item() - lt.item()) < 1e-4
    assert G.rel_err(gn, gt) < 5e-3
    # an image's feature does not depend on the batch it is evaluated in: same bits
    with torch.no_grad():
        assert torch.equal(fn[3:4], nat.extract_feats(x[3:4]))
        assert torch.equal(fn[5:7], nat.extract_feats(x[5:7]))


def test_there_is_no_cpu_path():
    m = IDLoss(ref=torch.zeros(1, 3, 256, 256), seed=0)
    with pytest.raises(RuntimeError):
        m.get_cosine_loss(torch.zeros(1, 3, 256, 256))
