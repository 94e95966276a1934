"""The oracle of the identity reward -- oracle/reward_nets.py evaluated on the product's
parameter container hedit.arcface.IDLoss -- against vectors produced by RUNNING the reference's IDLoss and IR-SE50
backbone (face-swapping/arcface/arcface_model.py:11-67, facial_recognition/model_irse.py, helpers.py) with hash-seeded
weights: tests/golden/g12_idloss.npz, generator tests/golden/make_golden.py::gen_idloss.  The product itself has no
CPU path (asserted here); the native executor is compared with the same vectors in tests/test_gpu_arcface.py."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers.tiny import hash_normal  # noqa: E402
from oracle import reward_nets as RN  # noqa: E402
from hedit.arcface import Backbone, IDLoss  # noqa: E402

G12 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g12_idloss.npz")


def irse_state_dict(shapes):
    sd = {}
    for name, shape in shapes.items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros(shape, dtype=torch.long)
            continue
        v = hash_normal(tuple(shape) if len(shape) else (1,), zlib.crc32(name.encode()) % 100003).reshape(shape)
        if name.endswith("running_var"):
            v = 1.0 + 0.2 * v.abs()
        elif len(shape) > 1:
            v = v * float(np.prod(shape[1:])) ** -0.5
        elif name.endswith("weight"):
            v = 1.0 + 0.1 * v
        else:
            v = 0.05 * v
        sd[name] = v
    return sd


@pytest.fixture(scope="module")
def idl(tmp_path_factory):
    from PIL import Image
    g = np.load(G12)
    path = str(tmp_path_factory.mktemp("face") / "ref.png")
    Image.fromarray(g["ref_rgb"]).save(path)
    sd = irse_state_dict({k: tuple(v.shape) for k, v in Backbone().state_dict().items()})
    return IDLoss(ref_path=path, weights=sd), g


def test_backbone_inventory():
    sd = Backbone().state_dict()
    assert sum(v.numel() for k, v in sd.items() if "running" not in k and "tracked" not in k) == 43797696
    for k in ("input_layer.0.weight", "input_layer.2.weight", "body.3.shortcut_layer.0.weight", "body.3.res_layer.5.fc1.weight",
              "body.23.res_layer.4.running_var", "output_layer.3.weight", "output_layer.4.bias"):
        assert k in sd, k
    assert "body.0.shortcut_layer.0.weight" not in sd and "body.1.shortcut_layer.0.weight" not in sd          # identity (strided max-pool) shortcut: no parameters


def test_reference_face_preprocessing(idl):
    m, g = idl
    assert m.ref.shape == (1, 3, 256, 256)
    assert np.allclose(m.ref[0, :, ::16, ::16].numpy(), g["ref_tensor_sub"], atol=1e-6)


@pytest.mark.parametrize("i,b,hw", [(0, 1, 256), (1, 2, 128)])
def test_features_loss_and_gradient_match_reference(idl, i, b, hw):
    m, g = idl
    x = (hash_normal((b, 3, hw, hw), 40 + i) * 0.4).requires_grad_(True)
    with torch.no_grad():
        feat = RN.idloss_extract_feats(m, x.detach())
        sim = RN.idloss_cosine_sim(m, x.detach())
    assert np.allclose(feat.numpy(), g[f"feat{i}"], atol=2e-5)
    assert np.allclose(sim.numpy(), g[f"sim{i}"], atol=2e-5)
    loss = RN.idloss_cosine_loss(m, x)
    (grad,) = torch.autograd.grad(loss, x)
    assert abs(loss.item() - g[f"loss{i}"][0]) < 2e-5
    want = g[f"grad_sub{i}"]
    assert np.allclose(grad[:, :, ::4, ::4].numpy(), want, atol=2e-3 * np.abs(want).max(), rtol=1e-3)


def test_product_has_no_cpu_path(idl):
    m, _ = idl
    x = hash_normal((1, 3, 256, 256), 3) * 0.4
    for call in (m.extract_feats, m.get_cosine_sim, m.get_cosine_loss):
        with pytest.raises(RuntimeError, match="HIP executor only"):
            call(x)
    assert type(m.facenet).forward is torch.nn.Module.forward
