"""Host-side image I/O (hedit/utils/utils.py) against vectors produced by the reference's own
load_512 (tests/golden/g8_load512.npz, generator tests/golden/make_golden.py:gen_load512)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
GOLD = os.path.join(ROOT, "tests", "golden")

from hedit.utils import image_grid, load_512, tensor_to_pil  # noqa: E402


def synthetic_rgb(h, w, seed):
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    ch = [((x * (3 + c) + y * (5 - c) + seed * 17) % 256 + ((x * y + c * 31) // 7) % 64) % 256 for c in range(3)]
    return np.stack(ch, -1).astype(np.uint8)


def test_load_512_matches_reference_vectors():
    g = np.load(os.path.join(GOLD, "g8_load512.npz"))
    n = sum(1 for k in g.files if k.startswith("case"))
    assert n >= 5
    for i in range(n):
        h, w, l, r, t, b = (int(v) for v in g[f"case{i}"])
        x = load_512(synthetic_rgb(h, w, i), l, r, t, b, torch.device("cpu"))
        assert x.shape == (1, 3, 512, 512) and x.dtype == torch.float32
        k = torch.round((x + 1) * 127.5).to(torch.int64)
        assert torch.equal(k[0, :, ::4, ::4], torch.from_numpy(g[f"sub{i}"].astype(np.int64)))   # bit-exact
        assert int(k.sum()) == int(g[f"sum{i}"][0])


def test_load_512_from_file(tmp_path):
    from PIL import Image
    arr = synthetic_rgb(96, 160, 3)
    p = tmp_path / "img.png"
    Image.fromarray(arr).save(p)
    assert torch.equal(load_512(str(p)), load_512(arr))


def test_tensor_to_pil_and_grid():
    x = torch.tensor([-1.5, -1.0, 0.0, 0.5, 1.0, 2.0]).reshape(1, 1, 1, 6).repeat(2, 3, 4, 1)
    pil = tensor_to_pil(x)
    assert len(pil) == 2 and pil[0].size == (6, 4)
    row = np.array(pil[0])[0, :, 0]
    assert row.tolist() == [0, 0, 127, 191, 255, 255]          # truncation, like ToPILImage
    assert tensor_to_pil([x[:1], x[1:]])[1].size == (6, 4)
    g = image_grid(x)
    assert g.size == (12, 4)
    g2 = image_grid(x, rows=2, cols=1, titles=["a", "b"])
    assert g2.size == (6, 2 * 24)
    assert np.array(g2)[0, 5].tolist() == [255, 255, 255]      # title strip
