"""h-edit_amd/evaluation/evaluation.py ("then the PieBench evaluator"): run-length mask
decoding against vectors produced by the reference's own function (g15), pixel metrics against their definitions, CSV layout, and the refusal of the network metrics."""
import csv
import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from evaluation import evaluation as EV  # noqa: E402


def test_mask_decode_known_answer():
    m = EV.mask_decode([10, 5, 30, 100], image_shape=(8, 8))
    flat = np.zeros(64)
    flat[10:15] = 1
    flat[30:64] = 1            # the run is clipped at the end of the image
    want = flat.reshape(8, 8)
    want[0, :] = want[-1, :] = 1
    want[:, 0] = want[:, -1] = 1
    assert np.array_equal(m, want)
    assert EV.mask_decode([], image_shape=(4, 4)).sum() == 12      # border only


def test_mask_decode_matches_reference_vectors():
    """g15: inputs and outputs of the reference's own mask_decode (tests/golden/make_golden_eval.py)"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g15_mask_decode.npz"))
    for i in range(6):
        want = np.unpackbits(g[f"mask{i}"])[:512 * 512].reshape(512, 512)
        assert np.array_equal(EV.mask_decode(g[f"enc{i}"].tolist()), want)


def test_pixel_metrics():
    from PIL import Image
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, size=a.shape), 0, 255).astype(np.uint8)
    ia, ib = Image.fromarray(a), Image.fromarray(b)
    mc = EV.MetricsCalculator()
    mse = mc.calculate_mse(ia, ib)
    assert abs(mse - ((a / 255.0 - b / 255.0) ** 2).mean()) < 1e-7
    assert abs(mc.calculate_psnr(ia, ib) - 10 * np.log10(1.0 / mse)) < 1e-4
    assert abs(mc.calculate_ssim(ia, ia) - 1.0) < 1e-6 and 0 < mc.calculate_ssim(ia, ib) < 1
    mask = np.zeros((64, 64, 3), dtype=np.float32)
    mask[:32] = 1
    m2 = mc.calculate_mse(ia, ib, mask, mask)
    assert abs(m2 - (((a / 255.0 - b / 255.0) * mask) ** 2).mean()) < 1e-7
    assert EV.calculate_metric(mc, "psnr_edit_part", ia, ib, np.zeros_like(mask), np.zeros_like(mask), "", "") == "nan"
    with pytest.raises(NotImplementedError):
        EV.calculate_metric(mc, "structure_distance", ia, ib, mask, mask, "", "")
    with pytest.raises(NotImplementedError):
        EV.calculate_metric(mc, "clip_similarity_target_image_edit_part", ia, ib, mask, mask, "", "")


def test_driver_writes_the_reference_csv_layout(tmp_path):
    from PIL import Image
    d = tmp_path / "data" / "annotation_images" / "0_x"
    d.mkdir(parents=True)
    out = tmp_path / "res" / "0_x"
    out.mkdir(parents=True)
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(32, 32, 3), dtype=np.uint8)
    Image.fromarray(img).save(d / "a.png")
    sheet = np.concatenate([np.zeros_like(img), img], 1)            # [source | edited] sheet: the right square is evaluated
    Image.fromarray(sheet).save(out / "a.png")
    mapping = {"000": dict(image_path="0_x/a.png", original_prompt="a [cat]", editing_prompt="a [dog]", editing_type_id="0",
                           mask=[100, 50]),
               "001": dict(image_path="0_x/a.png", original_prompt="a", editing_prompt="b", editing_type_id="7", mask=[])}
    mf = tmp_path / "data" / "mapping_file.json"
    json.dump(mapping, open(mf, "w"))
    res = tmp_path / "results.csv"
    n = EV.main(["--annotation_mapping_file", str(mf), "--src_image_folder", str(tmp_path / "data" / "annotation_images"),
                 "--tgt_methods", "h_edit", "--tgt_folders", str(tmp_path / "res"), "--result_path", str(res),
                 "--metrics", "psnr_unedit_part", "mse", "ssim_edit_part", "--edit_category_list", "0"])
    assert n == 1
    rows = list(csv.reader(open(res)))
    assert rows[0] == ["file_id", "h_edit|psnr_unedit_part", "h_edit|mse", "h_edit|ssim_edit_part"]
    assert rows[1][0] == "000" and float(rows[1][2]) == 0.0 and abs(float(rows[1][3]) - 1.0) < 1e-6
