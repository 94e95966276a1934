#!/usr/bin/env python3
"""PIE-Bench evaluation driver -- the CLI, dataset handling and CSV format of the reference's
text-guided/evaluation/evaluation.py (:9-25 run-length mask decoding, :27-92 metric dispatch, :103-215 main) for the
outputs of h-edit_amd/main_p2p.py.

Metrics.  The pixel metrics are computed here with the formulas of the torchmetrics classes the reference
instantiates (evaluation/matrics_calculator.py:272-279): ``psnr`` (PeakSignalNoiseRatio(data_range=1)), ``mse``
(MeanSquaredError), ``ssim`` (StructuralSimilarityIndexMeasure(data_range=1): 11 x 11 Gaussian window, sigma 1.5,
k1 0.01, k2 0.03, reflect padding, mean over the interior), each on the whole image / the edited part / the unedited
part exactly as the reference masks them (image * mask before the metric).  The network metrics need third-party
checkpoints that do not exist offline and are not part of the sampling path (SURVEY.md section 8 row f4 "then the
evaluator"): ``lpips*`` (torchmetrics LPIPS, SqueezeNet), ``clip_similarity_*`` / ``local_clip`` (CLIP ViT-L/14),
``structure_distance*`` (DINO ViT-B/8 self-similarity) -- asking for one of them raises with the name of the
missing checkpoint instead of writing a made-up number.
"""
import argparse
import csv
import json
import os

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

PIXEL_METRICS = ("psnr", "mse", "ssim")
NETWORK_METRICS = {"lpips": "torchmetrics LPIPS (SqueezeNet) weights", "structure_distance": "DINO ViT-B/8 weights",
                   "clip_similarity_source_image": "CLIP ViT-L/14 weights", "clip_similarity_target_image": "CLIP ViT-L/14 weights",
                   "clip_similarity_target_image_edit_part": "CLIP ViT-L/14 weights", "local_clip": "CLIP ViT-B/32 weights"}


def mask_decode(encoded_mask, image_shape=(512, 512)):
    """PIE-Bench run-length mask: pairs (start, length) over the flattened image; the one-pixel border is forced to 1
    ("to avoid annotation errors in boundary", evaluation.py:9-25)."""
    length = image_shape[0] * image_shape[1]
    flat = np.zeros((length,))
    enc = np.asarray(encoded_mask, dtype=np.int64).reshape(-1, 2) if len(encoded_mask) else np.zeros((0, 2), dtype=np.int64)
    for start, run in enc:
        flat[start:start + max(0, min(run, length - start))] = 1
    m = flat.reshape(image_shape[0], image_shape[1])
    m[0, :] = m[-1, :] = 1
    m[:, 0] = m[:, -1] = 1
    return m


def _pair(img_pred, img_gt, mask_pred, mask_gt, scale=255.0):
    a = np.array(img_pred).astype(np.float32) / scale
    b = np.array(img_gt).astype(np.float32) / scale
    assert a.shape == b.shape, "Image shapes should be the same."
    if mask_pred is not None:
        a = a * np.array(mask_pred).astype(np.float32)
    if mask_gt is not None:
        b = b * np.array(mask_gt).astype(np.float32)
    return torch.tensor(a).permute(2, 0, 1)[None], torch.tensor(b).permute(2, 0, 1)[None]


def _gauss(size=11, sigma=1.5):
    d = torch.arange((1 - size) / 2, (1 + size) / 2, dtype=torch.float32)
    g = torch.exp(-(d / sigma) ** 2 / 2)
    g = g / g.sum()
    return (g[:, None] @ g[None, :])


class MetricsCalculator:
    """The pixel-metric half of the reference's MetricsCalculator (matrics_calculator.py:271-390), same method names."""

    def __init__(self, device="cpu"):
        self.device = device

    def calculate_mse(self, img_pred, img_gt, mask_pred=None, mask_gt=None):
        a, b = _pair(img_pred, img_gt, mask_pred, mask_gt)
        return float(((a.to(self.device) - b.to(self.device)) ** 2).mean().item())

    def calculate_psnr(self, img_pred, img_gt, mask_pred=None, mask_gt=None):
        a, b = _pair(img_pred, img_gt, mask_pred, mask_gt)
        mse = ((a.to(self.device) - b.to(self.device)) ** 2).mean()
        return float((10.0 * torch.log10(1.0 / mse)).item())                 # data_range = 1

    def calculate_ssim(self, img_pred, img_gt, mask_pred=None, mask_gt=None):
        a, b = _pair(img_pred, img_gt, mask_pred, mask_gt)
        a, b = a.to(self.device), b.to(self.device)
        C = a.shape[1]
        k = _gauss().to(self.device)[None, None].expand(C, 1, 11, 11).contiguous()
        pad = 5
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        ap, bp = F.pad(a, (pad,) * 4, mode="reflect"), F.pad(b, (pad,) * 4, mode="reflect")
        stack = torch.cat([ap, bp, ap * ap, bp * bp, ap * bp])
        out = F.conv2d(stack, k, groups=C)
        mu_a, mu_b, saa, sbb, sab = out.split(1)
        va, vb, vab = saa - mu_a ** 2, sbb - mu_b ** 2, sab - mu_a * mu_b
        ssim = ((2 * mu_a * mu_b + c1) * (2 * vab + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (va + vb + c2))
        return float(ssim[..., pad:-pad, pad:-pad].mean().item())             # torchmetrics crops the padded border again


def calculate_metric(mc, metric, src_image, tgt_image, src_mask, tgt_mask, src_prompt, tgt_prompt):
    base = metric
    part = None
    for suffix in ("_unedit_part", "_edit_part"):
        if metric.endswith(suffix) and not metric.startswith("clip_similarity"):
            base, part = metric[:-len(suffix)], suffix
    if base in NETWORK_METRICS or metric in NETWORK_METRICS:
        raise NotImplementedError(f"metric {metric}: needs {NETWORK_METRICS.get(base, NETWORK_METRICS.get(metric))}, which this offline "
                                  "build does not have; pixel metrics: " + ", ".join(PIXEL_METRICS))
    if base not in PIXEL_METRICS:
        raise ValueError(f"unknown metric {metric}")
    fn = getattr(mc, "calculate_" + base)
    if part is None:
        return fn(src_image, tgt_image, None, None)
    if part == "_unedit_part":
        if (1 - src_mask).sum() == 0 or (1 - tgt_mask).sum() == 0:
            return "nan"
        return fn(src_image, tgt_image, 1 - src_mask, 1 - tgt_mask)
    if src_mask.sum() == 0 or tgt_mask.sum() == 0:
        return "nan"
    return fn(src_image, tgt_image, src_mask, tgt_mask)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--annotation_mapping_file', type=str, default="./PIE_Bench_Data/mapping_file.json")
    p.add_argument('--metrics', nargs='+', type=str, default=["psnr_unedit_part", "mse_unedit_part", "ssim_unedit_part"])
    p.add_argument('--src_image_folder', type=str, default="./PIE_Bench_Data/annotation_images")
    p.add_argument('--tgt_methods', nargs='+', type=str, default=["your_method"])
    p.add_argument('--tgt_folders', nargs='+', type=str, default=None,
                   help="one result directory per entry of --tgt_methods (the reference hard-codes this table in the script)")
    p.add_argument('--result_path', type=str, default="./results/results.csv")
    p.add_argument('--device', type=str, default="cpu")
    p.add_argument('--edit_category_list', nargs='+', type=str, default=[str(i) for i in range(10)])
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not args.tgt_folders or len(args.tgt_folders) != len(args.tgt_methods):
        raise SystemExit("give --tgt_folders DIR ... (one per method)")
    folders = dict(zip(args.tgt_methods, args.tgt_folders))
    mc = MetricsCalculator(args.device)
    os.makedirs(os.path.dirname(os.path.abspath(args.result_path)), exist_ok=True)
    with open(args.result_path, 'w', newline="") as f:
        csv.writer(f).writerow(["file_id"] + [f"{k}|{m}" for k in folders for m in args.metrics])
    with open(args.annotation_mapping_file) as f:
        annotation = json.load(f)
    rows = 0
    for key, item in annotation.items():
        if item["editing_type_id"] not in args.edit_category_list:
            continue
        src_image = Image.open(os.path.join(args.src_image_folder, item["image_path"])).convert("RGB")
        mask = mask_decode(item.get("mask", []), image_shape=(src_image.size[1], src_image.size[0]))
        mask = mask[:, :, np.newaxis].repeat([3], axis=2)
        src_p = item["original_prompt"].replace("[", "").replace("]", "")
        tgt_p = item["editing_prompt"].replace("[", "").replace("]", "")
        row = [key]
        for name, folder in folders.items():
            tgt = Image.open(os.path.join(folder, item["image_path"])).convert("RGB")
            if tgt.size[0] != tgt.size[1]:       # result sheets: the edited image is the right-most square (evaluation.py:203-205)
                s = src_image.size[0]
                tgt = tgt.crop((tgt.size[0] - s, tgt.size[1] - s, tgt.size[0], tgt.size[1]))
            for metric in args.metrics:
                row.append(calculate_metric(mc, metric, src_image, tgt, mask, mask, src_p, tgt_p))
        with open(args.result_path, 'a+', newline="") as f:
            csv.writer(f).writerow(row)
        rows += 1
    print(f"evaluated {rows} image(s) -> {args.result_path}")
    return rows


if __name__ == "__main__":
    main()
