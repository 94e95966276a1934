"""Checkpoint-directory plumbing (hedit/checkpoint.py) on the CPU."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit import checkpoint as CK  # noqa: E402


def test_component_round_trip(tmp_path):
    sd = {"a.weight": torch.randn(4, 3), "b.bias": torch.arange(5, dtype=torch.float32)}
    CK.write_component(str(tmp_path / "unet"), {"in_channels": 4, "block_out_channels": [64, 128]}, sd)
    cfg, back = CK.read_component(str(tmp_path / "unet"))
    assert cfg["block_out_channels"] == [64, 128]
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)
    os.remove(tmp_path / "unet" / "diffusion_pytorch_model.safetensors")
    torch.save(sd, tmp_path / "unet" / "diffusion_pytorch_model.bin")
    assert torch.equal(CK.read_component(str(tmp_path / "unet"))[1]["a.weight"], sd["a.weight"])
    os.remove(tmp_path / "unet" / "diffusion_pytorch_model.bin")
    with pytest.raises(FileNotFoundError):
        CK.read_component(str(tmp_path / "unet"))


def test_unet_config_filter():
    sd14 = dict(in_channels=4, out_channels=4, sample_size=64, block_out_channels=[320, 640, 1280, 1280],
                down_block_types=["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
                up_block_types=["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3, layers_per_block=2,
                cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32, act_fn="silu",
                use_linear_projection=False, only_cross_attention=False, _class_name="UNet2DConditionModel")
    c = CK.unet_config(sd14)
    assert c["block_out_channels"] == (320, 640, 1280, 1280) and "act_fn" not in c
    with pytest.raises(NotImplementedError):
        CK.unet_config(dict(sd14, use_linear_projection=True))                  # SD-2.x
    with pytest.raises(NotImplementedError):
        CK.unet_config(dict(sd14, attention_head_dim=[5, 10, 20, 20]))
    assert CK.unet_config(dict(sd14, attention_head_dim=[8, 8, 8, 8]))["attention_head_dim"] == 8


def test_scheduler_kwargs(tmp_path):
    assert CK.scheduler_kwargs(str(tmp_path)) == {}
    with open(tmp_path / "scheduler_config.json", "w") as f:
        json.dump(dict(beta_start=0.001, steps_offset=1, skip_prk_steps=True), f)
    assert CK.scheduler_kwargs(str(tmp_path)) == dict(beta_start=0.001, steps_offset=1)


def test_driver_flags_match_the_reference_cli():
    """h-edit_amd/main_p2p.py keeps the flag names and defaults of the reference driver
    (text-guided/main_p2p.py:38-70); the additions are separate flags."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("hedit_main_p2p_cli", os.path.join(ROOT, "h-edit_amd", "main_p2p.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = vars(m.build_parser().parse_args([]))
    reference_defaults = dict(device_num=0, data_path="./PIE_Bench_Data", output_path="./results/p2p",
                              edit_category_list=[str(i) for i in range(10)], mode="h_edit_R_p2p",
                              num_diffusion_steps=50, skip=0, eta=1.0, cfg_src=1.0, cfg_src_edit=5.0, cfg_tar=7.5,
                              implicit=False, optimization_steps=1, weight_reconstruction=0.1, xa=0.4, sa=0.35)
    for k, v in reference_defaults.items():
        assert a[k] == v, k
    assert set(a) - set(reference_defaults) == {"model_path", "random_init", "tiny", "seed", "batch"}
    assert a["batch"] == 1
    b = vars(m.build_parser().parse_args(["--implicit", "--mode", "h_edit_D_p2p", "--eta", "0.0", "--edit_category_list", "0", "3"]))
    assert b["implicit"] is True and b["eta"] == 0.0 and b["edit_category_list"] == ["0", "3"]


def test_style_driver_flags_match_the_reference_cli():
    """h-edit_amd/main_edit.py keeps the flag names and defaults of the reference's
    text-guided-n-style/main_edit.py:33-73 (note --implicit is store_false there: implicit by default)."""
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
    spec = importlib.util.spec_from_file_location("hedit_main_edit_cli", os.path.join(ROOT, "h-edit_amd", "main_edit.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = vars(m.build_parser().parse_args([]))
    reference_defaults = dict(device_num=0, dataset="./assets/demo/", output_path="./results/demo/", mode="h_edit_R_p2p",
                              num_diffusion_steps=50, skip=0, eta=1.0, cfg_src=1.0, cfg_src_edit=5.0, cfg_tar=7.5,
                              implicit=True, optimization_steps=1, xa=0.4, sa=0.35, weight_edit_clip=0.5,
                              weight_edit_clip_for_ef=1.5)
    for k, v in reference_defaults.items():
        assert a[k] == v, k
    assert set(a) - set(reference_defaults) == {"model_path", "clip_path", "random_init", "tiny", "seed", "batch"}


def test_masactrl_driver_flags_match_the_reference_cli():
    """h-edit_amd/main_masactrl.py keeps the flag names and defaults of text-guided/main_masactrl.py:55-88."""
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
    spec = importlib.util.spec_from_file_location("hedit_main_masactrl_cli", os.path.join(ROOT, "h-edit_amd", "main_masactrl.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = vars(m.build_parser().parse_args([]))
    reference_defaults = dict(device_num=0, data_path="./PIE_Bench_Data", output_path="./results/masactrl",
                              edit_category_list=[str(i) for i in range(10)], mode="h_edit_D_masactrl", num_diffusion_steps=50,
                              skip=0, eta=0.0, cfg_src=1.0, cfg_src_edit=5.0, cfg_tar=7.5, implicit=False, optimization_steps=1,
                              weight_reconstruction=0.1, layer=10, step=4)
    for k, v in reference_defaults.items():
        assert a[k] == v, k
    assert set(a) - set(reference_defaults) == {"model_path", "random_init", "tiny", "seed", "batch"}
    assert a["batch"] == 1


def test_pnp_driver_flags_match_the_reference_cli():
    """h-edit_amd/main_plugnplay.py keeps the flag names and defaults of text-guided/main_plugnplay.py:56-87."""
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
    spec = importlib.util.spec_from_file_location("hedit_main_pnp_cli", os.path.join(ROOT, "h-edit_amd", "main_plugnplay.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = vars(m.build_parser().parse_args([]))
    reference_defaults = dict(device_num=0, data_path="./PIE_Bench_Data", output_path="./results/pnp",
                              edit_category_list=[str(i) for i in range(10)], mode="h_edit_R_pnp", num_diffusion_steps=50,
                              skip=0, eta=1.0, cfg_src=1.0, cfg_src_edit=5.0, cfg_tar=7.5, implicit=False, optimization_steps=1,
                              weight_reconstruction=0.1, pnp_f_t=0.45, pnp_attn_t=0.35)
    for k, v in reference_defaults.items():
        assert a[k] == v, k
    assert set(a) - set(reference_defaults) == {"model_path", "random_init", "tiny", "seed", "batch"}
    assert a["batch"] == 1


def test_demo_driver_flags_match_the_reference_cli():
    """h-edit_amd/main_demo.py keeps the flag names and defaults of text-guided/main_demo.py:49-84."""
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
    spec = importlib.util.spec_from_file_location("hedit_main_demo_cli", os.path.join(ROOT, "h-edit_amd", "main_demo.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = vars(m.build_parser().parse_args([]))
    reference_defaults = dict(device_num=0, data_path="./assets/demo", output_path="./results/demo", mode="h_edit_R_p2p",
                              num_diffusion_steps=50, skip=0, eta=1.0, cfg_src=1.0, cfg_src_edit=5.0, cfg_tar=7.5, implicit=False,
                              optimization_steps=1, weight_reconstruction=0.1, xa=0.4, sa=0.35)
    for k, v in reference_defaults.items():
        assert a[k] == v, k
    assert set(a) - set(reference_defaults) == {"model_path", "random_init", "tiny", "seed", "batch"}
    assert a["batch"] == 1


def test_face_driver_flags_match_the_reference_cli():
    """h-edit_amd/main_edit_face.py keeps the flag names and defaults of face-swapping/main_edit.py:35-60
    (--post_processing is store_false there: on by default)."""
    import importlib.util
    sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
    spec = importlib.util.spec_from_file_location("hedit_main_face_cli", os.path.join(ROOT, "h-edit_amd", "main_edit_face.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    a = vars(m.build_parser().parse_args([]))
    reference_defaults = dict(device_num=0, json_file="./assets/demo/demo.json", image_path="./assets/demo/",
                              output_path="./results/demo/", mode="h_edit_R", num_diffusion_steps=100, skip=0, eta=1.0,
                              optimization_steps=3, post_processing=True, weight_edit_face=50.0)
    for k, v in reference_defaults.items():
        assert a[k] == v, k
    assert set(a) - set(reference_defaults) == {"ddpm_ckpt", "arcface_ckpt", "mask_dir", "random_init", "tiny", "seed", "batch", "lpips_ckpt", "no_lpips"}


def test_soft_erosion_and_segmentation_encoding():
    """face_utils: class ids -> (face, mouth, hair) maps; SoftErosion shrinks the mask, saturates the interior at 1"""
    import torch
    from hedit.arcface.face_utils import SoftErosion, encode_segmentation
    seg = torch.zeros(1, 1, 32, 32, dtype=torch.long)
    seg[..., 8:24, 8:24] = 1
    seg[..., 18:21, 12:20] = 10
    seg[..., 0:4, :] = 13
    enc = encode_segmentation(seg)
    assert enc.shape == (1, 3, 32, 32)
    assert int(enc[0, 0].sum()) == 16 * 16 and int(enc[0, 1].sum()) == 3 * 8 and int(enc[0, 2].sum()) == 4 * 32
    soft, hard = SoftErosion(kernel_size=5, threshold=0.9, iterations=2)(enc[:, 0, None].float())
    assert soft.max() == 1.0 and soft.min() >= 0.0 and hard.sum() < 16 * 16 and soft[0, 0, 16, 16] == 1.0
    assert soft[0, 0, 0, 0] == 0.0
