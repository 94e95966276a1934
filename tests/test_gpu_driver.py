"""End-to-end on the GPU: checkpoint directory -> pipeline, and the main_p2p.py driver from a
PIE-Bench-style mapping file to edited PNGs (reference text-guided/main_p2p.py:110-275)."""
import importlib.util
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import gpu as G  # noqa: E402

pytestmark = pytest.mark.gpu


def _driver():
    spec = importlib.util.spec_from_file_location("hedit_main_p2p", os.path.join(ROOT, "h-edit_amd", "main_p2p.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _dataset(tmp_path):
    from PIL import Image
    d = tmp_path / "data"
    (d / "annotation_images" / "0_random").mkdir(parents=True)
    y, x = np.mgrid[0:96, 0:128]
    mapping = {}
    for i, (src, tar, blend, cat) in enumerate([("a cat sitting on a bench", "a dog sitting on a bench", "cat dog", "0"),
                                                 ("a [red] car", "a [blue] car on a road", "", "1"),
                                                 ("a tree", "a tall tree", "tree tree", "7")]):
        img = np.stack([(x * (i + 2)) % 256, (y * 3 + i * 40) % 256, (x + y) % 256], -1).astype(np.uint8)
        rel = f"0_random/{i:012d}.png"
        Image.fromarray(img).save(d / "annotation_images" / rel)
        mapping[f"{i:012d}"] = dict(image_path=rel, original_prompt=src, editing_prompt=tar, editing_instruction="",
                                    editing_type_id=cat, blended_word=blend)
    with open(d / "mapping_file.json", "w") as f:
        json.dump(mapping, f)
    return d


@pytest.mark.parametrize("extra", [["--implicit", "--optimization_steps", "2"], [],
                                   ["--mode", "h_edit_D_p2p", "--eta", "0.0", "--implicit", "--sa", "0.6"],
                                   ["--mode", "h_edit_R", "--implicit"]])
def test_driver_writes_edited_images(tmp_path, extra):
    from PIL import Image
    d = _dataset(tmp_path)
    out = tmp_path / "results"
    written = _driver().main(["--data_path", str(d), "--output_path", str(out), "--random_init", "--tiny",
                              "--num_diffusion_steps", "4", "--edit_category_list", "0", "1"] + extra)
    assert len(written) == 2                       # category 7 filtered out
    for p in written:
        assert p.startswith(str(out)) and "_total_steps_4_skip_0_" in p and os.path.exists(p)
        im = np.array(Image.open(p))
        assert im.shape == (256, 256, 3) and im.std() > 0


def test_driver_batch_flag_edits_in_lock_step_with_identical_results(tmp_path):
    """--batch 2: the two entries go through the batched engine (VAE encode, DDIM inversion, loop, decode in 2-image
    launches).  The kernels are batch-invariant, so the PNGs are byte-identical to the one-image-at-a-time run
    (h-Edit-D: the DDIM inversion draws no random numbers)."""
    from PIL import Image
    d = _dataset(tmp_path)
    common = ["--data_path", str(d), "--random_init", "--tiny", "--num_diffusion_steps", "4", "--edit_category_list", "0", "1",
              "--mode", "h_edit_D_p2p", "--eta", "0.0", "--implicit", "--optimization_steps", "2"]
    one = _driver().main(common + ["--output_path", str(tmp_path / "r1")])
    two = _driver().main(common + ["--output_path", str(tmp_path / "r2"), "--batch", "2"])
    assert len(one) == len(two) == 2
    for a, b in zip(sorted(one), sorted(two)):
        assert os.path.basename(a) == os.path.basename(b)
        assert np.array_equal(np.array(Image.open(a)), np.array(Image.open(b)))


def test_driver_refuses_baselines(tmp_path):
    with pytest.raises(NotImplementedError):
        _driver().main(["--data_path", str(_dataset(tmp_path)), "--random_init", "--tiny", "--mode", "ef_p2p"])


def test_from_pretrained_round_trip(tmp_path):
    """weights written in the diffusers directory layout load back into executors that compute
    exactly what the original ones do"""
    from hedit.pipeline import HEditPipeline
    from hedit.unet import TINY_CONFIG, UNet2DConditionModel
    from hedit.vae import TINY_VAE_CONFIG, AutoencoderKL
    from hedit.text import ClipTextEncoder, WordTokenizer
    dev = G.dev()
    unet = UNet2DConditionModel(TINY_CONFIG, device=dev)
    usd = unet.init_random(5)
    vae = AutoencoderKL(TINY_VAE_CONFIG, device=dev)
    vsd = vae.init_random(6)
    enc = ClipTextEncoder(dim=64, layers=1, heads=4, seed=1).to(dev)
    pipe = HEditPipeline(unet, None, WordTokenizer(), enc, vae, dev)
    pipe.save_pretrained(str(tmp_path / "ckpt"), usd, vsd)
    os.makedirs(tmp_path / "ckpt" / "scheduler")
    with open(tmp_path / "ckpt" / "scheduler" / "scheduler_config.json", "w") as f:
        json.dump(dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                       clip_sample=False, set_alpha_to_one=False, steps_offset=1, _class_name="PNDMScheduler"), f)
    back = HEditPipeline.from_pretrained(str(tmp_path / "ckpt"), device=dev, tokenizer=WordTokenizer(), text_encoder=enc)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 32, 32, generator=g).to(dev)
    ctx = torch.randn(2, 77, 64, generator=g).to(dev)
    a = unet(x, 500, encoder_hidden_states=ctx).sample
    b = back.unet(x, 500, encoder_hidden_states=ctx).sample
    G.sync()
    assert torch.equal(a, b)
    z = torch.randn(1, 4, 16, 16, generator=g).to(dev)
    assert torch.equal(vae.decode(z).sample, back.vae.decode(z).sample)
    assert back.scheduler.config.steps_offset == 1
    with pytest.raises(FileNotFoundError):
        HEditPipeline.from_pretrained(str(tmp_path / "nope"), device=dev)


def test_style_driver_writes_edited_images(tmp_path, capsys):
    """main_edit.py (reference text-guided-n-style/main_edit.py:103-247): demo.json with a style image per
    entry -> encode, DDPM inversion, text + style editing, decode, PNG per entry; prints the CLIP loss."""
    from PIL import Image
    spec = importlib.util.spec_from_file_location("hedit_main_edit", os.path.join(ROOT, "h-edit_amd", "main_edit.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    d = tmp_path / "demo"
    (d / "styles").mkdir(parents=True)
    y, x = np.mgrid[0:96, 0:128]
    data = {}
    for i, (src, tar, blend) in enumerate([("an orange van with surfboards on top", "an orange van with flowers on top",
                                            "surfboards flowers"),
                                           ("a round cake on a plate", "a square cake on a wooden plate", "")]):
        img = np.stack([(x * (i + 2)) % 256, (y * 3 + i * 40) % 256, (x + y) % 256], -1).astype(np.uint8)
        Image.fromarray(img).save(d / f"{i:012d}.jpg")
        sty = np.stack([(x * y + i) % 256, (x * 7) % 256, (y * 5) % 256], -1).astype(np.uint8)
        Image.fromarray(sty).save(d / "styles" / f"s{i}.png")
        data[f"{i:012d}"] = dict(image_path=f"{i:012d}.jpg", original_prompt=src, editing_prompt=tar,
                                 editing_instruction="", blended_word=blend, style=f"styles/s{i}.png")
    with open(d / "demo.json", "w") as f:
        json.dump(data, f)
    out = tmp_path / "results"
    written = drv.main(["--dataset", str(d) + "/", "--output_path", str(out), "--random_init", "--tiny",
                        "--num_diffusion_steps", "4", "--weight_edit_clip", "0.5"])
    assert len(written) == 2
    for p in written:
        assert p.startswith(str(out)) and "_w_style_0.5_" in p and os.path.exists(p)
        im = np.array(Image.open(p))
        assert im.shape == (256, 256, 3) and im.std() > 0
    assert capsys.readouterr().out.count("loss from CLIP:") == 2
    with pytest.raises(NotImplementedError):
        drv.main(["--dataset", str(d) + "/", "--random_init", "--tiny", "--mode", "ef_p2p"])


@pytest.mark.parametrize("extra", [[], ["--mode", "h_edit_R_masactrl", "--eta", "1.0", "--optimization_steps", "2"]])
def test_masactrl_driver_writes_edited_images(tmp_path, extra):
    """main_masactrl.py (reference text-guided/main_masactrl.py:128-241) on the PIE-Bench-style dataset"""
    from PIL import Image
    spec = importlib.util.spec_from_file_location("hedit_main_masactrl", os.path.join(ROOT, "h-edit_amd", "main_masactrl.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    d = _dataset(tmp_path)
    out = tmp_path / "results"
    written = drv.main(["--data_path", str(d), "--output_path", str(out), "--random_init", "--tiny", "--num_diffusion_steps", "4",
                        "--edit_category_list", "0", "7", "--step", "1", "--layer", "2"] + extra)
    assert len(written) == 2
    for p in written:
        assert "_step_1_layer_2_" in p and os.path.exists(p)
        im = np.array(Image.open(p))
        assert im.shape == (256, 256, 3) and im.std() > 0
    with pytest.raises(NotImplementedError):
        drv.main(["--data_path", str(d), "--random_init", "--tiny", "--mode", "ef_masactrl"])


@pytest.mark.parametrize("extra", [[], ["--mode", "h_edit_D_pnp", "--eta", "0.0", "--pnp_f_t", "0.6", "--pnp_attn_t", "0.4"]])
def test_pnp_driver_writes_edited_images(tmp_path, extra):
    """main_plugnplay.py (reference text-guided/main_plugnplay.py:124-248) on the PIE-Bench-style dataset"""
    from PIL import Image
    spec = importlib.util.spec_from_file_location("hedit_main_pnp", os.path.join(ROOT, "h-edit_amd", "main_plugnplay.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    d = _dataset(tmp_path)
    out = tmp_path / "results"
    written = drv.main(["--data_path", str(d), "--output_path", str(out), "--random_init", "--tiny", "--num_diffusion_steps", "4",
                        "--edit_category_list", "0"] + extra)
    assert len(written) == 1
    p = written[0]
    assert "_f_t_" in p and "_attn_t_" in p and os.path.exists(p)
    im = np.array(Image.open(p))
    assert im.shape == (256, 256, 3) and im.std() > 0
    with pytest.raises(NotImplementedError):
        drv.main(["--data_path", str(d), "--random_init", "--tiny", "--mode", "ef_pnp"])


def test_demo_driver_writes_edited_images(tmp_path):
    """main_demo.py (reference text-guided/main_demo.py:124-262): demo.yaml list -> edited PNGs"""
    import yaml
    from PIL import Image
    spec = importlib.util.spec_from_file_location("hedit_main_demo", os.path.join(ROOT, "h-edit_amd", "main_demo.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    d = tmp_path / "demo"
    d.mkdir()
    y, x = np.mgrid[0:96, 0:128]
    Image.fromarray(np.stack([(x * 3) % 256, (y * 5) % 256, (x + y) % 256], -1).astype(np.uint8)).save(d / "lizard.jpg")
    with open(d / "demo.yaml", "w") as f:
        yaml.safe_dump([dict(image="/lizard.jpg", source_prompt="a green lizard is sitting on a branch",
                             target_prompt="a brown lizard is sitting on a branch", blended_word="lizard lizard",
                             editing_instruction="Change the color of the lizard to brown")], f)
    out = tmp_path / "results"
    for extra in (["--implicit"], ["--mode", "h_edit_D_p2p", "--eta", "0.0", "--sa", "0.6"]):
        written = drv.main(["--data_path", str(d), "--output_path", str(out), "--random_init", "--tiny",
                            "--num_diffusion_steps", "4"] + extra)
        assert len(written) == 1 and written[0].endswith("lizard.jpg") and os.path.exists(written[0])
        assert np.array(Image.open(written[0])).shape == (256, 256, 3)


def _load_driver(fname):
    spec = importlib.util.spec_from_file_location("hedit_" + fname.replace(".", "_"), os.path.join(ROOT, "h-edit_amd", fname))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    return drv


@pytest.mark.parametrize("fname,extra", [
    ("main_masactrl.py", ["--edit_category_list", "0", "7", "--step", "1", "--layer", "2", "--optimization_steps", "2"]),
    ("main_plugnplay.py", ["--edit_category_list", "0", "1", "--mode", "h_edit_D_pnp", "--eta", "0.0", "--pnp_f_t", "0.6",
                           "--pnp_attn_t", "0.4"])])
def test_masactrl_and_pnp_drivers_batch_flag(tmp_path, fname, extra):
    """--batch 2 on the MasaCtrl and Plug-and-Play drivers: two entries in lock-step on the batched engine; h-Edit-D draws no
    random numbers and the kernels are batch-invariant, so the PNGs are byte-identical to the one-at-a-time run."""
    from PIL import Image
    drv = _load_driver(fname)
    d = _dataset(tmp_path)
    common = ["--data_path", str(d), "--random_init", "--tiny", "--num_diffusion_steps", "4"] + extra
    one = drv.main(common + ["--output_path", str(tmp_path / "r1")])
    two = drv.main(common + ["--output_path", str(tmp_path / "r2"), "--batch", "2"])
    assert len(one) == len(two) == 2
    for a, b in zip(sorted(one), sorted(two)):
        assert os.path.basename(a) == os.path.basename(b)
        assert np.array_equal(np.array(Image.open(a)), np.array(Image.open(b)))


def test_demo_driver_batch_flag(tmp_path):
    """--batch 2 on the demo driver (its own replace / equalizer rules handed to the lock-step group)"""
    import yaml
    from PIL import Image
    drv = _load_driver("main_demo.py")
    d = tmp_path / "demo"
    d.mkdir()
    y, x = np.mgrid[0:96, 0:128]
    Image.fromarray(np.stack([(x * 3) % 256, (y * 5) % 256, (x + y) % 256], -1).astype(np.uint8)).save(d / "lizard.png")
    Image.fromarray(np.stack([(x * 7) % 256, (y * 2) % 256, (x * y) % 256], -1).astype(np.uint8)).save(d / "cat.png")
    with open(d / "demo.yaml", "w") as f:
        yaml.safe_dump([dict(image="/lizard.png", source_prompt="a green lizard is sitting on a branch",
                             target_prompt="a brown lizard is sitting on a branch", blended_word="lizard lizard",
                             editing_instruction=""),
                        dict(image="/cat.png", source_prompt="a cat", target_prompt="a cat wearing a big hat", blended_word="",
                             editing_instruction="")], f)
    common = ["--data_path", str(d), "--random_init", "--tiny", "--num_diffusion_steps", "4", "--mode", "h_edit_D_p2p",
              "--eta", "0.0", "--implicit"]
    one = drv.main(common + ["--output_path", str(tmp_path / "r1")])
    two = drv.main(common + ["--output_path", str(tmp_path / "r2"), "--batch", "2"])
    assert len(one) == len(two) == 2
    for a, b in zip(sorted(one), sorted(two)):
        assert os.path.basename(a) == os.path.basename(b)
        assert np.array_equal(np.array(Image.open(a)), np.array(Image.open(b)))


def test_face_driver_writes_swapped_images(tmp_path, capsys):
    """main_edit_face.py (reference face-swapping/main_edit.py:134-224): {idx, source, ref} pairs -> inversion, h_Edit_R with
    the identity reward, a [ref | source | result] sheet per pair; with and without the post-processing mask."""
    from PIL import Image
    spec = importlib.util.spec_from_file_location("hedit_main_face", os.path.join(ROOT, "h-edit_amd", "main_edit_face.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    d = tmp_path / "faces"
    (d / "masks").mkdir(parents=True)
    y, x = np.mgrid[0:80, 0:72]
    for i, name in enumerate(("1368.jpg", "7522.jpg")):
        Image.fromarray(np.stack([(x * (i + 2)) % 256, (y * 3 + i * 40) % 256, (x + y) % 256], -1).astype(np.uint8)).save(d / name)
    lab = np.zeros((32, 32), dtype=np.uint8)
    lab[8:24, 8:24] = 1
    lab[18:21, 12:20] = 10
    Image.fromarray(lab).save(d / "masks" / "7522.png")
    with open(d / "demo.json", "w") as f:
        json.dump([dict(idx=0, ref="1368.jpg", source="7522.jpg")], f)
    common = ["--json_file", str(d / "demo.json"), "--image_path", str(d) + "/", "--output_path", str(tmp_path / "out") + "/",
              "--random_init", "--tiny", "--num_diffusion_steps", "10", "--optimization_steps", "2", "--weight_edit_face", "4.0"]
    for extra in ([], ["--mask_dir", str(d / "masks")]):
        written = drv.main(common + extra)
        assert len(written) == 1 and written[0].endswith("item_1368_7522.png")
        assert "h_edit_R/steps_10_skip_0_weight_4.0_opts_2" in written[0]
        im = np.array(Image.open(written[0]))
        assert im.shape == (32, 96, 3) and im.std() > 0
    assert capsys.readouterr().out.count("Cosine Similarity:") == 2
    with pytest.raises(NotImplementedError):
        drv.main(common + ["--mode", "ef"])


def test_face_driver_batch_flag_swaps_pairs_in_lock_step(tmp_path):
    """--batch 2: two (source, reference) pairs through h_Edit_R in one lock-step batch, each with its own reference
    face (identity reward) and source image (LPIPS); the result sheets equal the pair-by-pair run (the eps-network and
    both reward networks are batch-invariant; with two images the 1/n and n factors of the batch mean are exact)."""
    from PIL import Image
    spec = importlib.util.spec_from_file_location("hedit_main_face", os.path.join(ROOT, "h-edit_amd", "main_edit_face.py"))
    drv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(drv)
    d = tmp_path / "faces"
    d.mkdir(parents=True)
    y, x = np.mgrid[0:80, 0:72]
    for i, name in enumerate(("a.jpg", "b.jpg", "c.jpg")):
        Image.fromarray(np.stack([(x * (i + 2)) % 256, (y * 3 + i * 40) % 256, (x + y * (i + 1)) % 256], -1).astype(np.uint8)).save(d / name)
    with open(d / "demo.json", "w") as f:
        json.dump([dict(idx=0, ref="a.jpg", source="b.jpg"), dict(idx=1, ref="c.jpg", source="a.jpg")], f)
    common = ["--json_file", str(d / "demo.json"), "--image_path", str(d) + "/", "--random_init", "--tiny",
              "--num_diffusion_steps", "8", "--optimization_steps", "2", "--weight_edit_face", "4.0"]
    one = drv.main(common + ["--output_path", str(tmp_path / "o1") + "/"])
    two = drv.main(common + ["--output_path", str(tmp_path / "o2") + "/", "--batch", "2"])
    assert len(one) == len(two) == 2
    for a, b in zip(sorted(one), sorted(two)):
        assert os.path.basename(a) == os.path.basename(b)
        ia, ib = np.array(Image.open(a)).astype(np.int32), np.array(Image.open(b)).astype(np.int32)
        assert ia.shape == ib.shape and np.abs(ia - ib).max() <= 1
