"""Image I/O either side of the editing loop (host side, PIL / numpy).

Mirrors the helpers the reference drivers call around the VAE:
``load_512`` (text-guided/p2p/ptp_classes.py:351-373, called at main_p2p.py:154), and
``tensor_to_pil`` / ``image_grid`` (text-guided/utils/utils.py:19-25, 48-85, called at
main_p2p.py:266).  Same names, argument meaning and pixel arithmetic; no plotting helpers.
"""
import numpy as np
import torch
from PIL import Image, ImageDraw


def load_512(image_path, left=0, right=0, top=0, bottom=0, device=None):
    """RGB image (path or HxWx3 uint8 array) -> float tensor (1, 3, 512, 512) in [-1, 1].

    Crop by the four offsets, centre-crop the longer side to a square, resize to 512x512 with
    PIL's default filter, scale by 1/127.5 - 1.  The offset clamps follow the reference exactly,
    including its use of ``left`` in the clamp of ``top`` (ptp_classes.py:358-360).
    """
    if isinstance(image_path, str):
        img = np.array(Image.open(image_path).convert("RGB"))[:, :, :3]
    else:
        img = image_path
    h, w, _ = img.shape
    left = min(left, w - 1)
    right = min(right, w - left - 1)
    top = min(top, h - left - 1)
    bottom = min(bottom, h - top - 1)
    img = img[top:h - bottom, left:w - right]
    h, w, _ = img.shape
    if h < w:
        o = (w - h) // 2
        img = img[:, o:o + h]
    elif w < h:
        o = (h - w) // 2
        img = img[o:o + w]
    img = np.array(Image.fromarray(img).resize((512, 512)))
    t = torch.from_numpy(img).float() / 127.5 - 1
    return t.permute(2, 0, 1).unsqueeze(0).to(device)


def tensor_to_pil(tensor_imgs):
    """(B, 3, H, W) in [-1, 1] (or a list of such) -> list of PIL images.  Uses ToPILImage's
    arithmetic: clamp(x / 2 + 0.5, 0, 1) * 255 truncated to uint8."""
    if isinstance(tensor_imgs, list):
        tensor_imgs = torch.cat(tensor_imgs)
    x = (tensor_imgs.detach().float().cpu() / 2 + 0.5).clamp(0, 1)
    arr = x.mul(255).byte().permute(0, 2, 3, 1).numpy()
    return [Image.fromarray(a) for a in arr]


def image_grid(imgs, rows=1, cols=None, size=None, titles=None, text_pos=(0, 0)):
    """Paste images (tensor batch, list of tensors or list of PIL images) on a rows x cols sheet.
    With ``titles`` every cell gets a 20-pixel white strip on top carrying its title."""
    if isinstance(imgs, list) and isinstance(imgs[0], torch.Tensor):
        imgs = torch.cat(imgs)
    if isinstance(imgs, torch.Tensor):
        imgs = tensor_to_pil(imgs)
    if size is not None:
        imgs = [im.resize((size, size)) for im in imgs]
    if cols is None:
        cols = len(imgs)
    assert len(imgs) >= rows * cols
    top = 20
    w, h = imgs[0].size
    delta = 0
    if len(imgs) > 1 and imgs[1].size[1] != h:
        delta = top
        h = imgs[1].size[1]
    if titles is not None:
        h = top + h
    grid = Image.new("RGB", size=(cols * w, rows * h + delta))
    for i, im in enumerate(imgs):
        if titles is not None:
            cell = Image.new(im.mode, (im.size[0], im.size[1] + top), (255, 255, 255))
            cell.paste(im, (0, top))
            ImageDraw.Draw(cell).text(text_pos, titles[i], (0, 0, 0))
            im = cell
        y = i // cols * h + (delta if delta and i > 0 else 0)
        grid.paste(im, box=(i % cols * w, y))
    return grid


def dataset_from_json(json_location):
    """text-guided-n-style/utils/utils.py:107-111."""
    import json
    with open(json_location, 'r') as stream:
        return json.load(stream)
