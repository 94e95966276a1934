"""``LPIPS_Loss`` of the reference's face-swapping/arcface/arcface_model.py:69-94: a thin wrapper over the third-party
``lpips.LPIPS(net='vgg')`` against the source image.  Importing this module needs the ``lpips`` package (absent in the
offline build image: the face driver then runs with ``lpipsloss=None``, which the loop guards like the reference)."""
import lpips
import torch.nn as nn


class LPIPS_Loss(nn.Module):
    def __init__(self, src_path=None, src=None):
        super().__init__()
        self.lpips_loss = lpips.LPIPS(net='vgg')
        if src is None:
            from .arcface_model import load_face_image
            src = load_face_image(src_path)
        self.register_buffer("src", src.float())

    def get_lpips_loss(self, x):
        return self.lpips_loss(x, self.src).mean()
