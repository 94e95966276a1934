"""Mask helpers of the face driver.  Same behaviour as the two helpers of the reference's
face-swapping/arcface/face_utils.py -- ``encode_segmentation`` (:5-24) and ``SoftErosion`` (:26-51) -- written
independently: the label image of the face-parsing network becomes three binary maps, and the soft erosion
shrinks a mask with an iterated min against a cone-weighted blur.  Host-side torch, once per image."""
import torch
import torch.nn.functional as F

_FACE_IDS = (1, 2, 3, 4, 5, 6, 7, 10, 11, 12)
_FACE_IDS_WITH_NECK = (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 14)
_MOUTH_ID, _HAIR_ID = 10, 13


def encode_segmentation(segmentation, no_neck=True):
    """(B,1,H,W) class ids -> (B,3,H,W): face parts, mouth, hair, each 0/1 in the dtype of the input."""
    ids = torch.tensor(_FACE_IDS if no_neck else _FACE_IDS_WITH_NECK, device=segmentation.device)
    planes = (torch.isin(segmentation, ids), segmentation == _MOUTH_ID, segmentation == _HAIR_ID)
    return torch.cat([p.to(segmentation.dtype) for p in planes], dim=1)


def _cone_kernel(size):
    r = size // 2
    ax = torch.arange(size, dtype=torch.float32) - r
    dist = torch.hypot(ax[None, :], ax[:, None])
    w = dist.max() - dist
    return (w / w.sum())[None, None]


class SoftErosion(torch.nn.Module):
    """forward(mask) -> (soft, hard): ``iterations - 1`` rounds of min(mask, blur(mask)), one more blur, then
    everything >= threshold becomes 1 and the rest is scaled so that its maximum is 1."""

    def __init__(self, kernel_size=15, threshold=0.6, iterations=1):
        super().__init__()
        self.pad, self.rounds, self.threshold = kernel_size // 2, iterations, threshold
        self.register_buffer("weight", _cone_kernel(kernel_size))

    def _blur(self, m):
        return F.conv2d(m, self.weight.expand(m.shape[1], -1, -1, -1), groups=m.shape[1], padding=self.pad)

    def forward(self, x):
        m = x.float()
        for _ in range(self.rounds - 1):
            m = torch.minimum(m, self._blur(m))
        m = self._blur(m)
        hard = m >= self.threshold
        soft = torch.where(hard, torch.ones_like(m), m / m[~hard].max())
        return soft, hard
