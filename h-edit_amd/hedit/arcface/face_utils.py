"""Mask helpers of the face driver -- mirrors face-swapping/arcface/face_utils.py: ``encode_segmentation`` (:5-24,
face / mouth / hair maps from a face-parsing label image) and ``SoftErosion`` (:26-51, iterated min with a cone
kernel, threshold, renormalise).  Host-side torch ops, once per image (outside the sampling loop)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


def encode_segmentation(segmentation, no_neck=True):
    face_ids = [1, 2, 3, 4, 5, 6, 7, 10, 11, 12] if no_neck else [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 14]
    face = torch.zeros_like(segmentation)
    for i in face_ids:
        face[segmentation == i] = 1
    mouth = (segmentation == 10).to(segmentation.dtype)
    hair = (segmentation == 13).to(segmentation.dtype)
    return torch.cat([face, mouth, hair], axis=1)


class SoftErosion(nn.Module):
    def __init__(self, kernel_size=15, threshold=0.6, iterations=1):
        super().__init__()
        r = kernel_size // 2
        self.padding, self.iterations, self.threshold = r, iterations, threshold
        yy, xx = torch.meshgrid(torch.arange(0., kernel_size), torch.arange(0., kernel_size), indexing="ij")
        dist = torch.sqrt((xx - r) ** 2 + (yy - r) ** 2)
        kernel = dist.max() - dist
        kernel /= kernel.sum()
        self.register_buffer("weight", kernel.view(1, 1, *kernel.shape))

    def forward(self, x):
        x = x.float()
        for _ in range(self.iterations - 1):
            x = torch.min(x, F.conv2d(x, weight=self.weight, groups=x.shape[1], padding=self.padding))
        x = F.conv2d(x, weight=self.weight, groups=x.shape[1], padding=self.padding)
        mask = x >= self.threshold
        x[mask] = 1.0
        x[~mask] /= x[~mask].max()
        return x, mask
