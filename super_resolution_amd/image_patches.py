"""Patch geometry: drop-in for the reference's image_patches.py.

Same public surface (reference image_patches.py:4-107): PatchInterpolationMode,
PatchHandler(images, base_patch_size=20, stride_percentage=0.9) with
.patch_locations, .stride, .scale_factors, iteration yielding one patch per
image, and .combine_patches(patches, mode).  The geometry is computed as a
row list x column list (patches always lie on that grid, duplicates included),
which is what the HIP gather / combine kernels consume; combine_patches runs
on the GPU (gpsr_combine) -- there is no host implementation of it here.
"""
from enum import Enum

import numpy as np


class PatchInterpolationMode(Enum):
    AVERAGE = 0
    BILINEAR = 1


def axis_starts(extent, base_patch_size, stride):
    """Patch corners along one axis: every `stride`, clamped so the patch stays
    inside (reference image_patches.py:37-40).  Clamping can repeat the last
    position; repeats are kept, as the reference keeps them."""
    starts = np.arange(0, extent, stride, dtype=np.int64)
    return np.minimum(starts, extent - base_patch_size).astype(np.int32)


def grid_of(image_shape, base_patch_size, stride_percentage=0.9):
    stride = int((base_patch_size - 2) * stride_percentage)
    return (axis_starts(image_shape[0], base_patch_size, stride),
            axis_starts(image_shape[1], base_patch_size, stride))


class PatchHandler:
    def __init__(self, images, base_patch_size=20, stride_percentage=0.9):
        self.image_shape = images[0].shape
        H, W = self.image_shape[0], self.image_shape[1]
        assert base_patch_size <= H and base_patch_size <= W, \
            f'Patch size {base_patch_size} is too big for image size {self.image_shape}'
        assert all(im.shape[0] / im.shape[1] == H / W for im in images), \
            'All images must have the same aspect ratio'
        assert all(im.shape[0] >= H for im in images), \
            'All further images must be at least as big as the first image'
        self.images = images
        self.scale_factors = [int(im.shape[0] / H) for im in images]
        self.base_patch_size = base_patch_size
        self.stride = int((base_patch_size - 2) * stride_percentage)
        self.rows = axis_starts(H, base_patch_size, self.stride)
        self.cols = axis_starts(W, base_patch_size, self.stride)
        rr, cc = np.meshgrid(self.rows, self.cols, indexing="ij")
        self.locations_array = np.stack([rr.reshape(-1), cc.reshape(-1)], axis=1).astype(np.int32)
        self.patch_locations = [(int(r), int(c)) for r, c in self.locations_array]

    def __len__(self):
        return len(self.patch_locations)

    def __iter__(self, image_indices=None):
        idx = range(len(self.images)) if image_indices is None else image_indices
        p = self.base_patch_size
        for r, c in self.patch_locations:
            out = []
            for i in idx:
                sf = self.scale_factors[i]
                ps, r0, c0 = int(p * sf), int(r * sf), int(c * sf)
                out.append(self.images[i][r0:r0 + ps, c0:c0 + ps].copy())
            yield out

    def combine_patches(self, patches, patch_interpolation=PatchInterpolationMode.AVERAGE):
        """Weighted recombination of overlapping patches (reference
        image_patches.py:44-76) on the GPU; bit-identical to the reference loop."""
        import torch
        from . import ops
        P = patches[0].shape[0]
        s = int(P / self.base_patch_size)
        dev = torch.device("cuda", torch.cuda.current_device())   # one process per GPU: the rank's own device
        stack = torch.as_tensor(np.ascontiguousarray(np.stack(patches), dtype=np.float64)).to(dev)
        rows = torch.as_tensor(self.rows).to(dev)
        cols = torch.as_tensor(self.cols).to(dev)
        mode = patch_interpolation.value if isinstance(patch_interpolation, Enum) else int(patch_interpolation)
        out = ops.combine(stack, None, self.image_shape[0], self.image_shape[1], s, self.base_patch_size,
                          rows, cols, mode, clip=False)
        return out.cpu().numpy()
