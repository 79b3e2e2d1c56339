"""Oracle: patch geometry (CPU, loops kept close to the reference on purpose).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates image_patches.py of the reference:
  patch_locations()  <- PatchHandler.__init__/_generate_patch_locations (image_patches.py:11-41)
  extract()          <- PatchIterator.__next__ (image_patches.py:92-107)
  combine()          <- PatchHandler.combine_patches (image_patches.py:44-76)
  neighbors()/features() <- extract_neighbors + list-comps (gprsr.py:34-35,148,150,192)
Pinned bit-for-bit by tests/golden/patches_*.npz (generated from the reference
module itself by tests/golden/make_patch_goldens.py).
"""
import numpy as np

AVERAGE, BILINEAR = 0, 1


def stride_of(base_patch_size, stride_percentage=0.9):
    return int((base_patch_size - 2) * stride_percentage)  # image_patches.py:29


def patch_locations(shape, base_patch_size, stride_percentage=0.9):
    """Row-major list of (row, col) LR patch corners; positions that would run
    off the image are clamped inward, duplicates are kept (image_patches.py:33-41)."""
    H, W = shape[0], shape[1]
    assert base_patch_size <= H and base_patch_size <= W
    stride = stride_of(base_patch_size, stride_percentage)
    locs = []
    for r in range(0, H, stride):
        r = min(r, H - base_patch_size)
        for c in range(0, W, stride):
            c = min(c, W - base_patch_size)
            locs.append((r, c))
    return locs


def extract(image, loc, base_patch_size, scale):
    """Copy of the (p*scale)^2 window at loc*scale (image_patches.py:99-102)."""
    ps = int(base_patch_size * scale)
    r, c = int(loc[0] * scale), int(loc[1] * scale)
    return image[r:r + ps, c:c + ps].copy()


# Order of the 8 neighbours, (dy, dx), gprsr.py:35
NEIGHBOR_OFFSETS = [(1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1), (1, -1), (-1, 1)]


def features(patch):
    """[ (side-2)^2, 8 ] neighbour features of every interior pixel, y-major
    (gprsr.py:148 with training_indices = range(1, side-1); gprsr.py:192)."""
    s = patch.shape[0]
    cols = [patch[1 + dy:s - 1 + dy, 1 + dx:s - 1 + dx].reshape(-1) for dy, dx in NEIGHBOR_OFFSETS]
    return np.stack(cols, axis=1)


def targets(patch):
    """Interior pixel values, y-major (gprsr.py:150)."""
    return patch[1:-1, 1:-1].reshape(-1).copy()


def grid_training_indices(patch_size):
    """USE_ALL_PIXELS_FOR_TRAINING=False pattern (gprsr.py:140)."""
    return [i for i in range(1, patch_size // 2, 2)] + [i for i in range(patch_size // 2, patch_size - 1, 2)]


def features_at(patch, idx):
    pts = [(y, x) for y in idx for x in idx]
    X = np.array([[patch[y + dy, x + dx] for dy, dx in NEIGHBOR_OFFSETS] for y, x in pts])
    return X


def targets_at(patch, idx):
    return np.array([patch[y, x] for y in idx for x in idx])


def combine(patches, locs, lr_shape, base_patch_size, mode=AVERAGE):
    """image_patches.py:44-76, same loop nest and same order of additions."""
    scale = int(patches[0].shape[0] / base_patch_size)
    P = base_patch_size * scale
    Hs, Ws = lr_shape[0] * scale, lr_shape[1] * scale
    result = np.zeros((Hs, Ws))
    weights = np.zeros((Hs, Ws))
    for (r0, c0), patch in zip(locs, patches):
        x, y = r0 * scale, c0 * scale
        x_start = 0 if x == 0 else 1
        y_start = 0 if y == 0 else 1
        x_end = P if x + P == Hs else P - 1
        y_end = P if y + P == Ws else P - 1
        if mode == AVERAGE:
            w = np.ones((P, P))
        else:
            i = np.arange(P)
            v = 1 - np.abs(1 / 2 - i / P)
            w = v[:, None] * v[None, :]
        sl = (slice(x_start, x_end), slice(y_start, y_end))
        result[x + x_start:x + x_end, y + y_start:y + y_end] += patch[sl] * w[sl]
        weights[x + x_start:x + x_end, y + y_start:y + y_end] += w[sl]
    result /= weights
    return result
