"""Oracle: AGPR inference, iterative back-projection and fit (CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates agpr.py of the reference:
  fit                       agpr.py:69-85   (100 Adam steps, linear kernel, default init)
  upscale_channel           agpr.py:160-190 (7x7 windows of the bicubic image -> GP mean, float32 test inputs)
  iterative_back_projection agpr.py:192-209
  downscale                 agpr.py:244-246 (GaussianBlur 3x3 sigma 1 + INTER_CUBIC by the MODULE-GLOBAL SRF)
  upscale                   agpr.py:148-158 (YCrCb: GP on Y, bicubic Cr/Cb)
The fit is pinned by the reference's checkpoints (tests/test_oracle.py); the
image part goes through the unpinned OpenCV restatement (oracle/cvops.py).
"""
import numpy as np
import torch

from . import cvops, gp

L = 7
PAD = L // 2
TAU = 0.01


def fit(train_x, train_y, iters=100, lr=0.1):
    h = gp.Hypers()
    gp.adam_fit(h, torch.as_tensor(train_x, dtype=torch.float64), torch.as_tensor(train_y, dtype=torch.float64),
                gp.LINEAR, iters=iters, lr=lr)
    return h


def downscale(image, global_srf):
    image = cvops.gaussian_blur(image, 3, 1)
    return cvops.resize_cubic(image, dsize=(image.shape[1] // global_srf, image.shape[0] // global_srf))


def iterative_back_projection(interpolated, original, global_srf):
    kern = np.array([[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]])
    cur = interpolated.copy()
    err = np.inf
    for _ in range(50):
        down = downscale(cur, global_srf)
        diff = original - down
        e = np.linalg.norm(diff) ** 2
        if err - 0.01 <= e:
            break
        err = e
        up = cvops.resize_cubic(diff, dsize=(cur.shape[1], cur.shape[0]))
        cur += cvops.filter2d_3x3(up, kern) * TAU
    return cur


def upscale_channel(image_u8, srf, h, train_x, train_y, global_srf=None):
    """-> float image * 255 (agpr.py:190); h: gp.Hypers of the trained model."""
    global_srf = srf if global_srf is None else global_srf
    image = image_u8 / 255.
    up = cvops.resize_cubic(image, dsize=(image.shape[1] * srf, image.shape[0] * srf))
    Hs, Ws = up.shape
    win = np.lib.stride_tricks.sliding_window_view(up, (L, L)).reshape(-1, L * L)
    test_x = torch.tensor(np.ascontiguousarray(win)).float().double()     # agpr.py:170 casts to float32
    X = torch.as_tensor(train_x, dtype=torch.float64)
    y = torch.as_tensor(train_y, dtype=torch.float64)
    mean, _ = gp.predictive_mean(h, X, y, test_x, gp.LINEAR, chunk=1000)
    up[PAD:-PAD, PAD:-PAD] = mean.numpy().reshape(Hs - 2 * PAD, Ws - 2 * PAD)
    up = iterative_back_projection(up, image, global_srf)
    return np.clip(up, 0, 1) * 255


def upscale(bgr, srf, h, train_x, train_y, grayscale=False, global_srf=None):
    if grayscale:
        return upscale_channel(cvops.bgr2gray(bgr), srf, h, train_x, train_y, global_srf)
    ycc = cvops.bgr2ycrcb(bgr)
    y = upscale_channel(ycc[..., 0], srf, h, train_x, train_y, global_srf).astype(np.uint8)
    cr = cvops.resize_cubic(ycc[..., 1], dsize=(y.shape[1], y.shape[0]))
    cb = cvops.resize_cubic(ycc[..., 2], dsize=(y.shape[1], y.shape[0]))
    return cvops.ycrcb2bgr(np.stack([y, cr, cb], axis=-1))
