"""Oracle: AGPR inference, iterative back-projection and fit (CPU).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates agpr.py of the reference:
  fit                       agpr.py:69-85   (100 Adam steps, linear kernel, default init)
  upscale_channel           agpr.py:160-190 (7x7 windows of the bicubic image -> GP mean, float32 test inputs)
  iterative_back_projection agpr.py:192-209
  downscale                 agpr.py:244-246 (GaussianBlur 3x3 sigma 1 + INTER_CUBIC by the MODULE-GLOBAL SRF)
  upscale                   agpr.py:148-158 (YCrCb: GP on Y, bicubic Cr/Cb)
  active_sampling           agpr.py:120-137, 211-222 (training-set selection; numpy is the reference's own library
                            here, so this part runs the same numpy calls on the same values: pinned by construction)
The fit is pinned by the reference's checkpoints (tests/test_oracle.py); inference, back-projection, downscale and
the YCrCb / uint8 stages are pinned by the reference's own *_AGPR2.png outputs, which those checkpoints reproduce
(tests/test_refpin.py: every pixel of img_006 at x2).
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
    """agpr.py:244-246.  float images (the back-projection loop, the training set): float64 paths; uint8 images of
    any channel count (agpr.py:286-287 hands it the uint8 BGR HR image): OpenCV's uint8 paths, channel by channel."""
    if image.dtype == np.uint8:
        b = cvops.gaussian_blur_u8(image, 3, 1)
        size = (image.shape[1] // global_srf, image.shape[0] // global_srf)
        if b.ndim == 2:
            return cvops.resize_cubic(b, dsize=size)
        return np.stack([cvops.resize_cubic(np.ascontiguousarray(b[..., c]), dsize=size) for c in range(b.shape[2])], -1)
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


def active_sampling(data, K=10, n_pick=500, bandwidth_coeff=0.2, tradeoff=0.2):
    """agpr.py:120-137 on the np.unique'd candidate rows `data` [N, D].

    -> (indices of the chosen rows in the order chosen, chosen rows, representativeness scores, bandwidth).
    The reference keeps rows, not indices; the indices are carried along here so that a device
    implementation can be compared exactly.  Kept as the reference has them:
      * :123  distances by broadcasting (memory O(N^2 D): small N only), numpy's summation order;
      * :125  bandwidth = coeff * median of the INDEX arrays np.where returns (not of the distances);
      * :126  K nearest by argsort of each row;
      * :133  score i uses representativeness_scores[i] with i the position in the SHRINKING pool (:136 deletes
              from the pool, the score array is never deleted from);
      * :222  diversity = min over the chosen rows of -exp(-d / 2bw), 0 while nothing is chosen (:218).
    """
    data = np.asarray(data, dtype=np.float64)
    dist = np.sum((data[:, None, :] - data[None, :, :]) ** 2, axis=-1)
    np.fill_diagonal(dist, np.inf)
    bandwidth = bandwidth_coeff * np.median(np.where(dist != np.inf))
    near = np.argsort(dist, axis=1)[:, :K]
    near_d = np.take_along_axis(dist, near, axis=-1)
    rep = np.mean(np.exp(-near_d / (2 * bandwidth)), axis=-1)

    pool = np.copy(data)
    pool_idx = np.arange(len(data))
    chosen, chosen_idx = [], []
    for _ in range(n_pick):
        scores = []
        for i, row in enumerate(pool):
            if len(chosen) == 0:
                diversity = 0
            else:
                sel_d = np.sum((chosen - row) ** 2, axis=1)
                diversity = np.min(-np.exp(-sel_d / (2 * bandwidth)))
            scores.append(tradeoff * rep[i] + (1 - tradeoff) * diversity)
        j = np.argmax(scores)
        chosen.append(pool[j])
        chosen_idx.append(int(pool_idx[j]))
        pool = np.delete(pool, j, axis=0)
        pool_idx = np.delete(pool_idx, j)
    return np.array(chosen_idx), np.array(chosen), rep, bandwidth
