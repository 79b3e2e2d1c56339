"""Image stages around the GP phases: the OpenCV calls of the reference, restated.

cv2 is not a dependency.  float64 images stay on the GPU: the host only builds
the small per-axis index / coefficient tables the way OpenCV's resize.cpp does
(float32 coefficient arithmetic, replicated border) and the HIP kernels
(gpsr_resample, gpsr_decimate2, gpsr_sepfilter_sym, gpsr_filter3x3) apply them.
uint8 colour conversion and the chroma resize are integer fixed-point kernels as well
(gpsr_bgr2gray / gpsr_bgr2yuv / gpsr_yuv2bgr / gpsr_bgr2ycrcb / gpsr_ycrcb2bgr / gpsr_resample_u8:
gprsr.py:53-63, agpr.py:150-156), so GPRSR.apply moves a colour image to the GPU once; the numpy
versions below remain for host-side callers (file loops, tests) and give the same bits.

Reference call sites: gprsr.py:60-61,106 (INTER_CUBIC), gprsr.py:115 (default
INTER_LINEAR, fx=1/s), gprsr.py:114 (GaussianBlur 5x5 sigma 0), agpr.py:245-246
(GaussianBlur 3x3 sigma 1 + INTER_CUBIC), agpr.py:206 (filter2D).
"""
from functools import lru_cache

import numpy as np
import torch

from . import ops

_F = np.float32


def cv_round(x):
    return int(np.rint(x))


@lru_cache(maxsize=64)
def _axis_tables(n_dst, n_src, inv_scale, kind):
    """(indices [n_dst,T] int32, weights [n_dst,T] float64) for one axis."""
    scale = 1.0 / inv_scale
    d = np.arange(n_dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(_F)
    sx = np.floor(fx)
    fr = (fx - sx.astype(_F)).astype(_F)
    sx = sx.astype(np.int64)
    if kind == "cubic":
        A = _F(-0.75)
        x1 = fr + _F(1)
        c0 = ((A * x1 - _F(5) * A) * x1 + _F(8) * A) * x1 - _F(4) * A
        c1 = ((A + _F(2)) * fr - (A + _F(3))) * fr * fr + _F(1)
        xm = _F(1) - fr
        c2 = ((A + _F(2)) * xm - (A + _F(3))) * xm * xm + _F(1)
        c3 = _F(1) - c0 - c1 - c2
        w = np.stack([c0, c1, c2, c3], axis=1).astype(_F)
        idx = sx[:, None] + np.arange(-1, 3)[None, :]
    else:
        lo = sx < 0
        fr = np.where(lo, _F(0), fr)
        sx = np.where(lo, 0, sx)
        hi = sx >= n_src - 1
        fr = np.where(hi, _F(0), fr).astype(_F)
        sx = np.where(hi, n_src - 1, sx)
        w = np.stack([_F(1) - fr, fr], axis=1).astype(_F)
        idx = sx[:, None] + np.arange(0, 2)[None, :]
    idx = np.clip(idx, 0, n_src - 1).astype(np.int32)
    return np.ascontiguousarray(idx), np.ascontiguousarray(w.astype(np.float64))


def _dev_tables(n_dst, n_src, inv_scale, kind, device):
    idx, w = _axis_tables(n_dst, n_src, float(inv_scale), kind)
    return torch.as_tensor(idx).to(device), torch.as_tensor(w).to(device)


def resize_cubic(img, fx=None, fy=None, dsize=None):
    """cv.resize(img, dsize or None, fx=, fy=, interpolation=INTER_CUBIC); img: device float64 [H,W]."""
    H, W = img.shape
    if dsize is None:
        Wd, Hd, isx, isy = cv_round(W * fx), cv_round(H * fy), fx, fy
    else:
        Wd, Hd = dsize
        isx, isy = Wd / W, Hd / H
    xi, xw = _dev_tables(Wd, W, isx, "cubic", img.device)
    yi, yw = _dev_tables(Hd, H, isy, "cubic", img.device)
    return ops.resample(img, Hd, Wd, xi, xw, yi, yw)


def resize_linear(img, fx, fy):
    """cv.resize(img, None, fx=, fy=) (INTER_LINEAR; exact 1/2 -> area fast path)."""
    H, W = img.shape
    Wd, Hd = cv_round(W * fx), cv_round(H * fy)
    eps = np.finfo(float).eps
    if abs(1.0 / fx - 2) < eps and abs(1.0 / fy - 2) < eps:
        return ops.decimate2(img)
    xi, xw = _dev_tables(Wd, W, fx, "linear", img.device)
    yi, yw = _dev_tables(Hd, H, fy, "linear", img.device)
    return ops.resample(img, Hd, Wd, xi, xw, yi, yw)


def gaussian_kernel(ksize, sigma):
    """cv.getGaussianKernel (sigma <= 0, odd ksize <= 7: fixed binomial table)."""
    table = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in table:
        k = np.array(table[ksize], dtype=np.float64)
    else:
        sg = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
        x = np.arange(ksize) - (ksize - 1) * 0.5
        k = np.exp((-0.5 / (sg * sg)) * x * x)
    return k * (1.0 / k.sum())


def gaussian_blur(img, ksize, sigma):
    k = torch.as_tensor(gaussian_kernel(ksize, sigma)).to(img.device)
    return ops.sepfilter_sym(img, k)


def gaussian_kernel_u8(ksize, sigma):
    """The 8-bit fixed-point taps of cv.GaussianBlur on CV_8U (getGaussianKernelFixedPoint_ED, 8 fraction bits): left
    half = float taps * 256 rounded with the error carried on, mirrored, centre = 256 - the rest."""
    k = gaussian_kernel(ksize, sigma)
    out = [0] * ksize
    err = 0.0
    for i in range(ksize // 2):
        adj = float(k[i]) * 256.0 + err
        v = cv_round(adj)
        err = adj - v
        out[i] = out[ksize - 1 - i] = v
    out[ksize // 2] = 256 - sum(out)
    return np.array(out, dtype=np.int32)


def gaussian_blur_u8_dev(img, ksize, sigma):
    """cv.GaussianBlur on a uint8 device plane (OpenCV's fixed-point path)."""
    return ops.sepfilter_u8(img, torch.as_tensor(gaussian_kernel_u8(ksize, sigma)).to(img.device))


def filter2d_3x3(img, kern):
    k = torch.as_tensor(np.ascontiguousarray(np.asarray(kern, dtype=np.float64).reshape(9))).to(img.device)
    return ops.filter3x3(img, k)


# ---------------------------------------------------------------- uint8 stages on the device
@lru_cache(maxsize=64)
def _axis_tables_u8(n_dst, n_src):
    """Taps of the uint8 bicubic path (the float32 model of the IPP-backed cv.resize the reference ran, see
    k_resample_u8): position and the A = -0.75 polynomials in float64, the four weights rounded to float32."""
    A = -0.75
    f = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
    sx = np.floor(f)
    x = f - sx
    c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
    c1 = ((A + 2) * x - (A + 3)) * x * x + 1
    c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
    w = np.stack([c0, c1, c2, 1 - c0 - c1 - c2], axis=1).astype(np.float32)
    idx = np.clip(sx.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1).astype(np.int32)
    return np.ascontiguousarray(idx), np.ascontiguousarray(w)


def _u8_dsize(H, W, fx, fy, dsize):
    return (cv_round(W * fx), cv_round(H * fy)) if dsize is None else tuple(dsize)


def resize_cubic_u8_dev(img, fx=None, fy=None, dsize=None):
    """cv.resize INTER_CUBIC on a uint8 device plane (the chroma planes, gprsr.py:60-61 / agpr.py:152-153)."""
    H, W = img.shape
    Wd, Hd = _u8_dsize(H, W, fx, fy, dsize)
    xi, xw = _axis_tables_u8(Wd, W)
    yi, yw = _axis_tables_u8(Hd, H)
    dev = img.device
    return ops.resample_u8(img, Hd, Wd, torch.as_tensor(xi).to(dev), torch.as_tensor(xw).to(dev),
                           torch.as_tensor(yi).to(dev), torch.as_tensor(yw).to(dev))


# ---------------------------------------------------------------- uint8 host code (host callers: file loops, tests)
def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def bgr2gray(bgr):
    """cv.cvtColor(COLOR_BGR2GRAY), uint8: OpenCV's 15-bit RGB2Gray table (3735, 19235, 9798, >> 15); the YUV /
    YCrCb conversions below use the 14-bit one.  Pinned by the reference's own *_GPR_matern52_gray.png outputs."""
    c = bgr.astype(np.int64)
    return _descale(c[..., 0] * 3735 + c[..., 1] * 19235 + c[..., 2] * 9798, 15).astype(np.uint8)


def _split_ycc(bgr, cr_coef, cb_coef):
    c = bgr.astype(np.int64)
    Y = _descale(c[..., 0] * 1868 + c[..., 1] * 9617 + c[..., 2] * 4899, 14)
    half = 128 << 14
    Cr = np.clip(_descale((c[..., 2] - Y) * cr_coef + half, 14), 0, 255)
    Cb = np.clip(_descale((c[..., 0] - Y) * cb_coef + half, 14), 0, 255)
    return np.clip(Y, 0, 255).astype(np.uint8), Cr.astype(np.uint8), Cb.astype(np.uint8)


def bgr2yuv(bgr):
    Y, V, U = _split_ycc(bgr, 14369, 8061)
    return Y, U, V


def bgr2ycrcb(bgr):
    return _split_ycc(bgr, 11682, 9241)


def _merge_ycc(Y, Cr, Cb, coef):
    c0, c1, c2, c3 = coef
    Y, Cr, Cb = Y.astype(np.int64), Cr.astype(np.int64) - 128, Cb.astype(np.int64) - 128
    b = Y + _descale(Cb * c3, 14)
    g = Y + _descale(Cb * c2 + Cr * c1, 14)
    r = Y + _descale(Cr * c0, 14)
    return np.clip(np.stack([b, g, r], axis=-1), 0, 255).astype(np.uint8)


def yuv2bgr(Y, U, V):
    return _merge_ycc(Y, V, U, (18678, -9519, -6472, 33292))


def ycrcb2bgr(Y, Cr, Cb):
    return _merge_ycc(Y, Cr, Cb, (22987, -11698, -5636, 29049))


def resize_cubic_u8(img, fx=None, fy=None, dsize=None):
    """cv.resize INTER_CUBIC on uint8, host form of k_resample_u8 (same float32 arithmetic, fused products included)."""
    H, W = img.shape
    Wd, Hd = _u8_dsize(H, W, fx, fy, dsize)
    xi, xw = _axis_tables_u8(Wd, W)
    yi, yw = _axis_tables_u8(Hd, H)

    def fma(a, b, c):
        """float32 fused multiply-add, exactly: the product of two float32 is exact in float64; the sum is taken in float64
        ROUNDED TO ODD (TwoSum gives the rounding error; an inexact even result moves to its odd neighbour on the side of the
        error), and a round-to-odd 53-bit value converts to float32 with the one rounding of a true fmaf -- a plain float64
        sum followed by the conversion rounds twice and can land on the other side of a float32 tie."""
        p = a.astype(np.float64) * b.astype(np.float64)
        c = np.broadcast_to(np.asarray(c, dtype=np.float64), np.broadcast(p, c).shape)
        p = np.broadcast_to(p, c.shape)
        s = p + c
        bb = s - p
        err = (p - (s - bb)) + (c - bb)
        even = (s.view(np.int64) & 1) == 0
        toward = np.where(err > 0, np.inf, -np.inf)
        s = np.where((err != 0) & even, np.nextafter(s, toward), s)
        return s.astype(np.float32)
    src = img.astype(np.float32)
    t = [src[:, xi[:, k]] for k in range(4)]
    w = [xw[None, :, k] for k in range(4)]
    hbuf = fma(t[0], w[0], t[1] * w[1]) + fma(t[2], w[2], t[3] * w[3])
    t = [hbuf[yi[:, k]] for k in range(4)]
    w = [yw[:, k, None] for k in range(4)]
    out = fma(t[0], w[0], t[1] * w[1]) + fma(t[2], w[2], t[3] * w[3])
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)
