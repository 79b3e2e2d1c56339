"""Oracle: the OpenCV / scikit-image operations on the path, restated in numpy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  cv2 and skimage are NOT
installed in the build container; the functions follow the libraries'
published algorithms (OpenCV 4.x imgproc: resize.cpp, smooth.dispatch.cpp /
filter.simd.hpp, color_rgb.simd.hpp / color_yuv.simd.hpp; skimage.metrics).
Pinned by output images the reference itself committed (tests/golden/refpin,
tests/test_refpin.py): bgr2gray, resize_cubic (float64), gaussian_blur(5, 0),
resize_linear (1/2 and 1/3) are on the path of GRAYSCALE SRGPR and the chained
oracle reproduces img_001_SRF_2_GPR_matern52_gray.png BIT FOR BIT; bgr2yuv /
yuv2bgr and the uint8 resize_cubic by the reference's colour outputs and uint8
downscales (every pixel at x2, x4 and 1/2; 99.89 % at x3); bgr2ycrcb / ycrcb2bgr,
gaussian_blur(3, 1), gaussian_blur_u8 and filter2d_3x3 by its AGPR outputs.
psnr / ssim are pinned by scikit-image 0.18.3 itself (tests/golden/skimage_metrics.json, made by running
skimage.metrics under the image's /opt/conda Python 3.9: agreement to 4e-15).

Call sites restated:
  resize_cubic        cv.resize(..., INTER_CUBIC)            gprsr.py:60-61,106; agpr.py:103,152-153,163,205,246
  resize_linear       cv.resize(img, None, fx=1/s, fy=1/s)   gprsr.py:115
  gaussian_blur       cv.GaussianBlur(img,(5,5),0) / ((3,3),1)  gprsr.py:114; agpr.py:245
  filter2d_3x3        cv.filter2D(img,-1,k)                  agpr.py:206
  bgr2gray / bgr2yuv / yuv2bgr / bgr2ycrcb / ycrcb2bgr       gprsr.py:53-63; agpr.py:150-154
  psnr / ssim         skimage.metrics                        evaluation.py:87,93
Written as explicit row/column loops over output lines (slow but plain).
"""
import numpy as np

f32 = np.float32


def cv_round(x):
    """cvRound: round-half-to-even."""
    return int(np.rint(x))


# ------------------------------------------------------------------ resize
def _cubic_coeffs(x):
    """interpolateCubic (resize.cpp), float32 arithmetic, A = -0.75."""
    A = f32(-0.75)
    x = f32(x)
    one = f32(1)
    c0 = ((A * (x + one) - f32(5) * A) * (x + one) + f32(8) * A) * (x + one) - f32(4) * A
    c1 = ((A + f32(2)) * x - (A + f32(3))) * x * x + one
    xm = one - x
    c2 = ((A + f32(2)) * xm - (A + f32(3))) * xm * xm + one
    c3 = one - c0 - c1 - c2
    return [f32(c0), f32(c1), f32(c2), f32(c3)]


def _cubic_tables(n_dst, n_src, inv_scale):
    scale = 1.0 / inv_scale
    idx = np.zeros((n_dst, 4), dtype=np.int64)
    cf = np.zeros((n_dst, 4), dtype=np.float32)
    for d in range(n_dst):
        fx = f32((d + 0.5) * scale - 0.5)
        sx = int(np.floor(fx))
        fx = f32(fx - f32(sx))
        cf[d] = _cubic_coeffs(fx)
        for k in range(4):
            idx[d, k] = min(max(sx - 1 + k, 0), n_src - 1)  # replicated border
    return idx, cf


def _cubic_tables_u8(n_dst, n_src):
    """Taps of the uint8 path: position and the A = -0.75 polynomials in float64, the four weights rounded to float32."""
    A = -0.75
    scale = n_src / n_dst
    idx = np.zeros((n_dst, 4), dtype=np.int64)
    cf = np.zeros((n_dst, 4), dtype=np.float32)
    for d in range(n_dst):
        f = (d + 0.5) * scale - 0.5
        sx = int(np.floor(f))
        x = f - sx
        c0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A
        c1 = ((A + 2) * x - (A + 3)) * x * x + 1
        c2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1
        cf[d] = [c0, c1, c2, 1 - c0 - c1 - c2]
        for k in range(4):
            idx[d, k] = min(max(sx - 1 + k, 0), n_src - 1)
    return idx, cf


def _fma32(a, b, c):
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


def _resize_cubic_u8(img, Wd, Hd):
    """cv.resize(INTER_CUBIC) on a uint8 plane AS THE REFERENCE'S OpenCV RAN IT.  Not resize.cpp's own fixed-point code
    (11-bit taps, (x + 2^21) >> 22): the opencv-python wheels route uint8 bicubic through IPP, whose arithmetic is
    closed source.  The model below was fitted to the images the reference committed, pixel by pixel and without any
    GP fit (tests/test_refpin.py): float32 taps (not quantised), float32 passes, horizontal then vertical, each
    pass (t0 w0 + t1 w1) + (t2 w2 + t3 w3) with the first product of a pair fused, ties to even.  It reproduces EVERY
    pixel of the reference's x2 and x4 chroma planes and of its uint8 downscales (498 -> 249 -> 124 columns, a
    non-dyadic ratio) and 99.89 % of its x3 chroma planes; resize.cpp's integer code: 4-5 pixels per image off at
    x2 / x4, 98.2 % on the 249 -> 124 downscale, 99.06 % at x3."""
    H, W = img.shape
    xi, xc = _cubic_tables_u8(Wd, W)
    yi, yc = _cubic_tables_u8(Hd, H)
    src = img.astype(np.float32)
    t = [src[:, xi[:, k]] for k in range(4)]
    w = [xc[None, :, k] for k in range(4)]
    hbuf = _fma32(t[0], w[0], t[1] * w[1]) + _fma32(t[2], w[2], t[3] * w[3])
    t = [hbuf[yi[:, k]] for k in range(4)]
    w = [yc[:, k, None] for k in range(4)]
    out = _fma32(t[0], w[0], t[1] * w[1]) + _fma32(t[2], w[2], t[3] * w[3])
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def resize_cubic(img, fx=None, fy=None, dsize=None):
    """cv.resize(INTER_CUBIC).  float64 images: float32 coefficients, float64
    accumulation, horizontal pass then vertical pass.  uint8 images: float32 model of the
    IPP-backed path the reference ran (_resize_cubic_u8)."""
    H, W = img.shape
    if dsize is None:
        Wd, Hd = cv_round(W * fx), cv_round(H * fy)
        isx, isy = fx, fy
    else:
        Wd, Hd = dsize  # (width, height) like cv
        isx, isy = Wd / W, Hd / H
    xi, xc = _cubic_tables(Wd, W, isx)
    yi, yc = _cubic_tables(Hd, H, isy)
    if img.dtype == np.uint8:
        return _resize_cubic_u8(img, Wd, Hd)
    src = img.astype(np.float64)
    xcd = xc.astype(np.float64)
    ycd = yc.astype(np.float64)
    hbuf = np.zeros((H, Wd))
    for r in range(H):
        row = src[r]
        hbuf[r] = (row[xi[:, 0]] * xcd[:, 0] + row[xi[:, 1]] * xcd[:, 1]
                   + row[xi[:, 2]] * xcd[:, 2] + row[xi[:, 3]] * xcd[:, 3])
    out = np.zeros((Hd, Wd))
    for r in range(Hd):
        out[r] = (hbuf[yi[r, 0]] * ycd[r, 0] + hbuf[yi[r, 1]] * ycd[r, 1]
                  + hbuf[yi[r, 2]] * ycd[r, 2] + hbuf[yi[r, 3]] * ycd[r, 3])
    return out


def _linear_tables(n_dst, n_src, inv_scale):
    scale = 1.0 / inv_scale
    idx = np.zeros((n_dst, 2), dtype=np.int64)
    cf = np.zeros((n_dst, 2), dtype=np.float32)
    for d in range(n_dst):
        fx = f32((d + 0.5) * scale - 0.5)
        sx = int(np.floor(fx))
        fx = f32(fx - f32(sx))
        if sx < 0:
            fx, sx = f32(0), 0
        if sx >= n_src - 1:
            fx, sx = f32(0), n_src - 1
        idx[d] = (sx, min(sx + 1, n_src - 1))
        cf[d] = (f32(1) - fx, fx)
    return idx, cf


def resize_linear(img, fx, fy):
    """cv.resize default INTER_LINEAR on float64 (gprsr.py:115).  OpenCV turns an
    exact 2x2 decimation into INTER_AREA (sum of the 4 pixels * 0.25)."""
    H, W = img.shape
    Wd, Hd = cv_round(W * fx), cv_round(H * fy)
    sx, sy = 1.0 / fx, 1.0 / fy
    if abs(sx - 2) < np.finfo(float).eps and abs(sy - 2) < np.finfo(float).eps:
        a = img[0:2 * Hd:2, 0:2 * Wd:2]
        b = img[0:2 * Hd:2, 1:2 * Wd:2]
        c = img[1:2 * Hd:2, 0:2 * Wd:2]
        d = img[1:2 * Hd:2, 1:2 * Wd:2]
        return (((a + b) + c) + d) * 0.25
    xi, xc = _linear_tables(Wd, W, fx)
    yi, yc = _linear_tables(Hd, H, fy)
    xcd, ycd = xc.astype(np.float64), yc.astype(np.float64)
    hbuf = img[:, xi[:, 0]] * xcd[None, :, 0] + img[:, xi[:, 1]] * xcd[None, :, 1]
    return hbuf[yi[:, 0]] * ycd[:, 0, None] + hbuf[yi[:, 1]] * ycd[:, 1, None]


# ------------------------------------------------------------------ filters
def _reflect101(i, n):
    if n == 1:
        return 0
    while i < 0 or i >= n:
        i = -i if i < 0 else 2 * (n - 1) - i
    return i


def gaussian_kernel(ksize, sigma):
    """cv.getGaussianKernel: sigma <= 0 and odd ksize <= 7 -> fixed binomial
    table ({1,4,6,4,1}/16 for 5); otherwise normalised exp(-x^2 / 2 sigma^2)."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in small:
        k = np.array(small[ksize], dtype=np.float64)
    else:
        s = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
        x = np.arange(ksize) - (ksize - 1) * 0.5
        k = np.exp(-0.5 / (s * s) * x * x)
    return k * (1.0 / k.sum())


def gaussian_kernel_u8(ksize, sigma):
    """The 8-bit fixed-point Gaussian taps cv.GaussianBlur uses on CV_8U images (smooth.dispatch.cpp,
    getGaussianKernelFixedPoint_ED with 8 fraction bits): the left half of the float kernel times 256, rounded (ties
    to even) with the rounding error carried to the next tap, mirrored; the centre tap takes what is left of 256."""
    k = gaussian_kernel(ksize, sigma)
    out = [0] * ksize
    err = 0.0
    for i in range(ksize // 2):
        adj = float(k[i]) * 256.0 + err
        v = cv_round(adj)
        err = adj - v
        out[i] = out[ksize - 1 - i] = v
    out[ksize // 2] = 256 - sum(out)
    return np.array(out, dtype=np.int64)


def gaussian_blur_u8(img, ksize, sigma):
    """cv.GaussianBlur on a uint8 image [H,W] or [H,W,C] (agpr.py:245 applied to the uint8 BGR image, agpr.py:286-287):
    OpenCV's bit-exact fixed-point path -- horizontal pass exact in 8.8 fixed point, vertical pass in 16.16, one
    rounding (half up) at the end; BORDER_REFLECT_101; channels independently.  Pinned by the reference's AGPR
    outputs (tests/test_refpin.py)."""
    if img.ndim == 3:
        return np.stack([gaussian_blur_u8(np.ascontiguousarray(img[..., c]), ksize, sigma) for c in range(img.shape[2])], -1)
    assert img.dtype == np.uint8
    k = gaussian_kernel_u8(ksize, sigma)
    H, W = img.shape
    r = ksize // 2
    ri = [_reflect101(i, H) for i in range(-r, H + r)]
    ci = [_reflect101(i, W) for i in range(-r, W + r)]
    p = img.astype(np.int64)[:, ci]
    h = sum(int(k[a]) * p[:, a:a + W] for a in range(ksize))
    q = h[ri]
    v = sum(int(k[a]) * q[a:a + H] for a in range(ksize))
    return ((v + (1 << 15)) >> 16).astype(np.uint8)


def gaussian_blur(img, ksize, sigma):
    """Separable symmetric filter, BORDER_REFLECT_101, rows then columns."""
    k = gaussian_kernel(ksize, sigma)
    h = ksize // 2
    H, W = img.shape
    ci = np.array([[_reflect101(x + d, W) for x in range(W)] for d in range(-h, h + 1)])
    tmp = img[:, ci[h]] * k[h]
    for j in range(1, h + 1):
        tmp = tmp + (img[:, ci[h + j]] + img[:, ci[h - j]]) * k[h + j]
    ri = np.array([[_reflect101(y + d, H) for y in range(H)] for d in range(-h, h + 1)])
    out = tmp[ri[h]] * k[h]
    for j in range(1, h + 1):
        out = out + (tmp[ri[h + j]] + tmp[ri[h - j]]) * k[h + j]
    return out


def filter2d_3x3(img, kern):
    """cv.filter2D (correlation), BORDER_REFLECT_101, taps added in row-major order."""
    H, W = img.shape
    ri = np.array([[_reflect101(y + d, H) for y in range(H)] for d in (-1, 0, 1)])
    ci = np.array([[_reflect101(x + d, W) for x in range(W)] for d in (-1, 0, 1)])
    out = np.zeros((H, W))
    for a in range(3):
        for b in range(3):
            if kern[a][b] != 0:
                out = out + float(kern[a][b]) * img[ri[a]][:, ci[b]]
    return out


# ------------------------------------------------------------------ colour
def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def bgr2gray(bgr):
    """cv.cvtColor(COLOR_BGR2GRAY) on uint8: OpenCV's RGB2Gray<uchar> uses its 15-bit table (BY15 = 3735,
    GY15 = 19235, RY15 = 9798, gray_shift = 15), NOT the 14-bit yuv_shift constants of the YUV / YCrCb
    conversions below (the two round differently on ~0.1 % of pixels).  Pinned: with this table the chained
    oracle reproduces the reference's committed img_001_SRF_2_GPR_matern52_gray.png bit for bit, with the
    14-bit table 99.17 % of the pixels (tests/test_refpin.py)."""
    b, g, r = [bgr[..., i].astype(np.int64) for i in range(3)]
    return _descale(b * 3735 + g * 19235 + r * 9798, 15).astype(np.uint8)


def _bgr2ycc(bgr, c_r, c_b, crcb_order):
    b, g, r = [bgr[..., i].astype(np.int64) for i in range(3)]
    delta = 128 << 14
    Y = _descale(b * 1868 + g * 9617 + r * 4899, 14)
    Cr = _descale((r - Y) * c_r + delta, 14)
    Cb = _descale((b - Y) * c_b + delta, 14)
    ch = (Y, Cr, Cb) if crcb_order else (Y, Cb, Cr)
    return np.stack([np.clip(c, 0, 255) for c in ch], axis=-1).astype(np.uint8)


def bgr2yuv(bgr):      # COLOR_BGR2YUV: U = 0.492 (B-Y), V = 0.877 (R-Y)
    return _bgr2ycc(bgr, 14369, 8061, False)


def bgr2ycrcb(bgr):    # COLOR_BGR2YCrCb: Cr = 0.713 (R-Y), Cb = 0.564 (B-Y)
    return _bgr2ycc(bgr, 11682, 9241, True)


def _ycc2bgr(Y, Cr, Cb, c0, c1, c2, c3):
    Y, Cr, Cb = Y.astype(np.int64), Cr.astype(np.int64) - 128, Cb.astype(np.int64) - 128
    b = Y + _descale(Cb * c3, 14)
    g = Y + _descale(Cb * c2 + Cr * c1, 14)
    r = Y + _descale(Cr * c0, 14)
    return np.stack([np.clip(c, 0, 255) for c in (b, g, r)], axis=-1).astype(np.uint8)


def yuv2bgr(yuv):      # V2R 1.140, V2G -0.581, U2G -0.395, U2B 2.032
    return _ycc2bgr(yuv[..., 0], yuv[..., 2], yuv[..., 1], 18678, -9519, -6472, 33292)


def ycrcb2bgr(ycc):    # Cr2R 1.403, Cr2G -0.714, Cb2G -0.344, Cb2B 1.773
    return _ycc2bgr(ycc[..., 0], ycc[..., 1], ycc[..., 2], 22987, -11698, -5636, 29049)


# ------------------------------------------------------------------ metrics
def preprocess_for_metrics(img, srf):
    """Evaluator.preprocess_image (evaluation.py:95-98)."""
    if img.ndim == 3:
        img = bgr2gray(img)
    return img[srf:-srf, srf:-srf]


def psnr(a, b, data_range=255):
    """skimage.metrics.peak_signal_noise_ratio."""
    err = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10(data_range ** 2 / err)


def _uniform7(x):
    """scipy.ndimage.uniform_filter(size=7, mode='reflect'), separable box mean."""
    from scipy.ndimage import uniform_filter
    return uniform_filter(x, size=7)


def ssim(a, b, data_range=255):
    """skimage.metrics.structural_similarity defaults: 7x7 uniform window,
    sample covariance, K1=.01, K2=.03, mean over the window-valid region."""
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    NP = 49
    cov_norm = NP / (NP - 1)
    ux, uy = _uniform7(a), _uniform7(b)
    uxx, uyy, uxy = _uniform7(a * a), _uniform7(b * b), _uniform7(a * b)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    return S[3:-3, 3:-3].mean()
