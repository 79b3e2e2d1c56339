"""Thin torch-tensor wrappers over the C ABI (include/gpsr.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; all
arithmetic on the path happens in libgpsr.so.  Every function requires CUDA
(ROCm) tensors and raises otherwise -- there is no CPU path.
"""
import numpy as np
import torch

from . import _lib

RBF, MATERN12, MATERN32, MATERN52, LINEAR = 0, 1, 2, 3, 4
AVERAGE, BILINEAR = 0, 1
BLOCK = 112


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.GpsrError("gpsr ops need device tensors (no CPU fallback)")
    if t.device.index != torch.cuda.current_device():
        # launches go on the current device's stream (_stream): a buffer of another GPU would fault there
        raise _lib.GpsrError(f"tensor on {t.device}, current device is cuda:{torch.cuda.current_device()} "
                             "(one process per GPU: torch.cuda.set_device(LOCAL_RANK) first)")
    if dtype is not None and t.dtype != dtype:
        raise _lib.GpsrError(f"expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.GpsrError("tensor must be contiguous")
    return t.data_ptr()


def padded_n(n):
    return ((n + BLOCK - 1) // BLOCK) * BLOCK


def block(A):
    """[B,N,N] row-major -> the library's tile-blocked layout [B][N/112][N/112][112][112] (same shape, a copy)."""
    B, N, _ = A.shape
    nb = N // BLOCK
    return A.view(B, nb, BLOCK, nb, BLOCK).permute(0, 1, 3, 2, 4).contiguous().view(B, N, N)


def unblock(Ab):
    """Inverse of block(): tile-blocked -> [B,N,N] row-major (a copy)."""
    B, N, _ = Ab.shape
    nb = N // BLOCK
    return Ab.view(B, nb, nb, BLOCK, BLOCK).permute(0, 1, 3, 2, 4).contiguous().view(B, N, N)


def gather(train_in, train_tgt, test_in, s, p, locs, tidx, out=None):
    """-> X [B,n,8], y [B,n], Xs [B,m,8]   (gpsr_gather).  out = (X, y, Xs): write into these (contiguous row slices of
    a batch buffer) instead of allocating."""
    lib = _lib.load()
    H, W = train_in.shape
    B, nt = locs.shape[0], tidx.shape[0]
    n, m = nt * nt, (p * s - 2) ** 2
    dev = train_in.device
    if out is None:
        X = torch.empty((B, n, 8), dtype=torch.float64, device=dev)
        y = torch.empty((B, n), dtype=torch.float64, device=dev)
        Xs = torch.empty((B, m, 8), dtype=torch.float64, device=dev)
    else:
        X, y, Xs = out
        if X.shape != (B, n, 8) or y.shape != (B, n) or Xs.shape != (B, m, 8):
            raise _lib.GpsrError(f"gather: out shapes {tuple(X.shape)}, {tuple(y.shape)}, {tuple(Xs.shape)} do not fit B={B}, n={n}, m={m}")
    _lib.check(lib.gpsr_gather(_p(train_in, torch.float64), _p(train_tgt, torch.float64),
                               _p(test_in, torch.float64), H, W, s, p, _p(locs, torch.int32), B,
                               _p(tidx, torch.int32), nt, _p(X, torch.float64), _p(y, torch.float64),
                               _p(Xs, torch.float64), _stream()), "gpsr_gather")
    return X, y, Xs


def init_hypers(y, kernel_id, init_raw_os=0.0, init_raw_ls=0.0, carry=None):
    """carry: None | [4] (every GP continues that one chain link) | [B,4] (GP b continues chain b)."""
    lib = _lib.load()
    B, n = y.shape
    raw = torch.empty((B, 4), dtype=torch.float32, device=y.device)
    stride = 0
    if carry is not None and carry.dim() == 2:
        if carry.shape != (B, 4):
            raise _lib.GpsrError(f"carry must be [4] or [{B},4], got {tuple(carry.shape)}")
        stride = 4
    _lib.check(lib.gpsr_init_hypers(_p(y, torch.float64), B, n, kernel_id, float(init_raw_os), float(init_raw_ls),
                                    _p(carry, torch.float32), stride, _p(raw), _stream()), "gpsr_init_hypers")
    return raw


def covariance(X, kernel_id, raw):
    lib = _lib.load()
    B, n, d = X.shape
    N = padded_n(n)
    K = torch.zeros((B, N, N), dtype=torch.float64, device=X.device)
    _lib.check(lib.gpsr_covariance(_p(X, torch.float64), B, n, d, kernel_id, _p(raw, torch.float32), _p(K),
                                   _stream()), "gpsr_covariance")
    return unblock(K)      # row-major for Python callers; the library's layout is tile-blocked (include/gpsr.h)


def covariance_grad(X, kernel_id, raw):
    """-> K, dK/dlengthscale (x outputscale), both [B,N,N] row-major, lower block-triangle valid (gpsr_covariance_grad)."""
    lib = _lib.load()
    B, n, d = X.shape
    N = padded_n(n)
    K = torch.zeros((B, N, N), dtype=torch.float64, device=X.device)
    dK = torch.zeros((B, N, N), dtype=torch.float64, device=X.device)
    _lib.check(lib.gpsr_covariance_grad(_p(X, torch.float64), B, n, d, kernel_id, _p(raw, torch.float32), _p(K),
                                        _p(dK), _stream()), "gpsr_covariance_grad")
    return unblock(K), unblock(dK)


def invert_(A):
    """In-place on a row-major [B,N,N] tensor: A[b] <- -inv(A[b]) (lower block-triangle).  Returns logdet, flags.
    (The library works on the tile-blocked layout: converted on the way in and out.)"""
    lib = _lib.load()
    B, N, _ = A.shape
    nbytes = lib.gpsr_invert_workspace_bytes(B, N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=A.device)
    logdet = torch.empty(B, dtype=torch.float64, device=A.device)
    flags = torch.empty(B, dtype=torch.int32, device=A.device)
    Ab = block(A)
    _lib.check(lib.gpsr_invert(_p(Ab, torch.float64), B, N, _p(logdet), _p(flags), _p(ws), nbytes, _stream()),
               "gpsr_invert")
    A.copy_(unblock(Ab))
    return logdet, flags


def fit_workspace(B, n, d, device):
    lib = _lib.load()
    nbytes = lib.gpsr_fit_workspace_bytes(B, n, d)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def fit(X, y, kernel_id, raw, iters=50, lr=0.1, workspace=None, want_loss=False):
    """Adam fit in place on raw [B,4]; returns alpha [B,N], flags [B] (, loss [B])."""
    lib = _lib.load()
    B, n, d = X.shape
    N = padded_n(n)
    if workspace is None:
        workspace = fit_workspace(B, n, d, X.device)
    alpha = torch.empty((B, N), dtype=torch.float64, device=X.device)
    flags = torch.empty(B, dtype=torch.int32, device=X.device)
    loss = torch.empty(B, dtype=torch.float64, device=X.device) if want_loss else None
    _lib.check(lib.gpsr_fit(_p(X, torch.float64), _p(y, torch.float64), B, n, d, kernel_id, _p(raw, torch.float32),
                            iters, float(lr), _p(alpha), _p(loss), _p(flags), _p(workspace), workspace.numel(),
                            _stream()), "gpsr_fit")
    return (alpha, flags, loss) if want_loss else (alpha, flags)


def set_dual_stream(on):
    """gpsr_set_dual_stream: two half-batches on two streams inside gpsr_fit (opt-in). Returns the previous setting."""
    return bool(_lib.load().gpsr_set_dual_stream(1 if on else 0))


def set_split_small(on):
    """gpsr_set_split_small: the small launches of a big batch as two halves on two streams (default on). Returns the previous setting."""
    return bool(_lib.load().gpsr_set_split_small(1 if on else 0))


def set_fused(on):
    """gpsr_set_fused: the one-launch per-GP route of gpsr_fit (opt-in, bit-identical, measured slower). Returns the previous setting."""
    return bool(_lib.load().gpsr_set_fused(1 if on else 0))


def loss_grad(X, y, kernel_id, raw, workspace=None):
    lib = _lib.load()
    B, n, d = X.shape
    if workspace is None:
        workspace = fit_workspace(B, n, d, X.device)
    loss = torch.empty(B, dtype=torch.float64, device=X.device)
    grad = torch.empty((B, 4), dtype=torch.float32, device=X.device)
    flags = torch.empty(B, dtype=torch.int32, device=X.device)
    _lib.check(lib.gpsr_loss_grad(_p(X, torch.float64), _p(y, torch.float64), B, n, d, kernel_id,
                                  _p(raw, torch.float32), _p(loss), _p(grad), _p(flags), _p(workspace),
                                  workspace.numel(), _stream()), "gpsr_loss_grad")
    return loss, grad, flags


def predict(X, Xs, alpha, raw, kernel_id):
    lib = _lib.load()
    B, n, d = X.shape
    if d != 8:
        raise _lib.GpsrError("gpsr_predict needs d == 8")
    m = Xs.shape[1]
    out = torch.empty((B, m), dtype=torch.float64, device=X.device)
    _lib.check(lib.gpsr_predict(_p(X, torch.float64), _p(Xs, torch.float64), _p(alpha, torch.float64),
                                _p(raw, torch.float32), B, n, m, kernel_id, _p(out), _stream()), "gpsr_predict")
    return out


def combine(pred, test_in, H, W, s, p, rows, cols, mode=AVERAGE, clip=True):
    """test_in=None: pred holds whole [B,P,P] patches (PatchHandler.combine_patches)."""
    lib = _lib.load()
    out = torch.empty((H * s, W * s), dtype=torch.float64, device=pred.device)
    _lib.check(lib.gpsr_combine(_p(pred, torch.float64), _p(test_in, torch.float64), H, W, s, p,
                                _p(rows, torch.int32), rows.numel(), _p(cols, torch.int32), cols.numel(), mode,
                                int(clip), _p(out), _stream()), "gpsr_combine")
    return out


def predict_pixels_fused(train_in, train_tgt, test_in, s, p, rows, cols, tidx, kernel_id, init_raw_os=0.0,
                         init_raw_ls=0.0, iters=50, lr=0.1, mode=AVERAGE, chunk=0):
    """One-call GPRSR.predict_pixels for one image (gpsr_predict_pixels) -> out [H*s,W*s], raw [B,4], flags [B]."""
    lib = _lib.load()
    H, W = train_in.shape
    nr, nc, nt = rows.numel(), cols.numel(), tidx.numel()
    B = nr * nc
    dev = train_in.device
    nbytes = lib.gpsr_predict_pixels_workspace_bytes(s, p, nt, B, chunk)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    out = torch.empty((H * s, W * s), dtype=torch.float64, device=dev)
    raw = torch.empty((B, 4), dtype=torch.float32, device=dev)
    flags = torch.empty(B, dtype=torch.int32, device=dev)
    _lib.check(lib.gpsr_predict_pixels(_p(train_in, torch.float64), _p(train_tgt, torch.float64),
                                       _p(test_in, torch.float64), H, W, s, p, _p(rows, torch.int32), nr,
                                       _p(cols, torch.int32), nc, _p(tidx, torch.int32), nt, kernel_id,
                                       float(init_raw_os), float(init_raw_ls), iters, float(lr), mode, chunk,
                                       _p(out), _p(raw), _p(flags), _p(ws), nbytes, _stream()), "gpsr_predict_pixels")
    return out, raw, flags


def agpr_weights(X, alpha, raw):
    lib = _lib.load()
    n, d = X.shape
    w = torch.empty(d, dtype=torch.float64, device=X.device)
    _lib.check(lib.gpsr_agpr_weights(_p(X, torch.float64), _p(alpha, torch.float64), _p(raw, torch.float32), n, d,
                                     _p(w), _stream()), "gpsr_agpr_weights")
    return w


def agpr_filter(img, L, w, raw):
    lib = _lib.load()
    Hs, Ws = img.shape
    out = torch.empty_like(img)
    _lib.check(lib.gpsr_agpr_filter(_p(img, torch.float64), Hs, Ws, L, _p(w, torch.float64), _p(raw, torch.float32),
                                    _p(out), _stream()), "gpsr_agpr_filter")
    return out


def selftest_mfma(A, Bm):
    lib = _lib.load()
    D = torch.empty((16, 16), dtype=torch.float64, device=A.device)
    _lib.check(lib.gpsr_selftest_mfma(_p(A, torch.float64), _p(Bm, torch.float64), _p(D), _stream()),
               "gpsr_selftest_mfma")
    return D


# ------------------------------------------------------------ image stages
def u8_to_unit(img_u8):
    lib = _lib.load()
    out = torch.empty(img_u8.shape, dtype=torch.float64, device=img_u8.device)
    _lib.check(lib.gpsr_u8_to_unit(_p(img_u8, torch.uint8), _p(out), img_u8.numel(), _stream()), "gpsr_u8_to_unit")
    return out


def unit_to_u8(img, scale=255.0):
    lib = _lib.load()
    out = torch.empty(img.shape, dtype=torch.uint8, device=img.device)
    _lib.check(lib.gpsr_unit_to_u8(_p(img, torch.float64), _p(out), img.numel(), float(scale), _stream()),
               "gpsr_unit_to_u8")
    return out


def resample(img, Hd, Wd, xi, xw, yi, yw):
    lib = _lib.load()
    Hs, Ws = img.shape
    taps = xi.shape[1]
    out = torch.empty((Hd, Wd), dtype=torch.float64, device=img.device)
    _lib.check(lib.gpsr_resample(_p(img, torch.float64), Hs, Ws, _p(out), Hd, Wd, _p(xi, torch.int32),
                                 _p(xw, torch.float64), _p(yi, torch.int32), _p(yw, torch.float64), taps,
                                 _stream()), "gpsr_resample")
    return out


def decimate2(img):
    lib = _lib.load()
    Hs, Ws = img.shape
    out = torch.empty((Hs // 2, Ws // 2), dtype=torch.float64, device=img.device)
    _lib.check(lib.gpsr_decimate2(_p(img, torch.float64), Hs, Ws, _p(out), _stream()), "gpsr_decimate2")
    return out


def sepfilter_sym(img, kern):
    lib = _lib.load()
    H, W = img.shape
    out = torch.empty_like(img)
    _lib.check(lib.gpsr_sepfilter_sym(_p(img, torch.float64), H, W, _p(out), _p(kern, torch.float64), kern.numel(),
                                      _stream()), "gpsr_sepfilter_sym")
    return out


def filter3x3(img, kern):
    lib = _lib.load()
    H, W = img.shape
    out = torch.empty_like(img)
    _lib.check(lib.gpsr_filter3x3(_p(img, torch.float64), H, W, _p(out), _p(kern, torch.float64), _stream()),
               "gpsr_filter3x3")
    return out


def axpy(a, b, alpha):
    lib = _lib.load()
    out = torch.empty_like(a)
    _lib.check(lib.gpsr_axpy(_p(a, torch.float64), _p(b, torch.float64), float(alpha), _p(out), a.numel(), _stream()),
               "gpsr_axpy")
    return out


# ------------------------------------------------------------ uint8 colour stages + metrics
def bgr2gray(bgr):
    """[H,W,3] uint8 device (cv.imread layout) -> [H,W] uint8   (gpsr_bgr2gray)."""
    lib = _lib.load()
    H, W, _ = bgr.shape
    out = torch.empty((H, W), dtype=torch.uint8, device=bgr.device)
    _lib.check(lib.gpsr_bgr2gray(_p(bgr, torch.uint8), _p(out), H * W, _stream()), "gpsr_bgr2gray")
    return out


def _split3(fn, name, bgr):
    H, W, _ = bgr.shape
    planes = [torch.empty((H, W), dtype=torch.uint8, device=bgr.device) for _ in range(3)]
    _lib.check(fn(_p(bgr, torch.uint8), _p(planes[0]), _p(planes[1]), _p(planes[2]), H * W, _stream()), name)
    return planes


def _merge3(fn, name, a, b, c):
    H, W = a.shape
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=a.device)
    _lib.check(fn(_p(a, torch.uint8), _p(b, torch.uint8), _p(c, torch.uint8), _p(out), H * W, _stream()), name)
    return out


def bgr2yuv(bgr):
    return _split3(_lib.load().gpsr_bgr2yuv, "gpsr_bgr2yuv", bgr)


def yuv2bgr(y, u, v):
    return _merge3(_lib.load().gpsr_yuv2bgr, "gpsr_yuv2bgr", y, u, v)


def bgr2ycrcb(bgr):
    return _split3(_lib.load().gpsr_bgr2ycrcb, "gpsr_bgr2ycrcb", bgr)


def ycrcb2bgr(y, cr, cb):
    return _merge3(_lib.load().gpsr_ycrcb2bgr, "gpsr_ycrcb2bgr", y, cr, cb)


def resample_u8(img, Hd, Wd, xi, xw, yi, yw):
    lib = _lib.load()
    Hs, Ws = img.shape
    out = torch.empty((Hd, Wd), dtype=torch.uint8, device=img.device)
    _lib.check(lib.gpsr_resample_u8(_p(img, torch.uint8), Hs, Ws, _p(out), Hd, Wd, _p(xi, torch.int32),
                                    _p(xw, torch.float32), _p(yi, torch.int32), _p(yw, torch.float32), _stream()),
               "gpsr_resample_u8")
    return out


def sepfilter_u8(img, kern):
    """uint8 device plane [H,W], int32 fixed-point taps (sum 256) -> blurred uint8 plane (gpsr_sepfilter_u8)."""
    lib = _lib.load()
    H, W = img.shape
    out = torch.empty_like(img)
    _lib.check(lib.gpsr_sepfilter_u8(_p(img, torch.uint8), H, W, _p(out), _p(kern, torch.int32), kern.numel(), _stream()),
               "gpsr_sepfilter_u8")
    return out


def metrics(a, b, crop):
    """Two [H,W] uint8 device images -> device float64 [3] = (PSNR dB, SSIM, MSE) after cropping (gpsr_metrics)."""
    lib = _lib.load()
    H, W = a.shape
    if b.shape != a.shape:
        raise _lib.GpsrError(f"metrics: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
    nbytes = lib.gpsr_metrics_workspace_bytes(H, W, crop)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=a.device)
    out = torch.empty(3, dtype=torch.float64, device=a.device)
    _lib.check(lib.gpsr_metrics(_p(a, torch.uint8), _p(b, torch.uint8), H, W, crop, _p(out), _p(ws), nbytes, _stream()),
               "gpsr_metrics")
    return out


def active_sample(data, K, n_pick, bandwidth, w_rep, w_div):
    """[N,D] float64 device rows -> (int32 [n_pick] chosen row indices in the order chosen, float64 [N]
    representativeness scores): the training-set selection of agpr.py:120-137 (gpsr_active_sample)."""
    lib = _lib.load()
    N, D = data.shape
    nbytes = lib.gpsr_active_sample_workspace_bytes(N)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=data.device)
    idx = torch.empty(n_pick, dtype=torch.int32, device=data.device)
    rep = torch.empty(N, dtype=torch.float64, device=data.device)
    _lib.check(lib.gpsr_active_sample(_p(data, torch.float64), N, D, int(K), int(n_pick), float(bandwidth), float(w_rep), float(w_div),
                                      _p(idx, torch.int32), _p(rep), _p(ws), nbytes, _stream()), "gpsr_active_sample")
    return idx, rep


# ------------------------------------------------------------ kernel timing
PROF_TAGS = ("update", "panel", "pivot", "build", "grad1", "grad2", "predict", "update_last", "update2", "update_sel", "fused")


def profile_begin(max_records=65536, tags=None):
    """tags: names out of PROF_TAGS to bracket (None: all).  Two event records per bracketed launch."""
    lib = _lib.load()
    mask = 0xFFFFFFFF if tags is None else sum(1 << PROF_TAGS.index(t) for t in tags)
    _lib.check(lib.gpsr_profile_select(mask), "gpsr_profile_select")
    _lib.check(lib.gpsr_profile_begin(int(max_records)), "gpsr_profile_begin")


def profile_end():
    """-> {tag: (launch_count, total_ms)} measured with HIP events on the launch stream."""
    import ctypes
    n = len(PROF_TAGS)
    counts = (ctypes.c_int32 * n)()
    ms = (ctypes.c_double * n)()
    _lib.check(_lib.load().gpsr_profile_end(ctypes.cast(counts, ctypes.c_void_p), ctypes.cast(ms, ctypes.c_void_p)),
               "gpsr_profile_end")
    return {t: (int(counts[i]), float(ms[i])) for i, t in enumerate(PROF_TAGS)}
