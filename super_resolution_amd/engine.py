"""Batched SRGPR engine: every patch GP of one predict_pixels call (and of many
images at once) is fitted in one batch on the GPU.

This is the reference's hot function GPRSR.predict_pixels (gprsr.py:126-210)
re-designed for MI355X: gather -> init -> fit (50 Adam steps on the batched
marginal likelihood) -> fused predictive mean -> ordered recombination, all in
HIP kernels behind include/gpsr.h.  Device tensors in, device tensors out.

Warm start.  In the reference the kernel objects are module-level singletons
(kernels.py:7-13) and only the RBF branch resets lengthscale/outputscale per
patch (gprsr.py:167-172); for the other kernels patch k starts Adam from patch
k-1's optimum -- a serial chain.
  warm_start="independent" (default): every patch starts from the
      fresh-process values (raw 0), patches are independent, batched, sharded.
  warm_start="chain": the reference's literal behaviour, one GP after the other
      (B=1 launches, hyper-parameters carried on the device).  Validation only.
For RBF both are the same thing.
"""
import numpy as np
import torch

from . import ops
from .image_patches import axis_starts


class NotPSDError(RuntimeError):
    """A patch covariance lost positive definiteness during the fit (a pivot <= 0 or NaN).

    The reference fails the same way: GPyTorch's psd_safe_cholesky raises NotPSDError once its
    jitter retries (1e-8, 1e-7, 1e-6 on float64) are exhausted.  With the
    likelihood's noise floor of 1e-4 on the diagonal a finite patch cannot get there in
    float64, and a jitter of <= 1e-6 cannot rescue one that does (NaN / Inf pixels), so the
    retries are not replayed: the failure is reported per GP in `flags` and raised here."""


class FitStats:
    """Per-call record (device tensors): raw hyper-parameters and failure flags."""

    def __init__(self):
        self.raw = []
        self.flags = []


def _geometry(H, W, s, p, stride_percentage):
    stride = int((p - 2) * stride_percentage)
    rows = axis_starts(H, p, stride)
    cols = axis_starts(W, p, stride)
    rr, cc = np.meshgrid(rows, cols, indexing="ij")
    locs = np.stack([rr.reshape(-1), cc.reshape(-1)], axis=1).astype(np.int32)
    return rows, cols, locs


def training_indices(p, use_all_pixels=True):
    """gprsr.py:137-140."""
    if use_all_pixels:
        return np.arange(1, p - 1, dtype=np.int32)
    return np.array([i for i in range(1, p // 2, 2)] + [i for i in range(p // 2, p - 1, 2)], dtype=np.int32)


class BatchedGP:
    """Fits/predicts B independent patch GPs, chunked so the workspace stays under a budget."""

    def __init__(self, kernel_id, iters=50, lr=0.1, max_workspace_bytes=48 << 30):
        self.kernel_id = kernel_id
        self.iters = iters
        self.lr = lr
        self.max_workspace_bytes = max_workspace_bytes
        self._ws = None

    def _workspace(self, B, n, d, device):
        from . import _lib
        need = _lib.load().gpsr_fit_workspace_bytes(B, n, d)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def chunk_size(self, B, n, d):
        from . import _lib
        per = _lib.load().gpsr_fit_workspace_bytes(1, n, d)
        return max(1, min(B, int(self.max_workspace_bytes // per)))

    def fit_predict(self, X, y, Xs, init_raw=(0.0, 0.0), chain_kernel=None):
        """X [B,n,8], y [B,n], Xs [B,m,8] -> pred [B,m], raw [B,4], flags [B]."""
        B, n, d = X.shape
        dev = X.device
        if chain_kernel is not None:
            return self._fit_predict_chain(X, y, Xs, chain_kernel)
        raw = ops.init_hypers(y, self.kernel_id, init_raw[0], init_raw[1])
        pred = torch.empty((B, Xs.shape[1]), dtype=torch.float64, device=dev)
        flags = torch.empty(B, dtype=torch.int32, device=dev)
        bc = self.chunk_size(B, n, d)
        ws = self._workspace(bc, n, d, dev)
        for b0 in range(0, B, bc):
            b1 = min(B, b0 + bc)
            alpha, fl = ops.fit(X[b0:b1], y[b0:b1], self.kernel_id, raw[b0:b1], self.iters, self.lr, ws)
            pred[b0:b1] = ops.predict(X[b0:b1], Xs[b0:b1], alpha, raw[b0:b1], self.kernel_id)
            flags[b0:b1] = fl
        return pred, raw, flags

    def _fit_predict_chain(self, X, y, Xs, kern):
        """Serial warm-start chain: GP b starts from GP b-1's outputscale/lengthscale;
        the chain enters and leaves through the kernel object's raw values."""
        B, n, d = X.shape
        dev = X.device
        raw = torch.empty((B, 4), dtype=torch.float32, device=dev)
        pred = torch.empty((B, Xs.shape[1]), dtype=torch.float64, device=dev)
        flags = torch.empty(B, dtype=torch.int32, device=dev)
        ws = self._workspace(1, n, d, dev)
        for b in range(B):
            if b == 0:
                r = ops.init_hypers(y[0:1], self.kernel_id, kern.raw_outputscale, kern.base_kernel.raw_parameter)
            else:
                r = ops.init_hypers(y[b:b + 1], self.kernel_id, carry=raw[b - 1])
            alpha, fl = ops.fit(X[b:b + 1], y[b:b + 1], self.kernel_id, r, self.iters, self.lr, ws)
            pred[b:b + 1] = ops.predict(X[b:b + 1], Xs[b:b + 1], alpha, r, self.kernel_id)
            raw[b] = r[0]
            flags[b] = fl[0]
        last = raw[B - 1].cpu()
        kern.raw_outputscale = float(last[2])
        kern.base_kernel.raw_parameter = float(last[3])
        return pred, raw, flags


def predict_pixels_batch(jobs, s, p, gp, stride_percentage=0.9, use_all_pixels=True, mode=ops.AVERAGE,
                         chain_kernel=None, stats=None, shard=None):
    """jobs: list of (training_input, training_target, test_input) device float64
    images (gprsr.py:126).  All patches of all jobs form ONE batch of GPs.
    Returns the list of recombined, clipped HR images.

    shard=(rank, world): this rank fits only its contiguous slice of the global
    patch list and the predicted patch interiors are all-gathered (RCCL) before
    recombination -- the only exchange on the path.
    """
    dev = jobs[0][0].device
    tidx = torch.as_tensor(training_indices(p, use_all_pixels)).to(dev)
    geo = []
    for tin, ttg, tst in jobs:
        assert tin.shape == ttg.shape
        assert tst.shape[0] == tin.shape[0] * s and tst.shape[1] == tin.shape[1] * s
        H, W = tin.shape
        assert p <= H and p <= W, f"Patch size {p} is too big for image size {(H, W)}"
        rows, cols, locs = _geometry(H, W, s, p, stride_percentage)
        geo.append((H, W, rows, cols, locs))
    B = sum(g[4].shape[0] for g in geo)
    rank, world = shard if shard is not None else (0, 1)
    from .dist import shard_range
    g0, g1 = shard_range(B, rank, world)
    # gather only this rank's slice of the global (job-major, row-major) patch list
    Xl, yl, Xsl, off = [], [], [], 0
    for (tin, ttg, tst), (H, W, rows, cols, locs) in zip(jobs, geo):
        nb = locs.shape[0]
        lo, hi = max(g0, off) - off, min(g1, off + nb) - off
        if hi > lo:
            X, y, Xs = ops.gather(tin, ttg, tst, s, p, torch.as_tensor(np.ascontiguousarray(locs[lo:hi])).to(dev), tidx)
            Xl.append(X); yl.append(y); Xsl.append(Xs)
        off += nb
    X = torch.cat(Xl) if len(Xl) > 1 else Xl[0]
    y = torch.cat(yl) if len(yl) > 1 else yl[0]
    Xs = torch.cat(Xsl) if len(Xsl) > 1 else Xsl[0]
    del Xl, yl, Xsl
    pred, raw, flags = gp.fit_predict(X, y, Xs, chain_kernel=chain_kernel)
    if world > 1:
        from .dist import all_gather_rows
        pred = all_gather_rows(pred, B, world)
        raw = all_gather_rows(raw, B, world)
        flags = all_gather_rows(flags, B, world)
    if stats is not None:
        stats.raw.append(raw)
        stats.flags.append(flags)
    outs, b0 = [], 0
    for (tin, ttg, tst), (H, W, rows, cols, locs) in zip(jobs, geo):
        nb = locs.shape[0]
        out = ops.combine(pred[b0:b0 + nb], tst, H, W, s, p, torch.as_tensor(rows).to(dev),
                          torch.as_tensor(cols).to(dev), mode, clip=True)
        outs.append(out)
        b0 += nb
    return outs
