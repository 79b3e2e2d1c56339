"""Batched SRGPR engine: every patch GP of one predict_pixels call (and of many
images at once) is fitted in one batch on the GPU.

This is the reference's hot function GPRSR.predict_pixels (gprsr.py:126-210)
re-designed for MI355X: gather -> init -> fit (50 Adam steps on the batched
marginal likelihood) -> fused predictive mean -> ordered recombination, all in
HIP kernels behind include/gpsr.h.  Device tensors in, device tensors out.

Warm start.  In the reference the kernel objects are module-level singletons
(kernels.py:7-13) and only the RBF branch resets lengthscale/outputscale per
patch (gprsr.py:167-172); for the other kernels patch k starts Adam from patch
k-1's optimum -- a serial chain.  predict_pixels_batch does both:
  chain_carry=None   every patch starts from the fresh-process values (raw 0):
      independent GPs, one batch, patch-sharded across GPUs;
  chain_carry=[jobs,4]   every job (image) is one chain over its patches; link k of
      all chains is one batched launch sequence (BatchedGP.fit_predict_chains),
      whole chains are dealt to the ranks.
GPRSR (gprsr.py) maps its warm_start modes onto these.  For RBF both are the same.
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


def owned_jobs(shapes, p, stride_percentage, shard, chained):
    """Which jobs (images of LR shape (H, W)) this rank fits patches of -- the table predict_pixels_batch itself
    derives: a contiguous slice of the global patch list (independent GPs) or whole chains dealt out longest first."""
    from . import dist as gdist
    if shard is None:
        return [True] * len(shapes)
    rank, world = shard
    stride = int((p - 2) * stride_percentage)
    lens = [len(axis_starts(H, p, stride)) * len(axis_starts(W, p, stride)) for H, W in shapes]
    if chained:
        owner = gdist.deal_chains(lens, world)
        return [owner[j] == rank for j in range(len(shapes))]
    offs = np.concatenate([[0], np.cumsum(lens)])
    lo, hi = gdist.shard_range(int(offs[-1]), rank, world)
    return [bool(lens[j]) and lo < offs[j + 1] and hi > offs[j] for j in range(len(shapes))]


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

    def fit_predict(self, X, y, Xs, init_raw=(0.0, 0.0)):
        """Independent GPs.  X [B,n,8], y [B,n], Xs [B,m,8] -> pred [B,m], raw [B,4], flags [B]."""
        B, n, d = X.shape
        dev = X.device
        raw = ops.init_hypers(y, self.kernel_id, init_raw[0], init_raw[1])
        pred = torch.empty((B, Xs.shape[1]), dtype=torch.float64, device=dev)
        flags = torch.empty(B, dtype=torch.int32, device=dev)
        bc = self.chunk_size(B, n, d)
        bc = -(-B // -(-B // bc))          # balanced chunks: 32 400 GPs under a 13 443-GP budget run as 3 x 10 800, not 13 443 + 13 443 + 5 514
        ws = self._workspace(bc, n, d, dev)
        for b0 in range(0, B, bc):
            b1 = min(B, b0 + bc)
            alpha, fl = ops.fit(X[b0:b1], y[b0:b1], self.kernel_id, raw[b0:b1], self.iters, self.lr, ws)
            pred[b0:b1] = ops.predict(X[b0:b1], Xs[b0:b1], alpha, raw[b0:b1], self.kernel_id)
            flags[b0:b1] = fl
        return pred, raw, flags

    def fit_predict_chains(self, X, y, Xs, lens, carry):
        """Warm-start chains: the GPs are the concatenation of len(lens) chains, chain j holding
        lens[j] consecutive GPs.  Inside a chain GP k starts Adam from GP k-1's raw outputscale / lengthscale
        (the reference's kernel singleton, kernels.py:7-13 + gprsr.py:153-174); the chains do not see each
        other, so link k of every chain that is still running is fitted in ONE batched launch sequence.
        carry [len(lens),4] float32 (device, columns 2,3 used): what each chain starts from; on return, what
        each chain ended with (fed to the next predict_pixels pass of the same image)."""
        B, n, d = X.shape
        dev = X.device
        m = Xs.shape[1]
        nch = len(lens)
        assert sum(lens) == B and carry.shape == (nch, 4)
        raw = torch.empty((B, 4), dtype=torch.float32, device=dev)
        flags = torch.empty(B, dtype=torch.int32, device=dev)
        if B == 0:
            return torch.empty((B, m), dtype=torch.float64, device=dev), raw, flags
        # longest chain first: the chains still running at link k are then a prefix of `order`
        order = sorted(range(nch), key=lambda j: -lens[j])
        offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
        steps, flat = [], []
        for k in range(max(lens)):
            act = [j for j in order if lens[j] > k]
            steps.append(len(act))
            flat.extend(int(offs[j]) + k for j in act)
        idx_all = torch.as_tensor(np.asarray(flat, dtype=np.int64)).to(dev)
        order_t = torch.as_tensor(np.asarray(order, dtype=np.int64)).to(dev)
        cur = carry.index_select(0, order_t).contiguous()      # [nch,4] in `order`
        # the chains of a link are independent GPs: more of them than the workspace budget holds (many images per
        # call) run in groups, like fit_predict's chunks
        bc = self.chunk_size(nch, n, d)
        ws = self._workspace(min(nch, bc), n, d, dev)
        alpha_all = torch.empty((B, ops.padded_n(n)), dtype=torch.float64, device=dev)   # 8 N bytes per GP: X / Xs, which
        pos = 0                                                                        # the caller holds, are 64 (n + m)
        for k, bk in enumerate(steps):
            for g0 in range(0, bk, bc):
                g1 = min(bk, g0 + bc)
                idx = idx_all[pos + g0:pos + g1]
                Xk, yk = X.index_select(0, idx), y.index_select(0, idx)
                r = ops.init_hypers(yk, self.kernel_id, carry=cur[g0:g1])
                alpha, fl = ops.fit(Xk, yk, self.kernel_id, r, self.iters, self.lr, ws)
                alpha_all.index_copy_(0, idx, alpha)
                raw.index_copy_(0, idx, r)
                flags.index_copy_(0, idx, fl)
                cur[g0:g1] = r
            pos += bk
        # the predictive means do not feed the chain: one batched launch for all links at the end (the mean of a GP
        # depends on its own X, alpha and hyper-parameters only), instead of one small launch per link
        pred = ops.predict(X, Xs, alpha_all, raw, self.kernel_id)
        carry.index_copy_(0, order_t, cur)
        return pred, raw, flags


def predict_pixels_batch(jobs, s, p, gp, stride_percentage=0.9, use_all_pixels=True, mode=ops.AVERAGE,
                         chain_carry=None, stats=None, shard=None, recombine="all"):
    """jobs: list of (training_input, training_target, test_input) device float64
    images (gprsr.py:126).  All patches of all jobs form ONE batch of GPs.
    Returns the list of recombined, clipped HR images.

    chain_carry=None: independent GPs.  chain_carry = float32 [len(jobs),4] device tensor: every job is a
    warm-start chain over its patches in PatchHandler order (BatchedGP.fit_predict_chains); row j is what chain
    j starts from and, on return, what it ended with.

    shard=(rank, world): independent GPs -- this rank fits a contiguous slice of the global patch list;
    chains -- this rank fits whole jobs (a chain is serial; the jobs are dealt out longest first).  The
    predicted patch interiors (+ raw hyper-parameters and flags, packed into the same rows) are exchanged
    with ONE all-gather (RCCL) before recombination -- the only collective on the path.

    recombine="owned" (sharded runs): only the jobs this rank fitted patches of are recombined; the others come
    back as None.  A rank needs the phase-1 image of a job only to gather its own phase-2 patches of that job
    (the sharding is the same in both phases), so with many images per call -- N copies of a dataset on N
    ranks -- every rank skips the image stages of the other ranks' images instead of repeating them.  A job
    this rank owns nothing of may then also be handed in with training_input / test_input = None (its geometry
    comes from training_target, which the caller always has).
    """
    from . import dist as gdist
    dev = jobs[0][1].device
    tidx = torch.as_tensor(training_indices(p, use_all_pixels)).to(dev)
    geo = []
    assert recombine in ("all", "owned")
    for tin, ttg, tst in jobs:
        H, W = ttg.shape
        if tin is not None:    # None: a job of another rank's share whose phase-1 image this rank skipped
            assert tin.shape == ttg.shape
        if tst is not None:
            assert tst.shape[0] == H * s and tst.shape[1] == W * s
        assert p <= H and p <= W, f"Patch size {p} is too big for image size {(H, W)}"
        rows, cols, locs = _geometry(H, W, s, p, stride_percentage)
        geo.append((H, W, rows, cols, locs))
    lens = [g[4].shape[0] for g in geo]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    B = int(offs[-1])
    rank, world = shard if shard is not None else (0, 1)
    nt = int(tidx.shape[0])
    n, m = nt * nt, (p * s - 2) ** 2
    # which global patch indices each rank owns (every rank computes the whole table: no communication)
    if chain_carry is None:
        owned = [np.arange(*gdist.shard_range(B, r, world), dtype=np.int64) for r in range(world)]
        my_jobs = None
    else:
        assert chain_carry.shape == (len(jobs), 4) and chain_carry.dtype == torch.float32
        owner = gdist.deal_chains(lens, world)
        job_lists = [[j for j in range(len(jobs)) if owner[j] == r] for r in range(world)]
        owned = [np.concatenate([np.arange(offs[j], offs[j + 1]) for j in jl]) if jl else np.zeros(0, np.int64)
                 for jl in job_lists]
        my_jobs = job_lists[rank]
    mine = owned[rank]
    # gather this rank's patches, job by job, straight into the batch buffers (one allocation, no concatenation pass)
    nmine = int(mine.size)
    X = torch.empty((nmine, n, 8), dtype=torch.float64, device=dev)   # a rank without work (fewer patches / chains than
    y = torch.empty((nmine, n), dtype=torch.float64, device=dev)      # ranks) still takes part in the exchange: 0 rows
    Xs = torch.empty((nmine, m, 8), dtype=torch.float64, device=dev)
    pos = 0
    for j, ((tin, ttg, tst), (H, W, rows, cols, locs)) in enumerate(zip(jobs, geo)):
        sel = mine[(mine >= offs[j]) & (mine < offs[j + 1])] - offs[j]
        if sel.size:
            assert tin is not None and tst is not None, f"job {j}: this rank owns patches of an image it was not given"
            k = int(sel.size)
            ops.gather(tin, ttg, tst, s, p, torch.as_tensor(np.ascontiguousarray(locs[sel])).to(dev), tidx,
                       out=(X[pos:pos + k], y[pos:pos + k], Xs[pos:pos + k]))
            pos += k
    assert pos == nmine
    if chain_carry is None:
        if X.shape[0]:
            pred, raw, flags = gp.fit_predict(X, y, Xs)
        else:
            pred = torch.empty((0, m), dtype=torch.float64, device=dev)
            raw = torch.empty((0, 4), dtype=torch.float32, device=dev)
            flags = torch.empty(0, dtype=torch.int32, device=dev)
    else:
        mj = torch.as_tensor(np.asarray(my_jobs, dtype=np.int64)).to(dev)
        carry_l = chain_carry.index_select(0, mj).contiguous()
        pred, raw, flags = gp.fit_predict_chains(X, y, Xs, [lens[j] for j in my_jobs], carry_l)
        if world > 1:
            carry_all = gdist.exchange_rows([carry_l], [np.asarray(jl, dtype=np.int64) for jl in job_lists],
                                            len(jobs), rank, world)[0]
            chain_carry.copy_(carry_all)
        else:
            chain_carry.index_copy_(0, mj, carry_l)
    if world > 1:
        pred, raw, flags = gdist.exchange_rows([pred, raw, flags], owned, B, rank, world)
    if stats is not None:
        stats.raw.append(raw)
        stats.flags.append(flags)
    outs = []
    for j, ((tin, ttg, tst), (H, W, rows, cols, locs)) in enumerate(zip(jobs, geo)):
        if tst is None or (recombine == "owned" and not np.any((mine >= offs[j]) & (mine < offs[j + 1]))):
            outs.append(None)
            continue
        out = ops.combine(pred[offs[j]:offs[j + 1]], tst, H, W, s, p, torch.as_tensor(rows).to(dev),
                          torch.as_tensor(cols).to(dev), mode, clip=True)
        outs.append(out)
    return outs
