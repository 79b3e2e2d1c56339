"""GPU parity tests, kernel by kernel, through the C ABI (libgpsr.so) against the
CPU oracle.  Bars: bit-exact for the index/copy kernels (gather, combine);
float64 tolerances written at each assert for the numerical kernels."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import gp as ogp
from oracle import patches as opatch


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def to_dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev())


def natural_image(H, W, seed):
    """Smooth + edges + noise, values k/255 (like a uint8 image / 255)."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    img = gaussian_filter(rng.random((H, W)), 3.0)
    img = (img - img.min()) / (img.max() - img.min()) * 219 + 16
    img += rng.normal(0, 2, (H, W))
    for _ in range(4):
        a, b, c = rng.normal(size=3)
        yy, xx = np.mgrid[0:H, 0:W]
        img += 25 * ((a * (yy - H / 2) + b * (xx - W / 2) + 5 * c) > 0)
    return np.clip(np.rint(img), 0, 255) / 255.0


def test_mfma_lane_maps():
    from super_resolution_amd import ops
    rng = np.random.default_rng(0)
    A = rng.integers(-8, 9, (16, 4)).astype(np.float64)
    B = rng.integers(-8, 9, (4, 16)).astype(np.float64)  # asymmetric on purpose
    D = ops.selftest_mfma(to_dev(A), to_dev(B)).cpu().numpy()
    assert np.array_equal(D, A @ B)


@pytest.mark.parametrize("H,W,p,s", [(47, 53, 20, 2), (61, 70, 30, 3), (52, 36, 20, 2)])
def test_gather_bit_exact(H, W, p, s):
    from super_resolution_amd import ops
    rng = np.random.default_rng(H)
    a, b, c = rng.random((H, W)), rng.random((H, W)), rng.random((H * s, W * s))
    locs = opatch.patch_locations((H, W), p)
    tidx = np.arange(1, p - 1, dtype=np.int32)
    X, y, Xs = ops.gather(to_dev(a), to_dev(b), to_dev(c), s, p, to_dev(np.array(locs, dtype=np.int32)), to_dev(tidx))
    for k, loc in enumerate(locs):
        assert np.array_equal(X[k].cpu().numpy(), opatch.features(opatch.extract(a, loc, p, 1)))
        assert np.array_equal(y[k].cpu().numpy(), opatch.targets(opatch.extract(b, loc, p, 1)))
        assert np.array_equal(Xs[k].cpu().numpy(), opatch.features(opatch.extract(c, loc, p, s)))


def test_gather_grid_training_points():
    from super_resolution_amd import ops
    H, W, p, s = 45, 50, 20, 2
    rng = np.random.default_rng(5)
    a, b, c = rng.random((H, W)), rng.random((H, W)), rng.random((H * s, W * s))
    locs = opatch.patch_locations((H, W), p)
    idx = opatch.grid_training_indices(p)
    X, y, _ = ops.gather(to_dev(a), to_dev(b), to_dev(c), s, p, to_dev(np.array(locs, dtype=np.int32)),
                         to_dev(np.array(idx, dtype=np.int32)))
    for k, loc in enumerate(locs):
        assert np.array_equal(X[k].cpu().numpy(), opatch.features_at(opatch.extract(a, loc, p, 1), idx))
        assert np.array_equal(y[k].cpu().numpy(), opatch.targets_at(opatch.extract(b, loc, p, 1), idx))


@pytest.mark.parametrize("tag", ["s2", "s3", "s2dup"])
@pytest.mark.parametrize("mode", [0, 1])
def test_combine_bit_exact_vs_reference_golden(golden_dir, tag, mode):
    """Golden = output of the reference's own PatchHandler.combine_patches."""
    from super_resolution_amd import ops
    from super_resolution_amd.image_patches import grid_of
    d = np.load(f"{golden_dir}/patches_iter_combine.npz")
    H, W, p, s = [int(v) for v in d[tag + "_meta"]]
    up = d[tag + "_up"]
    pred = np.ascontiguousarray(up[:, 1:-1, 1:-1].reshape(up.shape[0], -1))
    rows, cols = grid_of((H, W), p, 0.9)
    out = ops.combine(to_dev(pred), to_dev(d[tag + "_c"]), H, W, s, p, to_dev(rows), to_dev(cols), mode)
    want = np.clip(d[tag + ("_avg" if mode == 0 else "_bil")], 0, 1)
    assert np.array_equal(out.cpu().numpy(), want)


@pytest.mark.parametrize("kid", [0, 1, 2, 3])
@pytest.mark.parametrize("exact_hypers", [True, False])
def test_covariance(kid, exact_hypers):
    """exact_hypers: raw > 20 so softplus(raw) == raw with no float32 transcendental
    in the way -> the float64 kernel arithmetic itself is held to 1e-13 (relative).
    Otherwise the float32 softplus of GPU (ocml) and CPU (torch) may differ by an
    ulp of float32, so the bar is 5e-7 relative."""
    from super_resolution_amd import ops
    rng = np.random.default_rng(kid)
    B, n = 3, 324
    X = rng.random((B, n, 8)) * (30.0 if exact_hypers else 1.0)
    raw = rng.normal(0, 1, (B, 4)).astype(np.float32)
    if exact_hypers:
        raw = (21.0 + 4 * rng.random((B, 4))).astype(np.float32)
    tol = 1e-13 if exact_hypers else 5e-7
    K = ops.covariance(to_dev(X), kid, to_dev(raw)).cpu().numpy()
    N = K.shape[1]
    assert N == 336
    for b in range(B):
        h = ogp.Hypers(*raw[b])
        noise, _, os_, ls = [v.double() for v in h.constrained()]
        want = (os_ * ogp.base_kernel(torch.tensor(X[b]), torch.tensor(X[b]), ls, kid)
                + noise * torch.eye(n, dtype=torch.float64)).detach().numpy()
        got = K[b]
        for I in range(N // 112):          # only the lower block-triangle is defined
            for J in range(I + 1):
                g = got[I * 112:(I + 1) * 112, J * 112:(J + 1) * 112]
                w = np.zeros((112, 112))
                ii, jj = np.arange(I * 112, (I + 1) * 112), np.arange(J * 112, (J + 1) * 112)
                mi, mj = ii < n, jj < n
                w[np.ix_(mi, mj)] = want[np.ix_(ii[mi], jj[mj])]
                if I == J:
                    for t in range(112):
                        if ii[t] >= n:
                            w[t, t] = 1.0
                assert np.abs(g - w).max() <= tol * np.abs(want).max()


def _lower_blocks_to_full(M):
    """Symmetrise from the lower block-triangle (diagonal blocks are full)."""
    N = M.shape[0]
    out = np.array(M)
    for I in range(N // 112):
        for J in range(I):
            out[J * 112:(J + 1) * 112, I * 112:(I + 1) * 112] = M[I * 112:(I + 1) * 112, J * 112:(J + 1) * 112].T
    return out


@pytest.mark.parametrize("N", [112, 224, 336, 448, 560, 784, 1120])   # 1..10 blocks: no pair, pairs + one or two single steps
def test_invert_spd(N):
    from super_resolution_amd import ops
    rng = np.random.default_rng(N)
    B = 3
    A = np.zeros((B, N, N))
    for b in range(B):
        Q = rng.normal(size=(N, N + 8))
        A[b] = Q @ Q.T / N + 0.05 * np.eye(N)
    At = to_dev(A)
    logdet, flags = ops.invert_(At)
    assert int(flags.sum()) == 0
    for b in range(B):
        got = -_lower_blocks_to_full(At[b].cpu().numpy())
        want = np.linalg.inv(A[b])
        assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max()
        assert abs(logdet[b].item() - np.linalg.slogdet(A[b])[1]) < 1e-8 * N


def test_invert_flags_non_spd():
    from super_resolution_amd import ops
    N = 224
    A = np.stack([np.eye(N), np.eye(N)])
    A[1, 150, 150] = -1.0
    _, flags = ops.invert_(to_dev(A))
    assert flags.cpu().tolist() == [0, 1]


@pytest.mark.parametrize("kid,n,d", [(0, 324, 8), (1, 324, 8), (2, 324, 8), (3, 324, 8), (3, 784, 8), (3, 100, 8),
                                     (4, 500, 49), (2, 200, 8), (3, 420, 8), (0, 1024, 8)])
def test_loss_and_grad_vs_autograd(kid, n, d):
    """d loss / d raw from the HIP closed form vs torch autograd through Cholesky (the reference's route)."""
    from super_resolution_amd import ops
    rng = np.random.default_rng(100 + kid + n)
    B = 2
    if d == 8:
        img = natural_image(40, 40, n + kid)
        X = np.stack([rng.permutation(opatch.features(img[1:39, 1:39]))[:n] for _ in range(B)]) if n <= 1296 else None
        y = X[:, :, 0] * 0.5 + X[:, :, 3] * 0.5 + rng.normal(0, 0.01, (B, n))
    else:
        X = rng.random((B, n, d)); y = X.mean(-1) + rng.normal(0, 0.02, (B, n))
    raw = np.array([[-5.0, float(y[b].mean()), -1.0 + 0.3 * b, 0.2 - 0.4 * b] for b in range(B)], dtype=np.float32)
    loss, grad, flags = ops.loss_grad(to_dev(X), to_dev(y), kid, to_dev(raw))
    assert int(flags.sum()) == 0
    for b in range(B):
        h = ogp.Hypers(*raw[b])
        l = ogp.neg_mll(h, torch.tensor(X[b]), torch.tensor(y[b]), kid)
        l.backward()
        g = np.array([p.grad.item() for p in h.params()])
        assert abs(loss[b].item() - l.item()) < 1e-9 * max(1, abs(l.item()))
        got = grad[b].cpu().numpy()
        assert np.allclose(got, g, rtol=2e-5, atol=1e-7), (got, g)


@pytest.mark.parametrize("kid,n", [(0, 324), (3, 324), (2, 324), (1, 324), (3, 784)])
def test_fit_and_predict_vs_oracle(kid, n):
    """50 Adam steps + predictive mean on real patch geometry vs the oracle's
    torch/autograd loop.  Bar: predicted pixels within 1e-4 (<< 1/255); final raw
    hyper-parameters within 2e-3 (float32 Adam trajectories are not bit-comparable)."""
    from super_resolution_amd import ops
    s, p = (2, 20) if n == 324 else (3, 30)
    H, W = p + 6, p + 9
    lo = natural_image(H, W, 7 + kid)
    from oracle import cvops
    hi = cvops.resize_cubic(lo, fx=s, fy=s)
    locs = opatch.patch_locations((H, W), p)[:3]
    tidx = np.arange(1, p - 1, dtype=np.int32)
    X, y, Xs = ops.gather(to_dev(lo), to_dev(lo), to_dev(hi), s, p, to_dev(np.array(locs, dtype=np.int32)), to_dev(tidx))
    raw = ops.init_hypers(y, kid)
    raw0 = raw.cpu().numpy().copy()
    alpha, flags = ops.fit(X, y, kid, raw, iters=50, lr=0.1)
    pred = ops.predict(X, Xs, alpha, raw, kid).cpu().numpy()
    assert int(flags.sum()) == 0
    for k, loc in enumerate(locs):
        Xo = opatch.features(opatch.extract(lo, loc, p, 1))
        yo = opatch.targets(opatch.extract(lo, loc, p, 1))
        Xso = opatch.features(opatch.extract(hi, loc, p, s))
        h0 = ogp.init_hypers(torch.tensor(yo), kid)
        assert np.allclose(raw0[k], h0.as_array(), rtol=1e-5, atol=1e-6), (raw0[k], h0.as_array())
        mean, rawo = ogp.fit_predict_patch(Xo, yo, Xso, kid)
        assert np.allclose(raw[k].cpu().numpy(), rawo, rtol=2e-3, atol=2e-3), (raw[k].cpu().numpy(), rawo)
        assert np.abs(pred[k] - mean).max() < 1e-4


def test_fit_constant_patch_rbf():
    """A flat patch: var(y)=0 -> RBF outputscale init 0 (raw -inf); prediction must be the constant."""
    from super_resolution_amd import ops
    n, m = 324, 1444
    X = torch.full((1, n, 8), 0.5, dtype=torch.float64, device=dev())
    y = torch.full((1, n), 0.5, dtype=torch.float64, device=dev())
    Xs = torch.full((1, m, 8), 0.5, dtype=torch.float64, device=dev())
    for kid in (0, 3):
        raw = ops.init_hypers(y, kid)
        alpha, flags = ops.fit(X, y, kid, raw, iters=50)
        pred = ops.predict(X, Xs, alpha, raw, kid)
        assert torch.isfinite(pred).all()
        assert (pred - 0.5).abs().max().item() < 1e-6


@pytest.mark.parametrize("n,small", [(784, 1), (784, 4), (324, 3), (1120, 2), (200, 4), (784, 10), (784, 14), (324, 30)])
def test_small_batch_strip_mode_bit_identical(n, small):
    """Small batches run the tile products one wave per workgroup, seven workgroups per tile (strip mode), and leave
    tile (k,k) = -M^T M of a pivot step to the update launch: the arithmetic and its order are those of the large-batch
    route, so a GP fitted alone gives the same bits as the same GP inside a large batch.  Also the plain inversion."""
    from super_resolution_amd import ops
    # large enough that the big batch takes the other route everywhere: workgroup-form tile products, and tile (k,k) of
    # each pivot step formed by k_pivot itself instead of by the update launch (update_forms_ptile in gpsr_api.hip)
    B = {784: 24, 324: 96, 1120: 12, 200: 180}[n]   # (small = 10, 14, 30: panel / restricted update still in strip form, three workgroups per CU)
    rng = np.random.default_rng(n + small)
    X = to_dev(rng.random((B, n, 8)))
    y = to_dev(rng.random((B, n)))
    raw0 = ops.init_hypers(y, ops.MATERN52)
    raw_b = raw0.clone()
    alpha_b, flags_b, loss_b = ops.fit(X, y, ops.MATERN52, raw_b, iters=4, want_loss=True)
    raw_s = raw0[:small].clone()
    alpha_s, flags_s, loss_s = ops.fit(X[:small].contiguous(), y[:small].contiguous(), ops.MATERN52, raw_s, iters=4, want_loss=True)
    assert int(flags_b.sum()) == 0 and int(flags_s.sum()) == 0
    assert torch.equal(alpha_s, alpha_b[:small]) and torch.equal(raw_s, raw_b[:small]) and torch.equal(loss_s, loss_b[:small])
    N = -(-n // 112) * 112
    A = np.zeros((B, N, N))
    for b in range(B):
        Q = rng.normal(size=(N, N + 8))
        A[b] = Q @ Q.T / N + 0.05 * np.eye(N)
    Ab, As = to_dev(A), to_dev(A[:small])
    ld_b, _ = ops.invert_(Ab)
    ld_s, _ = ops.invert_(As)
    assert torch.equal(ld_s, ld_b[:small])
    for b in range(small):   # the lower block triangle is the result; the rest is scratch
        assert np.array_equal(_lower_blocks_to_full(As[b].cpu().numpy()), _lower_blocks_to_full(Ab[b].cpu().numpy()))


def test_small_batch_under_concurrent_load_bit_identical():
    """Four small-batch (strip-form) fits on four streams at once, each bit-identical to the same fit run alone.  The
    strip form has no barrier: what keeps a DMA from overwriting an LDS stage that is still being read is the s_waitcnt
    on the wave's own LDS reads before the stage is reused.  (The failure that wait fixed needed two PROCESSES sharing
    the GPU -- tests/test_gpu_dist.py::test_two_ranks_on_one_gpu_chained_mode_equals_one_rank reproduces it; this
    in-process version passes either way and is kept as the cheaper multi-stream check.)"""
    from super_resolution_amd import ops
    rng = np.random.default_rng(11)
    n, S = 784, 4
    Xs_ = [to_dev(rng.random((2 + (k % 2), n, 8))) for k in range(S)]
    ys_ = [to_dev(rng.random((x.shape[0], n))) for x in Xs_]
    raw0 = [ops.init_hypers(y, ops.MATERN52) for y in ys_]
    solo = []
    for k in range(S):
        raw = raw0[k].clone()
        alpha, flags, loss = ops.fit(Xs_[k], ys_[k], ops.MATERN52, raw, iters=6, want_loss=True)
        torch.cuda.synchronize()
        assert int(flags.sum()) == 0
        solo.append((alpha, raw, loss))
    streams = [torch.cuda.Stream() for _ in range(S)]
    for rep in range(3):
        got = []
        torch.cuda.synchronize()
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                raw = raw0[k].clone()
                alpha, flags, loss = ops.fit(Xs_[k], ys_[k], ops.MATERN52, raw, iters=6, want_loss=True)
                got.append((alpha, raw, loss, flags))
        torch.cuda.synchronize()
        for k in range(S):
            assert int(got[k][3].sum()) == 0
            for u, v in zip(got[k][:3], solo[k]):
                assert torch.equal(u, v)


def test_fit_dual_stream_bit_identical():
    """gpsr_set_dual_stream: the batch as two half-batches on two streams gives the same bits
    (per-GP arithmetic does not depend on the batch it runs in)."""
    from super_resolution_amd import ops
    B, n = 300, 324
    rng = np.random.default_rng(5)
    X = to_dev(rng.random((B, n, 8)))
    y = to_dev(rng.random((B, n)))
    raw0 = ops.init_hypers(y, ops.MATERN52)
    outs = []
    for dual in (False, True):
        prev = ops.set_dual_stream(dual)
        try:
            raw = raw0.clone()
            alpha, flags, loss = ops.fit(X, y, ops.MATERN52, raw, iters=3, want_loss=True)
            torch.cuda.synchronize()
            outs.append((alpha.clone(), raw.clone(), flags.clone(), loss.clone()))
        finally:
            ops.set_dual_stream(prev)
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    assert int(outs[0][2].sum()) == 0


@pytest.mark.parametrize("B,n", [(300, 324), (257, 784), (400, 196)])
def test_fit_split_small_bit_identical(B, n):
    """gpsr_set_split_small (the default schedule of a batch >= 256): the runs of small launches as two half-batches
    on two streams, the big launches on the whole batch -- the same bits as every launch on one stream, odd batch
    sizes and 2, 3 and 7 block matrices included."""
    from super_resolution_amd import ops
    rng = np.random.default_rng(B)
    X = to_dev(rng.random((B, n, 8)))
    y = to_dev(rng.random((B, n)))
    raw0 = ops.init_hypers(y, ops.MATERN32)
    outs = []
    for split in (False, True):
        prev = ops.set_split_small(split)
        try:
            raw = raw0.clone()
            alpha, flags, loss = ops.fit(X, y, ops.MATERN32, raw, iters=2, want_loss=True)
            torch.cuda.synchronize()
            outs.append((alpha.clone(), raw.clone(), flags.clone(), loss.clone()))
        finally:
            ops.set_split_small(prev)
    for u, v in zip(outs[0], outs[1]):
        assert torch.equal(u, v)
    assert int(outs[0][2].sum()) == 0


@pytest.mark.parametrize("B,n,kid,iters", [(37, 324, 2, 3), (9, 196, 0, 4), (5, 784, 3, 2), (1100, 324, 3, 2),
                                           (3, 196, 1, 131), (6, 500, 4, 2)])
def test_fit_fused_bit_identical(B, n, kid, iters):
    """gpsr_set_fused (opt-in): ONE launch per fit, one workgroup per GP running the role functions of the
    launch-per-step route back to back (k_fit_fused) -- the same bits: 2-, 3-, 5- and 7-block matrices, all five
    kernels, more GPs than resident workgroup slots (1100 > 512: a slot's workspace is reused), more iterations than
    one launch's table of Adam constants holds (131 > 128: the moments continue across launches)."""
    from super_resolution_amd import ops
    rng = np.random.default_rng(B + n)
    X = to_dev(rng.random((B, n, 8)))
    y = to_dev(rng.random((B, n)))
    raw0 = ops.init_hypers(y, kid)
    outs = []
    for fused in (False, True):
        prev = ops.set_fused(fused)
        try:
            raw = raw0.clone()
            alpha, flags, loss = ops.fit(X, y, kid, raw, iters=iters, want_loss=True)
            torch.cuda.synchronize()
            outs.append((alpha.clone(), raw.clone(), flags.clone(), loss.clone()))
        finally:
            ops.set_fused(prev)
    for name, u, v in zip(("alpha", "raw", "flags", "loss"), outs[0], outs[1]):
        assert torch.equal(u, v), (name, int((u != v).sum()))
    assert int(outs[0][2].sum()) == 0


def test_context_workspace_and_flag_count():
    """gpsr_create / gpsr_destroy: the library-owned workspace gives the same fit as a caller-owned one;
    gpsr_flag_count is the synchronising read-out of the per-GP failure flags."""
    from super_resolution_amd import _lib, ops
    lib = _lib.load()
    B, n = 5, 100
    rng = np.random.default_rng(9)
    X = to_dev(rng.random((B, n, 8)))
    y = to_dev(rng.random((B, n)))
    raw_a = ops.init_hypers(y, ops.RBF)
    raw_b = raw_a.clone()
    alpha_a, flags_a = ops.fit(X, y, ops.RBF, raw_a, iters=4)
    ctx = lib.gpsr_create(torch.cuda.current_device(), B, n, 0)
    assert ctx, lib.gpsr_last_error()
    try:
        assert lib.gpsr_ctx_workspace_bytes(ctx) >= lib.gpsr_fit_workspace_bytes(B, n, 8)
        N = ops.padded_n(n)
        alpha_b = torch.empty((B, N), dtype=torch.float64, device=X.device)
        flags_b = torch.empty(B, dtype=torch.int32, device=X.device)
        st = torch.cuda.current_stream().cuda_stream
        rc = lib.gpsr_fit(X.data_ptr(), y.data_ptr(), B, n, 8, ops.RBF, raw_b.data_ptr(), 4, 0.1, alpha_b.data_ptr(),
                          None, flags_b.data_ptr(), lib.gpsr_ctx_workspace(ctx), lib.gpsr_ctx_workspace_bytes(ctx), st)
        assert rc == 0
        assert lib.gpsr_flag_count(flags_b.data_ptr(), B, st) == 0
        assert torch.equal(alpha_a, alpha_b) and torch.equal(raw_a, raw_b)
        flags_b[1] = 7
        flags_b[3] = 1
        assert lib.gpsr_flag_count(flags_b.data_ptr(), B, st) == 2
    finally:
        lib.gpsr_destroy(ctx)
    assert lib.gpsr_create(torch.cuda.current_device(), 0, n, 0) is None   # bad size: NULL + error text
    assert b"gpsr_create" in lib.gpsr_last_error()


def test_context_workspace_runs_predict_pixels():
    """A context created with max_m > 0 holds enough for gpsr_predict_pixels (any chunk); one created with
    max_m = 0 is refused with "workspace too small" instead of being overrun."""
    from super_resolution_amd import _lib, engine, ops
    lib = _lib.load()
    s, p, H, W = 2, 20, 44, 40
    lr = natural_image(H, W, 31)
    from oracle import cvops
    tin, tst = to_dev(lr), to_dev(cvops.resize_cubic(lr, fx=s, fy=s))
    rows, cols, _ = engine._geometry(H, W, s, p, 0.9)
    tidx = engine.training_indices(p)
    rows_d, cols_d, tidx_d = to_dev(np.asarray(rows, np.int32)), to_dev(np.asarray(cols, np.int32)), to_dev(tidx)
    want, _, _ = ops.predict_pixels_fused(tin, tin, tst, s, p, rows_d, cols_d, tidx_d, ops.MATERN32, iters=5)
    B, n, m = len(rows) * len(cols), (p - 2) ** 2, (p * s - 2) ** 2
    st = torch.cuda.current_stream().cuda_stream
    for max_m, chunk in ((m, 0), (m, 4), (0, 0)):
        ctx = lib.gpsr_create(torch.cuda.current_device(), B, n, max_m)
        assert ctx, lib.gpsr_last_error()
        try:
            out = torch.empty((H * s, W * s), dtype=torch.float64, device=tin.device)
            flags = torch.empty(B, dtype=torch.int32, device=tin.device)
            rc = lib.gpsr_predict_pixels(tin.data_ptr(), tin.data_ptr(), tst.data_ptr(), H, W, s, p, rows_d.data_ptr(),
                                         len(rows), cols_d.data_ptr(), len(cols), tidx_d.data_ptr(), len(tidx),
                                         ops.MATERN32, 0.0, 0.0, 5, 0.1, ops.AVERAGE, chunk, out.data_ptr(), None,
                                         flags.data_ptr(), lib.gpsr_ctx_workspace(ctx), lib.gpsr_ctx_workspace_bytes(ctx), st)
            if max_m == 0:
                assert rc < 0 and b"workspace too small" in lib.gpsr_last_error()
            else:
                assert rc == 0, lib.gpsr_last_error()
                assert lib.gpsr_flag_count(flags.data_ptr(), B, st) == 0
                assert torch.equal(out, want)
        finally:
            lib.gpsr_destroy(ctx)


@pytest.mark.parametrize("kid", [0, 1, 2, 3, 4])
def test_covariance_grad_matches_oracle_and_finite_difference(kid):
    """gpsr_covariance_grad: dK/dl against the oracle's closed form and against a central difference
    of gpsr_covariance itself in the (exactly representable, raw > 20) lengthscale."""
    from super_resolution_amd import ops
    rng = np.random.default_rng(40 + kid)
    B, n = 2, 130
    X = rng.random((B, n, 8)) * 20.0
    raw = (21.0 + 4 * rng.random((B, 4))).astype(np.float32)
    K, dK = ops.covariance_grad(to_dev(X), kid, to_dev(raw))
    K0 = ops.covariance(to_dev(X), kid, to_dev(raw))
    assert torch.equal(K, K0)
    dK = dK.cpu().numpy()
    for b in range(B):
        os_, ls = float(raw[b, 2]), float(raw[b, 3])
        Xb = X[b]
        if kid == ogp.LINEAR:
            dk = Xb @ Xb.T                      # k = variance * <x, x'>: d k / d variance
        else:
            d2 = ((Xb[:, None, :] - Xb[None, :, :]) ** 2).sum(-1)
            _, dk = ogp.kernel_and_dl(d2, ls, kid)
        want = os_ * dk
        got = dK[b][:n, :n]
        lower = np.tril(np.ones((n, n), bool)) | (np.add.outer(np.arange(n) // 112, -(np.arange(n) // 112)) == 0)
        assert np.abs(got - want)[lower].max() <= 1e-12 * max(1.0, np.abs(want).max())
    # central difference in raw lengthscale (softplus(raw) == raw here, steps exact in float32)
    h = 0.25
    rp, rm = raw.copy(), raw.copy()
    rp[:, 3] += h
    rm[:, 3] -= h
    Kp = ops.covariance(to_dev(X), kid, to_dev(rp)).cpu().numpy()
    Km = ops.covariance(to_dev(X), kid, to_dev(rm)).cpu().numpy()
    fd = (Kp - Km) / (2 * h)
    blk = np.add.outer(np.arange(K.shape[1]) // 112, -(np.arange(K.shape[1]) // 112)) >= 0
    scale = max(1e-30, np.abs(fd[:, blk]).max())
    assert np.abs(fd - dK)[:, blk].max() <= 2e-3 * scale


# ----------------------------------------------------------------- uint8 colour stages + metrics (HIP, integer: bit-exact)
def test_colour_kernels_bit_exact_vs_oracle():
    """gpsr_bgr2gray / bgr2yuv / yuv2bgr / bgr2ycrcb / ycrcb2bgr / resample_u8 against oracle/cvops.py (OpenCV's
    fixed-point paths restated; the grey table is pinned by the reference's own *_gray.png outputs)."""
    from oracle import cvops
    from super_resolution_amd import imgproc, ops
    rng = np.random.default_rng(2)
    bgr = rng.integers(0, 256, (97, 131, 3)).astype(np.uint8)
    bgr[0, 0], bgr[0, 1], bgr[0, 2], bgr[0, 3], bgr[0, 4] = (0, 0, 0), (255, 255, 255), (255, 0, 0), (0, 255, 0), (0, 0, 255)
    d = to_dev(bgr)
    assert np.array_equal(ops.bgr2gray(d).cpu().numpy(), cvops.bgr2gray(bgr))
    assert np.array_equal(imgproc.bgr2gray(bgr), cvops.bgr2gray(bgr))
    y, u, v = ops.bgr2yuv(d)
    want = cvops.bgr2yuv(bgr)
    assert np.array_equal(torch.stack([y, u, v], -1).cpu().numpy(), want)
    assert np.array_equal(ops.yuv2bgr(y, u, v).cpu().numpy(), cvops.yuv2bgr(want))
    yy, cr, cb = ops.bgr2ycrcb(d)
    want = cvops.bgr2ycrcb(bgr)
    assert np.array_equal(torch.stack([yy, cr, cb], -1).cpu().numpy(), want)
    assert np.array_equal(ops.ycrcb2bgr(yy, cr, cb).cpu().numpy(), cvops.ycrcb2bgr(want))
    for kw in ({"fx": 3, "fy": 3}, {"fx": 2, "fy": 2}, {"dsize": (77, 50)}):
        got = imgproc.resize_cubic_u8_dev(u, **kw).cpu().numpy()
        assert np.array_equal(got, cvops.resize_cubic(u.cpu().numpy(), **kw)), kw


@pytest.mark.parametrize("H,W,crop", [(64, 80, 3), (33, 29, 2), (240, 250, 4), (9, 9, 1)])
def test_metrics_kernel_vs_oracle(H, W, crop):
    """gpsr_metrics (PSNR, SSIM as evaluation.py:83-98 / skimage) against the oracle's numpy restatement: 1e-10."""
    from oracle import cvops
    from super_resolution_amd import ops
    rng = np.random.default_rng(H * W)
    a = np.rint(natural_image(H, W, H) * 255).astype(np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-6, 7, a.shape), 0, 255).astype(np.uint8)
    got = ops.metrics(to_dev(a), to_dev(b), crop).cpu().numpy()
    ac, bc = a[crop:-crop, crop:-crop], b[crop:-crop, crop:-crop]
    assert abs(got[0] - cvops.psnr(ac, bc)) < 1e-10
    if min(ac.shape) >= 7:
        assert abs(got[1] - cvops.ssim(ac, bc)) < 1e-10
    assert abs(got[2] - np.mean((ac.astype(float) - bc) ** 2)) < 1e-12
    same = ops.metrics(to_dev(a), to_dev(a), crop).cpu().numpy()
    assert np.isinf(same[0]) and same[0] > 0 and (min(ac.shape) < 7 or abs(same[1] - 1.0) < 1e-12)
    # deterministic: a second launch gives the same bits
    assert np.array_equal(got, ops.metrics(to_dev(a), to_dev(b), crop).cpu().numpy())


def test_metrics_match_real_scikit_image(golden_dir):
    """gpsr_metrics against scikit-image 0.18.3 ITSELF (tests/golden/skimage_metrics.json: skimage.metrics on committed
    fixture pairs, tests/golden/make_skimage_metric_goldens.py): PSNR and SSIM of evaluation.py:83-98 to 1e-10."""
    import json
    from PIL import Image
    from super_resolution_amd import ops
    g = json.load(open(os.path.join(golden_dir, "skimage_metrics.json")))
    for p in g["pairs"]:
        a = np.array(Image.open(os.path.join(golden_dir, p["hr"])))
        b = np.array(Image.open(os.path.join(golden_dir, p["upscaled"])))
        got = ops.metrics(to_dev(a), to_dev(b), p["crop"]).cpu().numpy()
        assert abs(got[0] - p["psnr"]) < 1e-10 and abs(got[1] - p["ssim"]) < 1e-10 and abs(got[2] - p["mse"]) < 1e-9, (p, got)


@pytest.mark.parametrize("kid,n", [(3, 324), (0, 784)])
def test_near_singular_patch_needs_no_jitter(kid, n):
    """Failure semantics (SURVEY.md 5.9 "Solver"): GPyTorch's psd_safe_cholesky adds jitter 1e-8 / 1e-7 / 1e-6 only
    when the plain Cholesky FAILS.  The worst case the path can produce -- a patch whose feature rows repeat (a flat
    image region: K without noise is singular), the noise at its 1e-4 floor and a huge outputscale, cond(K) ~ 1e12 --
    factorises in float64 without any jitter, on the CPU (torch.linalg.cholesky does not throw) and here (no flag),
    and the two agree to what that conditioning allows.  So the jitter ladder is never entered on finite input; a
    non-finite patch is flagged and raised (test_non_finite_patch_raises_like_the_reference), GPyTorch's NanError."""
    from super_resolution_amd import ops
    rng = np.random.default_rng(n + kid)
    base = rng.random((n // 6, 8))
    X = np.concatenate([base] * 6 + [rng.random((n - 6 * (n // 6), 8))])[:n]      # every row 6 times over
    X[: n // 3] = X[0]                                                           # and a third of them identical
    y = rng.random(n)
    raw = np.array([[-30.0, 0.3, 2e5, 0.5]], dtype=np.float32)   # noise -> its 1e-4 floor, outputscale 2e5
    Xd, yd, rd = to_dev(X[None]), to_dev(y[None]), to_dev(raw)
    loss, grad, flags = ops.loss_grad(Xd, yd, kid, rd)
    assert int(flags.sum()) == 0 and torch.isfinite(loss).all() and torch.isfinite(grad).all()
    h = ogp.Hypers(*raw[0])
    noise, c, os_, ls = [v.detach().double() for v in h.constrained()]
    Xt, yt = torch.tensor(X), torch.tensor(y)
    K = os_ * ogp.base_kernel(Xt, Xt, ls, kid) + noise * torch.eye(n, dtype=torch.float64)
    cond = float(torch.linalg.cond(K))
    assert cond > 1e11, cond
    L = torch.linalg.cholesky(K)                     # the reference's first (jitter-free) attempt succeeds
    want = float(ogp.neg_mll(h, Xt, yt, kid))
    assert abs(float(loss[0]) - want) < 1e-3 * abs(want), (float(loss[0]), want)   # y^T K^-1 y carries cond x eps ~ 1e-4
    # alpha and the predictive mean at these hyper-parameters (iters = 0: factorisation only)
    alpha, fl = ops.fit(Xd, yd, kid, rd.clone(), iters=0)
    assert int(fl.sum()) == 0
    a_want = torch.cholesky_solve((yt - c)[:, None], L)[:, 0].numpy()
    a_got = alpha[0, :n].cpu().numpy()
    Xs = rng.random((50, 8))
    m_want = (os_ * ogp.base_kernel(torch.tensor(Xs), Xt, ls, kid)).numpy() @ a_want + float(c)
    m_got = ops.predict(Xd, to_dev(Xs[None]), alpha, rd, kid)[0].cpu().numpy()
    # two CPU routes (Cholesky solve vs explicit inverse) differ by 8e-6 / 4e-5 here: 1e-3 is a quarter grey level
    assert np.abs(m_got - m_want).max() < 1e-3, np.abs(m_got - m_want).max()


def test_library_loaded_before_torch_still_shares_one_hip_runtime():
    """`python __graft_entry__.py smoke` loads libgpsr.so (build()) before anything imports torch.  Two HIP runtimes in
    one process (/opt/rocm's, pulled in by the library, and torch's own) leave the second without a device, so
    _lib.load() imports torch first; a fresh process in that order must be able to launch."""
    import subprocess
    import sys
    code = ("from super_resolution_amd import _lib; _lib.load(); import torch; from super_resolution_amd import ops; "
            "x = torch.arange(64, dtype=torch.uint8, device='cuda'); y = ops.u8_to_unit(x); torch.cuda.synchronize(); "
            "assert abs(float(y[51]) - 51 / 255) < 1e-15; print('one-runtime-ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "one-runtime-ok" in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]
