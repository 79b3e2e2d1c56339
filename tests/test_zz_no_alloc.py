"""gpsr_fit_noalloc (include/gpsr.h): the entry point that allocates nothing -- its scratch is a static
__device__ array -- must give the bits gpsr_fit gives with a caller workspace.  Two calls in a row on the same
GPU (the second smaller: the array still holds the first call's matrices and Adam moments), then the size
limit.  Collected last (test_zz_*): it shares nothing with the other tests but must pass after them too."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fit_both(lib, ops, X, y, kid, iters):
    B, n, d = X.shape
    N = ops.padded_n(n)
    raw_a = ops.init_hypers(y, kid)
    raw_b = raw_a.clone()
    alpha_a, flags_a, loss_a = ops.fit(X, y, kid, raw_a, iters=iters, want_loss=True)
    alpha_b = torch.full((B, N), float("nan"), dtype=torch.float64, device=X.device)
    loss_b = torch.full((B,), float("nan"), dtype=torch.float64, device=X.device)
    flags_b = torch.full((B,), -1, dtype=torch.int32, device=X.device)
    st = torch.cuda.current_stream().cuda_stream
    raw_w = raw_b.clone()
    call = lambda r: lib.gpsr_fit_noalloc(X.data_ptr(), y.data_ptr(), B, n, d, kid, r.data_ptr(), iters, 0.1,
                                          alpha_b.data_ptr(), loss_b.data_ptr(), flags_b.data_ptr(), st)
    # "allocates nothing" at the DRIVER level: torch.cuda.memory_allocated() only sees torch's caching allocator and
    # would miss a hipMalloc inside libgpsr.so, so compare hipMemGetInfo's free bytes around a call -- the second one
    # of this shape (the first may make the runtime size its per-kernel scratch pool, which no caller can avoid)
    assert call(raw_w) == 0, lib.gpsr_last_error()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    before = torch.cuda.memory_allocated()
    rc = call(raw_b)
    assert rc == 0, lib.gpsr_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before, "gpsr_fit_noalloc changed the device's free memory"
    assert torch.equal(alpha_a, alpha_b), (alpha_a - alpha_b).abs().max()
    assert torch.equal(raw_a, raw_b) and torch.equal(loss_a, loss_b)
    assert torch.equal(flags_a, flags_b) and int(flags_b.sum()) == 0


def test_fit_noalloc_equals_fit_two_calls_and_limit():
    from super_resolution_amd import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    rng = np.random.default_rng(5)
    to = lambda a: torch.as_tensor(a).to(dev)
    assert lib.gpsr_noalloc_scratch_bytes() == 64 << 20
    # first call: 6 GPs of n = 324 (three pivot blocks: pair step + single step), Matern-5/2
    X1, y1 = to(rng.random((6, 324, 8))), to(rng.random((6, 324)))
    _fit_both(lib, ops, X1, y1, ops.MATERN52, 6)
    # second call, smaller in every dimension (one block, linear kernel, d = 49): stale scratch must not leak in
    X2, y2 = to(rng.random((3, 100, 49))), to(rng.random((3, 100)))
    _fit_both(lib, ops, X2, y2, ops.LINEAR, 4)
    # and the first shape again, after the small one
    _fit_both(lib, ops, X1, y1, ops.MATERN52, 6)
    # beyond the static array: rejected with a clear error, nothing launched
    Xb, yb = to(rng.random((40, 784, 8))), to(rng.random((40, 784)))
    raw = ops.init_hypers(yb, ops.RBF)
    alpha = torch.empty((40, 784), dtype=torch.float64, device=dev)
    flags = torch.zeros(40, dtype=torch.int32, device=dev)
    rc = lib.gpsr_fit_noalloc(Xb.data_ptr(), yb.data_ptr(), 40, 784, 8, ops.RBF, raw.data_ptr(), 2, 0.1,
                              alpha.data_ptr(), None, flags.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc < 0 and b"static array" in lib.gpsr_last_error()
