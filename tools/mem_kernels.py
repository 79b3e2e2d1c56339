"""Achieved HBM GB/s of the memory-side kernels (K1 gather, K2 covariance, K5 predict,
K6 combine) at the Set14-x3 and 4K-x2 patch geometries, through the C ABI with preallocated buffers,
timed with events on the launch stream.  Algorithmic bytes per patch are SURVEY.md 8(d)'s formulas.

usage: python tools/mem_kernels.py [out.json]
"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from super_resolution_amd import _lib, engine, ops  # noqa: E402

HBM_PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    st = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        fn()
    e1.record(st)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def case(H, W, s, kid, seed, maxB=None):
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    p = 10 * s
    P = p * s
    rng = np.random.default_rng(seed)
    lr = torch.as_tensor(rng.random((H, W))).to(dev)
    hi = torch.as_tensor(rng.random((H * s, W * s))).to(dev)
    rows, cols, locs = engine._geometry(H, W, s, p, 0.9)
    if maxB is not None and locs.shape[0] > maxB:      # keep whole patch rows
        nr = max(1, maxB // len(cols))
        rows = rows[:nr]
        locs = locs[:nr * len(cols)]
        H = int(rows[-1]) + p
        lr, hi = lr[:H].contiguous(), hi[:H * s].contiguous()
    tidx = torch.as_tensor(engine.training_indices(p)).to(dev)
    B, nt = locs.shape[0], tidx.numel()
    n, m, d = nt * nt, (P - 2) ** 2, 8
    N = ops.padded_n(n)
    st = torch.cuda.current_stream().cuda_stream
    locs_d = torch.as_tensor(np.ascontiguousarray(locs)).to(dev)
    rows_d = torch.as_tensor(np.asarray(rows, np.int32)).to(dev)
    cols_d = torch.as_tensor(np.asarray(cols, np.int32)).to(dev)
    X = torch.empty((B, n, d), dtype=torch.float64, device=dev)
    y = torch.empty((B, n), dtype=torch.float64, device=dev)
    Xs = torch.empty((B, m, d), dtype=torch.float64, device=dev)
    K = torch.zeros((B, N, N), dtype=torch.float64, device=dev)
    pred = torch.empty((B, m), dtype=torch.float64, device=dev)
    out = torch.empty((H * s, W * s), dtype=torch.float64, device=dev)
    ck = _lib.check
    g = lambda: ck(lib.gpsr_gather(lr.data_ptr(), lr.data_ptr(), hi.data_ptr(), H, W, s, p, locs_d.data_ptr(), B,
                                   tidx.data_ptr(), nt, X.data_ptr(), y.data_ptr(), Xs.data_ptr(), st), "gather")
    g()
    raw = ops.init_hypers(y, kid)
    alpha = torch.as_tensor(rng.standard_normal((B, n))).to(dev)
    c = lambda: ck(lib.gpsr_covariance(X.data_ptr(), B, n, d, kid, raw.data_ptr(), K.data_ptr(), st), "covariance")
    pr = lambda: ck(lib.gpsr_predict(X.data_ptr(), Xs.data_ptr(), alpha.data_ptr(), raw.data_ptr(), B, n, m, kid,
                                     pred.data_ptr(), st), "predict")
    cb = lambda: ck(lib.gpsr_combine(pred.data_ptr(), hi.data_ptr(), H, W, s, p, rows_d.data_ptr(), len(rows),
                                     cols_d.data_ptr(), len(cols), ops.AVERAGE, 1, out.data_ptr(), st), "combine")
    res = {"geometry": {"H": H, "W": W, "s": s, "p": p, "B": B, "n": n, "m": m, "kernel_id": kid}}
    byts = {
        "k_gather": B * ((2 * p * p + P * P) * 8 + (9 * n + 8 * m) * 8),
        "k_build (gpsr_covariance, full K written)": B * (n * n * 8 + n * d * 8),
        "k_predict": B * (m * d * 8 + n * d * 8 + n * 8 + m * 8),
        "k_combine": B * m * 8 + 2 * H * s * W * s * 8,
    }
    for (name, nbytes), fn in zip(byts.items(), (g, c, pr, cb)):
        t = timed(fn)
        gbs = nbytes / t * 1e-9
        res[name] = {"ms": t * 1e3, "algorithmic_bytes": nbytes, "GB_per_s": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK}
        if name == "k_predict":   # m*n kernel evaluations per patch: the VALU, not HBM, bounds this one
            res[name]["kernel_evaluations_per_s"] = B * m * n / t
        print(f"{name:45s} B={B:6d} {t*1e3:9.3f} ms {gbs:9.1f} GB/s ({gbs/HBM_PEAK:.3f} of peak)", flush=True)
    return res


def main():
    out = {"command": "python tools/mem_kernels.py", "hbm_peak_GB_per_s": HBM_PEAK,
           "bytes": "ALGORITHMIC bytes, SURVEY.md 8(d); durations = mean of 20 launches, events on the launch stream",
           "set14x3_geometry (480x640 LR, x3, Matern-5/2)": case(480, 640, 3, ops.MATERN52, 0),
           "4kx2_geometry (first 8192 patches of 2160x3840 LR, x2, Matern-3/2)": case(2160, 3840, 2, ops.MATERN32, 1, 8192)}
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
