"""Experiment: does the n = 324 path run faster when the batch is cut into chunks whose working
set stays inside the 256 MB Infinity Cache, with several chunks in flight on independent streams (one host thread and one
workspace per stream, no cross-stream events)?  A GP of n = 324 holds 1.9 MB of matrix + panels + pivot factors, so
4 streams x 32 GPs = 243 MB is the cache-sized point; larger chunks spill to HBM, smaller ones leave the chip emptier.
Baseline: ONE gpsr_fit call over the whole batch (what the engine does).  Prints microseconds per GP-evaluation.

usage: python tools/mall_chunks.py [total_gps] [iters]
"""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from super_resolution_amd import ops  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 8448
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, d, kid = 324, 8, ops.MATERN32
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
# patch-like features: smooth 8-neighbour values + noise, targets = their mean + noise
base = rng.random((T, n, 1))
X = torch.as_tensor(np.clip(base + 0.05 * rng.standard_normal((T, n, d)), 0, 1)).to(dev)
y = torch.as_tensor(np.clip(base[..., 0] + 0.02 * rng.standard_normal((T, n)), 0, 1)).to(dev)
raw0 = ops.init_hypers(y, kid)
evals = ITERS + 1


def whole():
    raw = raw0.clone()
    ws = ops.fit_workspace(T, n, d, dev)
    ops.fit(X, y, kid, raw, iters=ITERS, workspace=ws)          # warm
    torch.cuda.synchronize()
    raw = raw0.clone()
    t0 = time.perf_counter()
    alpha, flags = ops.fit(X, y, kid, raw, iters=ITERS, workspace=ws)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, raw, alpha


def chunked(S, c):
    raw = raw0.clone()
    alpha = torch.empty((T, ops.padded_n(n)), dtype=torch.float64, device=dev)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    wss = [ops.fit_workspace(c, n, d, dev) for _ in range(S)]
    chunks = [(lo, min(lo + c, T)) for lo in range(0, T, c)]
    start = threading.Barrier(S + 1)
    errs = []

    def worker(k):
        try:
            with torch.cuda.stream(streams[k]):
                start.wait()
                for lo, hi in chunks[k::S]:
                    a, _ = ops.fit(X[lo:hi], y[lo:hi], kid, raw[lo:hi], iters=ITERS, workspace=wss[k])
                    alpha[lo:hi].copy_(a)
        except Exception as e:   # noqa: BLE001 - report and let the main thread fail
            errs.append(repr(e))

    torch.cuda.synchronize()
    th = [threading.Thread(target=worker, args=(k,)) for k in range(S)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if errs:
        raise SystemExit("worker failed: " + errs[0])
    return dt, raw, alpha


dt0, raw_w, alpha_w = whole()
print(f"whole batch, one call: B = {T}: {dt0 * 1e3:8.2f} ms  = {dt0 * 1e6 / (T * evals):.3f} us per GP-evaluation", flush=True)
for S, c in ((4, 32), (4, 64), (4, 128), (4, 256), (4, 528), (2, 1056), (8, 32), (8, 64)):
    chunked(S, c)                                                 # warm (streams, workspaces, code)
    dt, raw_c, alpha_c = chunked(S, c)
    same = bool(torch.equal(raw_c, raw_w) and torch.equal(alpha_c, alpha_w))
    print(f"{S} streams x chunks of {c:4d} GPs ({S * c * 1.9:6.0f} MB in flight): {dt * 1e3:8.2f} ms  = "
          f"{dt * 1e6 / (T * evals):.3f} us per GP-evaluation   bit-identical to the whole batch: {same}", flush=True)
