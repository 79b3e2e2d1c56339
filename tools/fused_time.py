"""Time gpsr_fit on B GPs of n points, fused per-GP route against the launch-per-step route:
   python tools/fused_time.py B n kernel_id iters [modes...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from super_resolution_amd import ops  # noqa: E402

B, n, kid, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
modes = [int(v) for v in sys.argv[5:]] or [0, 1]
rng = np.random.default_rng(1)
X = torch.as_tensor(rng.random((B, n, 8))).cuda()
y = torch.as_tensor(rng.random((B, n))).cuda()
raw0 = ops.init_hypers(y, kid)
res = {}
for mode in modes:
    ops.set_fused(mode)
    ws = ops.fit_workspace(B, n, 8, X.device)
    for rep in range(2):
        raw = raw0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        alpha, flags = ops.fit(X, y, kid, raw, iters=iters, workspace=ws)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res[mode] = (alpha, raw, flags)
    print(f"mode {mode}: B={B} n={n} iters={iters}: {dt * 1e3:.1f} ms, {dt / B / (iters + 1) * 1e6:.3f} us per GP-evaluation, "
          f"workspace {ws.numel() / 2**30:.2f} GiB, flags {int(flags.sum())}", flush=True)
    del ws
if len(res) == 2:
    a, b = res.values()
    print("bit-identical:", all(torch.equal(u, v) for u, v in zip(a, b)))
