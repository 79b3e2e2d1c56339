"""Micro-workload for rocprofv3 --pmc passes: one batched inversion (7 pivot steps of
pivot / panel / update) of B SPD matrices of order N, as in one Adam iteration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from super_resolution_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 633
N = int(sys.argv[2]) if len(sys.argv) > 2 else 784
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mode = sys.argv[4] if len(sys.argv) > 4 else "invert"
dev = torch.device("cuda:0")
if mode == "eval":
    # one whole evaluation of the fit (covariance, sweeps, alpha, gradient statistics, loss/grad) of B GPs with n = N
    # training points (N need not be a multiple of 112 here), Matern-3/2, patch-like features
    import numpy as np
    rng = np.random.default_rng(0)
    X = torch.as_tensor(rng.random((B, N, 8)) * 0.3 + 0.3).to(dev)
    y = torch.as_tensor(rng.random((B, N))).to(dev)
    raw = ops.init_hypers(y, ops.MATERN32)
    ws = ops.fit_workspace(B, N, 8, dev)
    for _ in range(reps):
        loss, grad, flags = ops.loss_grad(X, y, ops.MATERN32, raw, ws)
    torch.cuda.synchronize()
    print("flags", int(flags.sum()), "loss0", float(loss[0]))
    sys.exit(0)
if mode == "fit1":
    # one Adam iteration + the final evaluation (two evaluations per GP) through gpsr_fit: the route GPSR_FUSED selects
    import numpy as np
    rng = np.random.default_rng(0)
    X = torch.as_tensor(rng.random((B, N, 8)) * 0.3 + 0.3).to(dev)
    y = torch.as_tensor(rng.random((B, N))).to(dev)
    ws = ops.fit_workspace(B, N, 8, dev)
    for _ in range(reps):
        raw = ops.init_hypers(y, ops.MATERN32)
        alpha, flags = ops.fit(X, y, ops.MATERN32, raw, iters=1, workspace=ws)
    torch.cuda.synchronize()
    print("flags", int(flags.sum()), "alpha0", float(alpha[0, 0]))
    sys.exit(0)
g = torch.Generator(device=dev); g.manual_seed(0)
Q = torch.randn((B, N, 64), dtype=torch.float64, device=dev, generator=g)
A0 = Q @ Q.transpose(1, 2) / 64 + 0.5 * torch.eye(N, dtype=torch.float64, device=dev)
for _ in range(reps):
    A = A0.clone()
    logdet, flags = ops.invert_(A)
torch.cuda.synchronize()
print("flags", int(flags.sum()), "logdet0", float(logdet[0]))
