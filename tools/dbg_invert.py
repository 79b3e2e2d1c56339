"""debug: per-16x16-block error map of gpsr_invert at N (lower block-triangle)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from super_resolution_amd import ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 224
rng = np.random.default_rng(N)
Q = rng.normal(size=(N, N + 8)); A = Q @ Q.T / N + 0.05 * np.eye(N)
At = torch.as_tensor(A[None]).cuda()
logdet, flags = ops.invert_(At)
got = -At[0].cpu().numpy(); want = np.linalg.inv(A)
print("flags", flags.tolist(), "logdet", logdet.item(), np.linalg.slogdet(A)[1])
nb = N // 16
for i in range(nb):
    row = ""
    for j in range(nb):
        if j // 7 > i // 7: row += " ."; continue
        e = np.abs(got[16*i:16*i+16, 16*j:16*j+16] - want[16*i:16*i+16, 16*j:16*j+16]).max() / np.abs(want).max()
        row += " %d" % min(9, max(0, int(-np.log10(e + 1e-30)))) if np.isfinite(e) else " N"
    print(row)
