"""HIP-event timing of the inversion kernels (pivot/panel/update) on B SPD matrices of order N."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from super_resolution_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 633
N = int(sys.argv[2]) if len(sys.argv) > 2 else 784
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
Q = torch.randn((B, N, 64), dtype=torch.float64, device=dev, generator=g)
A0 = Q @ Q.transpose(1, 2) / 64 + 0.5 * torch.eye(N, dtype=torch.float64, device=dev)
A = A0.clone(); ops.invert_(A); torch.cuda.synchronize()
ops.profile_begin(4096)
for _ in range(reps):
    A = A0.clone()
    logdet, flags = ops.invert_(A)
torch.cuda.synchronize()
p = ops.profile_end()
print({k: (v[0], round(v[1] / max(1, v[0]), 4)) for k, v in p.items() if v[0]}, "flags", int(flags.sum()))
