"""Where the cycles of a k_pivot workgroup go, by phase.  Needs the instrumented build:
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -ffp-contract=off -DGPSR_PIV_TRACE=1 \
          -o super_resolution_amd/libgpsr_trace.so super_resolution_amd/csrc/gpsr_api.hip
    GPSR_LIB=$PWD/super_resolution_amd/libgpsr_trace.so python tools/piv_trace.py [B N]
"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from super_resolution_amd import _lib, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
N = int(sys.argv[2]) if len(sys.argv) > 2 else 784
lib = _lib.load()
lib.gpsr_debug_piv_trace.restype = ctypes.c_int
lib.gpsr_debug_piv_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
Q = torch.randn((B, N, 64), dtype=torch.float64, device=dev, generator=g)
A0 = Q @ Q.transpose(1, 2) / 64 + 0.5 * torch.eye(N, dtype=torch.float64, device=dev)
A = A0.clone(); ops.invert_(A); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
buf2 = (ctypes.c_ulonglong * 8)()
lib.gpsr_debug_piv_trace2.restype = ctypes.c_int
lib.gpsr_debug_piv_trace2.argtypes = [ctypes.c_void_p]
lib.gpsr_debug_piv_trace(buf)          # clear
lib.gpsr_debug_piv_trace2(buf2)
ops.profile_begin(4096)
A = A0.clone(); ops.invert_(A); torch.cuda.synchronize()
p = ops.profile_end()
lib.gpsr_debug_piv_trace(buf)
n = max(1, buf[0])
names = ["workgroups", "load", "blocked Cholesky (7 x chol16 + panels + trailing)", "M = L^-1", "P = M^T M, A_kk stores", "U / M stores", "entry -> exit"]
print("k_pivot:", p.get("pivot"), "(launches, mean ms)")
for i, nm in enumerate(names):
    print(f"  {nm:52s} {buf[i] / n if i else buf[i]:10.0f}" + ("" if i == 0 else f" cycles per workgroup ({100.0 * buf[i] / max(1, buf[6]):.1f} %)"))
lib.gpsr_debug_piv_trace2(buf2)
n2 = max(1, buf2[0])
print("look-ahead wave q+1 inside a step of the blocked Cholesky (cycles per step):")
for i, nm in enumerate(["sub-panel product", "wait at the barrier after it", "diagonal block update", "chol16_inv + publish", "wait at the step's last barrier"]):
    print(f"  {nm:40s} {buf2[i + 1] / n2:8.0f}")
print(f"  wave 6 trailing update (when not look-ahead) {buf2[6] / max(1, buf2[7]):8.0f}")
