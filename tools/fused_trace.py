"""Where the cycles of a k_fit_fused workgroup go, by role.  Needs the instrumented build:
    hipcc -O3 --offload-arch=gfx950 -shared -fPIC -ffp-contract=off -DGPSR_FUSED_TRACE=1 \
          -o tools/bin/libgpsr_ftrace.so super_resolution_amd/csrc/gpsr_api.hip
    GPSR_LIB=$PWD/tools/bin/libgpsr_ftrace.so python tools/fused_trace.py [B n kernel_id iters]
"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from super_resolution_amd import _lib, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10800
n = int(sys.argv[2]) if len(sys.argv) > 2 else 324
kid = int(sys.argv[3]) if len(sys.argv) > 3 else 2
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
lib = _lib.load()
lib.gpsr_debug_fused_trace.restype = ctypes.c_int
lib.gpsr_debug_fused_trace.argtypes = [ctypes.c_void_p]
rng = np.random.default_rng(1)
X = torch.as_tensor(rng.random((B, n, 8))).cuda()
y = torch.as_tensor(rng.random((B, n))).cuda()
raw0 = ops.init_hypers(y, kid)
ops.set_fused(1)
ws = ops.fit_workspace(B, n, 8, X.device)
buf = (ctypes.c_ulonglong * 16)()
ops.fit(X, y, kid, raw0.clone(), iters=iters, workspace=ws); torch.cuda.synchronize()
lib.gpsr_debug_fused_trace(buf)   # clear
t0 = time.perf_counter()
ops.fit(X, y, kid, raw0.clone(), iters=iters, workspace=ws); torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.gpsr_debug_fused_trace(buf)
ne = max(1, buf[0])
print(f"B={B} n={n} iters={iters}: {dt * 1e3:.1f} ms, {dt / B / (iters + 1) * 1e6:.3f} us per GP-evaluation; {buf[0]} evaluations traced")
names = ["", "covariance", "pivots", "panels", "single-step / restricted updates", "two-pivot updates", "last update + alpha partials",
         "alpha reduce", "gradient statistics", "whole evaluation"]
for i in range(1, 10):
    print(f"  {names[i]:36s} {buf[i] / ne:10.0f} cycles per evaluation ({100.0 * buf[i] / max(1, buf[9]):5.1f} %)")
