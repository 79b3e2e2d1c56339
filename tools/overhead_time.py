"""Host + image-stage overhead of one step: the pipeline with iters=0 (one evaluation per phase instead of 51)
for 1 and 8 copies of the Set14 x3 workload (what every rank of an N-GPU run does besides its share of the fits)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from super_resolution_amd import kernels as K, ops
from super_resolution_amd.gprsr import GPRSR, ColorSpace
dev = torch.device("cuda", 0)
for copies in (1, 8):
    imgs_u8, s, _ = bench.workload("set14x3", copies)
    imgs = [ops.u8_to_unit(torch.as_tensor(im).to(dev)) for im in imgs_u8]
    g = GPRSR(s, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="independent", iters=0)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = g.apply_for_channel_device(imgs)
        outs = [ops.unit_to_u8(o, 255.0) for o in outs]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("copies", copies, "images", len(imgs), "step with iters=0: %.1f ms" % (dt * 1e3))

# AGPR active sampling at the reference's constants (N = 5000 candidates of D = 50, 500 picks)
import numpy as np
rng = np.random.default_rng(0)
data = torch.as_tensor(np.unique(rng.random((5000, 50)), axis=0)).to(dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx, _ = ops.active_sample(data, 10, 500, 0.2 * (len(data) - 1) / 2, 0.2, 0.8)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("active sampling N=5000 D=50 K=10, 500 picks: %.1f ms" % (dt * 1e3))
