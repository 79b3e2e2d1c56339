"""Time warm_start='chain' (the reference's literal serial warm start, B=1 fits one after the other)
against the batched independent mode on one small Set14 x3 image.  usage: python tools/chain_time.py [img_no]"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from PIL import Image
from super_resolution_amd import kernels as K
from super_resolution_amd.gprsr import GPRSR, ColorSpace
i = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lr = np.array(Image.open(f"tests/golden/set14_x3/lr_{i:03d}.png"))
for mode in ("independent", "chain"):
    g = GPRSR(3, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start=mode)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = g.apply_for_channel(lr)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nfit = sum(int(f.numel()) for f in g.stats.flags)
    print(f"{mode:12s} image {i} {lr.shape}: {dt*1e3:8.1f} ms for {nfit} fits = {dt*1e3/nfit:6.2f} ms/fit", flush=True)
