"""Bounded probe (round 3): does any further setting reproduce the reference's Set14/image_SRF_2/img_001_SRF_2_GPR_RBF.png?
geometry (today's / duplicate columns dropped) x initial values (gprsr.py:167-172's RBF branch / raw 0 as for the other
kernels / GPyTorch's noise and mean defaults) x recombination (AVERAGE / BILINEAR), YUV, per-patch re-initialisation.
Needs a copy of that file under tests/golden/_probe (not a committed fixture).  One line per hypothesis."""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from refset_probe import patch_grid, load_bgr  # noqa: E402
from super_resolution_amd import gprsr, kernels as K, ops  # noqa: E402
from super_resolution_amd.image_patches import PatchInterpolationMode  # noqa: E402

gprsr.DO_TIMING = False
ref = load_bgr(os.path.join(ROOT, "tests/golden/_probe/img_001_SRF_2_GPR_RBF.png"))
lr = load_bgr(os.path.join(ROOT, "tests/golden/refpin/x2_lr_001.png"))
orig_init = ops.init_hypers


def init_variant(kind):
    if kind == "rbf":
        return orig_init
    if kind == "raw0":
        return lambda y, kid, a=0.0, b=0.0, carry=None: orig_init(y, ops.MATERN52, a, b, None)

    def f(y, kid, a=0.0, b=0.0, carry=None):
        r = orig_init(y, kid, a, b, carry)
        r[:, 0] = 0.0
        r[:, 1] = 0.0
        return r
    return f


for geom in (None, "ucols"):
    for init in ("rbf", "raw0", "gpytorch_noise_mean"):
        for interp in ("AVERAGE", "BILINEAR"):
            for iters in (50,) if init != "rbf" else (50, 100):
                gprsr.PATCH_INTERPOLATION = PatchInterpolationMode[interp]
                ops.init_hypers = init_variant(init)
                K.rbf_kernel.reset()
                try:
                    with patch_grid(geom):
                        out = gprsr.GPRSR(2, K.rbf_kernel, gprsr.ColorSpace.YUV, warm_start="independent", iters=iters).apply(lr)
                    d = np.abs(out.astype(int) - ref.astype(int))
                    print(f"geom={geom} init={init} interp={interp} iters={iters}: identical {100 * (d == 0).mean():.3f} % "
                          f"<=1 {100 * (d <= 1).mean():.3f} %  max {d.max()}", flush=True)
                except Exception as e:   # a hypothesis may be numerically infeasible (flagged factorisation)
                    print(f"geom={geom} init={init} interp={interp} iters={iters}: failed {repr(e)[:200]}", flush=True)
                finally:
                    ops.init_hypers = orig_init
