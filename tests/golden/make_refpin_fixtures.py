"""Fixtures that pin the oracle and the HIP path to OUTPUT IMAGES THE REFERENCE ITSELF COMMITTED
(data files, not source).  Run in the build container only (reads /root/reference); the GPU box
only sees the PNGs written here.

  Set14/image_SRF_2/img_00{1,2}_SRF_2_LR.png                 -> refpin/x2_lr_00{1,2}.png   (colour, as cv.imread sees them)
  Set14/image_SRF_2/img_00{1,2}_SRF_2_HR.png                 -> refpin/x2_hr_00{1,2}.png   (grey, metric target)
  Set14/image_SRF_2/img_00{1,2}_SRF_2_GPR_matern52_gray.png  -> refpin/x2_ref_m52_gray_00{1,2}.png
        file name = gprsr.py:221-222 with the committed defaults gprsr.py:17-22 (SRF 2, matern52, GRAYSCALE):
        img_001 is the first image of a fresh process (kernel singletons at their initial values), img_002 the
        second image of the same process (the chain carries over, kernels.py:7-13)
  Set14/image_SRF_3/img_001_SRF_3_LR.png                     -> refpin/x3_lr_001.png       (colour)
  Set14/image_SRF_3/img_001_SRF_3_GPR_{RBF,matern12}.png     -> refpin/x3_ref_{RBF,matern12}_001.png
        (three equal channels: GRAYSCALE runs; RBF has no chain, gprsr.py:167-172)
The reference's three-equal-channel outputs are stored as 8-bit grey PNGs (one channel, no conversion).

COLOUR outputs (stored as they are, RGB PNGs).  Their file names carry only the kernel tag; the settings that produced
them were found by running the HIP path over the candidates (tools/refpin_colour_probe.py) -- each line below is the
one combination that reproduces the file, the others are far off (20-70 % identical pixels):
  Set14/image_SRF_2/img_001_SRF_2_GPR_matern12.png  -> refpin/x2_ref_matern12_yuv_001.png
        YUV (gprsr.py:56-64), exponential kernel, the warm-start chain from a fresh process, AVERAGE: today's code path
  Set14/image_SRF_2/img_00{1,2}_SRF_2_GPR_matern52.png -> refpin/x2_ref_matern52_yuv_00{1,2}.png
  Set14/image_SRF_4/img_001_SRF_4_{LR,GPR_matern32}.png -> refpin/x4_lr_001.png, refpin/x4_ref_matern32_yuv_001.png
  Set14/image_SRF_3/img_001_SRF_3_GPR_{matern32,linear}.png -> refpin/x3_ref_{matern32,linear}_yuv_001.png
        YUV, BILINEAR patch interpolation (image_patches.py:67-70) and EVERY PATCH FROM THE INITIAL HYPER-PARAMETERS:
        an earlier state of the reference in which the kernel was not yet a shared singleton -- exactly this
        repo's warm_start="independent".  (x3 linear: chain / AVERAGE, like matern12.)  x4 runs x2 twice (gprsr.py:94-96).
  Set14/image_SRF_{2,3}/img_XXX_SRF_s_{HR,AGPR2}.png -> refpin/agpr_x{s}_{hr,ref}_XXX.png
        AGPR (agpr.py:284-289) with the committed BSD100 checkpoints: LR = downscale(uint8 BGR HR), YCrCb mode.
  experiments/image{1,2}_{hr,lr,srgpr}.png -> refpin/exp_*.png   (experiment.py:9-28: two uint8 downscales, then
        GPRSR(2).apply -- matern52, YUV, the chain running from image 1 into image 2)
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cvops

REF = "/root/reference/Set14"
OUT = os.path.join(HERE, "refpin")
os.makedirs(OUT, exist_ok=True)


def rgb(path):
    return np.array(Image.open(path).convert("RGB"))


def save_grey3(src, dst):
    a = rgb(src)
    assert np.array_equal(a[..., 0], a[..., 1]) and np.array_equal(a[..., 0], a[..., 2]), src
    Image.fromarray(a[..., 0]).save(os.path.join(OUT, dst))


for i in (1, 2):
    Image.fromarray(rgb(f"{REF}/image_SRF_2/img_{i:03d}_SRF_2_LR.png")).save(f"{OUT}/x2_lr_{i:03d}.png")
    hr = rgb(f"{REF}/image_SRF_2/img_{i:03d}_SRF_2_HR.png")[:, :, ::-1]
    Image.fromarray(cvops.bgr2gray(hr)).save(f"{OUT}/x2_hr_{i:03d}.png")
    save_grey3(f"{REF}/image_SRF_2/img_{i:03d}_SRF_2_GPR_matern52_gray.png", f"x2_ref_m52_gray_{i:03d}.png")
Image.fromarray(rgb(f"{REF}/image_SRF_3/img_001_SRF_3_LR.png")).save(f"{OUT}/x3_lr_001.png")
for tag in ("RBF", "matern12"):
    save_grey3(f"{REF}/image_SRF_3/img_001_SRF_3_GPR_{tag}.png", f"x3_ref_{tag}_001.png")
def save_rgb(src, dst):
    Image.fromarray(rgb(src)).save(os.path.join(OUT, dst))


save_rgb(f"{REF}/image_SRF_2/img_001_SRF_2_GPR_matern12.png", "x2_ref_matern12_yuv_001.png")
for i in (1, 2):
    save_rgb(f"{REF}/image_SRF_2/img_{i:03d}_SRF_2_GPR_matern52.png", f"x2_ref_matern52_yuv_{i:03d}.png")
save_rgb(f"{REF}/image_SRF_4/img_001_SRF_4_LR.png", "x4_lr_001.png")
save_rgb(f"{REF}/image_SRF_4/img_001_SRF_4_GPR_matern32.png", "x4_ref_matern32_yuv_001.png")
for tag in ("matern32", "linear"):
    save_rgb(f"{REF}/image_SRF_3/img_001_SRF_3_GPR_{tag}.png", f"x3_ref_{tag}_yuv_001.png")
for i in (4, 5, 6, 8):   # more (small) images of the x3 Matern-3/2 loop over Set14: independent starts, any order
    save_rgb(f"{REF}/image_SRF_3/img_{i:03d}_SRF_3_LR.png", f"x3_lr_{i:03d}.png")
    save_rgb(f"{REF}/image_SRF_3/img_{i:03d}_SRF_3_GPR_matern32.png", f"x3_ref_matern32_yuv_{i:03d}.png")
# AGPR: the outputs the reference wrote with the checkpoints it committed (agpr.py:284-289: HR -> downscale -> upscale ->
# *_AGPR2.png; DATASET = BSD100).  The checkpoints are fixtures already (tests/golden/agpr_BSD100_SRF_{2,3}.npz).
for s_, i in ((2, 6), (2, 4), (3, 6)):
    save_rgb(f"{REF}/image_SRF_{s_}/img_{i:03d}_SRF_{s_}_HR.png", f"agpr_x{s_}_hr_{i:03d}.png")
    save_rgb(f"{REF}/image_SRF_{s_}/img_{i:03d}_SRF_{s_}_AGPR2.png", f"agpr_x{s_}_ref_{i:03d}.png")
# experiment.py of the reference: hr = downscale(cv.imread(HR x3 file)), lr = downscale(hr) on uint8 BGR images (global
# SRF = 2), srgpr = GPRSR(2).apply(lr) (defaults: matern52, YUV) for image 1 and then image 2 in ONE process.
EXP = "/root/reference/experiments"
for src, dst in (("image1_hr", "exp_hr_1"), ("image1_lr", "exp_lr_1"), ("image1_srgpr", "exp_srgpr_1"),
                 ("image2_lr", "exp_lr_2"), ("image2_srgpr", "exp_srgpr_2")):
    save_rgb(f"{EXP}/{src}.png", f"{dst}.png")
print("ok", sorted(os.listdir(OUT)))
