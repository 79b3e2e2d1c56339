"""Dataset-scale fixtures: ALL 14 Set14 images of the output sets the reference committed and this repo reproduces,
copied as DATA (PNG files, no source) into a Set14-shaped folder tests/golden/refset/image_SRF_<s>/ under the
reference's own file names, so that main.py --dataset-dir and the Evaluator can run on it as they would on the
reference's Set14/.  Run in the build container only (reads /root/reference):
    python tests/golden/make_refset_fixtures.py
  image_SRF_{2,3,4}/img_XXX_SRF_s_LR.png            inputs (colour, as cv.imread sees them)
  image_SRF_{2,3}/img_XXX_SRF_s_HR.png              metric target, and the input of the AGPR script (agpr.py:284-289)
  image_SRF_4/img_XXX_SRF_4_HR.png                  metric target only: stored as its BGR2GRAY plane (evaluation.py:96
                                                    converts to grey first; a grey file converts to itself)
  image_SRF_3/*_GPR_matern32.png, image_SRF_4/*_GPR_matern32.png, image_SRF_2/*_GPR_matern52.png
                                                    YUV, independent starts, BILINEAR (an earlier state of the reference)
  image_SRF_3/*_GPR_linear.png, image_SRF_2/*_GPR_matern12.png
                                                    YUV, the warm-start chain through all 14 images in file order
  image_SRF_{2,3}/*_AGPR2.png                       agpr.py:284-289 with the committed BSD100 checkpoints
tests/test_gpu_refset.py compares the HIP path with every one of them.
"""
import os
import shutil
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cvops

REF = "/root/reference/Set14"
OUT = os.path.join(HERE, "refset")
SETS = {2: ("LR", "HR", "GPR_matern52", "GPR_matern12", "AGPR2"),
        3: ("LR", "HR", "GPR_matern32", "GPR_linear", "AGPR2"),
        4: ("LR", "HR", "GPR_matern32")}
extra = sys.argv[1:]   # e.g. 2:GPR_linear to add a set for probing (tools/refset_probe.py; the x2 linear / matern32 /
                       # exponential / RBF sets were probed and are NOT reproduced by any tried setting: not kept)
for e in extra:
    s, t = e.split(":")
    SETS[int(s)] = SETS[int(s)] + (t,)
total = 0
for s, tags in SETS.items():
    d = os.path.join(OUT, f"image_SRF_{s}")
    os.makedirs(d, exist_ok=True)
    for i in range(1, 15):
        for t in tags:
            name = f"img_{i:03d}_SRF_{s}_{t}.png"
            src, dst = f"{REF}/image_SRF_{s}/{name}", f"{d}/{name}"
            if s == 4 and t == "HR":
                bgr = np.array(Image.open(src).convert("RGB"))[:, :, ::-1]
                Image.fromarray(cvops.bgr2gray(bgr)).save(dst)
            else:
                shutil.copyfile(src, dst)
            total += os.path.getsize(dst)
print(f"refset: {total / 2**20:.1f} MiB")
