"""Copy ONE of the reference's own committed output images (data, a loose golden:
SURVEY.md section 4 item 2) as an 8-bit grey PNG fixture:
    Set14/image_SRF_3/img_001_SRF_3_GPR_matern52.png -> tests/golden/set14_x3/refout_m52_001.png
Run in the build container only.  Its generating settings are unknown (work-in-progress
repo), so tests use it only as a loose pin (PSNR between outputs, PSNR-vs-HR delta)."""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cvops

im = np.array(Image.open("/root/reference/Set14/image_SRF_3/img_001_SRF_3_GPR_matern52.png").convert("RGB"))[:, :, ::-1]
Image.fromarray(cvops.bgr2gray(im)).save(os.path.join(HERE, "set14_x3", "refout_m52_001.png"))
print("ok", im.shape)
