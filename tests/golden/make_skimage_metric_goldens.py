"""PSNR / SSIM golden values from the REAL scikit-image (0.18.3), the library the reference's evaluation.py:87,93 calls:
    /opt/conda/bin/python3.9 tests/golden/make_skimage_metric_goldens.py
The build container's default interpreter has no scikit-image, but the image carries a Python 3.9 under /opt/conda whose
site-packages (not on its own sys.path: appended below) hold scikit-image 0.18.3 + numpy + scipy; nothing is installed.
For committed grey fixture pairs (HR image, upscaled image) the script does what Evaluator.get_psnr / get_ssim do after
preprocess_image (evaluation.py:83-98: crop `srf` pixels on every side, then skimage.metrics.peak_signal_noise_ratio and
structural_similarity with data_range=255) and stores the float64 results in tests/golden/skimage_metrics.json.
tests/test_oracle.py / tests/test_gpu_kernels.py hold the oracle's, the host functions' and the HIP metric kernels'
values against them.
"""
import json
import os
import sys

sys.path.append("/opt/conda/lib/python3.9/site-packages")
import numpy as np
import skimage
from PIL import Image
from skimage.metrics import peak_signal_noise_ratio, structural_similarity

HERE = os.path.dirname(os.path.abspath(__file__))
pairs = [(f"set14_x3/hr_{i:03d}.png", f"set14_x3/oracle_m52_{i:03d}.png", 3) for i in range(1, 15)]
pairs += [(f"refpin/x2_hr_{i:03d}.png", f"refpin/x2_ref_m52_gray_{i:03d}.png", 2) for i in (1, 2)]
pairs += [("set14_x3/hr_006.png", "set14_x3/oracle_m52_chain_006.png", 0)]   # no crop
out = {"skimage": skimage.__version__, "numpy": np.__version__, "pairs": []}
for hr, up, crop in pairs:
    a = np.array(Image.open(os.path.join(HERE, hr)))
    b = np.array(Image.open(os.path.join(HERE, up)))
    assert a.ndim == 2 and a.shape == b.shape and a.dtype == np.uint8
    if crop:
        a, b = a[crop:-crop, crop:-crop], b[crop:-crop, crop:-crop]
    out["pairs"].append({"hr": hr, "upscaled": up, "crop": crop,
                         "psnr": float(peak_signal_noise_ratio(a, b, data_range=255)),
                         "ssim": float(structural_similarity(a, b, data_range=255)),
                         "mse": float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))})
with open(os.path.join(HERE, "skimage_metrics.json"), "w") as f:
    json.dump(out, f, indent=1)
print("skimage", skimage.__version__, len(out["pairs"]), "pairs")
