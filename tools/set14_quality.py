"""PSNR / SSIM (evaluation.py:83-98 definition: grey, cropped by the factor) of the engine's Set14 x3 output against the
HR images, next to bicubic interpolation and the oracle's output -- on the fixtures committed under tests/golden/set14_x3
(the reference's Set14 images, greyscale).  Prints a small table; needs a GPU."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from super_resolution_amd import evaluation, imgproc, kernels as K  # noqa: E402
from super_resolution_amd.gprsr import GPRSR, ColorSpace  # noqa: E402


def read_gray(path):
    img = evaluation._imread(path)
    return img if img.ndim == 2 else imgproc.bgr2gray(img)


def main():
    d = os.path.join(ROOT, "tests", "golden", "set14_x3")
    ids = sorted(int(os.path.basename(p)[3:6]) for p in glob.glob(os.path.join(d, "lr_*.png")))
    lrs = [read_gray(os.path.join(d, f"lr_{i:03d}.png")) for i in ids]
    hrs = [read_gray(os.path.join(d, f"hr_{i:03d}.png")) for i in ids]
    ors = [read_gray(os.path.join(d, f"oracle_m52_{i:03d}.png")) for i in ids]
    outs = GPRSR(3, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="independent").apply_for_channel_batch(lrs)
    bic = [imgproc.resize_cubic_u8(lr, fx=3, fy=3) for lr in lrs]
    c = 3
    rows = []
    for name, imgs in (("bicubic", bic), ("SRGPR Matern-5/2, this engine", outs), ("SRGPR Matern-5/2, CPU oracle", ors)):
        ps = [evaluation.peak_signal_noise_ratio(h[c:-c, c:-c], o[c:-c, c:-c]) for h, o in zip(hrs, imgs)]
        ss = [evaluation.structural_similarity(h[c:-c, c:-c], o[c:-c, c:-c]) for h, o in zip(hrs, imgs)]
        rows.append((name, float(np.mean(ps)), float(np.mean(ss))))
    print("| method | PSNR (dB) | SSIM |\n|---|---|---|")
    for name, p, s in rows:
        print(f"| {name} | {p:.3f} | {s:.4f} |")


if __name__ == "__main__":
    main()
