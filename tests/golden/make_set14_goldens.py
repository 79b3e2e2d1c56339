"""Generate tests/golden/set14_x3/*: Set14 x3 inputs + the ORACLE's outputs.

Run in the build container only (reads the reference's Set14 PNG *data* files):
    python tests/golden/make_set14_goldens.py [--workers 6] [--images 1-14]
Writes, per image i (8-bit grey PNGs, lossless):
    lr_XXX.png      cv BGR2GRAY (restated, oracle.cvops.bgr2gray) of img_XXX_SRF_3_LR.png
    hr_XXX.png      the same of img_XXX_SRF_3_HR.png  (PSNR/SSIM target, evaluation.py:95-98)
    oracle_m52_XXX.png   oracle SRGPR x3, Matern-5/2, GRAYSCALE, independent start, uint8 (gprsr.py:100)
and hypers_m52.npz  (per image, per phase: raw hyperparameters after 50 Adam steps).
The per-patch fits of one predict_pixels call are independent in the oracle's
chain=False mode, so they are farmed out to a process pool; results are
identical to the serial loop.
No reference source is stored; the oracle output is this repo's own restatement
(GPyTorch/OpenCV absent: "parity unpinned", see oracle/__init__.py).
"""
import argparse
import os
import sys
from multiprocessing import Pool

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/Set14/image_SRF_3"


def _fit(args):
    import torch
    torch.set_num_threads(1)
    from oracle import gp
    X, y, Xs, kid = args
    return gp.fit_predict_patch(X, y, Xs, kid)


def predict_pixels_pool(pool, tin_img, ttg_img, tst_img, s, kid):
    from oracle import patches as P
    p = int(s * 10)
    locs = P.patch_locations(tin_img.shape, p)
    jobs, tsts = [], []
    for loc in locs:
        tin, ttg, tst = P.extract(tin_img, loc, p, 1), P.extract(ttg_img, loc, p, 1), P.extract(tst_img, loc, p, s)
        jobs.append((P.features(tin), P.targets(ttg), P.features(tst), kid))
        tsts.append(tst)
    res = pool.map(_fit, jobs, chunksize=1)
    for tst, (mean, raw) in zip(tsts, res):
        tst[1:-1, 1:-1] = mean.reshape(tst.shape[0] - 2, tst.shape[0] - 2)
    up = P.combine(tsts, locs, tin_img.shape, p, P.AVERAGE)
    return np.clip(up, 0, 1), np.array([r[1] for r in res])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=6)
    ap.add_argument("--images", default="1-14")
    a = ap.parse_args()
    lo, hi = [int(v) for v in a.images.split("-")]
    from oracle import cvops, gp
    out = os.path.join(HERE, "set14_x3")
    os.makedirs(out, exist_ok=True)
    hyp = {}
    hp = os.path.join(out, "hypers_m52.npz")
    if os.path.exists(hp):
        hyp = dict(np.load(hp))
    with Pool(a.workers) as pool:
        for i in range(lo, hi + 1):
            lr = np.array(Image.open(f"{REF}/img_{i:03d}_SRF_3_LR.png").convert("RGB"))[:, :, ::-1]
            hr = np.array(Image.open(f"{REF}/img_{i:03d}_SRF_3_HR.png").convert("RGB"))[:, :, ::-1]
            g, hg = cvops.bgr2gray(lr), cvops.bgr2gray(hr)
            Image.fromarray(g).save(f"{out}/lr_{i:03d}.png")
            Image.fromarray(hg).save(f"{out}/hr_{i:03d}.png")
            img = g.astype(np.double) / 255.
            s = 3
            bic = cvops.resize_cubic(img, fx=s, fy=s)
            bh, h1 = predict_pixels_pool(pool, img, img, bic, s, gp.MATERN52)
            bl = cvops.resize_linear(cvops.gaussian_blur(bh, 5, 0), 1 / s, 1 / s)
            high, h2 = predict_pixels_pool(pool, bl, img, bh, s, gp.MATERN52)
            u8 = (high * 255).astype(np.uint8)
            Image.fromarray(u8).save(f"{out}/oracle_m52_{i:03d}.png")
            hyp[f"p1_{i:03d}"], hyp[f"p2_{i:03d}"] = h1, h2
            np.savez_compressed(hp, **hyp)
            c = 3
            print(i, g.shape, "psnr oracle", cvops.psnr(hg[c:-c, c:-c], u8[c:-c, c:-c]),
                  "bicubic", cvops.psnr(hg[c:-c, c:-c], (np.clip(bic, 0, 1) * 255).astype(np.uint8)[c:-c, c:-c]), flush=True)


if __name__ == "__main__":
    main()
