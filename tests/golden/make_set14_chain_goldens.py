"""Generate tests/golden/set14_x3/oracle_m52_chain_XXX.png: the ORACLE's output with the reference's warm-start
chain (oracle chain=True), every image as the first image of a fresh process -- the golden of
GPRSR(warm_start="chain_per_image") and of `bench.py --warm-start chain_per_image`.

Run in the build container:   python tests/golden/make_set14_chain_goldens.py [--workers 8]
Inputs are the committed grey fixtures lr_XXX.png (no reference file is read).  A chain is serial, so the
parallelism is over images (one single-threaded process each).  The oracle in this mode reproduces the reference's
own committed outputs bit for bit (tests/test_refpin.py), so these files are what the reference would write for
`IMAGE_NUMS = [i]`, SRF 3, matern52, GRAYSCALE.
"""
import argparse
import os
import sys
from multiprocessing import Pool

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
D = os.path.join(HERE, "set14_x3")


def run(i):
    import torch
    torch.set_num_threads(1)
    from oracle import gp as ogp, srgpr as osr
    lr = np.array(Image.open(f"{D}/lr_{i:03d}.png"))
    o = osr.OracleGPRSR(3, ogp.MATERN52, "GRAYSCALE", chain=True)
    out = o.apply_for_channel(lr)
    Image.fromarray(out).save(f"{D}/oracle_m52_chain_{i:03d}.png")
    return i, [h.astype(np.float32) for h in o.last_hypers]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--images", default="1-14")
    a = ap.parse_args()
    lo, hi = [int(v) for v in a.images.split("-")]
    ids = sorted(range(lo, hi + 1), key=lambda i: -np.prod(Image.open(f"{D}/lr_{i:03d}.png").size))
    hyp = {}
    with Pool(a.workers) as pool:
        for i, hs in pool.imap_unordered(run, ids):
            hyp[f"p1_{i:03d}"], hyp[f"p2_{i:03d}"] = hs[0], hs[1]
            print("done", i, flush=True)
    np.savez_compressed(f"{D}/hypers_m52_chain.npz", **hyp)
