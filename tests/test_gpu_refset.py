"""DATASET-SCALE parity against the reference itself: the HIP path over ALL 14 Set14 images of every output set the
reference committed and this repo reproduces (tests/golden/refset: the reference's PNG data under its own file names,
tests/golden/make_refset_fixtures.py), compared pixel by pixel and through the Evaluator's metrics (evaluation.py:83-98:
BGR2GRAY, crop SRF px, PSNR / SSIM against *_HR.png) -- the north star's "Set14 PSNR parity within 0.01 dB" as a number
against the REFERENCE's files, not against this repo's oracle.  tools/refset_probe.py is the same run with a report.

What the sets are (settings found by probing):
  image_SRF_2/*_GPR_matern52.png   YUV, every patch from the initial hyper-parameters, BILINEAR  (x2, n = 324)
  image_SRF_3/*_GPR_matern32.png   the same at x3 (n = 784)          image_SRF_4/*_GPR_matern32.png   the same, x4 = x2 o x2
  image_SRF_2/*_GPR_matern12.png   YUV, today's code path: the warm-start chain running through all 14 images in file order
  image_SRF_3/*_GPR_linear.png     the same with the linear kernel at x3
  image_SRF_{2,3}/*_AGPR2.png      agpr.py:284-289 on the reference's committed BSD100 checkpoints
The two independent-start sets at x2 / x3 were written by a state of the reference whose patch grid did not repeat a
location equal to the previous one (clamped duplicate COLUMNS dropped, duplicate rows kept; tools/refset_probe.py
HISTORIC_GEOMETRY): with that grid all 14 images of both sets are reproduced, with today's image_patches.py:31-41 (which
the product follows, bit for bit) 13 of 14 -- the test runs them on the grid that wrote them.
Known exceptions, asserted as such: x4 matern32 img_014 (differences confined to the image's last patch rows: written
with duplicate ROWS dropped -- 99.71 %, max 1 on that grid -- while the other 13 images of the set need them kept) and
x2 matern12 img_013 (Set14's ppt3 slide: ~10 of its 330 patches, flat ones, where 50 Adam steps on a degenerate
likelihood amplify last-bit differences between GPyTorch's Cholesky and this engine's sweep; the chain continues
unharmed: img_014 is reproduced on every pixel again).  The x3 linear chain is reproduced for images 1-8 and drifts from
image 9 on (a chain carries every rounding difference forward).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# set -> (min identical share, max |diff|, max |dPSNR| dB, max |dSSIM|) for every image but the exceptions
BARS = {
    "x2_m52": (0.9999, 1, 1e-4, 1e-5),
    "x3_m32": (0.999, 3, 1e-4, 1e-5),     # luma on every pixel; what differs is the x3 uint8 chroma model
    "x4_m32": (0.9999, 1, 1e-4, 1e-5),
    "x2_m12": (0.9999, 1, 1e-4, 1e-5),
    "agpr2_x2": (0.9999, 1, 1e-4, 1e-5),
    "agpr2_x3": (0.999, 3, 1e-4, 1e-5),
}
EXCEPTIONS = {("x4_m32", 14): (0.95, 24, 0.01, 5e-4), ("x2_m12", 13): (0.96, 10, 0.01, 5e-4)}


@pytest.mark.parametrize("name", list(BARS))
def test_all_14_images_reproduce_the_references_committed_outputs(name):
    import refset_probe
    summ = refset_probe.run_set(name, verbose=False)
    assert summ["images"] == 14
    for r in summ["rows"]:
        ident, maxd, dps, dss = BARS[name] if (name, r["image"]) not in EXCEPTIONS else EXCEPTIONS[(name, r["image"])]
        tag = f"{name} img_{r['image']:03d}"
        assert r["identical"] >= ident, f"{tag}: {100 * r['identical']:.4f} % identical"
        assert r["max_abs_diff"] <= maxd, f"{tag}: max |diff| {r['max_abs_diff']}"
        assert abs(r["psnr_out"] - r["psnr_ref"]) <= dps, f"{tag}: dPSNR {r['psnr_out'] - r['psnr_ref']:+.5f} dB"
        assert abs(r["ssim_out"] - r["ssim_ref"]) <= dss, f"{tag}: dSSIM {r['ssim_out'] - r['ssim_ref']:+.6f}"
    # the north star's bar on the data set: mean PSNR within 0.01 dB of the reference's own files
    assert abs(summ["mean_dpsnr"]) <= 0.01 * 0.2, f"{name}: mean dPSNR {summ['mean_dpsnr']:+.5f} dB"
    print(f"{name}: min identical {100 * summ['min_identical']:.4f} %, mean PSNR {summ['mean_psnr_out']:.4f} dB vs the "
          f"reference's {summ['mean_psnr_ref']:.4f} dB")


def test_x3_linear_chain_reproduces_the_first_eight_images():
    """The linear kernel's chain at x3: images 1-8 as the reference wrote them (luma on every pixel), then the chain
    drifts -- 99.3-99.5 % identical for images 9-12 and 14, mean PSNR of those within 0.001 dB -- and img_013 differs."""
    import refset_probe
    summ = refset_probe.run_set("x3_linear", verbose=False)
    rows = {r["image"]: r for r in summ["rows"]}
    for i in range(1, 9):
        assert rows[i]["identical"] >= 0.999 and rows[i]["max_abs_diff"] <= 3, (i, rows[i])
        assert abs(rows[i]["psnr_out"] - rows[i]["psnr_ref"]) <= 1e-4
    for i in (9, 10, 11, 12, 14):
        assert rows[i]["identical"] >= 0.98 and rows[i]["max_abs_diff"] <= 4, (i, rows[i])
        assert abs(rows[i]["psnr_out"] - rows[i]["psnr_ref"]) <= 0.002
