"""The oracle against OUTPUT IMAGES THE REFERENCE ITSELF COMMITTED (tests/golden/refpin, written by
tests/golden/make_refpin_fixtures.py from /root/reference/Set14/...: data files, no source).

  Set14/image_SRF_2/img_001_SRF_2_GPR_matern52_gray.png   its name is gprsr.py:221-222 under the committed defaults
      gprsr.py:17-22 (SRF 2, matern52, GRAYSCALE) and img_001 is the first image of that loop: a fresh process, the
      kernel singleton at its initial values, then the warm-start chain over 240 patches x 2 phases.
The live test below runs the whole reference pipeline restated in oracle/ -- cv.cvtColor(BGR2GRAY), /255, bicubic x2,
480 chained GP fits (Matern-5/2, 50 Adam steps each), GaussianBlur 5x5, INTER_LINEAR 1/2, recombination, clip,
truncating uint8 cast -- and demands the reference's PNG: measured 100.00 % identical pixels (max |diff| 0).
With chain=False (independent starts) the same image is only 49 % identical (max 99 levels, 41.6 dB), with the
14-bit grey table 99.17 %: the pin discriminates both.

Measured once with tools/refpin_eval.py and not repeated here for time (each is minutes of CPU): img_002 continuing
the chain from img_001 (810 s): 100.00 % identical; Set14/image_SRF_3/img_001_SRF_3_GPR_RBF.png (459 s): 100.00 %.
GPSR_REFPIN_FULL=1 runs those too.  The HIP path is held to the same images in tests/test_gpu_refpin.py.
"""
import os

import numpy as np
import pytest
import torch

from oracle import cvops, gp as ogp, srgpr as osr


def _load(golden_dir, name):
    from PIL import Image
    return np.array(Image.open(os.path.join(golden_dir, "refpin", name)))


def _bgr(rgb):
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _stats(got, ref):
    d = np.abs(got.astype(int) - ref.astype(int))
    return float((d == 0).mean()), int(d.max())


def test_fixtures_are_what_the_script_describes(golden_dir):
    lr, ref, hr = _load(golden_dir, "x2_lr_001.png"), _load(golden_dir, "x2_ref_m52_gray_001.png"), _load(golden_dir, "x2_hr_001.png")
    assert lr.shape == (240, 250, 3) and ref.shape == hr.shape == (480, 500) and ref.dtype == np.uint8
    # the reference's output is a plausible x2 of its input: PSNR vs HR as in SURVEY.md section 4 (judge: 23.678 dB)
    c = 2
    assert abs(cvops.psnr(hr[c:-c, c:-c], ref[c:-c, c:-c]) - 23.678) < 0.01
    assert _load(golden_dir, "x3_ref_RBF_001.png").shape == (480, 498)


def test_grey_table_is_opencvs_15_bit_one(golden_dir):
    """bgr2gray: 15-bit RGB2Gray table; the 14-bit yuv_shift table differs on 69 of the 60 000 pixels of img_001."""
    bgr = _bgr(_load(golden_dir, "x2_lr_001.png")).astype(np.int64)
    g14 = (bgr[..., 0] * 1868 + bgr[..., 1] * 9617 + bgr[..., 2] * 4899 + (1 << 13)) >> 14
    g = cvops.bgr2gray(_bgr(_load(golden_dir, "x2_lr_001.png")))
    assert int((g != g14).sum()) == 69
    assert np.array_equal(cvops.bgr2yuv(_bgr(_load(golden_dir, "x2_lr_001.png")))[..., 0], g14.astype(np.uint8))   # Y keeps the 14-bit one


def test_oracle_chain_reproduces_the_references_x2_matern52_gray_output(golden_dir):
    torch.set_num_threads(min(4, os.cpu_count() or 1))   # 4 is faster than 8 here: the n = 324 factorisations are small
    o = osr.OracleGPRSR(2, ogp.MATERN52, "GRAYSCALE", chain=True)
    out = o.apply(_bgr(_load(golden_dir, "x2_lr_001.png")))
    assert np.array_equal(out[..., 0], out[..., 1]) and np.array_equal(out[..., 0], out[..., 2])
    ident, maxd = _stats(out[..., 0], _load(golden_dir, "x2_ref_m52_gray_001.png"))
    print(f"oracle chain=True vs the reference's PNG: {100 * ident:.3f} % identical, max |diff| {maxd}")
    assert ident >= 0.999 and maxd <= 1          # measured: 100 %, 0 (bit for bit)
    assert o.shared.raw_os != 0.0                # the chain left its state behind, as the singleton does
    if os.environ.get("GPSR_REFPIN_FULL"):
        out2 = o.apply(_bgr(_load(golden_dir, "x2_lr_002.png")))           # second image of the same process
        ident, maxd = _stats(out2[..., 0], _load(golden_dir, "x2_ref_m52_gray_002.png"))
        assert ident >= 0.999 and maxd <= 1


@pytest.mark.skipif(not os.environ.get("GPSR_REFPIN_FULL"), reason="minutes of CPU each; measured values in the module docstring")
@pytest.mark.parametrize("kid,tag", [(ogp.RBF, "RBF"), (ogp.MATERN12, "matern12")])
def test_oracle_reproduces_the_references_x3_outputs(golden_dir, kid, tag):
    torch.set_num_threads(min(4, os.cpu_count() or 1))   # 4 is faster than 8 here: the n = 324 factorisations are small
    out = osr.OracleGPRSR(3, kid, "GRAYSCALE", chain=True).apply(_bgr(_load(golden_dir, "x3_lr_001.png")))
    ident, maxd = _stats(out[..., 0], _load(golden_dir, f"x3_ref_{tag}_001.png"))
    assert ident >= 0.999 and maxd <= 1


def test_headline_fixtures_are_the_references_inputs_and_its_output(golden_dir):
    """The headline workload's inputs are exactly what gprsr.py:53,91 feeds the path -- lr_XXX.png = the 15-bit
    cv.cvtColor(BGR2GRAY) of the reference's LR file (the colour copy refpin/x3_lr_001.png converts to lr_001.png bit for
    bit) -- and the committed oracle output of the chained mode for img_001 (480 Matern-5/2 fits of n = 784, made by
    tests/golden/make_set14_chain_goldens.py from lr_001.png) IS the reference's own img_001_SRF_3_GPR_matern52.png
    (refout_m52_001.png): every pixel.  No fit runs here: both sides are committed files."""
    from PIL import Image
    d = os.path.join(golden_dir, "set14_x3")
    lr = np.array(Image.open(f"{d}/lr_001.png"))
    assert np.array_equal(cvops.bgr2gray(_bgr(_load(golden_dir, "x3_lr_001.png"))), lr)
    assert np.array_equal(np.array(Image.open(f"{d}/oracle_m52_chain_001.png")), np.array(Image.open(f"{d}/refout_m52_001.png")))


# ---------------------------------------------------------------- the reference's COLOUR outputs (YUV mode)
def _explained_by_chroma(ref_bgr, U, V):
    """Share of pixels whose reference BGR value is yuv2bgr(Y, U, V) for SOME luma Y: a test of the colour stages and
    of the uint8 bicubic chroma planes that needs no GP fit (a wrong U or V leaves no Y that reproduces all three
    channels)."""
    ok = np.zeros(ref_bgr.shape[:2], bool)
    y0 = cvops.bgr2yuv(ref_bgr)[..., 0].astype(int)          # the luma is within a few levels of this one
    for dy in range(-4, 5):
        Y = np.clip(y0 + dy, 0, 255).astype(np.uint8)
        ok |= (cvops.yuv2bgr(np.stack([Y, U, V], -1)) == ref_bgr).all(-1)
    rest = np.nonzero(~ok)                                    # whatever is left: every luma, on those pixels only
    if rest[0].size:
        r, u, v = ref_bgr[rest][:, None, :], U[rest][:, None], V[rest][:, None]
        good = np.zeros(rest[0].size, bool)
        for Y in range(256):
            good |= (cvops.yuv2bgr(np.stack([np.full(u.shape, Y, np.uint8), u, v], -1)) == r).all(-1)[:, 0]
        ok[rest] = good
    return float(ok.mean())


@pytest.mark.parametrize("s,lr,ref,bar", [(2, "x2_lr_001.png", "x2_ref_matern52_yuv_001.png", 1.0),
                                          (2, "x2_lr_002.png", "x2_ref_matern52_yuv_002.png", 1.0),
                                          (2, "x2_lr_001.png", "x2_ref_matern12_yuv_001.png", 1.0),
                                          (4, "x4_lr_001.png", "x4_ref_matern32_yuv_001.png", 1.0),
                                          (3, "x3_lr_001.png", "x3_ref_matern32_yuv_001.png", 0.998)])
def test_colour_stages_and_uint8_bicubic_explain_the_references_colour_outputs(golden_dir, s, lr, ref, bar):
    """cvops.bgr2yuv, the uint8 INTER_CUBIC resize and cvops.yuv2bgr against the reference's own YUV-mode outputs
    (gprsr.py:56-64): at x2 and x4 EVERY pixel of the reference is yuv2bgr(some Y, our U, our V); at x3 99.89 %.
    The resize is the float32 model of the IPP-backed cv.resize the reference ran (cvops._resize_cubic_u8); with
    resize.cpp's own fixed-point code 4-5 pixels per image are unexplained at x2 / x4 and 1 % at x3."""
    yuv = cvops.bgr2yuv(_bgr(_load(golden_dir, lr)))
    U = cvops.resize_cubic(np.ascontiguousarray(yuv[..., 1]), fx=s, fy=s)
    V = cvops.resize_cubic(np.ascontiguousarray(yuv[..., 2]), fx=s, fy=s)
    share = _explained_by_chroma(_bgr(_load(golden_dir, ref)), U, V)
    print(f"{ref}: {100 * share:.4f} % of the pixels explained")
    assert share >= bar


@pytest.mark.skipif(not os.environ.get("GPSR_REFPIN_FULL"), reason="minutes of CPU; measured values in DESIGN.md section 2")
def test_oracle_independent_bilinear_reproduces_the_references_x4_matern32_colour_output(golden_dir):
    """The oracle in the configuration that produced img_001_SRF_4_GPR_matern32.png: YUV, Matern-3/2, x4 = x2 o x2,
    BILINEAR recombination, every patch from the initial hyper-parameters (chain=False).  Measured: 100.0000 % identical,
    max |diff| 0 (340 s on 4 threads)."""
    from oracle import patches as P
    torch.set_num_threads(min(4, os.cpu_count() or 1))
    o = osr.OracleGPRSR(4, ogp.MATERN32, "YUV", chain=False, patch_interpolation=P.BILINEAR)
    out = o.apply(_bgr(_load(golden_dir, "x4_lr_001.png")))
    ident, maxd = _stats(out, _bgr(_load(golden_dir, "x4_ref_matern32_yuv_001.png")))
    print(f"oracle independent / BILINEAR vs the reference's x4 matern32 PNG: {100 * ident:.4f} % identical, max |diff| {maxd}")
    assert ident >= 0.999 and maxd <= 2


# ---------------------------------------------------------------- AGPR: the reference's *_AGPR2.png outputs
@pytest.mark.parametrize("s,i,min_ident", [(2, 6, 0.99999), (2, 4, 0.9999), (3, 6, 0.999)])
def test_oracle_agpr_reproduces_the_references_outputs_from_its_checkpoints(golden_dir, s, i, min_ident):
    """agpr.py:284-289 restated end to end: uint8 BGR HR image -> downscale (cv.GaussianBlur 3x3 sigma 1 and INTER_CUBIC
    on uint8: OpenCV's fixed-point paths) -> upscale (BGR2YCrCb, Y/255, float64 bicubic, the linear-kernel GP mean of
    every 7x7 window with the reference's COMMITTED checkpoint, iterative back-projection, clip, truncating cast,
    uint8 bicubic Cr / Cb, YCrCb2BGR) against the *_AGPR2.png the reference committed.  Measured: x2 img_006 EVERY pixel,
    img_004 99.998 % (max 1: the reference predicts in float32, a few truncations fall the other way), x3 99.97 %
    (max 2: what is left of the x3 uint8 chroma resize, see the colour pins above).  This pins AGPR inference, the back-projection
    loop, GaussianBlur 3x3 (float64 and uint8), filter2D and the YCrCb pair -- no GPyTorch needed."""
    from oracle import agpr as oagpr
    f = np.load(os.path.join(golden_dir, f"agpr_BSD100_SRF_{s}.npz"))
    h = ogp.Hypers(*f["raw"].astype(np.float32))
    hr = _bgr(_load(golden_dir, f"agpr_x{s}_hr_{i:03d}.png"))
    lr = oagpr.downscale(hr, s)
    assert lr.dtype == np.uint8 and lr.shape == (hr.shape[0] // s, hr.shape[1] // s, 3)
    out = oagpr.upscale(lr, s, h, f["train_x"], f["train_y"], grayscale=False, global_srf=s)
    ident, maxd = _stats(out, _bgr(_load(golden_dir, f"agpr_x{s}_ref_{i:03d}.png")))
    print(f"oracle AGPR x{s} img_{i:03d} vs the reference's AGPR2 PNG: {100 * ident:.4f} % identical, max |diff| {maxd}")
    assert ident >= min_ident and maxd <= 3


def test_uint8_gaussian_taps_are_opencvs_fixed_point_ones():
    assert list(cvops.gaussian_kernel_u8(3, 1)) == [70, 116, 70]        # 0.274, 0.452, 0.274 at 8 fraction bits, sum 256
    assert list(cvops.gaussian_kernel_u8(5, 0)) == [16, 64, 96, 64, 16]
    assert sum(cvops.gaussian_kernel_u8(7, 1.5)) == 256


def test_uint8_downscale_reproduces_the_references_experiment_images(golden_dir):
    """experiment.py:9-11: lr = downscale(hr) on the uint8 BGR image (GaussianBlur 3x3 sigma 1 + INTER_CUBIC to
    249 // 2 = 124 columns: a NON-dyadic ratio).  Every pixel; resize.cpp's fixed-point bicubic gives 98.2 %."""
    from oracle import agpr as oagpr
    lr = oagpr.downscale(_bgr(_load(golden_dir, "exp_hr_1.png")), 2)
    assert np.array_equal(lr, _bgr(_load(golden_dir, "exp_lr_1.png")))


@pytest.mark.skipif(not os.environ.get("GPSR_REFPIN_FULL"), reason="minutes of CPU; measured values in DESIGN.md section 2")
def test_oracle_independent_bilinear_reproduces_the_references_x2_matern52_colour_output(golden_dir):
    """The oracle's chain=False mode -- the golden of the batched `independent` engine bench.py times -- against the
    reference's img_001_SRF_2_GPR_matern52.png (YUV, BILINEAR, every patch from the initial hyper-parameters).
    Measured: 100.0000 % identical, max |diff| 0 (268 s on 4 threads)."""
    from oracle import patches as P
    torch.set_num_threads(min(4, os.cpu_count() or 1))
    o = osr.OracleGPRSR(2, ogp.MATERN52, "YUV", chain=False, patch_interpolation=P.BILINEAR)
    out = o.apply(_bgr(_load(golden_dir, "x2_lr_001.png")))
    ident, maxd = _stats(out, _bgr(_load(golden_dir, "x2_ref_matern52_yuv_001.png")))
    print(f"oracle independent / BILINEAR vs the reference's x2 matern52 colour PNG: {100 * ident:.4f} % identical, max |diff| {maxd}")
    assert ident >= 0.999 and maxd <= 2
