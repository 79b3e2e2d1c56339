"""HIP path vs OUTPUT IMAGES THE REFERENCE ITSELF COMMITTED (tests/golden/refpin, made by
tests/golden/make_refpin_fixtures.py from /root/reference/Set14/...): this is HIP against reference goldens,
not against the oracle.  The reference's final cast truncates (gprsr.py:100), so a 1e-13 difference at an
integer boundary flips a grey level: the bar is the share of identical pixels and max |diff| <= 1, not bit
equality of the PNG (the float64 oracle does reach bit equality, tests/test_refpin.py).

  x2  img_001, img_002  *_GPR_matern52_gray.png   gprsr.py:17-22 defaults (SRF 2, matern52, GRAYSCALE), one process,
                                                  images in order: the warm-start chain runs through both
  x3  img_001           *_GPR_RBF.png             RBF re-initialises per patch (gprsr.py:167-172): no chain
  x3  img_001           *_GPR_matern12.png        chain, first image of a process
  x3  img_001           *_GPR_matern52.png        the headline configuration (grey, chained)
Colour outputs (YUV mode; settings found with tools/refpin_colour_probe.py, see tests/golden/make_refpin_fixtures.py):
  x2  img_001           *_GPR_matern12.png        chained, AVERAGE
  x2  img_001, img_002  *_GPR_matern52.png        warm_start="independent" + BILINEAR (an earlier state of the reference)
  x4  img_001           *_GPR_matern32.png        the same, x4 = x2 o x2, Matern-3/2
  x3  img_001,4,5,6,8   *_GPR_matern32.png, img_001 *_GPR_linear.png
  experiments/image{1,2}_srgpr.png               GPRSR(2) defaults (matern52, YUV), the chain across two images
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cvops


def _load(golden_dir, name):
    from PIL import Image
    return np.array(Image.open(os.path.join(golden_dir, "refpin", name)))


def _bgr(rgb):
    return np.ascontiguousarray(rgb[:, :, ::-1])


def _compare(got_bgr, ref, min_identical, tag):
    assert got_bgr.ndim == 3 and np.array_equal(got_bgr[..., 0], got_bgr[..., 1]) and np.array_equal(got_bgr[..., 0], got_bgr[..., 2])
    got = got_bgr[..., 0]
    assert got.shape == ref.shape and got.dtype == np.uint8
    d = np.abs(got.astype(int) - ref.astype(int))
    ident = float((d == 0).mean())
    psnr = cvops.psnr(ref, got) if d.any() else float("inf")
    print(f"{tag}: identical {100 * ident:.3f} %, max |diff| {d.max()}, PSNR(out, ref) {psnr:.2f} dB")
    assert d.max() <= 1, f"{tag}: max |diff| {d.max()} grey levels"
    assert ident >= min_identical, f"{tag}: only {100 * ident:.2f} % identical"
    assert psnr >= 65.0
    return ident


def test_x2_matern52_chain_reproduces_the_references_gray_outputs(golden_dir):
    """warm_start="chain" (the default of the drop-in): img_001 from the fresh singleton, img_002 continuing it."""
    from super_resolution_amd import gprsr, kernels as K
    gprsr.DO_TIMING = False
    K.matern52_kernel.reset()
    try:
        g = gprsr.GPRSR(2, K.matern52_kernel, gprsr.ColorSpace.GRAYSCALE)
        assert g.warm_start == "chain"
        for i in (1, 2):
            out = g.apply(_bgr(_load(golden_dir, f"x2_lr_{i:03d}.png")))
            _compare(out, _load(golden_dir, f"x2_ref_m52_gray_{i:03d}.png"), 0.985, f"x2 m52 chain img_{i:03d}")
            assert all(int(f.sum()) == 0 for f in g.stats.flags)
        assert K.matern52_kernel.raw_outputscale != 0.0      # the chain left its state on the singleton
    finally:
        K.matern52_kernel.reset()


def test_x2_matern52_chain_per_image_equals_fresh_process_runs(golden_dir):
    """warm_start="chain_per_image": both images at once, each as the first image of a fresh process.  img_001 must
    reproduce the reference's output (which WAS the first image of its process); img_002 must equal what the literal
    chain gives from a reset singleton, to the bit (same kernels, same order inside the chain)."""
    from super_resolution_amd import gprsr, imgproc, kernels as K
    gprsr.DO_TIMING = False
    lrs = [imgproc.bgr2gray(_bgr(_load(golden_dir, f"x2_lr_{i:03d}.png"))) for i in (1, 2)]
    K.matern52_kernel.reset()
    g = gprsr.GPRSR(2, K.matern52_kernel, gprsr.ColorSpace.GRAYSCALE, warm_start="chain_per_image")
    outs = g.apply_for_channel_batch(lrs)
    assert K.matern52_kernel.raw_outputscale == 0.0 and K.matern52_kernel.base_kernel.raw_parameter == 0.0
    _compare(np.stack([outs[0]] * 3, -1), _load(golden_dir, "x2_ref_m52_gray_001.png"), 0.985, "x2 m52 chain_per_image img_001")
    try:
        lit = gprsr.GPRSR(2, K.matern52_kernel, gprsr.ColorSpace.GRAYSCALE, warm_start="chain").apply_for_channel(lrs[1])
    finally:
        K.matern52_kernel.reset()
    d = np.abs(lit.astype(int) - outs[1].astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.999   # B=1 and B=2 launches of the same kernels: identical up to nothing


def test_x3_rbf_reproduces_the_references_output(golden_dir):
    from super_resolution_amd import gprsr, kernels as K
    gprsr.DO_TIMING = False
    out = gprsr.GPRSR(3, K.rbf_kernel, gprsr.ColorSpace.GRAYSCALE).apply(_bgr(_load(golden_dir, "x3_lr_001.png")))
    _compare(out, _load(golden_dir, "x3_ref_RBF_001.png"), 0.985, "x3 RBF img_001")


def test_x3_matern12_chain_reproduces_the_references_output(golden_dir):
    from super_resolution_amd import gprsr, kernels as K
    gprsr.DO_TIMING = False
    K.exponential_kernel.reset()
    try:
        out = gprsr.GPRSR(3, K.exponential_kernel, gprsr.ColorSpace.GRAYSCALE).apply(_bgr(_load(golden_dir, "x3_lr_001.png")))
    finally:
        K.exponential_kernel.reset()
    _compare(out, _load(golden_dir, "x3_ref_matern12_001.png"), 0.985, "x3 matern12 chain img_001")


# ---------------------------------------------------------------- the reference's COLOUR outputs (YUV mode)
def _compare_colour(got, ref_rgb, min_identical, max_diff, tag):
    ref = _bgr(ref_rgb)
    assert got.shape == ref.shape and got.dtype == np.uint8
    d = np.abs(got.astype(int) - ref.astype(int))
    ident = float((d == 0).mean())
    print(f"{tag}: identical {100 * ident:.4f} %, max |diff| {d.max()}")
    assert d.max() <= max_diff, f"{tag}: max |diff| {d.max()}"
    assert ident >= min_identical, f"{tag}: only {100 * ident:.3f} % identical"


def test_x2_matern12_yuv_chain_reproduces_the_references_colour_output(golden_dir):
    """img_001_SRF_2_GPR_matern12.png: today's code path in YUV mode (gprsr.py:56-64) -- BGR2YUV, the GP on Y (exponential
    kernel, warm-start chain from a fresh process), uint8 bicubic U and V, YUV2BGR.  Pins the colour stages and the
    uint8 cubic resize against a reference golden (measured: every pixel)."""
    from super_resolution_amd import gprsr, kernels as K
    gprsr.DO_TIMING = False
    K.exponential_kernel.reset()
    try:
        out = gprsr.GPRSR(2, K.exponential_kernel, gprsr.ColorSpace.YUV).apply(_bgr(_load(golden_dir, "x2_lr_001.png")))
    finally:
        K.exponential_kernel.reset()
    _compare_colour(out, _load(golden_dir, "x2_ref_matern12_yuv_001.png"), 0.9995, 2, "x2 matern12 YUV chain img_001")


def _independent_bilinear(s, kernel, lr_bgr):
    """The state of the reference that produced its *_GPR_matern52.png / *_GPR_matern32.png colour files: YUV, BILINEAR
    patch interpolation, every patch from the initial hyper-parameters (no shared kernel singleton yet)."""
    from super_resolution_amd import gprsr
    from super_resolution_amd.image_patches import PatchInterpolationMode
    gprsr.DO_TIMING = False
    old = gprsr.PATCH_INTERPOLATION
    gprsr.PATCH_INTERPOLATION = PatchInterpolationMode.BILINEAR
    try:
        return gprsr.GPRSR(s, kernel, gprsr.ColorSpace.YUV, warm_start="independent").apply(lr_bgr)
    finally:
        gprsr.PATCH_INTERPOLATION = old


def test_x2_matern52_yuv_independent_reproduces_the_references_colour_outputs(golden_dir):
    """warm_start="independent" -- the batched mode bench.py times -- IS an algorithm of the reference: its committed
    img_00{1,2}_SRF_2_GPR_matern52.png are reproduced by independent starts + BILINEAR recombination in YUV mode
    (chained starts give 49 % identical pixels on the same file)."""
    from super_resolution_amd import kernels as K
    for i in (1, 2):
        out = _independent_bilinear(2, K.matern52_kernel, _bgr(_load(golden_dir, f"x2_lr_{i:03d}.png")))
        _compare_colour(out, _load(golden_dir, f"x2_ref_matern52_yuv_{i:03d}.png"), 0.9995, 2, f"x2 matern52 YUV independent img_{i:03d}")


def test_x4_matern32_yuv_independent_reproduces_the_references_colour_output(golden_dir):
    """img_001_SRF_4_GPR_matern32.png: Matern-3/2 (BASELINE config 4's kernel), x4 = x2 o x2 (gprsr.py:94-96), BILINEAR,
    independent starts, YUV with x4 chroma."""
    from super_resolution_amd import kernels as K
    out = _independent_bilinear(4, K.matern32_kernel, _bgr(_load(golden_dir, "x4_lr_001.png")))
    _compare_colour(out, _load(golden_dir, "x4_ref_matern32_yuv_001.png"), 0.9995, 2, "x4 matern32 YUV independent img_001")


def test_x3_matern32_and_linear_yuv_reproduce_the_references_colour_outputs(golden_dir):
    """x3 (n = 784): Matern-3/2 independent / BILINEAR, and the linear kernel chained / AVERAGE.  The luma is reproduced
    on every pixel; the differing pixels (max 3) are chroma: the x3 uint8 bicubic U / V planes of the reference's
    IPP-backed cv.resize are modelled to 99.89 % (tests/test_refpin.py shows it without any GP)."""
    from super_resolution_amd import gprsr, kernels as K
    lr = _bgr(_load(golden_dir, "x3_lr_001.png"))
    out = _independent_bilinear(3, K.matern32_kernel, lr)
    _compare_colour(out, _load(golden_dir, "x3_ref_matern32_yuv_001.png"), 0.999, 3, "x3 matern32 YUV independent img_001")
    for i in (4, 5, 6, 8):   # four more images of the same loop over Set14 (independent starts: the order is immaterial)
        out = _independent_bilinear(3, K.matern32_kernel, _bgr(_load(golden_dir, f"x3_lr_{i:03d}.png")))
        _compare_colour(out, _load(golden_dir, f"x3_ref_matern32_yuv_{i:03d}.png"), 0.999, 3, f"x3 matern32 YUV independent img_{i:03d}")
    K.linear_kernel.reset()
    try:
        out = gprsr.GPRSR(3, K.linear_kernel, gprsr.ColorSpace.YUV).apply(lr)
    finally:
        K.linear_kernel.reset()
    _compare_colour(out, _load(golden_dir, "x3_ref_linear_yuv_001.png"), 0.999, 3, "x3 linear YUV chain img_001")


def test_x3_matern52_gray_chain_reproduces_the_references_output(golden_dir):
    """The headline configuration itself (Set14 x3, Matern-5/2, GRAYSCALE): the reference's img_001_SRF_3_GPR_matern52.png
    (tests/golden/set14_x3/refout_m52_001.png) is the chained run from a fresh process (measured: every pixel)."""
    from PIL import Image
    from super_resolution_amd import gprsr, kernels as K
    gprsr.DO_TIMING = False
    ref = np.array(Image.open(os.path.join(golden_dir, "set14_x3", "refout_m52_001.png")))
    K.matern52_kernel.reset()
    try:
        out = gprsr.GPRSR(3, K.matern52_kernel, gprsr.ColorSpace.GRAYSCALE).apply(_bgr(_load(golden_dir, "x3_lr_001.png")))
    finally:
        K.matern52_kernel.reset()
    _compare(out, ref, 0.999, "x3 m52 chain img_001")


def test_experiment_srgpr_yuv_chain_across_two_images(golden_dir):
    """experiment.py:7,12,22 of the reference: GPRSR(2) -- defaults matern52, YUV -- applied to image 1 and then image 2
    in one process: today's chained code path in YUV mode, the chain running from one image into the next."""
    from super_resolution_amd import gprsr, kernels as K
    gprsr.DO_TIMING = False
    K.matern52_kernel.reset()
    try:
        g = gprsr.GPRSR(2)
        assert g.warm_start == "chain" and g.color_mode == gprsr.ColorSpace.YUV and g.used_kernel is K.matern52_kernel
        for k in (1, 2):
            out = g.apply(_bgr(_load(golden_dir, f"exp_lr_{k}.png")))
            _compare_colour(out, _load(golden_dir, f"exp_srgpr_{k}.png"), 0.9999, 2, f"experiment image{k} SRGPR YUV chain")
    finally:
        K.matern52_kernel.reset()
