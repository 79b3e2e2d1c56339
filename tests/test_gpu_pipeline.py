"""GPU parity tests of the image stages and of the whole GPRSR pipeline against
the CPU oracle, and against the committed Set14 x3 oracle goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cvops, gp as ogp, srgpr as osr
from test_gpu_kernels import natural_image, to_dev


# ----------------------------------------------------------------- image stages: bit-exact vs oracle
@pytest.mark.parametrize("H,W,s", [(37, 41, 2), (30, 53, 3), (64, 48, 4)])
def test_resize_cubic_bit_exact(H, W, s):
    from super_resolution_amd import imgproc
    img = natural_image(H, W, H + s)
    got = imgproc.resize_cubic(to_dev(img), fx=s, fy=s).cpu().numpy()
    assert np.array_equal(got, cvops.resize_cubic(img, fx=s, fy=s))
    got = imgproc.resize_cubic(to_dev(img), dsize=(W // 2, H // 2)).cpu().numpy()
    assert np.array_equal(got, cvops.resize_cubic(img, dsize=(W // 2, H // 2)))


@pytest.mark.parametrize("H,W,s", [(74, 82, 2), (90, 159, 3), (64, 48, 4)])
def test_blur_downsample_bit_exact(H, W, s):
    from super_resolution_amd import imgproc
    img = natural_image(H, W, H * s)
    b = imgproc.gaussian_blur(to_dev(img), 5, 0)
    assert np.array_equal(b.cpu().numpy(), cvops.gaussian_blur(img, 5, 0))
    got = imgproc.resize_linear(b, 1 / s, 1 / s).cpu().numpy()
    assert np.array_equal(got, cvops.resize_linear(cvops.gaussian_blur(img, 5, 0), 1 / s, 1 / s))
    b3 = imgproc.gaussian_blur(to_dev(img), 3, 1)
    assert np.array_equal(b3.cpu().numpy(), cvops.gaussian_blur(img, 3, 1))


def test_filter3x3_and_casts_bit_exact():
    from super_resolution_amd import imgproc, ops
    img = natural_image(33, 47, 3)
    k = np.array([[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]])
    assert np.array_equal(imgproc.filter2d_3x3(to_dev(img), k).cpu().numpy(), cvops.filter2d_3x3(img, k))
    u8 = np.random.default_rng(0).integers(0, 256, (20, 31)).astype(np.uint8)
    f = ops.u8_to_unit(to_dev(u8))
    assert np.array_equal(f.cpu().numpy(), u8.astype(np.double) / 255.)
    x = np.random.default_rng(1).random((20, 31))
    assert np.array_equal(ops.unit_to_u8(to_dev(x)).cpu().numpy(), (x * 255).astype(np.uint8))


def test_colour_host_code_matches_oracle():
    from super_resolution_amd import imgproc
    bgr = np.random.default_rng(2).integers(0, 256, (24, 30, 3)).astype(np.uint8)
    assert np.array_equal(imgproc.bgr2gray(bgr), cvops.bgr2gray(bgr))
    y, u, v = imgproc.bgr2yuv(bgr)
    assert np.array_equal(np.stack([y, u, v], -1), cvops.bgr2yuv(bgr))
    assert np.array_equal(imgproc.yuv2bgr(y, u, v), cvops.yuv2bgr(cvops.bgr2yuv(bgr)))
    yy, cr, cb = imgproc.bgr2ycrcb(bgr)
    assert np.array_equal(np.stack([yy, cr, cb], -1), cvops.bgr2ycrcb(bgr))
    assert np.array_equal(imgproc.ycrcb2bgr(yy, cr, cb), cvops.ycrcb2bgr(cvops.bgr2ycrcb(bgr)))
    assert np.array_equal(imgproc.resize_cubic_u8(u, fx=3, fy=3), cvops.resize_cubic(u, fx=3, fy=3))


# ----------------------------------------------------------------- whole pipeline vs oracle
KERNELS = {}


def _kernels():
    from super_resolution_amd import kernels as K
    return {0: K.rbf_kernel, 1: K.exponential_kernel, 2: K.matern32_kernel, 3: K.matern52_kernel}


def _lr_image(H, W, seed):
    return np.rint(natural_image(H, W, seed) * 255).astype(np.uint8)


@pytest.mark.parametrize("kid,s,H,W", [(0, 2, 44, 40), (3, 2, 44, 40), (2, 2, 41, 52), (3, 3, 47, 39), (0, 4, 24, 26)])
def test_apply_for_channel_vs_oracle_independent(kid, s, H, W):
    """Bar (north star): max abs pixel error <= 1/255 on the float image; here we
    hold 2e-3/255 and at most one grey level on the truncated uint8."""
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    lr = _lr_image(H, W, 11 * kid + s)
    o = osr.OracleGPRSR(s, kid, "GRAYSCALE", chain=False)
    want = o.apply_for_channel_float(lr)
    g = GPRSR(s, _kernels()[kid], ColorSpace.GRAYSCALE, warm_start="independent")
    got = g.apply_for_channel_device([__import__("super_resolution_amd").ops.u8_to_unit(to_dev(lr))])[0].cpu().numpy()
    assert got.shape == want.shape == (H * s, W * s)
    assert np.abs(got - want).max() < 2e-3 / 255
    u8 = g.apply_for_channel(lr)
    assert np.abs(u8.astype(int) - (want * 255).astype(np.uint8).astype(int)).max() <= 1
    assert all(int(f.sum()) == 0 for f in g.stats.flags)


def test_apply_for_channel_vs_oracle_chain():
    """warm_start='chain' reproduces the reference's singleton warm-start order."""
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    from super_resolution_amd import kernels as K, ops
    lr = _lr_image(44, 40, 5)
    o = osr.OracleGPRSR(2, ogp.MATERN52, "GRAYSCALE", chain=True)
    want = o.apply_for_channel_float(lr)
    K.matern52_kernel.reset()
    g = GPRSR(2, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="chain")
    got = g.apply_for_channel_device([ops.u8_to_unit(to_dev(lr))])[0].cpu().numpy()
    K.matern52_kernel.reset()
    assert np.abs(got - want).max() < 1e-2 / 255
    # and the chain really differs from the independent start on this image
    ind = osr.OracleGPRSR(2, ogp.MATERN52, "GRAYSCALE", chain=False).apply_for_channel_float(lr)
    assert np.abs(ind - want).max() > 10 * np.abs(got - want).max()


@pytest.mark.parametrize("mode", ["YUV", "BGR", "GRAYSCALE"])
def test_apply_colour_modes_vs_oracle(mode):
    from super_resolution_amd import gprsr
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    from super_resolution_amd import kernels as K
    rng = np.random.default_rng(3)
    base = _lr_image(40, 44, 9).astype(int)
    bgr = np.clip(np.stack([base, base + rng.integers(-20, 20, base.shape), 255 - base], -1), 0, 255).astype(np.uint8)
    gprsr.DO_TIMING = False
    got = GPRSR(2, K.rbf_kernel, ColorSpace[mode]).apply(bgr)
    want = osr.OracleGPRSR(2, ogp.RBF, mode).apply(bgr)
    assert got.shape == want.shape == (80, 88, 3) and got.dtype == np.uint8
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 2  # truncation flips of one level, through the colour matrix


def test_batch_of_images_equals_one_by_one():
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    from super_resolution_amd import kernels as K
    imgs = [_lr_image(44, 40, 1), _lr_image(40, 47, 2), _lr_image(52, 41, 3)]
    g = GPRSR(2, K.matern32_kernel, ColorSpace.GRAYSCALE, warm_start="independent")
    together = g.apply_for_channel_batch(imgs)
    for im, t in zip(imgs, together):
        assert np.array_equal(GPRSR(2, K.matern32_kernel, ColorSpace.GRAYSCALE, warm_start="independent").apply_for_channel(im), t)


def test_patch_handler_combine_on_gpu(golden_dir):
    """Drop-in PatchHandler.combine_patches (GPU) vs the reference's own output."""
    from super_resolution_amd.image_patches import PatchHandler, PatchInterpolationMode
    d = np.load(f"{golden_dir}/patches_iter_combine.npz")
    for tag in ("s2", "s3", "s2dup"):
        H, W, p, s = [int(v) for v in d[tag + "_meta"]]
        h = PatchHandler([d[tag + "_a"], d[tag + "_b"], d[tag + "_c"]], base_patch_size=p)
        ups = list(d[tag + "_up"])
        assert np.array_equal(h.combine_patches(ups, PatchInterpolationMode.AVERAGE), d[tag + "_avg"])
        assert np.array_equal(h.combine_patches(ups, PatchInterpolationMode.BILINEAR), d[tag + "_bil"])


# ----------------------------------------------------------------- Set14 x3 Matern-5/2 (config 2) vs committed oracle goldens
def _set14(golden_dir):
    from PIL import Image
    d = os.path.join(golden_dir, "set14_x3")
    ids = [i for i in range(1, 15) if os.path.exists(f"{d}/oracle_m52_{i:03d}.png")]
    load = lambda n: np.array(Image.open(f"{d}/{n}"))
    return [(i, load(f"lr_{i:03d}.png"), load(f"hr_{i:03d}.png"), load(f"oracle_m52_{i:03d}.png")) for i in ids]


def test_set14_x3_matern52_parity(golden_dir):
    """North-star bar: max abs pixel error <= 1/255, PSNR within 0.01 dB and SSIM
    within 0.001 of the oracle's output, per image and on the Set14 average
    (metrics as evaluation.py:83-98: grey, cropped by srf)."""
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    from super_resolution_amd import kernels as K
    data = _set14(golden_dir)
    assert len(data) >= 1
    g = GPRSR(3, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="independent")
    outs = g.apply_for_channel_batch([lr for _, lr, _, _ in data])
    dp, ds = [], []
    for (i, lr, hr, want), got in zip(data, outs):
        assert got.shape == want.shape == hr.shape
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 1, f"img {i}"
        c = 3
        p_got, p_want = cvops.psnr(hr[c:-c, c:-c], got[c:-c, c:-c]), cvops.psnr(hr[c:-c, c:-c], want[c:-c, c:-c])
        s_got, s_want = cvops.ssim(hr[c:-c, c:-c], got[c:-c, c:-c]), cvops.ssim(hr[c:-c, c:-c], want[c:-c, c:-c])
        assert abs(p_got - p_want) <= 0.01 and abs(s_got - s_want) <= 0.001, f"img {i}"
        dp.append(p_got - p_want); ds.append(s_got - s_want)
    assert abs(np.mean(dp)) <= 0.01 and abs(np.mean(ds)) <= 0.001
    assert all(int(f.sum()) == 0 for f in g.stats.flags)


def test_set14_x3_final_hypers_close_to_oracle(golden_dir):
    """The committed oracle run also recorded the raw hyper-parameters after 50 Adam steps
    (per image, per phase).  float32 Adam trajectories are not bit-comparable; most patches
    must still end within 1e-2 of the oracle and none may be wildly off."""
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    from super_resolution_amd import kernels as K
    data = _set14(golden_dir)[:3]
    hyp = np.load(os.path.join(golden_dir, "set14_x3", "hypers_m52.npz"))
    for i, lr, _, _ in data:
        g = GPRSR(3, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="independent")
        g.apply_for_channel(lr)
        for phase, key in enumerate([f"p1_{i:03d}", f"p2_{i:03d}"]):
            got, want = g.stats.raw[phase].cpu().numpy(), hyp[key]
            assert got.shape == want.shape
            err = np.abs(got - want).max(1)
            assert np.median(err) < 1e-3 and (err < 1e-2).mean() > 0.95 and err.max() < 0.5, (i, phase, err.max())


def test_set14_x3_chain_per_image_vs_oracle_chain_goldens(golden_dir):
    """warm_start="chain_per_image" on ALL 14 Set14 x3 images at once (14 chains of 16..80 patches x 2 phases, link k
    of every chain in one batched launch sequence) against the oracle's chain=True output per image
    (tests/golden/set14_x3/oracle_m52_chain_*.png, tests/golden/make_set14_chain_goldens.py): north-star bar."""
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    from super_resolution_amd import kernels as K
    from PIL import Image
    d = os.path.join(golden_dir, "set14_x3")
    load = lambda n: np.array(Image.open(f"{d}/{n}"))
    ids = list(range(1, 15))
    g = GPRSR(3, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="chain_per_image")
    outs = g.apply_for_channel_batch([load(f"lr_{i:03d}.png") for i in ids])
    ident = []
    for i, got in zip(ids, outs):
        want, hr = load(f"oracle_m52_chain_{i:03d}.png"), load(f"hr_{i:03d}.png")
        dd = np.abs(got.astype(int) - want.astype(int))
        assert dd.max() <= 1, f"img {i}: max |diff| {dd.max()}"
        ident.append((dd == 0).mean())
        c = 3
        assert abs(cvops.psnr(hr[c:-c, c:-c], got[c:-c, c:-c]) - cvops.psnr(hr[c:-c, c:-c], want[c:-c, c:-c])) <= 0.01
        assert abs(cvops.ssim(hr[c:-c, c:-c], got[c:-c, c:-c]) - cvops.ssim(hr[c:-c, c:-c], want[c:-c, c:-c])) <= 0.001
    assert min(ident) > 0.999, ident
    assert all(int(f.sum()) == 0 for f in g.stats.flags)
    # the reference's own committed img_001_SRF_3_GPR_matern52.png IS this run (chained, first image of a fresh process)
    # and lr_001.png is exactly what gprsr.py:53,91 feeds it (15-bit BGR2GRAY of the reference's LR file): the oracle's
    # golden equals that PNG bit for bit (tests/test_refpin.py), the HIP output to the truncating cast's last level
    ref = load("refout_m52_001.png")
    assert np.array_equal(load("oracle_m52_chain_001.png"), ref)
    dd = np.abs(outs[0].astype(int) - ref.astype(int))
    assert dd.max() <= 1 and (dd == 0).mean() > 0.9999, (dd.max(), (dd == 0).mean())


def test_chained_batch_under_a_small_workspace_budget_bit_identical():
    """fit_predict_chains respects max_workspace_bytes: with a budget that holds 3 GPs the 7 chains of a call advance in
    groups of 3 per link (the chains of a link are independent GPs), and the result is the unchunked one to the bit."""
    from super_resolution_amd import kernels as K, ops
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    imgs = [ops.u8_to_unit(to_dev(_lr_image(40 + 3 * k, 44 + 2 * k, 50 + k))) for k in range(7)]
    per = ops._lib.load().gpsr_fit_workspace_bytes(1, 324, 8)
    outs = []
    for budget in (48 << 30, 3 * per + 1024):
        g = GPRSR(2, K.matern32_kernel, ColorSpace.GRAYSCALE, warm_start="chain_per_image", iters=6, max_workspace_bytes=budget)
        assert g._gp.chunk_size(7, 324, 8) == (7 if budget > (1 << 30) else 3)
        outs.append(g.apply_for_channel_device(imgs))
        assert all(int(f.sum()) == 0 for f in g.stats.flags)
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("kid,s,chunk", [(3, 2, 0), (0, 3, 3), (2, 2, 4)])
def test_fused_predict_pixels_entry_point_equals_staged_path(kid, s, chunk):
    """gpsr_predict_pixels (one C call: gather -> init -> fit -> predict -> combine, chunked)
    gives the same bits as the staged engine path, and stays within the oracle bar."""
    from super_resolution_amd import engine, ops
    H, W, p = 47, 44, 10 * s
    lr = natural_image(H, W, 21 + kid)
    hi = cvops.resize_cubic(lr, fx=s, fy=s)
    tin, tst = to_dev(lr), to_dev(hi)
    gp = engine.BatchedGP(kid)
    staged = engine.predict_pixels_batch([(tin, tin, tst)], s, p, gp)[0]
    rows, cols, _ = engine._geometry(H, W, s, p, 0.9)
    tidx = engine.training_indices(p)
    fused, raw, flags = ops.predict_pixels_fused(tin, tin, tst, s, p, to_dev(np.asarray(rows, np.int32)),
                                                 to_dev(np.asarray(cols, np.int32)), to_dev(tidx), kid, chunk=chunk)
    assert int(flags.sum()) == 0
    assert torch.equal(fused, staged)
    o = osr.OracleGPRSR(s, kid, "GRAYSCALE", chain=False)
    o.current_scaling_factor = s
    want = o.predict_pixels(lr, lr, hi)
    assert np.abs(fused.cpu().numpy() - want).max() < 2e-3 / 255


def test_non_default_options_vs_oracle():
    """gprsr.py:17-26 options off their defaults, end to end: grid training points (n = 100 at p = 20: the
    one-block N = 112 path of the fit), BILINEAR recombination, another stride and learning rate."""
    from super_resolution_amd import gprsr, kernels as K, ops
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    from super_resolution_amd.image_patches import PatchInterpolationMode
    from oracle import patches as opatch
    lr = _lr_image(46, 52, 17)
    saved = (gprsr.USE_ALL_PIXELS_FOR_TRAINING, gprsr.PATCH_INTERPOLATION, gprsr.STRIDE_PERCENTAGE, gprsr.LEARNING_RATE)
    try:
        gprsr.USE_ALL_PIXELS_FOR_TRAINING = False
        gprsr.PATCH_INTERPOLATION = PatchInterpolationMode.BILINEAR
        gprsr.STRIDE_PERCENTAGE = 0.7
        gprsr.LEARNING_RATE = 0.05
        g = GPRSR(2, K.matern32_kernel, ColorSpace.GRAYSCALE, warm_start="independent")
        got = g.apply_for_channel_device([ops.u8_to_unit(to_dev(lr))])[0].cpu().numpy()
    finally:
        (gprsr.USE_ALL_PIXELS_FOR_TRAINING, gprsr.PATCH_INTERPOLATION, gprsr.STRIDE_PERCENTAGE,
         gprsr.LEARNING_RATE) = saved
    o = osr.OracleGPRSR(2, ogp.MATERN32, "GRAYSCALE", chain=False, use_all_pixels=False, learning_rate=0.05,
                        stride_percentage=0.7, patch_interpolation=opatch.BILINEAR)
    want = o.apply_for_channel_float(lr)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-3 / 255
    assert all(int(f.sum()) == 0 for f in g.stats.flags)


def test_non_finite_patch_raises_like_the_reference():
    """A GP whose factorisation breaks down is flagged per patch and raised as NotPSDError
    (GPyTorch's failure mode after its jitter retries); the other patches are unaffected."""
    from super_resolution_amd import engine, kernels as K, ops
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    lr = _lr_image(44, 40, 3).astype(np.float64) / 255.
    lr[5, 7] = np.nan
    g = GPRSR(2, K.rbf_kernel, ColorSpace.GRAYSCALE)
    with pytest.raises(engine.NotPSDError):
        g.apply_for_channel_device([to_dev(lr)])
    flags = g.stats.flags[0].cpu().numpy()
    assert flags.sum() >= 1 and flags.sum() < flags.size   # only the patches that contain the pixel
    g.check_failures = False
    out = g.apply_for_channel_device([to_dev(lr)])[0]
    assert out.shape == (88, 80)


@pytest.mark.parametrize("H,W", [(20, 20), (20, 37), (33, 21)])
def test_edge_geometries_vs_oracle(H, W):
    """Smallest legal images: one distinct patch (H = W = p, four times over), one patch row, duplicate patch locations
    (min(r, H-p) maps several grid points to the same corner, image_patches.py:31-41 keeps them)."""
    from super_resolution_amd import ops
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    from super_resolution_amd.image_patches import PatchHandler
    lr = _lr_image(H, W, H + W)
    locs = PatchHandler([np.zeros((H, W)), np.zeros((2 * H, 2 * W))], base_patch_size=20).patch_locations
    assert len(locs) >= 1
    if (H, W) == (20, 20):
        assert [tuple(l) for l in locs] == [(0, 0)] * 4    # range(0, 20, 16) = [0, 16], both clamped to 0
    assert len(set(map(tuple, locs))) < len(locs)          # duplicates really occur here
    want = osr.OracleGPRSR(2, ogp.RBF, "GRAYSCALE", chain=False).apply_for_channel_float(lr)
    g = GPRSR(2, _kernels()[0], ColorSpace.GRAYSCALE)
    got = g.apply_for_channel_device([ops.u8_to_unit(to_dev(lr))])[0].cpu().numpy()
    assert got.shape == want.shape == (2 * H, 2 * W)
    assert np.abs(got - want).max() < 2e-3 / 255


def test_patch_larger_than_image_is_rejected():
    """gprsr.py asserts through PatchHandler (image_patches.py:18-20): same failure mode here."""
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    with pytest.raises(AssertionError):
        GPRSR(2, _kernels()[0], ColorSpace.GRAYSCALE).apply_for_channel(_lr_image(19, 30, 1))


def test_main_cli_end_to_end(golden_dir, tmp_path, capsys):
    """main.py (README.md:22 of the reference names it) over a Set14-shaped folder: reads
    *_LR.png, writes *_GPR_matern52_gray.png, --evaluate prints the PSNR/SSIM lines of
    evaluation.py.  The written image must be the committed oracle output to one grey level."""
    import shutil
    from PIL import Image
    import main
    src = os.path.join(golden_dir, "set14_x3")
    dst = tmp_path / "Set14" / "image_SRF_3"
    dst.mkdir(parents=True)
    nums = [5, 6]
    for i in nums:
        shutil.copy(f"{src}/lr_{i:03d}.png", dst / f"img_{i:03d}_SRF_3_LR.png")
        shutil.copy(f"{src}/hr_{i:03d}.png", dst / f"img_{i:03d}_SRF_3_HR.png")
    rc = main.main(["--srf", "3", "--kernel", "matern52", "--color-space", "grayscale", "--images", "5,6",
                    "--dataset-dir", str(tmp_path / "Set14"), "--batch-images", "--warm-start", "independent", "--evaluate"])
    assert rc == 0
    out = capsys.readouterr().out
    assert "Average - PSNR:" in out and "Upscaling 2 images took" in out
    for i in nums:
        got = np.array(Image.open(dst / f"img_{i:03d}_SRF_3_GPR_matern52_gray.png"))
        want = np.array(Image.open(f"{src}/oracle_m52_{i:03d}.png"))
        assert got.ndim == 3 and np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])
        assert np.abs(got[..., 0].astype(int) - want.astype(int)).max() <= 1
