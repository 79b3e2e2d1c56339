"""GPU tests at the size of the BASELINE.json configurations that no other test reaches:
  config 4  synthetic 3840x2160 LR x2, Matern-3/2: 32 400 patches at full size, the engine's workspace chunking
  config 3  BSD100-shaped batches (100 images of 160x107 at x3 and of 240x160 at x2) sharded over two ranks
  config 1  one Set14 image_SRF_2 image, x2, RBF, GRAYSCALE, through main.py
  config 5  "n ~ 1k": p = 34 -> n = 1024 training points per GP (10 pivot blocks)
Full-size checks use properties + sampled patches against the oracle (the oracle needs ~0.5 s per n=324 fit)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cvops, gp as ogp, patches as opatch
from test_gpu_kernels import natural_image, to_dev

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_4k_x2_matern32_full_size_chunked():
    """BASELINE config 4, phase 1 (upsample) of the 2160x3840 image: 32 400 GPs of n = 324 in one call."""
    from bench import synthetic_lr
    from super_resolution_amd import engine, imgproc, ops
    H, W, s, p = 2160, 3840, 2, 20
    lr_u8 = synthetic_lr(H, W, 1234)
    lr = ops.u8_to_unit(to_dev(lr_u8))
    hi = imgproc.resize_cubic(lr, fx=s, fy=s)
    rows, cols, locs = engine._geometry(H, W, s, p, 0.9)
    assert locs.shape[0] == 32400 and len({(int(r), int(c)) for r, c in locs}) == 32400      # no duplicates at this size
    st = engine.FitStats()
    gp_small = engine.BatchedGP(ops.MATERN32, max_workspace_bytes=6 << 30)
    assert gp_small.chunk_size(32400, 324, 8) * 3 <= 32400                                  # >= 4 chunks
    out = engine.predict_pixels_batch([(lr, lr, hi)], s, p, gp_small, stats=st)[0]
    torch.cuda.synchronize()
    del gp_small
    assert out.shape == (H * s, W * s) and bool(torch.isfinite(out).all())
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0
    assert int(st.flags[0].sum()) == 0 and st.raw[0].shape == (32400, 4)
    # the chunked result equals the default-budget result to the bit (per-GP arithmetic does not depend on the batch)
    out2 = engine.predict_pixels_batch([(lr, lr, hi)], s, p, engine.BatchedGP(ops.MATERN32))[0]
    assert torch.equal(out, out2)
    del out2
    # 16 patches spread over the image (corners, edges, interior) against the oracle's per-patch loop
    lr_h, hi_h = lr.cpu().numpy(), hi.cpu().numpy()
    pick = np.unique(np.concatenate([[0, len(cols) - 1, 32400 - len(cols), 32399],
                                     np.random.default_rng(0).integers(0, 32400, 12)]))
    sel = to_dev(np.ascontiguousarray(locs[pick]))
    X, y, Xs = ops.gather(lr, lr, hi, s, p, sel, to_dev(engine.training_indices(p)))
    raw = ops.init_hypers(y, ops.MATERN32)
    alpha, fl = ops.fit(X, y, ops.MATERN32, raw)
    pred = ops.predict(X, Xs, alpha, raw, ops.MATERN32).cpu().numpy()
    assert torch.equal(raw, st.raw[0][to_dev(pick)])                                        # same GPs as in the big batch
    for k, b in enumerate(pick[:16]):
        loc = tuple(int(v) for v in locs[b])
        Xo = opatch.features(opatch.extract(lr_h, loc, p, 1))
        yo = opatch.targets(opatch.extract(lr_h, loc, p, 1))
        Xso = opatch.features(opatch.extract(hi_h, loc, p, s))
        mean, _ = ogp.fit_predict_patch(Xo, yo, Xso, ogp.MATERN32)
        assert np.abs(pred[k] - mean).max() < 1e-4, (b, np.abs(pred[k] - mean).max())


CHILD = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from super_resolution_amd import dist as gdist, kernels as K, ops
from super_resolution_amd.gprsr import GPRSR, ColorSpace
from bench import synthetic_lr
rank, world, local = gdist.init_from_env()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
# BSD100 HR is 321x481: LR 107x160 at x3 (35 patches per image, n = 784), 160x240 at x2 (150 patches per image, n = 324)
for s, (H, W), per in ((3, (107, 160), 35), (2, (160, 240), 150)):
    imgs = [ops.u8_to_unit(torch.as_tensor(synthetic_lr(H, W, seed)).to(dev)) for seed in range(100)]
    kern = K.matern52_kernel if s == 3 else K.matern32_kernel
    for mode in ("independent", "chain_per_image"):
        g = GPRSR(s, kern, ColorSpace.GRAYSCALE, warm_start=mode, shard=(rank, world))
        sharded = g.apply_for_channel_device(imgs)
        assert g.stats.raw[0].shape == (100 * per, 4) and all(int(f.sum()) == 0 for f in g.stats.flags)
        if rank == 0:
            whole = GPRSR(s, kern, ColorSpace.GRAYSCALE, warm_start=mode).apply_for_channel_device(imgs)
            assert all(torch.equal(a, b) for a, b in zip(sharded, whole)), mode + ": sharded != unsharded"
            assert all(o.shape == (H * s, W * s) and bool(torch.isfinite(o).all()) for o in whole)
            sys.stdout.write("x%%d-%%s-ok\n" %% (s, mode))
        dist.barrier()
    del imgs, sharded
dist.destroy_process_group()
"""


def test_bsd100_shaped_batch_two_ranks(tmp_path):
    """BASELINE config 3 ("BSD100 x2 and x3") rehearsed on the one GPU of the box: 100 generated images of BSD100's LR
    shape at x3 (160x107: 3 500 patches, 7 000 fits of n = 784) AND at x2 (240x160: 15 000 patches, 30 000 fits of
    n = 324) over two ranks -- patches sharded (independent) and whole chains dealt out (chain_per_image) -- equal the
    one-rank result to the bit.  Collectives ride on gloo here (RCCL refuses two ranks on one device)."""
    script = tmp_path / "child.py"
    script.write_text(CHILD % (ROOT, ROOT))
    env = dict(os.environ, GPSR_LOCAL_DEVICE="0", GPSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and all(f"x{s}-{m}-ok" in r.stdout for s in (3, 2) for m in ("independent", "chain_per_image")), \
        r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("kid", [ogp.MATERN52, ogp.RBF])
def test_n1024_fit_predict_vs_oracle(kid):
    """Config 5's "n ~ 1k": patch 34 -> n = 32^2 = 1024 (N = 1120, ten pivot blocks: four pair steps + two singles).
    Beyond GPyTorch's max_cholesky_size = 800 the reference itself would leave the Cholesky path, so the target is
    the oracle's dense-Cholesky loop (SURVEY.md section 6 note)."""
    from super_resolution_amd import ops
    s, p = 2, 34
    H, W = 40, 43
    lo = natural_image(H, W, 3 + kid)
    hi = cvops.resize_cubic(lo, fx=s, fy=s)
    locs = [(0, 0), (6, 9)]
    tidx = np.arange(1, p - 1, dtype=np.int32)
    X, y, Xs = ops.gather(to_dev(lo), to_dev(lo), to_dev(hi), s, p, to_dev(np.array(locs, dtype=np.int32)), to_dev(tidx))
    assert X.shape == (2, 1024, 8) and Xs.shape == (2, 66 * 66, 8)
    raw = ops.init_hypers(y, kid)
    alpha, flags = ops.fit(X, y, kid, raw, iters=50, lr=0.1)
    pred = ops.predict(X, Xs, alpha, raw, kid).cpu().numpy()
    assert int(flags.sum()) == 0
    for k, loc in enumerate(locs):
        Xo = opatch.features(opatch.extract(lo, loc, p, 1))
        yo = opatch.targets(opatch.extract(lo, loc, p, 1))
        Xso = opatch.features(opatch.extract(hi, loc, p, s))
        mean, rawo = ogp.fit_predict_patch(Xo, yo, Xso, kid)
        assert np.allclose(raw[k].cpu().numpy(), rawo, rtol=2e-3, atol=2e-3), (raw[k].cpu().numpy(), rawo)
        assert np.abs(pred[k] - mean).max() < 1e-4


def test_config1_set14_x2_rbf_through_main(golden_dir, tmp_path, capsys):
    """BASELINE config 1 as written: `python main.py --srf 2 --kernel rbf --color-space grayscale --images 1` on the
    reference's Set14/image_SRF_2/img_001_SRF_2_LR.png (fixture refpin/x2_lr_001.png): 240 patches, 480 fits of n = 324,
    RBF re-initialised per patch (gprsr.py:167-172: no chain, the warm-start modes coincide).  The written
    img_001_SRF_2_GPR_RBF_gray.png must be the CPU oracle's output (refpin/oracle_x2_rbf_gray_001.png,
    tests/golden/make_config1_golden.py) to one grey level, PSNR / SSIM against the HR image within 0.01 dB / 0.001."""
    import shutil
    from PIL import Image
    import main
    src = os.path.join(golden_dir, "refpin")
    dst = tmp_path / "Set14" / "image_SRF_2"
    dst.mkdir(parents=True)
    shutil.copy(f"{src}/x2_lr_001.png", dst / "img_001_SRF_2_LR.png")
    shutil.copy(f"{src}/x2_hr_001.png", dst / "img_001_SRF_2_HR.png")
    rc = main.main(["--srf", "2", "--kernel", "rbf", "--color-space", "grayscale", "--images", "1",
                    "--dataset-dir", str(tmp_path / "Set14"), "--evaluate"])
    assert rc == 0
    out = capsys.readouterr().out
    assert "Average - PSNR:" in out and "Upscaling took" in out
    got = np.array(Image.open(dst / "img_001_SRF_2_GPR_RBF_gray.png"))
    want = np.array(Image.open(f"{src}/oracle_x2_rbf_gray_001.png"))
    hr = np.array(Image.open(f"{src}/x2_hr_001.png"))
    assert got.shape == want.shape + (3,) and np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])
    d = np.abs(got[..., 0].astype(int) - want.astype(int))
    assert d.max() <= 1 and (d == 0).mean() > 0.999, (d.max(), (d == 0).mean())
    c = 2
    assert abs(cvops.psnr(hr[c:-c, c:-c], got[c:-c, c:-c, 0]) - cvops.psnr(hr[c:-c, c:-c], want[c:-c, c:-c])) <= 0.01
    assert abs(cvops.ssim(hr[c:-c, c:-c], got[c:-c, c:-c, 0]) - cvops.ssim(hr[c:-c, c:-c], want[c:-c, c:-c])) <= 0.001
