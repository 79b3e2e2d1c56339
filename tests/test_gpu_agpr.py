"""GPU parity tests of the AGPR path (BASELINE config 5) through the reference-shaped
API, against the oracle and against the reference's committed checkpoints."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import agpr as oagpr, gp as ogp
from test_gpu_kernels import natural_image, to_dev


def _write_checkpoint(tmp, golden_dir, name):
    f = np.load(f"{golden_dir}/agpr_{name.replace('/', '_')}.npz")
    folder = os.path.join(tmp, "checkpoints", name)
    os.makedirs(folder)
    torch.save(torch.tensor(f["train_x"]), os.path.join(folder, "train_x.pt"))
    torch.save(torch.tensor(f["train_y"]), os.path.join(folder, "train_y.pt"))
    r = f["raw"].astype(np.float32)
    sd = {"likelihood.noise_covar.raw_noise": torch.tensor([r[0]]),
          "likelihood.noise_covar.raw_noise_constraint.lower_bound": torch.tensor(1e-4),
          "likelihood.noise_covar.raw_noise_constraint.upper_bound": torch.tensor(float("inf")),
          "mean_module.raw_constant": torch.tensor(r[1]),
          "covar_module.raw_outputscale": torch.tensor(r[2]),
          "covar_module.base_kernel.raw_variance": torch.tensor([[r[3]]]),
          "covar_module.base_kernel.raw_variance_constraint.lower_bound": torch.tensor(0.0),
          "covar_module.base_kernel.raw_variance_constraint.upper_bound": torch.tensor(float("inf")),
          "covar_module.raw_outputscale_constraint.lower_bound": torch.tensor(0.0),
          "covar_module.raw_outputscale_constraint.upper_bound": torch.tensor(float("inf"))}
    torch.save(sd, os.path.join(folder, "model.pt"))
    assert sorted(sd) == sorted(str(k) for k in f["state_dict_keys"])   # the reference's key set
    return f


@pytest.mark.parametrize("name,srf", [("BSD100/SRF_2", 2), ("BSD100/SRF_3", 3)])
def test_agpr_fit_reproduces_reference_checkpoint(golden_dir, name, srf):
    """100 Adam steps of the HIP fit on the checkpoint's train_x/train_y must land on the
    reference's model.pt (produced by real GPyTorch): the known-answer test of the fit path."""
    from super_resolution_amd import kernels as K
    f = np.load(f"{golden_dir}/agpr_{name.replace('/', '_')}.npz")
    K.linear_kernel.reset()
    m = K.GeneralModel(K.linear_kernel, f["train_x"], f["train_y"])
    m.fit(iters=100, lr=0.1)
    K.linear_kernel.reset()
    got = m.raw[0].cpu().numpy()
    want = f["raw"]
    assert abs(got[0] - want[0]) < 1e-4 and abs(got[1] - want[1]) < 1e-5
    assert abs(got[2] - want[2]) < 1e-4 and abs(got[3] - want[3]) < 1e-4
    sd = m.state_dict()
    assert sorted(sd) == sorted(str(k) for k in f["state_dict_keys"])
    assert sd["covar_module.base_kernel.raw_variance"].shape == (1, 1)
    assert sd["likelihood.noise_covar.raw_noise"].shape == (1,) and sd["mean_module.raw_constant"].shape == ()


@pytest.mark.parametrize("name,srf", [("BSD100/SRF_2", 2), ("BSD100/SRF_3", 3)])
def test_agpr_upscale_channel_vs_oracle(tmp_path, golden_dir, name, srf):
    from super_resolution_amd import agpr
    f = _write_checkpoint(str(tmp_path), golden_dir, name)
    agpr.BASE_DIR, agpr.SRF, agpr.DATASET = str(tmp_path), srf, agpr.TrainingDataset.BSD100
    a = agpr.AGPRSuperResolution(srf, use_existing_model=True)
    lr = np.rint(natural_image(31, 37, 5) * 255).astype(np.uint8)
    got = a.upscale_channel(lr)
    h = ogp.Hypers(*f["raw"].astype(np.float32))
    want = oagpr.upscale_channel(lr, srf, h, f["train_x"], f["train_y"], global_srf=srf)
    assert got.shape == want.shape == (31 * srf, 37 * srf)
    assert np.abs(got - want).max() < 1e-6 * 255          # float64 path, << 1/255 of a grey level
    # the quirk: downscale() divides by the module-global SRF, not self.srf (agpr.py:246)
    agpr.COLOR_SPACE = agpr.ColorSpace.YCRCB
    bgr = np.stack([lr, np.clip(lr.astype(int) + 10, 0, 255).astype(np.uint8), 255 - lr], -1)
    out = a.upscale(bgr)
    ref = oagpr.upscale(bgr, srf, h, f["train_x"], f["train_y"], global_srf=srf)
    assert out.shape == ref.shape and out.dtype == np.uint8
    assert np.abs(out.astype(int) - ref.astype(int)).max() <= 2
    agpr.BASE_DIR, agpr.SRF = ".", 2


def test_agpr_filter_equals_per_pixel_gp_mean(golden_dir):
    """The 49-tap filter IS the GP posterior mean of every 7x7 window (agpr.py:176)."""
    from super_resolution_amd import kernels as K, ops
    f = np.load(f"{golden_dir}/agpr_BSD100_SRF_2.npz")
    m = K.GeneralModel(K.linear_kernel, f["train_x"], f["train_y"])
    m.load_state_dict({"likelihood.noise_covar.raw_noise": torch.tensor([f["raw"][0]]),
                       "mean_module.raw_constant": torch.tensor(f["raw"][1]),
                       "covar_module.raw_outputscale": torch.tensor(f["raw"][2]),
                       "covar_module.base_kernel.raw_variance": torch.tensor([[f["raw"][3]]])})
    alpha = m.ensure_alpha()
    w = ops.agpr_weights(m.train_x, alpha[0], m.raw[0].contiguous())
    img = natural_image(20, 23, 1)
    out = ops.agpr_filter(to_dev(img), 7, w, m.raw[0].contiguous()).cpu().numpy()
    win = np.lib.stride_tricks.sliding_window_view(img, (7, 7)).reshape(-1, 49)
    h = ogp.Hypers(*f["raw"].astype(np.float32))
    mean, _ = ogp.predictive_mean(h, torch.tensor(f["train_x"]), torch.tensor(f["train_y"]),
                                  torch.tensor(np.ascontiguousarray(win)).float().double(), ogp.LINEAR)
    assert np.abs(out[3:-3, 3:-3].reshape(-1) - mean.numpy()).max() < 1e-8
    assert np.array_equal(out[:3], img[:3]) and np.array_equal(out[:, -3:], img[:, -3:])


def _candidate_rows(N, seed, D=50):
    """Candidate rows shaped like agpr.py:113-118: 7x7 windows of a smooth image + one more pixel, np.unique'd."""
    rng = np.random.default_rng(seed)
    img = natural_image(120, 130, seed)
    rows = []
    for _ in range(N):
        y, x = rng.integers(3, 117), rng.integers(3, 127)
        rows.append(np.concatenate((img[y - 3:y + 4, x - 3:x + 4].reshape(-1)[:D - 1], [img[y, x] + 0.01 * rng.random()])))
    return np.unique(np.array(rows), axis=0)


@pytest.mark.parametrize("N,K,n_pick,seed", [(300, 10, 30, 0), (257, 3, 257, 1), (640, 10, 64, 2)])
def test_active_sampling_vs_oracle(N, K, n_pick, seed):
    """gpsr_active_sample against the oracle's line-by-line restatement of agpr.py:120-137 (which runs the
    reference's own numpy calls): the SAME rows in the SAME order, scores to a few ulp (device exp vs numpy exp)."""
    from super_resolution_amd import ops
    data = _candidate_rows(N, seed)
    n = len(data)
    n_pick = min(n_pick, n)
    want_idx, want_rows, want_rep, bw = oagpr.active_sampling(data, K=K, n_pick=n_pick)
    assert bw == 0.2 * (n - 1) / 2          # the median of np.where's index arrays, whatever the data
    idx, rep = ops.active_sample(to_dev(data), K, n_pick, bw, 0.2, 1 - 0.2)
    idx, rep = idx.cpu().numpy(), rep.cpu().numpy()
    assert np.abs(rep / want_rep - 1).max() < 1e-14
    assert np.array_equal(idx, want_idx)
    assert np.array_equal(data[idx], want_rows)


def test_active_sampling_full_size_and_drop_in():
    """N = 5000 candidates, 500 picks (the reference's constants): the drop-in's _active_sampling against a
    vectorised numpy running-minimum form of the same loop (distances from numpy's own broadcast sum, in row
    blocks), plus the properties the loop guarantees."""
    from super_resolution_amd import agpr
    data = _candidate_rows(5000, 7)
    n = len(data)
    a = agpr.AGPRSuperResolution.__new__(agpr.AGPRSuperResolution)
    chosen = a._active_sampling(data)
    idx = a.active_indices
    assert len(chosen) == 500 and len(set(idx.tolist())) == 500 and a.bandwidth == 0.2 * (n - 1) / 2
    dist = np.empty((n, n))
    for r0 in range(0, n, 100):
        dist[r0:r0 + 100] = np.sum((data[r0:r0 + 100, None, :] - data[None, :, :]) ** 2, axis=-1)
    np.fill_diagonal(dist, np.inf)
    near = np.sort(dist, axis=1)[:, :agpr.K]
    rep = np.mean(np.exp(-near / (2 * a.bandwidth)), axis=-1)
    assert np.abs(a.representativeness_scores / rep - 1).max() < 1e-14
    pool = np.arange(n)
    div = np.zeros(n)
    for it in range(500):
        j = int(np.argmax(0.2 * rep[:len(pool)] + (1 - 0.2) * div))
        assert idx[it] == pool[j], f"pick {it}"
        cand = -np.exp(-dist[pool[j]] / (2 * a.bandwidth))
        pool = np.delete(pool, j)
        div = np.delete(div, j)
        div = cand[pool] if it == 0 else np.minimum(div, cand[pool])
    assert np.array_equal(np.array(chosen), data[idx])


def test_active_sampling_rejects_bad_sizes():
    from super_resolution_amd import _lib, ops
    data = to_dev(_candidate_rows(50, 3))
    with pytest.raises(_lib.GpsrError):
        ops.active_sample(data, 64, 5, 1.0, 0.2, 0.8)         # K >= N
    with pytest.raises(_lib.GpsrError):
        ops.active_sample(data, 5, len(data) + 1, 1.0, 0.2, 0.8)
    with pytest.raises(_lib.GpsrError):
        ops.active_sample(data, 5, 5, 0.0, 0.2, 0.8)          # bandwidth


@pytest.mark.parametrize("s,i,min_ident", [(2, 6, 0.99999), (2, 4, 0.9999), (3, 6, 0.999)])
def test_agpr_drop_in_reproduces_the_references_outputs_from_its_checkpoints(tmp_path, golden_dir, s, i, min_ident):
    """HIP path vs a REFERENCE golden: agpr.py:284-289 through the drop-in (downscale of the uint8 BGR HR image on the GPU,
    AGPRSuperResolution(use_existing_model=True) on the reference's committed checkpoint, upscale in YCrCb mode)
    against the *_AGPR2.png the reference committed.  Same bars as the oracle's test (tests/test_refpin.py)."""
    from PIL import Image
    from super_resolution_amd import agpr
    _write_checkpoint(str(tmp_path), golden_dir, f"BSD100/SRF_{s}")
    old = (agpr.BASE_DIR, agpr.SRF, agpr.DATASET, agpr.COLOR_SPACE)
    agpr.BASE_DIR, agpr.SRF, agpr.DATASET, agpr.COLOR_SPACE = str(tmp_path), s, agpr.TrainingDataset.BSD100, agpr.ColorSpace.YCRCB
    try:
        load = lambda n: np.ascontiguousarray(np.array(Image.open(os.path.join(golden_dir, "refpin", n)))[..., ::-1])
        hr, ref = load(f"agpr_x{s}_hr_{i:03d}.png"), load(f"agpr_x{s}_ref_{i:03d}.png")
        lr = agpr.downscale(hr)
        assert np.array_equal(lr, oagpr.downscale(hr, s))             # uint8 blur + uint8 bicubic: bit for bit
        out = agpr.AGPRSuperResolution(s, use_existing_model=True).upscale(lr)
    finally:
        agpr.BASE_DIR, agpr.SRF, agpr.DATASET, agpr.COLOR_SPACE = old
    d = np.abs(out.astype(int) - ref.astype(int))
    ident = float((d == 0).mean())
    print(f"AGPR x{s} img_{i:03d} vs the reference's AGPR2 PNG: {100 * ident:.4f} % identical, max |diff| {d.max()}")
    assert out.shape == ref.shape and ident >= min_ident and d.max() <= 3


def test_sepfilter_u8_bit_exact():
    from super_resolution_amd import imgproc
    rng = np.random.default_rng(3)
    for (H, W), ks, sg in (((37, 53), 3, 1), ((5, 4), 5, 0), ((64, 31), 7, 1.5), ((1, 9), 3, 1)):
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        got = imgproc.gaussian_blur_u8_dev(torch.as_tensor(img).cuda(), ks, sg).cpu().numpy()
        from oracle import cvops
        assert np.array_equal(got, cvops.gaussian_blur_u8(img, ks, sg)), (H, W, ks)
