"""Dataset-scale comparison with the reference's own committed outputs (tests/golden/refset, a Set14-shaped folder of
the reference's PNG data): runs the HIP path over ALL 14 images of a set with the settings that produced it and prints,
per image, the share of identical pixels, max |diff|, and PSNR / SSIM against the HR image (evaluation.py:83-98: grey,
crop SRF px) for this run and for the reference's PNG.
    python tools/refset_probe.py [set ...]      sets: x3_m32 x4_m32 x2_m52 x3_linear x2_m12 agpr2_x2 agpr2_x3 (+ probes)
Writes gpurun_out/refset_<set>.json.
"""
import json
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from super_resolution_amd import agpr, gprsr, kernels as K, ops  # noqa: E402
from super_resolution_amd.image_patches import PatchInterpolationMode  # noqa: E402

REFSET = os.path.join(ROOT, "tests", "golden", "refset")
gprsr.DO_TIMING = False

# set -> (srf, reference tag, kernel, colour mode, warm start, patch interpolation)
SETS = {
    "x3_m32": (3, "GPR_matern32", K.matern32_kernel, "YUV", "independent", "BILINEAR"),
    "x4_m32": (4, "GPR_matern32", K.matern32_kernel, "YUV", "independent", "BILINEAR"),
    "x2_m52": (2, "GPR_matern52", K.matern52_kernel, "YUV", "independent", "BILINEAR"),
    "x3_linear": (3, "GPR_linear", K.linear_kernel, "YUV", "chain", "AVERAGE"),
    "x2_m12": (2, "GPR_matern12", K.exponential_kernel, "YUV", "chain", "AVERAGE"),
    # probes (no fixture committed unless they reproduce)
    "x2_linear": (2, "GPR_linear", K.linear_kernel, "YUV", "chain", "AVERAGE"),
    "x2_linear_ib": (2, "GPR_linear", K.linear_kernel, "YUV", "independent", "BILINEAR"),
    "x2_m32_ib": (2, "GPR_matern32", K.matern32_kernel, "YUV", "independent", "BILINEAR"),
    "x2_m32_chain": (2, "GPR_matern32", K.matern32_kernel, "YUV", "chain", "AVERAGE"),
    "x2_exp_ib": (2, "GPR_exponential", K.exponential_kernel, "YUV", "independent", "BILINEAR"),
    "x2_exp_chain": (2, "GPR_exponential", K.exponential_kernel, "YUV", "chain", "AVERAGE"),
    "x2_rbf": (2, "GPR_RBF", K.rbf_kernel, "YUV", "independent", "AVERAGE"),
    "x2_rbf_b": (2, "GPR_RBF", K.rbf_kernel, "YUV", "independent", "BILINEAR"),
}


def load_bgr(path):
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])


def metrics_vs_hr(hr_bgr, img_bgr, s):
    dev = torch.device("cuda")
    a = ops.bgr2gray(torch.as_tensor(hr_bgr).to(dev))
    b = ops.bgr2gray(torch.as_tensor(img_bgr).to(dev))
    m = ops.metrics(a, b, s).cpu().numpy()
    return float(m[0]), float(m[1])


def run_set(name, ids=range(1, 15), verbose=True):
    if os.environ.get("REFSET_IDS"):
        ids = [int(v) for v in os.environ["REFSET_IDS"].split(",")]
    rows = []
    t0 = time.time()
    if name.startswith("agpr2_x"):
        s = int(name[-1])
        outs = run_agpr(s, ids)
        tag = "AGPR2"
    else:
        s, tag, kern, cmode, warm, interp = SETS[name]
        old = gprsr.PATCH_INTERPOLATION
        gprsr.PATCH_INTERPOLATION = PatchInterpolationMode[interp]
        kern.reset()
        try:
            g = gprsr.GPRSR(s, kern, gprsr.ColorSpace[cmode], warm_start=warm)
            with patch_grid(os.environ.get("REFSET_GEOM", HISTORIC_GEOMETRY.get(name))):
                outs = [g.apply(load_bgr(f"{REFSET}/image_SRF_{s}/img_{i:03d}_SRF_{s}_LR.png")) for i in ids]
        finally:
            gprsr.PATCH_INTERPOLATION = old
            kern.reset()
    secs = time.time() - t0
    for i, out in zip(ids, outs):
        ref = load_bgr(f"{REFSET}/image_SRF_{s}/img_{i:03d}_SRF_{s}_{tag}.png")
        hr = load_bgr(f"{REFSET}/image_SRF_{s}/img_{i:03d}_SRF_{s}_HR.png")
        d = np.abs(out.astype(int) - ref.astype(int))
        po, so = metrics_vs_hr(hr, out, s)
        pr, sr = metrics_vs_hr(hr, ref, s)
        if d.max() > 3 and os.environ.get("REFSET_SAVE"):   # for diff maps of the images that do not reproduce
            Image.fromarray(np.ascontiguousarray(out[:, :, ::-1])).save(os.path.join(ROOT, "gpurun_out", f"refset_{name}_{i:03d}.png"))
        rows.append({"image": i, "identical": float((d == 0).mean()), "max_abs_diff": int(d.max()),
                     "psnr_out": po, "psnr_ref": pr, "ssim_out": so, "ssim_ref": sr})
        if verbose:
            print(f"  {name} img_{i:03d}: identical {100 * rows[-1]['identical']:8.4f} %  max|d| {d.max():3d}  "
                  f"PSNR out {po:.4f} ref {pr:.4f} (d {po - pr:+.5f})  SSIM d {so - sr:+.6f}", flush=True)
    summ = {"set": name, "seconds": round(secs, 2), "images": len(rows),
            "min_identical": min(r["identical"] for r in rows), "max_abs_diff": max(r["max_abs_diff"] for r in rows),
            "mean_psnr_out": float(np.mean([r["psnr_out"] for r in rows])),
            "mean_psnr_ref": float(np.mean([r["psnr_ref"] for r in rows])),
            "mean_dpsnr": float(np.mean([r["psnr_out"] - r["psnr_ref"] for r in rows])),
            "max_abs_dpsnr": float(np.max(np.abs([r["psnr_out"] - r["psnr_ref"] for r in rows]))),
            "max_abs_dssim": float(np.max(np.abs([r["ssim_out"] - r["ssim_ref"] for r in rows]))), "rows": rows}
    print(f"{name}: {len(rows)} images in {secs:.1f} s; min identical {100 * summ['min_identical']:.4f} %, max |diff| "
          f"{summ['max_abs_diff']}, mean PSNR {summ['mean_psnr_out']:.4f} vs reference {summ['mean_psnr_ref']:.4f} "
          f"(mean d {summ['mean_dpsnr']:+.5f} dB, max |d| {summ['max_abs_dpsnr']:.5f}), max |dSSIM| {summ['max_abs_dssim']:.6f}",
          flush=True)
    return summ


def run_agpr(s, ids):
    import tempfile
    ck = np.load(os.path.join(ROOT, "tests", "golden", f"agpr_BSD100_SRF_{s}.npz"))
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "checkpoints", "BSD100", f"SRF_{s}")
        os.makedirs(d)
        torch.save(torch.as_tensor(ck["train_x"]), f"{d}/train_x.pt")
        torch.save(torch.as_tensor(ck["train_y"]), f"{d}/train_y.pt")
        r = ck["raw"].astype(np.float32)   # the reference's model.pt, as GPyTorch's state_dict lays it out
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
        torch.save(sd, f"{d}/model.pt")
        old = (agpr.BASE_DIR, agpr.SRF, agpr.DATASET, agpr.COLOR_SPACE)
        agpr.BASE_DIR, agpr.SRF, agpr.DATASET, agpr.COLOR_SPACE = tmp, s, agpr.TrainingDataset.BSD100, agpr.ColorSpace.YCRCB
        try:
            model = agpr.AGPRSuperResolution(s, use_existing_model=True)
            outs = []
            for i in ids:
                hr = load_bgr(f"{REFSET}/image_SRF_{s}/img_{i:03d}_SRF_{s}_HR.png")
                outs.append(model.upscale(agpr.downscale(hr)))
        finally:
            agpr.BASE_DIR, agpr.SRF, agpr.DATASET, agpr.COLOR_SPACE = old
    return outs


# The state of the reference that wrote its *_GPR_matern52.png (x2) and *_GPR_matern32.png (x3) sets placed its patches
# like today's image_patches.py:31-41 except that a location equal to the PREVIOUS one was not emitted again: clamping
# can repeat the last column position (consecutive in row-major order: dropped), and the last row position (a whole
# row of locations apart: kept).  Found by probing (REFSET_GEOM = unique | ucols | urows on images 5, 13, 14: "ucols"
# reproduces all 14 images of both sets, today's rule 13 of 14); the x4 set was written with today's rule.
HISTORIC_GEOMETRY = {"x3_m32": "ucols", "x2_m52": "ucols"}


class patch_grid:
    """with patch_grid(mode): the engine's patch grid with duplicate positions dropped on both axes ("unique"), on the
    columns ("ucols") or on the rows ("urows"); None: today's rule (duplicates kept)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from super_resolution_amd import engine
        self.engine, self.orig = engine, engine._geometry
        mode, orig = self.mode, self.orig
        if mode is None:
            return self

        def geometry(H, W, s, p, stride_percentage):
            rows, cols, _ = orig(H, W, s, p, stride_percentage)
            if mode in ("unique", "urows"):
                rows = np.unique(rows).astype(np.int32)
            if mode in ("unique", "ucols"):
                cols = np.unique(cols).astype(np.int32)
            rr, cc = np.meshgrid(rows, cols, indexing="ij")
            return rows, cols, np.stack([rr.reshape(-1), cc.reshape(-1)], axis=1).astype(np.int32)
        engine._geometry = geometry
        return self

    def __exit__(self, *exc):
        self.engine._geometry = self.orig
        return False


if __name__ == "__main__":
    names = sys.argv[1:] or ["x3_m32", "x4_m32", "x2_m52", "x3_linear", "x2_m12", "agpr2_x2", "agpr2_x3"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for nm in names:
        summ = run_set(nm)
        with open(os.path.join(ROOT, "gpurun_out", f"refset_{nm}.json"), "w") as f:
            json.dump(summ, f, indent=1)
