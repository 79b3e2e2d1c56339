#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched SRGPR engine.

Metric (BASELINE.json): output megapixels / second of SR x3 with the Matern-5/2
kernel.  Workload at N=1 = BASELINE.json configs[1]: the full Set14 x3 set
(14 LR images of the Set14 image_SRF_3 shapes, GRAYSCALE, 633 patches, 1266 GP
fits of n=784, m=7744 each, 3.2066 output Mpx) -- synthetic images of exactly
those shapes (the reference's PNGs do not travel to the GPU box).

A "step" = one pass of GPRSR.apply_for_channel over the whole batch (both
phases: gather -> 50-step Adam fit -> predictive mean -> recombination, plus
the bicubic / blur / decimation stages between them), inputs resident in HBM as
float64 images when the clock starts, uint8 HR images in HBM when it stops.

N > 1 (weak scaling, default): the batch is N copies of the workload (different seeds);
the global patch list is sharded over the ranks (super_resolution_amd/dist.py)
and the predicted patches are all-gathered over RCCL; value = all output
megapixels / max-over-ranks time.  --scaling strong shards ONE copy instead
(BASELINE config 4: --workload 4kx2 --scaling strong at 1/2/4/8 GPUs).

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline     : the dominant kernel (k_update2, f64 MFMA tile products), HIP-event timed
  cpu_baseline : the CPU oracle (torch f64 restatement of the GPyTorch loop) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# LR shapes (H, W) of Set14/image_SRF_3/img_001..014_SRF_3_LR.png
SET14_X3_LR_SHAPES = [(160, 166), (192, 240), (170, 170), (96, 117), (120, 83), (92, 92), (120, 166),
                      (96, 117), (170, 170), (170, 170), (170, 256), (170, 170), (218, 176), (130, 195)]
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix peak (AMD spec; the microarch guide lists no f64 row)
NB = 112


def synthetic_lr(H, W, seed):
    """Natural-like uint8 image (SURVEY.md 8(d) generator): smooth field + edges + noise."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    img = gaussian_filter(rng.random((H, W)), 3.0)
    img = (img - img.min()) / (img.max() - img.min()) * 219 + 16
    img += rng.normal(0, 2, (H, W))
    yy, xx = np.mgrid[0:H, 0:W]
    for _ in range(8):
        a, b, c = rng.normal(size=3)
        img += rng.uniform(-30, 30) * ((a * (yy - H / 2) + b * (xx - W / 2) + 0.2 * c * max(H, W)) > 0)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def workload(name, copies):
    if name == "set14x3":
        shapes, s = SET14_X3_LR_SHAPES, 3
    elif name == "4kx2":
        shapes, s = [(2160, 3840)], 2
    elif name == "tiny":
        shapes, s = [(47, 39), (40, 52)], 3
    else:
        raise SystemExit(f"unknown workload {name}")
    imgs = []
    for c in range(copies):
        for i, (H, W) in enumerate(shapes):
            imgs.append(synthetic_lr(H, W, 1234 + 100 * c + i))
    return imgs, s


def patch_count(shape, s):
    p = 10 * s
    stride = int((p - 2) * 0.9)
    return len(range(0, shape[0], stride)) * len(range(0, shape[1], stride))


def cpu_baseline(kernel_id, s, seconds_budget=20.0):
    """Oracle (CPU restatement of the reference's per-patch GPyTorch loop) timed on a
    bounded sample of the same workload: whole patch fits (n=(10s-2)^2, m=(10s^2-2)^2)."""
    from oracle import gp as ogp, patches as opatch, cvops
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    p = 10 * s
    lr = synthetic_lr(4 * p, 4 * p, 77).astype(np.double) / 255.   # 25+ patches: the time budget, not the image, ends the sample
    hi = cvops.resize_cubic(lr, fx=s, fy=s)
    locs = opatch.patch_locations(lr.shape, p)
    t0, done = time.time(), 0
    while done < len(locs) and (done == 0 or (time.time() - t0) * (done + 1) / done < seconds_budget):
        loc = locs[done]
        X = opatch.features(opatch.extract(lr, loc, p, 1))
        y = opatch.targets(opatch.extract(lr, loc, p, 1))
        Xs = opatch.features(opatch.extract(hi, loc, p, s))
        ogp.fit_predict_patch(X, y, Xs, kernel_id)
        done += 1
    dt = time.time() - t0
    return dt / done, done, cores


def pmc_traffic(kernel, workload, B_local, N):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950
    correction of MI355X_MICROARCH.md, + WRITE_SIZE), which were taken on 633 GPs of order 784 = this bench's
    default shape.  Counters cannot be read from inside this process: None for any other shape or if the
    summary is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_invert_633x784.json")
    if workload != "set14x3" or B_local != 633 or N != 784 or not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            return round(json.load(f)["hbm_bytes_per_launch"][kernel]["traffic_corrected"])
    except (KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="set14x3")
    ap.add_argument("--kernel", default="matern52")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N > 1: weak = N copies of the workload (default, the driver's scaling run); strong = ONE copy "
                         "whose patches are sharded over the N ranks (BASELINE config 4: one 4K image at 1/2/4/8 GPUs)")
    ap.add_argument("--max-workspace-gb", type=float, default=48.0)
    ap.add_argument("--dual-stream", action="store_true",
                    help="gpsr_set_dual_stream(1): two half-batches on two streams (+5 %% throughput; per-launch "
                         "durations then overlap, so the roofline object describes launches that share the GPU)")
    a = ap.parse_args()

    from super_resolution_amd import dist as gdist
    rank, world, local = gdist.init_from_env()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from super_resolution_amd import kernels as K, ops
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    kern = {"matern52": K.matern52_kernel, "matern32": K.matern32_kernel, "rbf": K.rbf_kernel,
            "exponential": K.exponential_kernel}[a.kernel]
    if a.workload == "4kx2" and a.kernel == "matern52":
        kern = K.matern32_kernel  # BASELINE.json config 4
    ops.set_dual_stream(a.dual_stream)
    imgs_u8, s = workload(a.workload, world if a.scaling == "weak" else 1)
    out_px = sum(im.shape[0] * s * im.shape[1] * s for im in imgs_u8)
    n_patches = sum(patch_count(im.shape, s) for im in imgs_u8)
    g = GPRSR(s, kern, ColorSpace.GRAYSCALE, warm_start="independent", shard=(rank, world) if world > 1 else None,
              max_workspace_bytes=int(a.max_workspace_gb * (1 << 30)))
    imgs = [ops.u8_to_unit(torch.as_tensor(im).to(dev)) for im in imgs_u8]

    def step():
        outs = g.apply_for_channel_device(imgs)
        return [ops.unit_to_u8(o, 255.0) for o in outs]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    if rank == 0:
        ops.profile_begin(1 << 16)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        outs = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = ops.profile_end() if rank == 0 else None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64,
                         device=dev if torch.distributed.get_backend() == "nccl" else "cpu")   # gloo: one-GPU rehearsal
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    flags_bad = int(sum(int(f.sum()) for f in g.stats.flags))

    if rank != 0:
        return
    ms_per_step = dt / a.steps * 1e3
    value = out_px / 1e6 / (dt / a.steps)

    # roofline of the dominant kernel.  Pivots are taken two at a time: k_update2 applies both rank-112 updates of
    # a pair in one pass over A (batched 112x112x224 f64 MFMA tile products).  Algorithmic flops of one launch, per
    # GP, with R = N - 2*112 rows outside the pair: symmetric rank-224 update of the R x R trailing matrix R^2*224,
    # pivot column of the first step + its update by the second 3*R*112^2, the first pivot block's symmetric update
    # 112^3, the triangular pivot-column product of the second step (N-112)*112^2.  (Matrices of fewer than 3
    # blocks have no pair: then k_update, (N-112)^2*112 + (N-112)*112^2.)
    n = (10 * s - 2) ** 2
    N = -(-n // NB) * NB
    B_local = -(-n_patches // world)
    # A launch covers the whole local batch unless the engine cut it into workspace-sized chunks (4kx2) or the
    # dual-stream scheduler into halves: GPs per launch = local batch x launches a whole-batch run would make / launches
    # made.  Whole-batch launches per step: 2 phases x 51 evaluations (50 Adam steps + the final one) x passes per sweep.
    nblk = N // NB
    if prof.get("update2", (0, 0.0))[0]:
        kname_dom, (cnt, ms) = "k_update2", prof["update2"]
        R = N - 2 * NB
        flops_gp = R * R * 2 * NB + 3 * R * NB * NB + NB ** 3 + (N - NB) * NB * NB
        passes = (nblk - 1) // 2 if nblk % 2 else (nblk - 2) // 2
    else:
        kname_dom, (cnt, ms) = "k_update", prof["update"]
        flops_gp = (N - NB) ** 2 * NB + (N - NB) * NB * NB
        passes = nblk
    gps_per_launch = B_local * (2 * (g.iters + 1) * passes * a.steps) / cnt if cnt else 0.0
    flops_launch = int(gps_per_launch * flops_gp)
    avg_s = (ms / cnt) * 1e-3 if cnt else float("nan")
    achieved = flops_launch / avg_s / 1e12 if cnt else float("nan")
    kernel_ms = {k: round(v[1] / a.steps, 3) for k, v in prof.items()}
    roofline = {"kernel": kname_dom, "bound": "mfma", "achieved": round(achieved, 3),
                "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
                "traffic": pmc_traffic(kname_dom, a.workload, int(round(gps_per_launch)), N),
                "traffic_note": "HBM bytes per launch, profiles/r01_pmc_invert_633x784.json (separate rocprofv3 --pmc passes)",
                "launches_per_step": cnt // max(1, a.steps), "gps_per_launch": round(gps_per_launch, 1),
                "avg_launch_ms": round(avg_s * 1e3, 4),
                "flops_per_launch": flops_launch, "kernel_ms_per_step": kernel_ms}

    cpu = None
    if not a.no_cpu_baseline and world == 1:   # reported at N=1 only (rank 0 would stall the other ranks' exit)
        sec_per_fit, nfit, cores = cpu_baseline(g._gp.kernel_id, s)
        copies = world if a.scaling == "weak" else 1
        fits = 2 * n_patches // copies  # per copy of the workload
        cpu_val = (out_px / copies) / 1e6 / (sec_per_fit * fits)
        cpu = {"value": round(cpu_val, 6), "unit": "output Mpx/s", "cores": cores, "kind": "port",
               "sample": f"{nfit} whole patch fits (n={n}, 50 Adam steps + predictive mean, m={(10 * s * s - 2) ** 2}) "
                         f"of the oracle's torch-f64 per-patch loop, {sec_per_fit:.2f} s/fit, extrapolated "
                         f"linearly to the workload's {fits} fits (patch gather/recombination not counted)"}

    kname = "matern32" if (a.workload == "4kx2" and a.kernel == "matern52") else a.kernel
    line = {"metric": "output megapixels/sec (SR x3, Matern-5/2)" if (a.workload == "set14x3" and kname == "matern52")
            else f"output megapixels/sec ({a.workload}, {kname})",
            "value": round(value, 4), "unit": "Mpx/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {len(imgs_u8)} images, {n_patches} patches, "
                                   f"{2 * n_patches} GP fits (n={n}), {out_px / 1e6:.4f} output Mpx",
                       "kernel": kname,
                       "color": "GRAYSCALE", "warm_start": "independent", "dual_stream": bool(a.dual_stream),
                       "parallelism": f"patch-sharded x{world}"},
            "nonpositive_pivot_flags": flags_bad,
            "roofline": roofline, "cpu_baseline": cpu}
    if cpu:
        line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
