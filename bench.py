#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched SRGPR engine.

Metric (BASELINE.json): output megapixels / second of SR x3 with the Matern-5/2 kernel (+ PSNR parity).
Workload at N=1 = BASELINE.json configs[1]: the full Set14 x3 set -- the 14 LR images of
Set14/image_SRF_3 as grey PNG fixtures (tests/golden/set14_x3/lr_*.png; the reference's files do not travel to
the GPU box), GRAYSCALE, 633 patches, 1266 GP fits of n=784, m=7744 each, 3.2066 output Mpx.

A "step" = one pass of GPRSR.apply_for_channel over the whole batch (both phases: gather -> 50-step Adam fit
-> predictive mean -> recombination, plus the bicubic / blur / decimation stages between them), inputs resident
in HBM as float64 images when the clock starts, uint8 HR images in HBM when it stops.

--warm-start independent (default: the north star's batched engine, every patch from the initial values),
chain_per_image (the reference's warm-start chain, every image as the first image of a fresh reference process;
the 14 chains advance together) or chain (the literal reference process: one chain through all images, serial).
The `parity` object compares this run's outputs with the committed CPU-oracle outputs OF THE SAME MODE
(tests/golden/set14_x3/oracle_m52_*.png / oracle_m52_chain_*.png) with the HIP metric kernels.

--gpus N > 1: bench.py starts its N ranks itself (one process per GPU, RCCL), unless it already runs under
torch.distributed.run (RANK set).  weak scaling (default for set14x3): N copies of the workload; strong (default
for 4kx2 = BASELINE config 4): one copy sharded.  value = all output megapixels / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with
  roofline     : the dominant kernel (k_update2, f64 MFMA tile products), HIP-event timed, + whole_step_frac
  cpu_baseline : the CPU oracle (torch f64 restatement of the GPyTorch loop) on a bounded sample
  parity       : max |diff| in grey levels, dPSNR / dSSIM against the committed goldens of the mode
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# LR shapes (H, W) of Set14/image_SRF_3/img_001..014_SRF_3_LR.png
SET14_X3_LR_SHAPES = [(160, 166), (192, 240), (170, 170), (96, 117), (120, 83), (92, 92), (120, 166),
                      (96, 117), (170, 170), (170, 170), (170, 256), (170, 170), (218, 176), (130, 195)]
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X FP64 matrix peak (AMD spec; the microarch guide lists no f64 row)
NB = 112
GOLDEN = os.path.join(ROOT, "tests", "golden", "set14_x3")


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


def _png(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name)))


def workload(name, copies, synthetic=False):
    """-> (list of uint8 LR images, scale factor, data description, LR patch edge or None = the reference's 10 * s)."""
    p = None
    if name == "set14x3":
        s = 3
        real = (not synthetic) and all(os.path.exists(os.path.join(GOLDEN, f"lr_{i:03d}.png")) for i in range(1, 15))
        if real:
            base = [_png(f"lr_{i:03d}.png") for i in range(1, 15)]
            assert [b.shape for b in base] == SET14_X3_LR_SHAPES
            return base * copies, s, "Set14 x3 LR images (grey PNG fixtures tests/golden/set14_x3/lr_*.png)", p
        shapes = SET14_X3_LR_SHAPES
    elif name == "4kx2":                      # BASELINE config 4
        shapes, s = [(2160, 3840)], 2
    elif name == "bsd100x3":                  # BASELINE config 3: BSD100 is not in the container:
        shapes, s = [(107, 160)] * 100, 3     # 100 generated images of BSD100's LR shape (HR 321 x 481)
    elif name == "bsd100x2":
        shapes, s = [(160, 240)] * 100, 2
    elif name == "p34":                       # BASELINE config 5, the n ~ 1k case: p = 34 -> n = 1024 (perf only: above
        shapes, s, p = [(700, 700)], 3, 34    # GPyTorch's max_cholesky_size = 800 the reference has no deterministic result)
    elif name == "tiny":
        shapes, s = [(47, 39), (40, 52)], 3
    else:
        raise SystemExit(f"unknown workload {name}")
    imgs = []
    for c in range(copies):
        for i, (H, W) in enumerate(shapes):
            imgs.append(synthetic_lr(H, W, 1234 + 100 * c + i))
    return imgs, s, "synthetic", p


def patch_count(shape, s, p=None):
    p = p or 10 * s
    stride = int((p - 2) * 0.9)
    return len(range(0, shape[0], stride)) * len(range(0, shape[1], stride))


def cpu_baseline(kernel_id, s, cores, seconds_budget=12.0, p=None):
    """Oracle (CPU restatement of the reference's per-patch GPyTorch loop) timed on a
    bounded sample of the same workload: whole patch fits (n=(p-2)^2, m=(p s-2)^2) on `cores` threads."""
    import torch
    from oracle import gp as ogp, patches as opatch, cvops
    torch.set_num_threads(cores)
    p = p or 10 * s
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
    return dt / done, done


def pmc_traffic(kernel, B_local, N):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950
    correction of MI355X_MICROARCH.md, + WRITE_SIZE), taken on B_local GPs of order N.  Counters cannot be
    read from inside this process: None for a shape without a committed summary."""
    for rnd in ("r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_invert_{B_local}x{N}.json")
        if os.path.exists(path):
            try:
                with open(path) as f:
                    return round(json.load(f)["hbm_bytes_per_launch"][kernel]["traffic_corrected"]), os.path.relpath(path, ROOT)
            except (KeyError, ValueError):
                pass
    return None, None


def fit_flops(n, m, iters=50):
    """algorithmic flops of one patch fit (potrf + inverse for the trace + kernel/derivative
    evaluations and solves per Adam iteration, final factorisation + predictive mean)."""
    f_iter = n ** 3 / 3 + 2 * n ** 3 / 3 + 38 * n * n
    return iters * f_iter + n ** 3 / 3 + 38 * m * n


# ------------------------------------------------------------------ rank launcher
def launch_ranks(n, argv, script=None, extra_env=None, timeout=None):
    """Start n ranks of `script` (default: this file) as child processes, one per GPU: RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, backend chosen by the child (nccl = RCCL on GPUs).
    The parent touches no GPU; it relays rank 0's stdout (the JSON line) and returns non-zero if any rank fails
    (the others are terminated).  -> (returncode, rank-0 stdout)."""
    script = script or os.path.abspath(__file__)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, "-u", script] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    t0 = time.time()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write(f"bench.py: rank {r} exited with {code}; stopping the other ranks\n")
                for q in alive:
                    procs[q].terminate()
        if timeout is not None and time.time() - t0 > timeout and alive:
            rc = rc or 124
            for q in alive:
                procs[q].kill()
        time.sleep(0.05)
    out = procs[0].stdout.read() if procs[0].stdout else ""
    return rc, out


def bench_agpr(a):
    """BASELINE config 5, the part that exists in the reference: AGPR (agpr.py:148-209) -- ONE global linear-kernel GP,
    here the reference's committed BSD100 x2 checkpoint -- upscaling the 14 Set14 images (LR = agpr.py's downscale of
    the HR fixtures, tests/golden/refset).  A step = AGPRSuperResolution.upscale over the 14 images (YCrCb: bicubic, the
    49-tap GP mean, iterative back-projection, uint8 chroma), uint8 BGR images resident in HBM when the clock starts and
    stops.  The stages are HBM-bound stencils: `roofline` is k_agpr_filter (the GP mean of every HR pixel) timed alone
    with HIP events; `stages` lists the back-projection kernels the same way."""
    import tempfile
    import torch
    from PIL import Image
    from super_resolution_amd import agpr, imgproc, ops
    if a.gpus != 1:
        raise SystemExit("--workload agpr is a one-GPU line (one global GP, 14 small images)")
    s = 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    refset = os.path.join(ROOT, "tests", "golden", "refset", f"image_SRF_{s}")
    ck = np.load(os.path.join(ROOT, "tests", "golden", f"agpr_BSD100_SRF_{s}.npz"))
    tmp = tempfile.mkdtemp()
    d = os.path.join(tmp, "checkpoints", "BSD100", f"SRF_{s}")
    os.makedirs(d)
    torch.save(torch.as_tensor(ck["train_x"]), f"{d}/train_x.pt")
    torch.save(torch.as_tensor(ck["train_y"]), f"{d}/train_y.pt")
    r = ck["raw"].astype(np.float32)
    torch.save({"likelihood.noise_covar.raw_noise": torch.tensor([r[0]]), "mean_module.raw_constant": torch.tensor(r[1]),
                "covar_module.raw_outputscale": torch.tensor(r[2]),
                "covar_module.base_kernel.raw_variance": torch.tensor([[r[3]]])}, f"{d}/model.pt")
    agpr.BASE_DIR, agpr.SRF, agpr.DATASET, agpr.COLOR_SPACE = tmp, s, agpr.TrainingDataset.BSD100, agpr.ColorSpace.YCRCB
    model = agpr.AGPRSuperResolution(s, use_existing_model=True)
    hrs = [np.ascontiguousarray(np.array(Image.open(f"{refset}/img_{i:03d}_SRF_{s}_HR.png").convert("RGB"))[:, :, ::-1])
           for i in range(1, 15)]
    lrs = [torch.as_tensor(agpr.downscale(h)).to(dev) for h in hrs]
    out_px = sum(int(l.shape[0]) * s * int(l.shape[1]) * s for l in lrs)

    def step():
        return [model.upscale_device(l) for l in lrs]
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        outs = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    # parity against the reference's own *_AGPR2.png of the same images
    ident, maxd = [], 0
    for i, o in zip(range(1, 15), outs):
        ref = np.ascontiguousarray(np.array(Image.open(f"{refset}/img_{i:03d}_SRF_{s}_AGPR2.png").convert("RGB"))[:, :, ::-1])
        dd = np.abs(o.cpu().numpy().astype(int) - ref.astype(int))
        ident.append(float((dd == 0).mean())); maxd = max(maxd, int(dd.max()))
    # the stage kernels alone, on the largest image's HR plane: algorithmic bytes (read + write of the float64 plane) / time
    big = max(lrs, key=lambda l: l.numel())
    H, W = int(big.shape[0]) * s, int(big.shape[1]) * s
    plane = torch.rand((H, W), dtype=torch.float64, device=dev)
    w, raw0 = model.filter_weights(), model.model.raw[0].contiguous()
    kern3 = [[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]]

    def timed(fn, reps=50):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    byts = 2 * H * W * 8
    stages = {}
    for name, fn in (("k_agpr_filter (49-tap GP mean)", lambda: ops.agpr_filter(plane, agpr.L, w, raw0)),
                     ("k_sepfilter_sym (GaussianBlur 3x3)", lambda: imgproc.gaussian_blur(plane, 3, 1)),
                     ("k_filter3x3 (sharpen)", lambda: imgproc.filter2d_3x3(plane, kern3)),
                     ("k_axpy", lambda: ops.axpy(plane, plane, 0.01))):
        t = timed(fn)
        stages[name] = {"us": round(t * 1e6, 2), "GB/s": round(byts / t / 1e9, 1)}
    tf = stages["k_agpr_filter (49-tap GP mean)"]
    line = {"metric": "output megapixels/sec (AGPR x2, Set14, the reference's BSD100 checkpoint)", "value": round(out_px / 1e6 / dt, 3),
            "unit": "Mpx/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "Set14 x2 HR fixtures (tests/golden/refset), LR = agpr.py:244-246 downscale",
            "config": {"workload": f"agpr: 14 images, {out_px / 1e6:.4f} output Mpx, 1 global GP (n=500, d=49) from checkpoint",
                       "color": "YCrCb"},
            "parity": {"golden": "the reference's Set14/image_SRF_2/*_AGPR2.png", "min_identical_pixel_frac": round(min(ident), 6),
                       "max_abs_diff": maxd},
            "roofline": {"kernel": "k_agpr_filter", "bound": "hbm", "achieved": tf["GB/s"], "peak": 8000.0, "unit": "GB/s",
                         "frac": round(tf["GB/s"] / 8000.0, 4), "traffic": None,
                         "note": f"one {H}x{W} float64 plane read + written ({byts / 1e6:.1f} MB) per launch of {tf['us']} us: at "
                                 f"Set14 sizes the launch is latency-, not bandwidth-bound",
                         "stages": stages},
            "cpu_baseline": None}
    print(json.dumps(line), flush=True)
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="set14x3",
                    help="set14x3 (headline, BASELINE config 2) | 4kx2 (config 4) | bsd100x3 | bsd100x2 (config 3: 100 generated "
                         "images of BSD100's LR shape) | p34 (config 5's n ~ 1k case: p = 34, n = 1024, perf only) | agpr "
                         "(config 5: AGPR x2 inference on the Set14 fixtures from the reference's BSD100 checkpoint) | tiny")
    ap.add_argument("--kernel", default="matern52")
    ap.add_argument("--warm-start", default="independent", choices=("independent", "chain_per_image", "chain"))
    ap.add_argument("--synthetic", action="store_true", help="synthetic images of the Set14 shapes instead of the fixtures")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-all", action="store_true",
                    help="bracket every tagged launch inside the timed region (default: only the dominant kernel; the other "
                         "kernels' milliseconds then come from one extra untimed step)")
    ap.add_argument("--no-kernel-breakdown", action="store_true", help="skip that extra untimed step")
    ap.add_argument("--no-reference-check", action="store_true",
                    help="skip the untimed runs against the reference's own PNGs (parity.*vs_reference_png): under rocprofv3 "
                         "their launches would enter the per-kernel averages of the profile")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="N > 1: weak = N copies of the workload (default for set14x3); strong = ONE copy whose patches "
                         "are sharded over the N ranks (default for 4kx2 = BASELINE config 4: one 4K image at 1/2/4/8 GPUs)")
    ap.add_argument("--max-workspace-gb", type=float, default=48.0)
    ap.add_argument("--no-split-small", action="store_true",
                    help="gpsr_set_split_small(0): every launch takes the whole batch on one stream (the default issues the "
                         "runs of small launches -- pivot / panel / restricted update -- as two halves on two streams)")
    ap.add_argument("--dual-stream", action="store_true",
                    help="gpsr_set_dual_stream(1): two half-batches on two streams (+5 %% throughput; per-launch "
                         "durations then overlap, so the roofline object describes launches that share the GPU)")
    a = ap.parse_args(argv)
    if a.scaling is None:
        a.scaling = "strong" if a.workload == "4kx2" else "weak"
    if a.workload == "agpr":
        return bench_agpr(a)

    import torch
    if a.gpus > 1 and "RANK" not in os.environ:
        # launcher: start the ranks before anything touches a GPU (device_count does not initialise it)
        have = torch.cuda.device_count()
        if have < a.gpus and "GPSR_LOCAL_DEVICE" not in os.environ:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} GPU(s) visible -- refusing to report a "
                             f"{a.gpus}-GPU number from fewer devices")
        rc, out = launch_ranks(a.gpus, argv)
        sys.stdout.write(out)
        sys.stdout.flush()
        if rc == 0 and '"n_gpus": %d' % a.gpus not in out:
            sys.stderr.write("bench.py: rank 0 printed no JSON line for the requested rank count\n")
            rc = 1
        return rc

    from super_resolution_amd import dist as gdist
    rank, world, local = gdist.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from super_resolution_amd import kernels as K, ops
    from super_resolution_amd.gprsr import GPRSR, ColorSpace
    kname = "matern32" if (a.workload == "4kx2" and a.kernel == "matern52") else a.kernel   # BASELINE.json config 4
    kern = {"matern52": K.matern52_kernel, "matern32": K.matern32_kernel, "rbf": K.rbf_kernel,
            "exponential": K.exponential_kernel}[kname]
    ops.set_dual_stream(a.dual_stream)
    ops.set_split_small(not a.no_split_small)
    copies = world if a.scaling == "weak" else 1
    imgs_u8, s, data_desc, p_edge = workload(a.workload, copies, a.synthetic)
    n_base = len(imgs_u8) // copies
    out_px = sum(im.shape[0] * s * im.shape[1] * s for im in imgs_u8)
    n_patches = sum(patch_count(im.shape, s, p_edge) for im in imgs_u8)
    # weak scaling = N copies of the data set on N ranks: the HR images stay with the rank that fitted them (no rank
    # repeats the image stages / recombination of another rank's images); strong scaling: every rank holds the result
    g = GPRSR(s, kern, ColorSpace.GRAYSCALE, warm_start=a.warm_start, shard=(rank, world) if world > 1 else None,
              max_workspace_bytes=int(a.max_workspace_gb * (1 << 30)),
              outputs="owned" if (world > 1 and a.scaling == "weak") else "all", patch_size=p_edge)
    imgs = [ops.u8_to_unit(torch.as_tensor(im).to(dev)) for im in imgs_u8]

    def step():
        if a.warm_start == "chain":
            kern.reset()     # every step = one fresh reference process over the image list
        outs = g.apply_for_channel_device(imgs)
        return [None if o is None else ops.unit_to_u8(o, 255.0) for o in outs]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    gdist.reset_stats()
    # HIP-event brackets inside the timed region around the launches of the kernel the roofline object describes only
    # (two event records per bracketed launch: bracketing all 188 000 launches of a chained Set14 x3 step costs a
    # quarter of it, bracketing all 2 600 of an independent step < 1 %); --profile-all brackets every tagged launch
    nblk_w = -(-((p_edge or 10 * s) - 2) ** 2 // NB)
    dom_tags = ["update2"] if nblk_w >= 3 else ["update", "update_last"]
    chained = a.warm_start != "independent" and kname != "rbf"
    live = a.profile_all or not chained      # chained modes: no brackets at all in the timed region (see the roofline note)
    if rank == 0 and live:
        ops.profile_begin(1 << 18, tags=None if a.profile_all else dom_tags)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        outs = step()
    barrier()
    dt = time.perf_counter() - t0
    prof = ops.profile_end() if (rank == 0 and live) else None
    prof_all = None
    if rank == 0 and not a.profile_all and not a.no_kernel_breakdown and world == 1:
        # per-kernel milliseconds of ONE more step, untimed, every tagged launch bracketed (roofline.kernel_ms_per_step)
        ops.profile_begin(1 << 18)
        step()
        torch.cuda.synchronize()
        prof_all = ops.profile_end()
    steps_prof = a.steps
    if rank == 0 and prof is None:
        if prof_all is None:
            raise SystemExit("chained modes take the roofline from the extra profiled step: drop --no-kernel-breakdown")
        prof, steps_prof = prof_all, 1
    ranks_seen, backend = 1, None
    if world > 1:
        backend = torch.distributed.get_backend()
        ranks_seen = torch.distributed.get_world_size()
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")   # gloo: one-GPU rehearsal
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    flags_bad = int(sum(int(f.sum()) for f in g.stats.flags))
    if a.warm_start == "chain":
        kern.reset()
    # per-rank account of the run (a scaling run should explain itself): GP fits this rank ran, what it put into and
    # took out of the all-gathers, the time inside them (HIP events around the collective, summed over the timed steps)
    rank_info = None
    if world > 1:
        mine = dict(gdist.stats_summary(), rank=rank)
        mine["gp_fits_per_step"] = mine.pop("rows_owned") // max(1, a.steps)
        mine["allgather_bytes_sent_per_pass"] = mine["bytes_sent"] // max(1, mine["exchange_calls"])
        mine["exchange_ms_per_step"] = round(mine.pop("exchange_ms") / max(1, a.steps), 3)
        rank_info = [None] * world
        torch.distributed.all_gather_object(rank_info, mine)

    if rank != 0:
        return 0
    ms_per_step = dt / a.steps * 1e3
    value = out_px / 1e6 / (dt / a.steps)

    # ---- parity against the committed oracle outputs of this mode (HIP metric kernels; Set14 fixtures only)
    parity = None
    gold_tag = {"independent": "oracle_m52", "chain_per_image": "oracle_m52_chain"}.get(a.warm_start)
    if (a.workload == "set14x3" and kname == "matern52" and data_desc != "synthetic" and gold_tag
            and all(os.path.exists(os.path.join(GOLDEN, f"{gold_tag}_{i:03d}.png")) for i in range(1, n_base + 1))):
        maxd, ident, npx_all, dps, dss, ps, pg = 0, 0, 0, [], [], [], []
        for k in range(len(outs)):      # N > 1, weak scaling: the images rank 0 holds (one copy's worth, any copy)
            if outs[k] is None:
                continue
            i = k % n_base
            gold = torch.as_tensor(_png(f"{gold_tag}_{i + 1:03d}.png")).to(dev)
            hr = torch.as_tensor(_png(f"hr_{i + 1:03d}.png")).to(dev)
            d = (outs[k].to(torch.int16) - gold.to(torch.int16)).abs()
            maxd = max(maxd, int(d.max()))
            ident += int((d == 0).sum())
            npx_all += d.numel()
            mo, mg = ops.metrics(hr, outs[k], s).cpu().numpy(), ops.metrics(hr, gold, s).cpu().numpy()
            dps.append(mo[0] - mg[0]); dss.append(mo[1] - mg[1]); ps.append(mo[0]); pg.append(mg[0])
        parity = {"golden": f"tests/golden/set14_x3/{gold_tag}_*.png (CPU oracle, same warm-start mode)",
                  "images_compared": len(ps),
                  "max_abs_diff_grey_levels": maxd, "identical_pixel_frac": round(ident / npx_all, 6),
                  "mean_psnr_db": round(float(np.mean(ps)), 4), "mean_psnr_db_golden": round(float(np.mean(pg)), 4),
                  "max_abs_dpsnr_db": round(float(np.max(np.abs(dps))), 5),
                  "max_abs_dssim": round(float(np.max(np.abs(dss))), 6),
                  "bar": "<= 1 grey level, 0.01 dB, 0.001 (north star)",
                  "ok": bool(maxd <= 1 and np.max(np.abs(dps)) <= 0.01 and np.max(np.abs(dss)) <= 0.001)}

    # independent starts: the reference output this MODE reproduces is a x2 YUV / BILINEAR one (an earlier state of the
    # reference) -- not this workload's configuration, recorded as the mode's pin (untimed, 0.1 s)
    ref_lr, ref_png = (os.path.join(ROOT, "tests", "golden", "refpin", n) for n in ("x2_lr_001.png", "x2_ref_matern52_yuv_001.png"))
    if (parity is not None and not a.no_reference_check and a.warm_start == "independent"
            and os.path.exists(ref_lr) and os.path.exists(ref_png)):
        from PIL import Image
        from super_resolution_amd import gprsr as _gprsr
        from super_resolution_amd.image_patches import PatchInterpolationMode
        old_pi, old_t = _gprsr.PATCH_INTERPOLATION, _gprsr.DO_TIMING
        _gprsr.PATCH_INTERPOLATION, _gprsr.DO_TIMING = PatchInterpolationMode.BILINEAR, False
        try:
            o2 = GPRSR(2, kern, ColorSpace.YUV, warm_start="independent").apply(
                np.ascontiguousarray(np.array(Image.open(ref_lr).convert("RGB"))[:, :, ::-1]))
        finally:
            _gprsr.PATCH_INTERPOLATION, _gprsr.DO_TIMING = old_pi, old_t
        d2 = np.abs(o2.astype(int) - np.array(Image.open(ref_png).convert("RGB"))[:, :, ::-1].astype(int))
        parity["mode_vs_reference_png"] = {
            "image": "Set14/image_SRF_2/img_001_SRF_2_GPR_matern52.png (the reference's own output; x2, YUV, BILINEAR, "
                     "independent starts)",
            "identical_pixel_frac": round(float((d2 == 0).mean()), 6), "max_abs_diff_grey_levels": int(d2.max())}

    # roofline of the dominant kernel.  Pivots are taken two at a time: k_update2 applies both rank-112 updates of
    # a pair in one pass over A (batched 112x112x224 f64 MFMA tile products).  Algorithmic flops of one launch, per
    # GP, with R = N - 2*112 rows outside the pair: symmetric rank-224 update of the R x R trailing matrix R^2*224,
    # pivot column of the first step + its update by the second 3*R*112^2, the first pivot block's symmetric update
    # 112^3, the triangular pivot-column product of the second step (N-112)*112^2.  (Matrices of fewer than 3
    # blocks have no pair: then k_update, (N-112)^2*112 + (N-112)*112^2.)
    pe = p_edge or 10 * s
    n = (pe - 2) ** 2
    m = (pe * s - 2) ** 2
    N = -(-n // NB) * NB
    nblk = N // NB
    if prof.get("update2", (0, 0.0))[0]:
        kname_dom, (cnt, ms) = "k_update2", prof["update2"]
        R = N - 2 * NB
        flops_gp = R * R * 2 * NB + 3 * R * NB * NB + NB ** 3 + (N - NB) * NB * NB
    else:
        kname_dom, (cnt, ms) = "k_update", prof["update"]
        flops_gp = (N - NB) ** 2 * NB + (N - NB) * NB * NB
    # GPs per launch: every GP of rank 0's share passes through the kernel `passes` times per evaluation, 2 phases x
    # 51 evaluations per step, however the engine cut the batch (workspace chunks, chain links, dual-stream halves)
    passes = ((nblk - 1) // 2 if nblk % 2 else (nblk - 2) // 2) if kname_dom == "k_update2" else nblk
    B_local = -(-n_patches // world)
    gps_per_launch = B_local * (2 * (g.iters + 1) * passes * steps_prof) / cnt if cnt else 0.0
    flops_launch = int(gps_per_launch * flops_gp)
    avg_s = (ms / cnt) * 1e-3 if cnt else float("nan")
    achieved = flops_launch / avg_s / 1e12 if cnt else float("nan")
    kernel_ms = ({k: round(v[1] / steps_prof, 3) for k, v in prof.items()} if a.profile_all
                 else {k: round(v[1], 3) for k, v in prof_all.items()} if prof_all else None)
    traffic, traffic_src = pmc_traffic(kname_dom, int(round(gps_per_launch)), N)
    whole = (n_patches / world) * 2 * fit_flops(n, m, g.iters) / (dt / a.steps) / 1e12    # per GPU
    roofline = {"kernel": kname_dom, "bound": "mfma", "achieved": round(achieved, 3),
                "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
                "traffic": traffic,
                "traffic_note": f"HBM bytes per launch, {traffic_src} (separate rocprofv3 --pmc passes)" if traffic_src
                else "no committed PMC summary for this shape",
                "whole_step_frac": round(whole / FP64_MFMA_PEAK_TFLOPS, 4),
                "whole_step_note": "SURVEY 8(d) algorithmic flops of all fits of one GPU / step time / FP64 peak",
                "launches_per_step": cnt // max(1, steps_prof), "gps_per_launch": round(gps_per_launch, 1),
                "measured": "HIP events on the launch stream inside the timed region" if live else
                            "HIP events in one extra step after the timed region: a chained step is ~190 000 small launches and "
                            "two event records per launch would cost a quarter of the time it is meant to measure",
                "avg_launch_ms": round(avg_s * 1e3, 4),
                "flops_per_launch": flops_launch, "kernel_ms_per_step": kernel_ms,
                "kernel_ms_note": "every tagged launch bracketed inside the timed region" if a.profile_all else
                                  "from one extra untimed step with every tagged launch bracketed (the timed region brackets "
                                  "only the dominant kernel)"}

    cpu = cpu1 = None
    if not a.no_cpu_baseline and world == 1:   # reported at N=1 only (rank 0 would stall the other ranks' exit)
        fits = 2 * n_patches // copies  # per copy of the workload
        allc = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)

        def leg(cores):
            sec_per_fit, nfit = cpu_baseline(g._gp.kernel_id, s, cores, p=p_edge)
            return {"value": round((out_px / copies) / 1e6 / (sec_per_fit * fits), 6), "unit": "output Mpx/s", "cores": cores,
                    "kind": "port",
                    "sample": f"{nfit} whole patch fits (n={n}, 50 Adam steps + predictive mean, m={m}) "
                              f"of the oracle's torch-f64 per-patch loop on {cores} thread(s), {sec_per_fit:.2f} s/fit, "
                              f"extrapolated linearly to the workload's {fits} fits (patch gather/recombination not "
                              f"counted; the warm-start mode does not change the cost of a fit)"}
        cpu = leg(allc)
        cpu1 = leg(1) if allc > 1 else cpu    # all host cores AND one

    headline = a.workload == "set14x3" and kname == "matern52"
    # the reference-faithful route next to the headline: `independent` starts are the north star's batched engine (and an
    # earlier state of the reference), today's reference chains its patches -- one untimed-for-the-headline step of
    # chain_per_image, with its own clock, and its output against the PNG the reference committed for this very workload
    ref_sem = None
    ref_png = os.path.join(GOLDEN, "refout_m52_001.png")
    if headline and world == 1 and not a.no_reference_check and data_desc != "synthetic" and os.path.exists(ref_png):
        if a.warm_start == "independent":
            gc = GPRSR(s, kern, ColorSpace.GRAYSCALE, warm_start="chain_per_image")
            gc.apply_for_channel_device(imgs)      # warm-up (workspace, first-launch costs)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            oc = [ops.unit_to_u8(o, 255.0) for o in gc.apply_for_channel_device(imgs)]
            torch.cuda.synchronize()
            tc = time.perf_counter() - tc
        else:
            oc, tc = outs, dt / a.steps
        refimg = torch.as_tensor(_png("refout_m52_001.png")).to(dev)
        d1 = (oc[0].to(torch.int16) - refimg.to(torch.int16)).abs()
        ref_sem = {"warm_start": "chain_per_image" if a.warm_start == "independent" else a.warm_start,
                   "note": "the reference's own semantics (kernels.py:7-13 warm-start chain), every image as the first image "
                           "of a fresh reference process; same workload, same step definition",
                   "ms_per_step": round(tc * 1e3, 2), "value": round(out_px / 1e6 / tc, 4), "unit": "Mpx/s",
                   "vs_reference_png": {"image": "Set14/image_SRF_3/img_001_SRF_3_GPR_matern52.png (the reference's own output)",
                                        "identical_pixel_frac": round(float((d1 == 0).float().mean()), 6),
                                        "max_abs_diff_grey_levels": int(d1.max())}}
    line = {"metric": f"output megapixels/sec (SR x3, Matern-5/2; warm_start={a.warm_start})" if headline
            else f"output megapixels/sec ({a.workload}, {kname}; warm_start={a.warm_start})",
            "value": round(value, 4), "unit": "Mpx/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f64", "data": data_desc,
            "config": {"workload": f"{a.workload}: {len(imgs_u8)} images, {n_patches} patches, "
                                   f"{2 * n_patches} GP fits (n={n}), {out_px / 1e6:.4f} output Mpx",
                       "kernel": kname,
                       "color": "GRAYSCALE", "warm_start": a.warm_start, "dual_stream": bool(a.dual_stream), "split_small": not a.no_split_small,
                       "parallelism": (f"patch-sharded x{world}" if a.warm_start == "independent"
                                       else f"whole chains dealt to {world} rank(s)"),
                       "ranks_in_process_group": ranks_seen, "backend": backend},
            "nonpositive_pivot_flags": flags_bad,
            "parity": parity, "reference_semantics": ref_sem, "roofline": roofline, "cpu_baseline": cpu,
            "cpu_baseline_1_thread": cpu1, "ranks": rank_info}
    if cpu:
        line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
    print(json.dumps(line), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
