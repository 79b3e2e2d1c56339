"""Patch sharding across the GPUs of one node (one process per GPU).

The per-patch GPs of a predict_pixels call are independent (independent
warm-start mode), so the global patch list is cut into `world` contiguous,
balanced slices; rank r gathers, fits and predicts only its slice.  The ONE
exchange on the path is an all-gather of the predicted patch interiors (plus
the tiny hyper-parameter / flag arrays) so that every rank can recombine the
full HR image -- it needs the whole phase-1 image for the blur/downsample that
feeds phase 2 (gprsr.py:85-87).  torch.distributed backend "nccl" is RCCL over
xGMI on ROCm; "gloo" is used for the CPU tests of this logic.
"""
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous balanced slice [lo, hi) of B items: the first B % world ranks get one extra."""
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_rows(local, B, world, group=None, force_collective=False):
    """local: this rank's [hi-lo, ...] slice (shard_range order) -> the full [B, ...] tensor on every rank.
    force_collective: run the collective even for world == 1 (used by the single-GPU RCCL smoke test)."""
    if world == 1 and not force_collective:
        return local
    cap = (B + world - 1) // world
    tail = local.shape[1:]
    buf = torch.zeros((cap,) + tuple(tail), dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    if dist.get_backend(group) == "gloo" and buf.is_cuda:   # rehearsal path: gloo moves host memory
        hbuf = buf.cpu()
        hparts = [torch.empty_like(hbuf) for _ in range(world)]
        dist.all_gather(hparts, hbuf, group=group)
        parts = [p.to(buf.device) for p in hparts]
    else:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
    out = torch.empty((B,) + tuple(tail), dtype=local.dtype, device=local.device)
    for r in range(world):
        lo, hi = shard_range(B, r, world)
        out[lo:hi] = parts[r][:hi - lo]
    return out


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun). Returns (rank, world, local_rank)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: GPSR_LOCAL_DEVICE=0 puts every rank on that device and
    # GPSR_DIST_BACKEND=gloo carries the collectives (RCCL refuses two ranks on one GPU)
    if "GPSR_LOCAL_DEVICE" in os.environ:
        local = int(os.environ["GPSR_LOCAL_DEVICE"])
    if backend is None and "GPSR_DIST_BACKEND" in os.environ:
        backend = os.environ["GPSR_DIST_BACKEND"]
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local
