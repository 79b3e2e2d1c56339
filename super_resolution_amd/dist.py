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
import numpy as np
import torch
import torch.distributed as dist


def shard_range(B, rank, world):
    """Contiguous balanced slice [lo, hi) of B items: the first B % world ranks get one extra."""
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def deal_chains(lens, world):
    """Owner rank of each warm-start chain (a chain is serial and stays on one GPU): longest chain first,
    each to the rank with the least work so far.  Deterministic: every rank computes the same table."""
    load = [0] * world
    owner = [0] * len(lens)
    for j in sorted(range(len(lens)), key=lambda j: (-lens[j], j)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[j] = r
        load[r] += lens[j]
    return owner


# Diagnostics of the exchange (bench.py --gpus N prints them per rank, so that a scaling run explains itself): number
# of calls, bytes this rank contributed / received, and -- on CUDA tensors -- HIP events around the collective.
STATS = {"calls": 0, "bytes_sent": 0, "bytes_received": 0, "rows_owned": 0, "events": [], "ms": 0.0}
_MAX_PENDING_EVENTS = 32      # a process that never asks for the summary must not pile up event pairs


def reset_stats():
    STATS.update(calls=0, bytes_sent=0, bytes_received=0, rows_owned=0, events=[], ms=0.0)


def _fold_events(keep):
    """Move the oldest event pairs beyond `keep` into the millisecond total (they are waited for: long finished)."""
    ev = STATS["events"]
    while len(ev) > keep:
        a, b = ev.pop(0)
        b.synchronize()
        STATS["ms"] += a.elapsed_time(b)


def stats_summary():
    """-> dict with the counters and the total milliseconds inside the collectives (synchronises on their events)."""
    _fold_events(0)
    ms = STATS["ms"]
    return {"exchange_calls": STATS["calls"], "bytes_sent": STATS["bytes_sent"], "bytes_received": STATS["bytes_received"],
            "rows_owned": STATS["rows_owned"], "exchange_ms": round(ms, 3)}


def _as_range(idx):
    """(lo, hi) if the index array is the contiguous range lo..hi-1, else None."""
    n = len(idx)
    if n == 0:
        return (0, 0)
    lo = int(idx[0])
    return (lo, lo + n) if int(idx[-1]) == lo + n - 1 and (n < 3 or np.array_equal(idx, np.arange(lo, lo + n))) else None


def exchange_rows(locals_, owned, B, rank, world, group=None, force_collective=False):
    """The one collective of the path.  Every rank holds, for each tensor of `locals_`, the rows it computed
    (row i of the local tensor = global row owned[rank][i]); returns the full [B, ...] tensors on every rank.
    The row payloads of all tensors are packed byte-wise into one buffer of max-rows-per-rank rows, exchanged
    with a single all_gather_into_tensor (RCCL all-gather over xGMI; no list API, no per-tensor collectives)
    and scattered to their global rows: a rank's rows go to their place with one strided copy per tensor when the rank
    owns a contiguous range (independent GPs: always), with an index_copy_ otherwise (whole chains dealt out).
    owned: list over ranks of int64 index arrays (known to every rank).  A rank that owns nothing sends padding
    (the padding rows of the buffer are never read, so the buffer is not cleared).
    force_collective: also for world == 1 (RCCL smoke)."""
    cnt = len(owned[rank])
    for t in locals_:
        assert t.shape[0] == cnt, (t.shape, cnt)
    if world == 1 and not force_collective:
        outs = []
        for t in locals_:
            o = torch.empty((B,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            o.index_copy_(0, torch.as_tensor(owned[0]).to(t.device), t)
            outs.append(o)
        return outs
    dst_all = np.concatenate([np.asarray(o, dtype=np.int64) for o in owned])
    assert dst_all.size == B and np.array_equal(np.sort(dst_all), np.arange(B)), "every global row must have exactly one owner"
    dev = locals_[0].device
    cap = max(1, max(len(o) for o in owned))
    widths = [int(np_prod(t.shape[1:])) * t.element_size() for t in locals_]
    row_bytes = (sum(widths) + 15) // 16 * 16
    buf = torch.empty((cap, row_bytes), dtype=torch.uint8, device=dev)
    col = 0
    for t, w in zip(locals_, widths):
        if cnt:
            buf[:cnt, col:col + w] = t.contiguous().view(cnt, -1).view(torch.uint8).view(cnt, w)
        col += w
    out = torch.empty((world * cap, row_bytes), dtype=torch.uint8, device=dev)
    ev = None
    if buf.is_cuda:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    if dist.get_backend(group) == "gloo" and buf.is_cuda:   # one-GPU rehearsal: gloo moves host memory
        hout = torch.empty(out.shape, dtype=torch.uint8)
        dist.all_gather_into_tensor(hout, buf.cpu(), group=group)
        out = hout.to(dev)
    else:
        dist.all_gather_into_tensor(out, buf, group=group)
    if ev is not None:
        ev[1].record()
        STATS["events"].append(ev)
        _fold_events(_MAX_PENDING_EVENTS)
    STATS["calls"] += 1
    STATS["bytes_sent"] += cap * row_bytes
    STATS["bytes_received"] += world * cap * row_bytes
    STATS["rows_owned"] += cnt
    outs, col = [], 0
    for t, w in zip(locals_, widths):
        full = torch.empty((B, w), dtype=torch.uint8, device=dev)
        for r, o in enumerate(owned):
            k = len(o)
            if k == 0:
                continue
            src = out[r * cap:r * cap + k, col:col + w]
            rng = _as_range(np.asarray(o))
            if rng is not None:
                full[rng[0]:rng[1]] = src
            else:
                full.index_copy_(0, torch.as_tensor(np.asarray(o, dtype=np.int64)).to(dev), src.contiguous())
        outs.append(full.view(t.dtype).view((B,) + tuple(t.shape[1:])))
        col += w
    return outs


def np_prod(shape):
    p = 1
    for v in shape:
        p *= int(v)
    return p


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
