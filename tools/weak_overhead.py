"""What rank 0 of an N-rank weak-scaling run does per step, without a process group: N copies of the Set14 x3
workload, shard=(0, N), the all-gather replaced by tiling this rank's rows (the copies are identical and the
contiguous patch slices are copy-aligned, so rank r's rows ARE rank 0's rows).  The difference to the 1-copy,
1-rank step is the replicated work (image stages, recombination, host loops) -- the part of the weak-scaling
loss that is not communication."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from super_resolution_amd import kernels as K, ops, dist as gdist
from super_resolution_amd.gprsr import GPRSR, ColorSpace

dev = torch.device("cuda", 0)
world = [1]


def fake_exchange(locals_, owned, B, rank, world_, group=None, force_collective=False):
    outs = []
    for t in locals_:
        assert t.shape[0] * world_ == B
        outs.append(t.repeat((world_,) + (1,) * (t.dim() - 1)))
    return outs


gdist.exchange_rows = fake_exchange
OUTPUTS = sys.argv[1] if len(sys.argv) > 1 else "owned"   # what bench.py uses for weak scaling; "all": every rank recombines everything
res = {}
for N in (1, 2, 4, 8):
    imgs_u8, s, _ = bench.workload("set14x3", N)
    imgs = [ops.u8_to_unit(torch.as_tensor(im).to(dev)) for im in imgs_u8]
    g = GPRSR(s, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="independent", shard=(0, N) if N > 1 else None,
              outputs=OUTPUTS)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = [None if o is None else ops.unit_to_u8(o, 255.0) for o in g.apply_for_channel_device(imgs)]
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[N] = dt
    print("ranks %d: rank-0 step %.1f ms (+%.1f ms over one rank, weak-scaling efficiency bound %.3f)"
          % (N, dt * 1e3, (dt - res[1]) * 1e3, res[1] / dt), flush=True)
