"""RCCL smoke test on the one GPU of the box: a 1-rank "nccl" process group running the
same collective call the sharded path makes (one all_gather_into_tensor of the packed f64 / f32 /
int32 rows).  The world-2 logic itself is covered on CPU with gloo (tests/test_host.py);
8-GPU runs are the driver's."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from super_resolution_amd import dist as gdist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29653", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
pred = torch.randn((37, 1444), dtype=torch.float64, device=dev, generator=g)
raw = torch.randn((37, 4), dtype=torch.float32, device=dev, generator=g)
flags = torch.arange(37, dtype=torch.int32, device=dev)
import numpy as np
outs = gdist.exchange_rows([pred, raw, flags], [np.arange(37, dtype=np.int64)], 37, 0, 1, force_collective=True)
for out, t in zip(outs, (pred, raw, flags)):
    assert out.shape == t.shape and out.dtype == t.dtype and torch.equal(out, t)
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
sys.stdout.write("rccl-ok\n")
"""


@pytest.mark.gpu
def test_rccl_all_gather_single_rank():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


CHILD2 = r"""
import os, sys
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from super_resolution_amd import dist as gdist, kernels as K, ops
from super_resolution_amd.gprsr import GPRSR, ColorSpace
from test_gpu_kernels import natural_image
rank, world, local = gdist.init_from_env()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
imgs = [np.rint(natural_image(H, W, s) * 255).astype(np.uint8) for H, W, s in ((44, 40, 1), (47, 52, 2), (40, 41, 3))]
dimgs = [ops.u8_to_unit(torch.as_tensor(im).to(dev)) for im in imgs]
sharded = GPRSR(2, K.matern32_kernel, ColorSpace.GRAYSCALE, warm_start="independent", shard=(rank, world)).apply_for_channel_device(dimgs)
if rank == 0:
    whole = GPRSR(2, K.matern32_kernel, ColorSpace.GRAYSCALE, warm_start="independent").apply_for_channel_device(dimgs)
    assert all(torch.equal(a, b) for a, b in zip(sharded, whole)), "sharded != unsharded"
# outputs="owned": a rank returns only the images it fitted patches of, and skips the stages of the others
owned = GPRSR(2, K.matern32_kernel, ColorSpace.GRAYSCALE, warm_start="independent", shard=(rank, world), outputs="owned").apply_for_channel_device(dimgs)
assert any(o is None for o in owned) and any(o is not None for o in owned), [o is None for o in owned]
assert all(o is None or torch.equal(o, a) for o, a in zip(owned, sharded)), "owned != all"
have = torch.tensor([float(o is not None) for o in owned]); dist.all_reduce(have)
assert bool((have >= 1).all()), "an image nobody returned"
if rank == 0:
    sys.stdout.write("shard-ok\n")
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_sharded_equals_unsharded(tmp_path):
    """The whole sharded path on the device (gather of a slice, HIP fit, all-gather, recombination on every
    rank) with two ranks sharing the box's one GPU; the collectives ride on gloo for this rehearsal (RCCL
    refuses two ranks on one device), everything else is the production path.  Bit-identical to one rank."""
    script = tmp_path / "child2.py"
    script.write_text(CHILD2 % (ROOT, ROOT))
    env = dict(os.environ)
    env.update(GPSR_LOCAL_DEVICE="0", GPSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "shard-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-8000:]


CHILD3 = r"""
import os, sys
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import numpy as np, torch, torch.distributed as dist
from super_resolution_amd import dist as gdist, kernels as K, ops
from super_resolution_amd.gprsr import GPRSR, ColorSpace
from test_gpu_kernels import natural_image
rank, world, local = gdist.init_from_env()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
imgs = [np.rint(natural_image(58 + (s * 5) %% 7, 57 + (s * 3) %% 8, s) * 255).astype(np.uint8) for s in range(1, 15)]   # 14 chains: 7 per rank
dimgs = [ops.u8_to_unit(torch.as_tensor(im).to(dev)) for im in imgs]
dist.barrier()
sharded = GPRSR(3, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="chain_per_image", shard=(rank, world)).apply_for_channel_device(dimgs)
if rank == 0:
    whole = GPRSR(3, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="chain_per_image").apply_for_channel_device(dimgs)
    assert all(torch.equal(a, b) for a, b in zip(sharded, whole)), "sharded != unsharded"
owned = GPRSR(3, K.matern52_kernel, ColorSpace.GRAYSCALE, warm_start="chain_per_image", shard=(rank, world), outputs="owned").apply_for_channel_device(dimgs)
assert sum(o is not None for o in owned) == 7 and all(o is None or torch.equal(o, a) for o, a in zip(owned, sharded)), "owned != all"
if rank == 0:
    sys.stdout.write("chain-shard-ok\n")
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_chained_mode_equals_one_rank(tmp_path):
    """Chained warm start, whole chains dealt to two ranks that share the one GPU: both processes run small-batch
    (strip-form) launches of n = 784 at the same time on the same CUs.  This is the configuration in which the strip
    form's missing wait on its own LDS reads showed (garbage operands, thousands of flagged GPs) while every
    single-process test passed; bit-identical to one rank."""
    script = tmp_path / "child3.py"
    script.write_text(CHILD3 % (ROOT, ROOT))
    env = dict(os.environ)
    env.update(GPSR_LOCAL_DEVICE="0", GPSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29633", str(script)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "chain-shard-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-8000:]


@pytest.mark.gpu
@pytest.mark.parametrize("workload,scaling", [("tiny", "weak"), ("tiny", "strong")])
def test_bench_two_ranks_on_one_gpu_prints_the_line_and_the_per_rank_account(workload, scaling):
    """`bench.py --gpus 2` as the driver's scaling run starts it (its own launcher, one process per rank), rehearsed on
    the one GPU of the box with gloo collectives: rank 0 prints ONE JSON line for n_gpus = 2 with the per-rank account
    of the exchange (GP fits per step, bytes into the all-gather, milliseconds inside it)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(GPSR_LOCAL_DEVICE="0", GPSR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", workload, "--scaling", scaling,
                        "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == scaling and line["value"] > 0
    assert line["config"]["ranks_in_process_group"] == 2 and line["config"]["backend"] == "gloo"
    ranks = line["ranks"]
    assert [q["rank"] for q in ranks] == [0, 1]
    n_fits = int(line["config"]["workload"].split(" GP fits")[0].split(", ")[-1])
    assert sum(q["gp_fits_per_step"] for q in ranks) == n_fits           # every fit on exactly one rank
    assert all(q["exchange_calls"] == 2 and q["allgather_bytes_sent_per_pass"] > 0 and q["exchange_ms_per_step"] > 0 for q in ranks)
