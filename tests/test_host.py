"""CPU tests (-m "not gpu") of the product's host side: the C-ABI library loads and
exports every symbol include/gpsr.h declares (no compute calls), the drop-in
host logic (PatchHandler geometry, kernel registry, resize tables, sharding), and
the N>1 exchange on world_size-2 gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------- C ABI
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpsr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpsr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from super_resolution_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = _lib.load()
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gpsr.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.gpsr_version().decode() == "gpsr 0.1 gfx950"
    assert [lib.gpsr_padded_n(n) for n in (1, 100, 112, 113, 324, 784, 1024)] == [112, 112, 112, 224, 336, 784, 1120]
    assert lib.gpsr_fit_workspace_bytes(0, 324, 8) == 0
    assert lib.gpsr_fit_workspace_bytes(2, 784, 8) > 2 * 784 * 784 * 8


def test_library_is_gfx950_code_object():
    from super_resolution_amd import _lib
    import glob
    before = set(glob.glob(_lib.LIB_PATH + ".*"))
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", _lib.LIB_PATH],
                         capture_output=True, text=True)
    for f in set(glob.glob(_lib.LIB_PATH + ".*")) - before:   # llvm-objdump leaves the unbundled code objects beside the input
        os.remove(f)
    if out.returncode == 0:
        assert "gfx950" in out.stdout


def test_empty_batches_are_no_ops():
    """B = 0 (an image rank with no patches in its shard, an empty job list) returns 0 from every batched entry
    point before anything is dereferenced or launched -- so this runs without a GPU."""
    from super_resolution_amd import _lib
    lib = _lib.load()
    N = None
    assert lib.gpsr_gather(N, N, N, 64, 64, 2, 20, N, 0, N, 18, N, N, N, N) == 0
    assert lib.gpsr_init_hypers(N, 0, 324, 3, 0.0, 0.0, N, 0, N, N) == 0
    assert lib.gpsr_covariance(N, 0, 324, 8, 3, N, N, N) == 0
    assert lib.gpsr_covariance_grad(N, 0, 324, 8, 3, N, N, N, N) == 0
    assert lib.gpsr_invert(N, 0, 336, N, N, N, 0, N) == 0
    assert lib.gpsr_fit(N, N, 0, 324, 8, 3, N, 50, 0.1, N, N, N, N, 0, N) == 0
    assert lib.gpsr_loss_grad(N, N, 0, 324, 8, 3, N, N, N, N, N, 0, N) == 0
    assert lib.gpsr_predict(N, N, N, N, 0, 324, 1444, 3, N, N) == 0
    assert lib.gpsr_invert_workspace_bytes(0, 336) == 0


def test_ops_refuse_cpu_tensors():
    from super_resolution_amd import _lib, ops
    with pytest.raises(_lib.GpsrError):
        ops.covariance(torch.zeros(1, 4, 8, dtype=torch.float64), 0, torch.zeros(1, 4))


# ----------------------------------------------------------------- PatchHandler / kernels drop-in surface
def test_patch_handler_geometry_matches_reference(golden_dir):
    from super_resolution_amd.image_patches import PatchHandler, PatchInterpolationMode
    g = np.load(f"{golden_dir}/patches_locations.npz")
    for k in g.files:
        parts = k.split("_")
        H, W, p = map(int, parts[:3])
        sp = float(parts[3]) if len(parts) > 3 else 0.9
        h = PatchHandler([np.zeros((H, W))], base_patch_size=p, stride_percentage=sp)
        assert np.array_equal(np.array(h.patch_locations), g[k]), k
        assert h.stride == int((p - 2) * sp) and h.scale_factors == [1]
    d = np.load(f"{golden_dir}/patches_iter_combine.npz")
    for tag in ("s2", "s3", "s2dup"):
        H, W, p, s = [int(v) for v in d[tag + "_meta"]]
        h = PatchHandler([d[tag + "_a"], d[tag + "_b"], d[tag + "_c"]], base_patch_size=p)
        assert h.scale_factors == [1, 1, s]
        trip = list(h)
        assert len(trip) == len(h.patch_locations)
        assert np.array_equal(np.stack([t[0] for t in trip]), d[tag + "_pa"])
        assert np.array_equal(np.stack([t[2] for t in trip]), d[tag + "_pc"])
        trip[0][0][:] = -1                     # patches are copies (image_patches.py:102)
        assert d[tag + "_a"].min() >= 0
    assert PatchInterpolationMode.AVERAGE.value == 0 and PatchInterpolationMode.BILINEAR.value == 1
    with pytest.raises(AssertionError):
        PatchHandler([np.zeros((10, 30))], base_patch_size=20)


def test_kernel_registry_names_and_ids():
    from super_resolution_amd import kernels as K, ops
    names = [K.get_kernel_name(k) for k in (K.rbf_kernel, K.matern52_kernel, K.matern32_kernel, K.exponential_kernel,
                                            K.linear_kernel, K.periodic_kernel, K.spectral_mixture_kernel)]
    assert names == ["RBF", "matern52", "matern32", "matern12", "linear", "periodic", "spectral_mixture"]
    assert K.get_kernel_name(K.AdditiveKernel(K.rbf_kernel, K.linear_kernel)) == "add(RBF,linear)"
    assert K.get_kernel_name(K.ProductKernel(K.rbf_kernel, K.matern32_kernel)) == "prod(RBF,matern32)"
    assert [K.kernel_id_of(k) for k in (K.rbf_kernel, K.exponential_kernel, K.matern32_kernel, K.matern52_kernel,
                                        K.linear_kernel)] == [ops.RBF, ops.MATERN12, ops.MATERN32, ops.MATERN52, ops.LINEAR]
    with pytest.raises(NotImplementedError):
        K.kernel_id_of(K.periodic_kernel)
    with pytest.raises(Exception):
        K.get_kernel_name(object())


def test_gprsr_module_surface():
    import gprsr, kernels, image_patches, main  # the top-level drop-in modules  # noqa: F401
    for name in ("GPRSR", "ColorSpace", "SRF", "IMAGE_NUMS", "DO_TIMING", "USED_COLOR_SPACE", "USED_KERNEL",
                 "USE_ALL_PIXELS_FOR_TRAINING", "LEARNING_RATE", "STRIDE_PERCENTAGE", "PATCH_INTERPOLATION",
                 "extract_patch", "extract_neighbors"):
        assert hasattr(gprsr, name), name
    assert gprsr.LEARNING_RATE == 0.1 and gprsr.STRIDE_PERCENTAGE == 0.9 and gprsr.SRF == 2
    assert [c.name for c in gprsr.ColorSpace] == ["BGR", "YUV", "GRAYSCALE"]
    img = np.arange(25.).reshape(5, 5)
    assert gprsr.extract_neighbors(img, 2, 2) == [17.0, 7.0, 13.0, 11.0, 18.0, 6.0, 16.0, 8.0]
    g = gprsr.GPRSR(3)
    assert g.used_kernel is kernels.matern52_kernel and g.color_mode == gprsr.ColorSpace.YUV
    for m in ("apply", "apply_for_channel", "upsample", "blur_downsample", "deblur", "predict_pixels"):
        assert callable(getattr(g, m))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):          # no silent CPU fallback
            g.apply_for_channel(np.zeros((40, 40), dtype=np.uint8))


def test_main_options_parse():
    import main
    a = main.parse_args(["--srf", "3", "--kernel", "matern32", "--color-space", "yuv", "--images", "2-4",
                         "--stride-percentage", "0.8", "--patch-interpolation", "bilinear", "--grid-training",
                         "--warm-start", "chain", "--no-timing", "--dataset-dir", "/tmp/x"])
    assert a.srf == 3 and a.kernel == "matern32" and a.color_space == "yuv" and a.images == "2-4"
    assert a.stride_percentage == 0.8 and a.patch_interpolation == "bilinear" and a.grid_training
    assert main.image_numbers("2-4") == [2, 3, 4] and main.image_numbers("1,3,14") == [1, 3, 14]
    assert main.output_path("Set14", 3, 7, "matern52", True) == "Set14/image_SRF_3/img_007_SRF_3_GPR_matern52_gray.png"
    assert main.output_path("Set14", 2, 1, "RBF", False) == "Set14/image_SRF_2/img_001_SRF_2_GPR_RBF.png"


# ----------------------------------------------------------------- resize tables vs the oracle's scalar restatement
@pytest.mark.parametrize("n_src,inv_scale", [(37, 2.0), (53, 3.0), (40, 4.0), (90, 1 / 3), (75, 0.4)])
def test_resize_tables_match_oracle(n_src, inv_scale):
    from oracle import cvops
    from super_resolution_amd import imgproc
    n_dst = imgproc.cv_round(n_src * inv_scale)
    i1, w1 = imgproc._axis_tables(n_dst, n_src, float(inv_scale), "cubic")
    i2, w2 = cvops._cubic_tables(n_dst, n_src, inv_scale)
    assert np.array_equal(i1, i2) and np.array_equal(w1, w2.astype(np.float64))
    i1, w1 = imgproc._axis_tables(n_dst, n_src, float(inv_scale), "linear")
    i2, w2 = cvops._linear_tables(n_dst, n_src, inv_scale)
    assert np.array_equal(i1, i2) and np.array_equal(w1, w2.astype(np.float64))
    assert np.array_equal(imgproc.gaussian_kernel(5, 0), cvops.gaussian_kernel(5, 0))
    assert np.array_equal(imgproc.gaussian_kernel(3, 1), cvops.gaussian_kernel(3, 1))


# ----------------------------------------------------------------- sharding
def test_shard_ranges_partition_the_patch_list():
    from super_resolution_amd.dist import shard_range
    for B in (1, 7, 8, 633, 32400):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(B, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from super_resolution_amd.dist import shard_range, exchange_rows, deal_chains, init_from_env
rank, world, _ = init_from_env("gloo")
# contiguous slices (independent GPs), three tensors of different dtypes packed into ONE all-gather
for B in (11, 1, 2):          # B = 1: rank 1 owns nothing and still takes part
    full = torch.arange(B * 3, dtype=torch.float64).reshape(B, 3) * 1.5
    raw = (torch.arange(B * 4, dtype=torch.float32).reshape(B, 4) - 7) / 3
    owned = [np.arange(*shard_range(B, r, world), dtype=np.int64) for r in range(world)]
    lo, hi = shard_range(B, rank, world)
    out, r4, fl = exchange_rows([full[lo:hi].clone(), raw[lo:hi].clone(), torch.arange(lo, hi, dtype=torch.int32)],
                                owned, B, rank, world)
    assert torch.equal(out, full) and torch.equal(r4, raw), (rank, B, out)
    assert torch.equal(fl, torch.arange(B, dtype=torch.int32))
# chains dealt out whole: interleaved ownership
lens = [5, 2, 7, 1]
owner = deal_chains(lens, world)
assert sorted(set(owner)) == list(range(world)) and owner[2] != owner[0]      # the two longest go to different ranks
offs = np.concatenate([[0], np.cumsum(lens)])
owned = [np.concatenate([np.arange(offs[j], offs[j + 1]) for j in range(len(lens)) if owner[j] == r]) for r in range(world)]
B = int(offs[-1])
full = torch.arange(B * 2, dtype=torch.float64).reshape(B, 2)
(out,) = exchange_rows([full[torch.as_tensor(owned[rank])]], owned, B, rank, world)
assert torch.equal(out, full)
dist.barrier()
sys.stdout.write("rank%dok\n" % rank); sys.stdout.flush()
'''


def test_exchange_rows_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank0ok" in r.stdout and "rank1ok" in r.stdout


_WORKER_ENGINE = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from super_resolution_amd import engine, ops, dist as gdist
rank, world, _ = gdist.init_from_env("gloo")

# Stand-ins for the device ops (no GPU here): gather = real patch features in numpy,
# the "GP" = a deterministic function of each patch, combine = oracle recombination.
from oracle import patches as P
def fake_gather(tin, ttg, tst, s, p, locs, tidx, out=None):
    tin, ttg, tst = tin.numpy(), ttg.numpy(), tst.numpy()
    X = np.stack([P.features(P.extract(tin, l, p, 1)) for l in locs.numpy()])
    y = np.stack([P.targets(P.extract(ttg, l, p, 1)) for l in locs.numpy()])
    Xs = np.stack([P.features(P.extract(tst, l, p, s)) for l in locs.numpy()])
    if out is not None:   # the engine gathers straight into slices of its batch buffers
        out[0].copy_(torch.tensor(X)); out[1].copy_(torch.tensor(y)); out[2].copy_(torch.tensor(Xs))
        return out
    return torch.tensor(X), torch.tensor(y), torch.tensor(Xs)
def fake_combine(pred, tst, H, W, s, p, rows, cols, mode, clip=True):
    locs = [(int(r), int(c)) for r in rows for c in cols]
    ups = []
    for k, l in enumerate(locs):
        q = P.extract(tst.numpy(), l, p, s); q[1:-1, 1:-1] = pred[k].numpy().reshape(p * s - 2, p * s - 2); ups.append(q)
    return torch.tensor(np.clip(P.combine(ups, locs, (H, W), p, mode), 0, 1))
ops.gather, ops.combine = fake_gather, fake_combine
class FakeGP:
    def fit_predict(self, X, y, Xs):
        pred = Xs.mean(-1) * 0.5 + y.mean(-1, keepdim=True) * 0.5
        raw = torch.stack([y.mean(-1), y.std(-1), X.mean((1, 2)), X.amax((1, 2))], 1).float()
        return pred, raw, torch.zeros(X.shape[0], dtype=torch.int32)
    def fit_predict_chains(self, X, y, Xs, lens, carry):
        # a serial recurrence along each chain, so a wrong order / wrong owner shows
        pred, raw, fl = self.fit_predict(X, y, Xs)
        o = 0
        for j, L in enumerate(lens):
            c = carry[j, 2].double()
            for k in range(L):
                c = 0.5 * c + y[o + k].mean()
                pred[o + k] += c
                raw[o + k, 2] = c.float()
            carry[j, 2] = c.float()
            o += L
        return pred, raw, fl
rng = np.random.default_rng(0)
jobs = []
for (H, W) in [(47, 53), (52, 36)]:
    a, b, c = rng.random((H, W)), rng.random((H, W)), rng.random((2 * H, 2 * W))
    jobs.append((torch.tensor(a), torch.tensor(b), torch.tensor(c)))
st = engine.FitStats()
outs = engine.predict_pixels_batch(jobs, 2, 20, FakeGP(), stats=st, shard=(rank, world))
ref = engine.predict_pixels_batch(jobs, 2, 20, FakeGP(), shard=None) if rank == 0 else None
if rank == 0:
    for o, r in zip(outs, ref):
        assert torch.equal(o, r)
    assert st.raw[0].shape == (24, 4)
# warm-start chains: whole jobs per rank, carries exchanged
c1 = torch.tensor([[0., 0., 0.25, 0.], [0., 0., -1.0, 0.]])
c2 = c1.clone()
outs = engine.predict_pixels_batch(jobs, 2, 20, FakeGP(), chain_carry=c1, shard=(rank, world))
ref = engine.predict_pixels_batch(jobs, 2, 20, FakeGP(), chain_carry=c2, shard=None)
for o, r in zip(outs, ref):
    assert torch.equal(o, r)
assert torch.equal(c1, c2) and not torch.equal(c1[:, 2], torch.tensor([0.25, -1.0]))
# fewer patches than ranks: one 20x20 image has 4 (duplicate) patches; 1 job as a chain -> rank 1 owns nothing
small = [(torch.tensor(rng.random((20, 20))), torch.tensor(rng.random((20, 20))), torch.tensor(rng.random((40, 40))))]
c3 = torch.zeros((1, 4))
o1 = engine.predict_pixels_batch(small, 2, 20, FakeGP(), chain_carry=c3, shard=(rank, world))
o2 = engine.predict_pixels_batch(small, 2, 20, FakeGP(), chain_carry=torch.zeros((1, 4)), shard=None)
assert torch.equal(o1[0], o2[0])
# recombine="owned": two copies of the two images on two ranks -> the patch slices are copy-aligned, a rank recombines
# its own copy only and may be handed None for the inputs of the other rank's images (what GPRSR does in phase 2)
jobs4 = jobs + [tuple(t.clone() for t in j) for j in jobs]
own = engine.owned_jobs([tuple(j[1].shape) for j in jobs4], 20, 0.9, (rank, world), False)
assert own == ([True, True, False, False] if rank == 0 else [False, False, True, True]), own
ref4 = engine.predict_pixels_batch(jobs4, 2, 20, FakeGP(), shard=None)
sparse = [j if o else (None, j[1], None) for j, o in zip(jobs4, own)]
upsample_like = [j if o else (j[0], j[1], None) for j, o in zip(jobs4, own)]
for given in (jobs4, sparse, upsample_like):
    o4 = engine.predict_pixels_batch(given, 2, 20, FakeGP(), shard=(rank, world), recombine="owned")
    for o, r, mine_ in zip(o4, ref4, own):
        assert (o is None) == (not mine_) and (o is None or torch.equal(o, r))
ownc = engine.owned_jobs([tuple(j[1].shape) for j in jobs4], 20, 0.9, (rank, world), True)
assert sum(ownc) == 2
c4 = torch.zeros((4, 4)); c5 = torch.zeros((4, 4))
o4 = engine.predict_pixels_batch(jobs4, 2, 20, FakeGP(), chain_carry=c4, shard=(rank, world), recombine="owned")
r4 = engine.predict_pixels_batch(jobs4, 2, 20, FakeGP(), chain_carry=c5, shard=None)
for o, r, mine_ in zip(o4, r4, ownc):
    assert (o is None) == (not mine_) and (o is None or torch.equal(o, r))
assert torch.equal(c4, c5)
dist.barrier()
sys.stdout.write("rank%dok\n" % rank); sys.stdout.flush()
'''


def test_sharded_predict_pixels_equals_unsharded_world2_gloo(tmp_path):
    """The N>1 path: each rank fits its slice of the patch list, all-gather, identical recombination."""
    script = tmp_path / "e.py"
    script.write_text(_WORKER_ENGINE.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29612", str(script)],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "rank0ok" in r.stdout and "rank1ok" in r.stdout


# ----------------------------------------------------------------- bench.py starts its own ranks
_RANK_STUB = r'''
import os, sys, json, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo", rank=rank, world_size=world)
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
if "--fail" in sys.argv and rank == 1:
    sys.exit(3)
if rank == 0:
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t.item())}), flush=True)
dist.barrier()
'''


def test_bench_launcher_starts_ranks_relays_rank0_and_propagates_failures(tmp_path):
    import json
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(_RANK_STUB)
    rc, out = bench.launch_ranks(2, [], script=str(stub), timeout=120)
    assert rc == 0 and json.loads(out.strip().splitlines()[-1]) == {"n_gpus": 2, "sum": 3.0}
    rc, out = bench.launch_ranks(3, [], script=str(stub), timeout=120)
    assert rc == 0 and json.loads(out.strip().splitlines()[-1]) == {"n_gpus": 3, "sum": 6.0}
    rc, _ = bench.launch_ranks(2, ["--fail"], script=str(stub), timeout=120)
    assert rc != 0


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` on a box with fewer GPUs must fail loudly, not print n_gpus: 1."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "GPSR_LOCAL_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and '"n_gpus"' not in r.stdout


def test_deal_chains_balances_whole_chains():
    from super_resolution_amd.dist import deal_chains
    lens = [49, 80, 49, 20, 20, 16, 35, 20, 49, 49, 77, 49, 72, 48]      # patches per Set14 x3 image (633)
    for world in (1, 2, 4, 8):
        owner = deal_chains(lens, world)
        load = [sum(l for l, o in zip(lens, owner) if o == r) for r in range(world)]
        assert sum(load) == sum(lens) and max(load) <= max(max(lens), -(-sum(lens) // world) + max(lens) // 2)
