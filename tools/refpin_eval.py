"""Oracle vs the reference-held output PNGs (tests/golden/refpin): prints identical-pixel share, max |diff|, PSNR.
usage: refpin_eval.py {x2m52|x3rbf|x3m12|x3m52} {14|15} [chain|indep] [threads]"""
import sys, os, time
import numpy as np, torch
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cvops, gp as ogp, srgpr as osr

case, bits = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "chain"
torch.set_num_threads(int(sys.argv[4]) if len(sys.argv) > 4 else 3)
G = os.path.join(ROOT, "tests", "golden", "refpin")
load = lambda n: np.array(Image.open(os.path.join(G, n)))

def gray(rgb):
    b, g, r = [rgb[..., i].astype(np.int64) for i in (2, 1, 0)]
    if bits == 14:
        return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)

def report(tag, got, ref):
    d = np.abs(got.astype(int) - ref.astype(int))
    print(f"{tag}: identical {100 * (d == 0).mean():.2f}%  <=1 {100 * (d <= 1).mean():.3f}%  max {d.max()}  "
          f"PSNR(out,ref) {cvops.psnr(ref, got):.2f} dB", flush=True)

t0 = time.time()
if case == "x2m52":
    o = osr.OracleGPRSR(2, ogp.MATERN52, "GRAYSCALE", chain=(mode == "chain"))
    for i in (1, 2):
        got = o.apply_for_channel(gray(load(f"x2_lr_{i:03d}.png")))
        report(f"x2 m52 {mode} gray{bits} img{i} ({time.time() - t0:.0f}s)", got, load(f"x2_ref_m52_gray_{i:03d}.png"))
        np.save(os.path.join(ROOT, "gpurun_out", f"oracle_x2m52_{mode}_{bits}_{i}.npy"), got)
else:
    kid, tag = {"x3rbf": (ogp.RBF, "RBF"), "x3m12": (ogp.MATERN12, "matern12"), "x3m52": (ogp.MATERN52, "matern52")}[case]
    o = osr.OracleGPRSR(3, kid, "GRAYSCALE", chain=(mode == "chain"))
    got = o.apply_for_channel(gray(load("x3_lr_001.png")))
    ref = load(f"x3_ref_{tag}_001.png") if tag != "matern52" else np.array(Image.open(os.path.join(ROOT, "tests/golden/set14_x3/refout_m52_001.png")))
    report(f"x3 {tag} {mode} gray{bits} ({time.time() - t0:.0f}s)", got, ref)
    np.save(os.path.join(ROOT, "gpurun_out", f"oracle_{case}_{mode}_{bits}.npy"), got)
