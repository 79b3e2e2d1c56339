"""Generate tests/golden/patches_*.npz by RUNNING the reference's image_patches.py.

Run in the build container only (needs /root/reference; numpy-only import):
    python tests/golden/make_patch_goldens.py
The fixtures are data (inputs + the reference's outputs); no reference source
is stored.  Cases follow SURVEY.md 8(c) G1-G3.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sys.path.insert(0, REF)
    import image_patches as ref  # the reference module (numpy only)

    # G1: patch locations, incl. duplicate-producing shapes (image_patches.py:33-41)
    g1 = {}
    for (H, W, p) in [(128, 128, 20), (180, 125, 20), (195, 293, 20), (218, 176, 30),
                      (130, 195, 30), (47, 53, 20), (20, 20, 20), (30, 31, 30), (240, 250, 20),
                      (85, 120, 30), (64, 64, 20)]:
        h = ref.PatchHandler([np.zeros((H, W))], base_patch_size=p, stride_percentage=0.9)
        g1[f"{H}_{W}_{p}"] = np.array(h.patch_locations, dtype=np.int64)
    for (H, W, p, sp) in [(100, 90, 20, 0.8), (77, 91, 20, 0.5), (64, 80, 12, 1.0)]:
        h = ref.PatchHandler([np.zeros((H, W))], base_patch_size=p, stride_percentage=sp)
        g1[f"{H}_{W}_{p}_{sp}"] = np.array(h.patch_locations, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "patches_locations.npz"), **g1)

    # G2/G3: iterator + combine on seeded cases (scale 2 and 3), both modes.
    out = {}
    for tag, (H, W, p, s) in {"s2": (47, 53, 20, 2), "s3": (61, 70, 30, 3), "s2dup": (52, 36, 20, 2)}.items():
        rng = np.random.default_rng(1000 + s + H)
        a = rng.integers(0, 256, (H, W)) / 255.0
        b = rng.integers(0, 256, (H, W)) / 255.0
        c = rng.integers(0, 256, (H * s, W * s)) / 255.0
        h = ref.PatchHandler([a, b, c], base_patch_size=p, stride_percentage=0.9)
        trip = list(h)
        out[f"{tag}_a"], out[f"{tag}_b"], out[f"{tag}_c"] = a, b, c
        out[f"{tag}_meta"] = np.array([H, W, p, s])
        out[f"{tag}_locs"] = np.array(h.patch_locations, dtype=np.int64)
        out[f"{tag}_pa"] = np.stack([t[0] for t in trip])
        out[f"{tag}_pb"] = np.stack([t[1] for t in trip])
        out[f"{tag}_pc"] = np.stack([t[2] for t in trip])
        # "predicted" patches: a seeded perturbation, rim marked, as gprsr.py:201 leaves them
        up = []
        for k, t in enumerate(trip):
            q = t[2].copy()
            q[1:-1, 1:-1] = rng.integers(0, 1024, (p * s - 2, p * s - 2)) / 1023.0
            up.append(q)
        out[f"{tag}_up"] = np.stack(up)
        out[f"{tag}_avg"] = h.combine_patches(up, ref.PatchInterpolationMode.AVERAGE)
        out[f"{tag}_bil"] = h.combine_patches(up, ref.PatchInterpolationMode.BILINEAR)
    np.savez_compressed(os.path.join(HERE, "patches_iter_combine.npz"), **out)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
