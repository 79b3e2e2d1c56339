#!/usr/bin/env python3
"""main.py -- the entry point the reference's README names (README.md:22) but never
shipped: its options are the module-level constants of gprsr.py:17-26, exposed here
as flags.  Loops over Set14 like gprsr.py:217-227 and writes
<dataset>/image_SRF_<s>/img_XXX_SRF_<s>_GPR_<kernel>[_gray].png.

    python main.py --srf 3 --kernel matern52 --color-space grayscale --images 1-14
"""
import argparse
import os
import sys
import time

import numpy as np

KERNELS = ("rbf", "matern52", "matern32", "exponential", "linear")


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="GPR super-resolution (SRGPR) on MI355X")
    ap.add_argument("--srf", type=int, default=2, help="SRF: scaling factor (2, 3, 4)")
    ap.add_argument("--images", default="1-14", help="IMAGE_NUMS: e.g. 1-14 or 1,3,5")
    ap.add_argument("--no-timing", action="store_true", help="DO_TIMING = False")
    ap.add_argument("--color-space", default="grayscale", choices=["bgr", "yuv", "grayscale"], help="USED_COLOR_SPACE")
    ap.add_argument("--kernel", default="matern52", choices=KERNELS, help="USED_KERNEL")
    ap.add_argument("--grid-training", action="store_true", help="USE_ALL_PIXELS_FOR_TRAINING = False")
    ap.add_argument("--learning-rate", type=float, default=0.1, help="LEARNING_RATE")
    ap.add_argument("--stride-percentage", type=float, default=0.9, help="STRIDE_PERCENTAGE")
    ap.add_argument("--patch-interpolation", default="average", choices=["average", "bilinear"],
                    help="PATCH_INTERPOLATION")
    ap.add_argument("--warm-start", default="chain", choices=["chain", "chain_per_image", "independent"],
                    help="chain (default): the reference's semantics -- its kernel objects are singletons, so every patch "
                         "starts Adam from the previous patch's optimum, across patches, phases and images (serial); "
                         "chain_per_image: every image as the first image of a fresh reference process (images run "
                         "concurrently); independent: every patch from the initial values -- the batched, patch-sharded "
                         "engine, a different algorithm at pixel level for non-RBF kernels")
    ap.add_argument("--batch-images", action="store_true", help="fit the patches of all images in one GP batch")
    ap.add_argument("--dataset-dir", default="Set14", help="folder holding image_SRF_<s>/img_XXX_SRF_<s>_LR.png")
    ap.add_argument("--evaluate", action="store_true", help="print PSNR/SSIM against *_HR.png (evaluation.py)")
    return ap.parse_args(argv)


def image_numbers(spec):
    out = []
    for part in spec.split(","):
        if "-" in part:
            lo, hi = part.split("-")
            out.extend(range(int(lo), int(hi) + 1))
        else:
            out.append(int(part))
    return out


def output_path(dataset_dir, srf, i, kernel_name, gray):
    return f'{dataset_dir}/image_SRF_{srf}/img_{i:03d}_SRF_{srf}_GPR_{kernel_name}{"_gray" if gray else ""}.png'


def imread_bgr(path):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])


def imwrite_bgr(path, bgr):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path)


def main(argv=None):
    a = parse_args(argv)
    from super_resolution_amd import gprsr, kernels, imgproc
    from super_resolution_amd.image_patches import PatchInterpolationMode
    kern = {"rbf": kernels.rbf_kernel, "matern52": kernels.matern52_kernel, "matern32": kernels.matern32_kernel,
            "exponential": kernels.exponential_kernel, "linear": kernels.linear_kernel}[a.kernel]
    cs = {"bgr": gprsr.ColorSpace.BGR, "yuv": gprsr.ColorSpace.YUV, "grayscale": gprsr.ColorSpace.GRAYSCALE}[a.color_space]
    gprsr.SRF, gprsr.USED_KERNEL, gprsr.USED_COLOR_SPACE = a.srf, kern, cs
    gprsr.DO_TIMING = not a.no_timing
    gprsr.USE_ALL_PIXELS_FOR_TRAINING = not a.grid_training
    gprsr.LEARNING_RATE = a.learning_rate
    gprsr.STRIDE_PERCENTAGE = a.stride_percentage
    gprsr.PATCH_INTERPOLATION = PatchInterpolationMode[a.patch_interpolation.upper()]
    nums = image_numbers(a.images)
    gprsr.IMAGE_NUMS = nums
    g = gprsr.GPRSR(a.srf, kern, cs, warm_start=a.warm_start)
    name = kernels.get_kernel_name(kern)
    gray = cs == gprsr.ColorSpace.GRAYSCALE
    paths = [(i, f"{a.dataset_dir}/image_SRF_{a.srf}/img_{i:03d}_SRF_{a.srf}_LR.png") for i in nums]
    if a.batch_images and gray and a.warm_start != "chain":
        t0 = time.time()
        outs = g.apply_for_channel_batch([imgproc.bgr2gray(imread_bgr(p)) for _, p in paths])
        if gprsr.DO_TIMING:
            print("Upscaling %d images took %s seconds" % (len(outs), time.time() - t0), flush=True)
        results = [(i, np.stack([o, o, o], axis=-1)) for (i, _), o in zip(paths, outs)]
    else:
        results = [(i, g.apply(imread_bgr(p))) for i, p in paths]
    for i, img in results:
        imwrite_bgr(output_path(a.dataset_dir, a.srf, i, name, gray), img)
    if a.evaluate:
        from super_resolution_amd.evaluation import Evaluator
        ev = Evaluator(a.srf, image_nums=nums, verbose=True, dataset_dir=a.dataset_dir)
        ev.evaluate_method(f'GPR_{name}{"_gray" if gray else ""}')
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
