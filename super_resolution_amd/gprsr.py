"""GPRSR: drop-in for the reference's gprsr.py on the batched MI355X engine.

Same surface as the reference (gprsr.py:12-26, 37-210): the ColorSpace enum, the
module-level option constants, extract_patch / extract_neighbors, and
GPRSR(scaling_factor, used_kernel, color_mode) with apply / apply_for_channel /
upsample / blur_downsample / deblur / predict_pixels.  numpy arrays in, fresh
numpy arrays out; internally the image goes to the GPU once and stays there
through upsample -> blur/downsample -> deblur and through x4 = x2 o x2.

Additional (keyword-only) knobs of the engine: warm_start ("chain" = the reference's
semantics and the default | "chain_per_image" | "independent", see class GPRSR), iters, and the batch entry points
apply_for_channel_batch / apply_batch that fit the patches of many images in
one GP batch.
"""
import time
from enum import Enum

import numpy as np
import torch

from . import engine, imgproc, ops
from .image_patches import PatchHandler, PatchInterpolationMode  # noqa: F401  (re-exported like the reference)
from .kernels import (GeneralModel, get_kernel_name, kernel_id_of, linear_kernel,  # noqa: F401
                      matern32_kernel, matern52_kernel)


class ColorSpace(Enum):
    BGR = 0
    YUV = 1
    GRAYSCALE = 2


# Options (module constants in the reference, gprsr.py:17-26; main.py exposes them as flags)
SRF = 2
IMAGE_NUMS = range(1, 15)
DO_TIMING = True
USED_COLOR_SPACE = ColorSpace.GRAYSCALE
USED_KERNEL = matern52_kernel
USE_ALL_PIXELS_FOR_TRAINING = True
LEARNING_RATE = 0.1
STRIDE_PERCENTAGE = 0.9
PATCH_INTERPOLATION = PatchInterpolationMode.AVERAGE


def extract_patch(img, y, x, patch_size):
    return img[y:y + patch_size, x:x + patch_size].copy()


def extract_neighbors(img, y, x):
    """8-neighbour feature vector in the reference's order (gprsr.py:35)."""
    return [img[y + 1, x], img[y - 1, x], img[y, x + 1], img[y, x - 1],
            img[y + 1, x + 1], img[y - 1, x - 1], img[y + 1, x - 1], img[y - 1, x + 1]]


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("GPRSR needs an MI355X (ROCm) device: there is no CPU path in this package")
    return torch.device("cuda", torch.cuda.current_device())


class GPRSR:
    """warm_start (keyword-only; the reference has no such switch, it IS "chain"):
      "chain"            the reference's behaviour: its kernel objects are module singletons (kernels.py:7-13)
                         and only RBF re-initialises outputscale / lengthscale per patch (gprsr.py:167-172), so every
                         patch starts Adam from the previous patch's optimum -- across patches, the two phases,
                         channels, images and GPRSR instances, through `used_kernel`.  Serial by definition.
      "chain_per_image"  every image is processed as the reference processes the FIRST image of a fresh process
                         (chain from the initial raw 0, the singleton is neither read nor written): the images of
                         a batch are independent chains and link k of all of them is one batched launch sequence.
      "independent"      every patch starts from the initial raw 0: the batched / patch-sharded engine of the
                         north star.  A different algorithm at pixel level for non-RBF kernels.
    For RBF the three coincide."""

    def __init__(self, scaling_factor, used_kernel=matern52_kernel, color_mode=ColorSpace.YUV, *,
                 warm_start="chain", iters=50, shard=None, max_workspace_bytes=48 << 30, outputs="all",
                 patch_size=None):
        assert warm_start in ("independent", "chain", "chain_per_image")
        assert outputs in ("all", "owned")
        # outputs (sharded runs only): "all" = every rank returns every HR image; "owned" = a rank returns the
        # images it fitted patches of and None for the others (data-parallel deployments, bench.py's weak scaling:
        # the ranks then skip recombination and image stages of each other's images in both phases)
        self.outputs = outputs
        # patch_size: LR patch edge instead of the reference's int(10 * s) (gprsr.py:132) -- an engine knob for larger
        # GPs (p = 34 -> n = 1024, bench.py --workload p34); None = the reference's rule
        self.patch_size = patch_size
        self.scaling_factor = scaling_factor
        self.current_scaling_factor = None
        self.used_kernel = used_kernel
        self.color_mode = color_mode
        self.warm_start = warm_start
        self.iters = iters
        self.shard = shard
        self.stats = engine.FitStats()
        self.check_failures = True
        self._carry = None          # [n_images,4] float32 device tensor while a chained call runs
        self._scoped_carry = None   # chain_per_image: the carry of the image apply() is working on (BGR: 3 channels)
        self._last_round = True     # x4 = x2 o x2: False while the first x2 round runs
        self._gp = engine.BatchedGP(kernel_id_of(used_kernel), iters=iters, lr=LEARNING_RATE,
                                    max_workspace_bytes=max_workspace_bytes)

    def _chained(self):
        return self.warm_start != "independent" and self._gp.kernel_id != ops.RBF

    # ------------------------------------------------------------------ colour handling (gprsr.py:48-74)
    def apply(self, lr_image):
        if DO_TIMING:
            start_time = time.time()
        if self.warm_start == "chain_per_image" and self._chained():
            self._scoped_carry = torch.zeros((1, 4), dtype=torch.float32, device=_device())
        try:
            # the colour image goes to the GPU once; conversion, chroma resize and merge are HIP kernels
            bgr = torch.as_tensor(np.ascontiguousarray(lr_image, dtype=np.uint8)).to(_device())
            sf = self.scaling_factor
            if self.color_mode == ColorSpace.GRAYSCALE:
                g = self._channel_u8(ops.bgr2gray(bgr))
                img_out = torch.stack([g, g, g], dim=-1)                      # cv.COLOR_GRAY2BGR
            elif self.color_mode == ColorSpace.YUV:
                y, u, v = ops.bgr2yuv(bgr)
                y = self._channel_u8(y)
                u = imgproc.resize_cubic_u8_dev(u, fx=sf, fy=sf)
                v = imgproc.resize_cubic_u8_dev(v, fx=sf, fy=sf)
                img_out = ops.yuv2bgr(y, u, v)
            else:
                img_out = torch.stack([self._channel_u8(bgr[..., i].contiguous()) for i in range(3)], dim=-1)
            img_out = img_out.cpu().numpy()
        finally:
            self._scoped_carry = None
        if DO_TIMING:
            print("Upscaling took %s seconds" % (time.time() - start_time), flush=True)
        return img_out

    # ------------------------------------------------------------------ one channel (gprsr.py:79-100)
    def apply_for_channel(self, lr_image_channel):
        return self.apply_for_channel_batch([lr_image_channel])[0]

    def _channel_u8(self, ch):
        """uint8 device plane -> uint8 device HR plane (gprsr.py:91-100 without leaving the GPU)."""
        return ops.unit_to_u8(self.apply_for_channel_device([ops.u8_to_unit(ch)])[0], 255.0)

    def apply_for_channel_batch(self, channels):
        """Several uint8 channels/images at once: their patches share the GP batches."""
        dev = _device()
        imgs = [ops.u8_to_unit(torch.as_tensor(np.ascontiguousarray(c, dtype=np.uint8)).to(dev)) for c in channels]
        outs = self.apply_for_channel_device(imgs)
        res = [ops.unit_to_u8(o, 255.0).cpu().numpy() for o in outs]
        return res

    def apply_for_channel_device(self, imgs):
        """Device float64 images in [0,1] -> device float64 HR images (before the uint8 cast).
        self.stats holds the raw hyper-parameters and failure flags of this call's predict_pixels passes;
        a flagged GP raises engine.NotPSDError (the reference's failure mode) unless check_failures is False."""
        self.stats = engine.FitStats()
        if not self._chained():
            outs = self._run(imgs, None)
        elif self.warm_start == "chain":
            # image after image, each continuing the chain the kernel singleton holds (gprsr.py:217-227 loop)
            outs = []
            for im in imgs:
                carry = self._singleton_carry(im.device)
                outs.append(self._run([im], carry)[0])
                self._store_singleton(carry)
        else:
            carry = self._scoped_carry if (self._scoped_carry is not None and len(imgs) == 1) else \
                torch.zeros((len(imgs), 4), dtype=torch.float32, device=imgs[0].device)
            outs = self._run(imgs, carry)
        if self.check_failures:
            bad = [int(f.sum()) for f in self.stats.flags]   # one device sync, at the end of the call
            if any(bad):
                raise engine.NotPSDError(f"{sum(bad)} patch GP(s) hit a non-positive pivot "
                                         f"(per predict_pixels pass: {bad}); see GPRSR.stats.flags")
        return outs

    def _singleton_carry(self, dev):
        k = self.used_kernel
        return torch.tensor([[0.0, 0.0, k.raw_outputscale, k.base_kernel.raw_parameter]], dtype=torch.float32, device=dev)

    def _store_singleton(self, carry):
        last = carry[0].cpu()
        self.used_kernel.raw_outputscale = float(last[2])
        self.used_kernel.base_kernel.raw_parameter = float(last[3])

    def _run(self, imgs, carry):
        """gprsr.py:91-98 on device images: x s recursion, upsample -> blur/downsample -> deblur."""
        self._carry = carry
        try:
            scaling_factor = self.scaling_factor
            self._last_round = False
            while scaling_factor >= 4:
                scaling_factor /= 2
                imgs = self._do_algorithm(imgs, 2)   # every rank needs every intermediate image: the next round starts from it
            self._last_round = True
            return self._do_algorithm(imgs, scaling_factor)
        finally:
            self._carry = None

    def _do_algorithm(self, imgs, scaling_factor):
        self.current_scaling_factor = scaling_factor
        # outputs="owned": a rank then needs the phase-1 image only where it has phase-2 patches of its own (None
        # elsewhere); with outputs="all" it recombines every image in phase 2, on top of the phase-1 image
        rc = self.outputs if self._last_round else "all"
        blurred_high = self._upsample(imgs, recombine=rc)
        blurred_low = [None if b is None else self._blur_downsample(b) for b in blurred_high]
        return self._predict_pixels(list(zip(blurred_low, imgs, blurred_high)), recombine=rc)

    def _s_int(self):
        s = self.current_scaling_factor
        assert float(s) == int(s), "the patch pipeline needs an integer step factor (2 or 3)"
        return int(s)

    def _p(self):
        return int(self.patch_size) if self.patch_size is not None else int(self.current_scaling_factor * 10)

    def _upsample(self, imgs, recombine="all"):
        s = self.current_scaling_factor
        own = [True] * len(imgs)
        if recombine == "owned" and self.shard is not None:   # the bicubic image of another rank's image is never read
            own = engine.owned_jobs([tuple(im.shape) for im in imgs], self._p(), STRIDE_PERCENTAGE, self.shard,
                                    self._carry is not None)
        bic = [imgproc.resize_cubic(im, fx=s, fy=s) if o else None for im, o in zip(imgs, own)]
        return self._predict_pixels(list(zip(imgs, imgs, bic)), recombine=recombine)

    def _blur_downsample(self, img):
        s = self.current_scaling_factor
        return imgproc.resize_linear(imgproc.gaussian_blur(img, 5, 0), 1 / s, 1 / s)

    def _predict_pixels(self, jobs, recombine="all"):
        s = self._s_int()
        p = self._p()
        mode = PATCH_INTERPOLATION.value
        self._gp.lr = LEARNING_RATE     # the reference reads the constant per patch (gprsr.py:178): read per call, not at construction
        return engine.predict_pixels_batch(jobs, s, p, self._gp, STRIDE_PERCENTAGE, USE_ALL_PIXELS_FOR_TRAINING,
                                           mode, chain_carry=self._carry, stats=self.stats, shard=self.shard,
                                           recombine=recombine if self.shard is not None else "all")

    # ------------------------------------------------------------------ reference-named numpy methods
    def _to_dev(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(_device())

    def _stage(self, fn):
        """A reference-named stage called on its own: chained modes read and write the kernel singleton
        ("chain") or start from raw 0 ("chain_per_image"), like one predict_pixels call of the reference."""
        if not self._chained() or self._carry is not None:
            return fn()
        dev = _device()
        self._carry = self._singleton_carry(dev) if self.warm_start == "chain" else \
            torch.zeros((1, 4), dtype=torch.float32, device=dev)
        try:
            out = fn()
            if self.warm_start == "chain":
                self._store_singleton(self._carry)
            return out
        finally:
            self._carry = None

    def upsample(self, img):
        return self._stage(lambda: self._upsample([self._to_dev(img)])[0].cpu().numpy())

    def blur_downsample(self, img):
        return self._blur_downsample(self._to_dev(img)).cpu().numpy()

    def deblur(self, blurred_high, blurred_low, low):
        return self.predict_pixels(blurred_low, low, blurred_high)

    def predict_pixels(self, training_input, training_target, test_input):
        assert training_input.shape == training_target.shape
        assert test_input.shape[0] == training_input.shape[0] * self.current_scaling_factor
        job = (self._to_dev(training_input), self._to_dev(training_target), self._to_dev(test_input))
        return self._stage(lambda: self._predict_pixels([job])[0].cpu().numpy())
