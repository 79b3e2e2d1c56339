"""Oracle: the SRGPR pipeline (CPU, serial per-patch loop).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates gprsr.py of the reference:
  predict_pixels      gprsr.py:126-210
  upsample/blur_downsample/deblur   gprsr.py:105-122
  apply_for_channel   gprsr.py:79-100
  apply               gprsr.py:48-74
`chain=True` reproduces the warm-start chain of the shared kernel singletons
(kernels.py:7-13): non-RBF kernels start each patch from the
previous patch's outputscale / lengthscale.  `chain=False` starts every patch
from the fresh-process values (raw 0) -- the definition the batched engine uses.
"""
import numpy as np

from . import cvops, gp
from . import patches as P


class OracleGPRSR:
    def __init__(self, scaling_factor, kernel_id=gp.MATERN52, color_mode="YUV", chain=False,
                 use_all_pixels=True, learning_rate=0.1, stride_percentage=0.9,
                 patch_interpolation=P.AVERAGE, iters=50, patch_filter=None):
        self.scaling_factor = scaling_factor
        self.kernel_id = kernel_id
        self.color_mode = color_mode
        self.chain = chain
        self.shared = gp.SharedKernel(kernel_id) if chain else None
        self.use_all_pixels = use_all_pixels
        self.lr = learning_rate
        self.stride_percentage = stride_percentage
        self.patch_interpolation = patch_interpolation
        self.iters = iters
        self.current_scaling_factor = None
        self.last_hypers = []          # per predict_pixels call: [B,4] raw hypers
        self.patch_filter = patch_filter  # optional callable(index)->bool, for sampled timing

    # gprsr.py:126-210
    def predict_pixels(self, training_input, training_target, test_input):
        s = self.current_scaling_factor
        assert training_input.shape == training_target.shape
        assert test_input.shape[0] == training_input.shape[0] * s
        p = int(s * 10)
        locs = P.patch_locations(training_input.shape, p, self.stride_percentage)
        idx = list(range(1, p - 1)) if self.use_all_pixels else P.grid_training_indices(p)
        sc = int(test_input.shape[0] / training_input.shape[0])
        ups, hyp = [], []
        for k, loc in enumerate(locs):
            tin = P.extract(training_input, loc, p, 1)
            ttg = P.extract(training_target, loc, p, 1)
            tst = P.extract(test_input, loc, p, sc)
            if self.patch_filter is not None and not self.patch_filter(k):
                ups.append(tst)
                continue
            if self.use_all_pixels:
                X, y = P.features(tin), P.targets(ttg)
            else:
                X, y = P.features_at(tin, idx), P.targets_at(ttg, idx)
            Xs = P.features(tst)
            mean, raw = gp.fit_predict_patch(X, y, Xs, self.kernel_id, self.shared, self.iters, self.lr)
            tst[1:-1, 1:-1] = mean.reshape(tst.shape[0] - 2, tst.shape[0] - 2)
            ups.append(tst)
            hyp.append(raw)
        self.last_hypers.append(np.array(hyp))
        up = P.combine(ups, locs, training_input.shape, p, self.patch_interpolation)
        return np.clip(up, 0, 1)

    def upsample(self, img):
        s = self.current_scaling_factor
        bicubic = cvops.resize_cubic(img, fx=s, fy=s)
        return self.predict_pixels(img, img, bicubic)

    def blur_downsample(self, img):
        s = self.current_scaling_factor
        img = cvops.gaussian_blur(img, 5, 0)
        return cvops.resize_linear(img, 1 / s, 1 / s)

    def deblur(self, blurred_high, blurred_low, low):
        return self.predict_pixels(blurred_low, low, blurred_high)

    def apply_for_channel_float(self, lr_u8):
        """gprsr.py:91-98: the float image before the truncating uint8 cast."""
        def do_algorithm(img, s):
            self.current_scaling_factor = s
            blurred_high = self.upsample(img)
            blurred_low = self.blur_downsample(blurred_high)
            return self.deblur(blurred_high, blurred_low, img)

        img = lr_u8.astype(np.double) / 255.
        s = self.scaling_factor
        while s >= 4:
            s /= 2
            img = do_algorithm(img, 2)
        return do_algorithm(img, s)

    def apply_for_channel(self, lr_u8):
        return (self.apply_for_channel_float(lr_u8) * 255).astype(np.uint8)  # gprsr.py:100

    def apply(self, lr_bgr):
        s = self.scaling_factor
        if self.color_mode == "GRAYSCALE":
            g = self.apply_for_channel(cvops.bgr2gray(lr_bgr))
            return np.stack([g, g, g], axis=-1)
        if self.color_mode == "YUV":
            yuv = cvops.bgr2yuv(lr_bgr)
            y = self.apply_for_channel(yuv[..., 0])
            u = cvops.resize_cubic(yuv[..., 1], fx=s, fy=s)
            v = cvops.resize_cubic(yuv[..., 2], fx=s, fy=s)
            return cvops.yuv2bgr(np.stack([y, u, v], axis=-1))
        chans = [self.apply_for_channel(lr_bgr[..., i]) for i in range(3)]
        return np.stack(chans, axis=-1)
