"""Evaluation: drop-in for the reference's evaluation.py without cv2 / scikit-image.

Same surface (reference evaluation.py:12-149): PerceptualSimilarityMetric,
PerceptualSimilarity, Evaluator(srf, image_nums, verbose, generate_gpr_images) with
get_metrics / get_psnr / get_ssim / preprocess_image / evaluate_image /
evaluate_method / evaluate.  PSNR and SSIM follow skimage.metrics
(peak_signal_noise_ratio, structural_similarity with its defaults: 7x7 uniform
window, sample covariance, K1=.01, K2=.03) on the grey image cropped by srf pixels.
Evaluator.get_metrics runs them on the GPU (gpsr_bgr2gray + gpsr_metrics: integer window sums,
fixed-order reductions); the numpy functions peak_signal_noise_ratio / structural_similarity below
are the host-side equivalents for callers that hold numpy arrays and no device.
"""
import os
from collections import defaultdict
from enum import Enum
from typing import NamedTuple

import numpy as np

from . import imgproc
from .kernels import (exponential_kernel, get_kernel_name, linear_kernel, matern32_kernel, matern52_kernel,
                      rbf_kernel, spectral_mixture_kernel, ScaleKernel, _Base)


class PerceptualSimilarityMetric(Enum):
    PSNR = 1
    SSIM = 2
    ALL = 3


class PerceptualSimilarity(NamedTuple):
    psnr: float
    ssim: float
    number_of_images: int = 1

    def __add__(self, other):
        return PerceptualSimilarity(self.psnr + other.psnr, self.ssim + other.ssim,
                                    self.number_of_images + other.number_of_images)

    def average(self):
        if self.number_of_images == 0:
            return PerceptualSimilarity(0, 0, 0)
        return PerceptualSimilarity(self.psnr / self.number_of_images, self.ssim / self.number_of_images, 1)


def _is_kernel(obj):
    return isinstance(obj, (ScaleKernel, _Base))


def _box_mean(x, win):
    """Mean over win x win windows with scipy's 'reflect' boundary, via an integral image."""
    pad = win // 2
    xp = np.pad(x, pad, mode="symmetric")
    c = np.cumsum(np.cumsum(xp, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    H, W = x.shape
    s = c[win:win + H, win:win + W] - c[:H, win:win + W] - c[win:win + H, :W] + c[:H, :W]
    return s / (win * win)


def peak_signal_noise_ratio(a, b, data_range=255):
    err = np.mean((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) ** 2)
    return 10 * np.log10((data_range ** 2) / err)


def structural_similarity(a, b, data_range=255, win_size=7):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    NP = win_size * win_size
    cov_norm = NP / (NP - 1)
    ux, uy = _box_mean(a, win_size), _box_mean(b, win_size)
    vx = cov_norm * (_box_mean(a * a, win_size) - ux * ux)
    vy = cov_norm * (_box_mean(b * b, win_size) - uy * uy)
    vxy = cov_norm * (_box_mean(a * b, win_size) - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2))
    p = (win_size - 1) // 2
    return S[p:-p, p:-p].mean()


def _imread(path):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])


def _imwrite(path, bgr):
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path)


class Evaluator:
    def __init__(self, srf, image_nums=range(1, 15), verbose=False, generate_gpr_images=False, dataset_dir="Set14"):
        self.srf = srf
        self.image_nums = image_nums
        self.verbose = verbose
        self.metrics = defaultdict(lambda: PerceptualSimilarity(0, 0, 0))
        self.generate_gpr_images = generate_gpr_images
        self.dataset_dir = dataset_dir

    def clear(self):
        self.metrics = defaultdict(lambda: PerceptualSimilarity(0, 0, 0))

    def _path(self, i, method):
        return f'{self.dataset_dir}/image_SRF_{self.srf}/img_{i:03d}_SRF_{self.srf}_{method}.png'

    def get_hr_image(self, i):
        return self.get_image(i, 'HR', fail_on_nonexistent=True)

    def get_bicubic_image(self, i):
        return self.get_image(i, 'bicubic', fail_on_nonexistent=True)

    def get_agpr_image(self, i):
        return self.get_image(i, 'AGPR')

    def get_gpr_image(self, i, kernel):
        kernel_name = get_kernel_name(kernel)
        image = self.get_image(i, f'GPR_{kernel_name}')
        if image is None and self.generate_gpr_images:
            from .gprsr import GPRSR, USED_COLOR_SPACE
            image = GPRSR(self.srf, kernel, USED_COLOR_SPACE).apply(_imread(self._path(i, 'LR')))
            _imwrite(self._path(i, f'GPR_{kernel_name}'), image)
        return image

    def get_image(self, i, method, fail_on_nonexistent=False):
        path = self._path(i, method)
        if not os.path.exists(path):
            if fail_on_nonexistent:
                raise Exception(f'Image {path} does not exist')
            return None
        return _imread(path)

    def get_metrics(self, hr_image, upsampled_image, name):
        if upsampled_image is None:
            return PerceptualSimilarity(0, 0, 0)
        psnr, ssim = self._metrics_gpu(hr_image, upsampled_image)
        if self.verbose:
            print(f'{name} - PSNR: {psnr:.3f}, SSIM: {ssim:.3f}')
        return PerceptualSimilarity(psnr, ssim)

    def get_psnr(self, hr_image, upsampled_image, preprocess=False):
        if preprocess:
            hr_image, upsampled_image = self.preprocess_image(hr_image), self.preprocess_image(upsampled_image)
        return peak_signal_noise_ratio(hr_image, upsampled_image, data_range=255)

    def get_ssim(self, hr_image, upsampled_image, preprocess=False):
        if preprocess:
            hr_image, upsampled_image = self.preprocess_image(hr_image), self.preprocess_image(upsampled_image)
        return structural_similarity(hr_image, upsampled_image, data_range=255)

    def _metrics_gpu(self, hr_image, upsampled_image):
        """PSNR and SSIM of two images as read from disk (BGR or grey uint8): grey conversion, crop by srf and both
        reductions on the device (evaluation.py:83-98)."""
        import torch
        from . import ops
        dev = torch.device("cuda", torch.cuda.current_device())

        def grey(img):
            t = torch.as_tensor(np.ascontiguousarray(img, dtype=np.uint8)).to(dev)
            return ops.bgr2gray(t) if t.dim() == 3 else t
        m = ops.metrics(grey(hr_image), grey(upsampled_image), self.srf).cpu().numpy()
        return float(m[0]), float(m[1])

    def preprocess_image(self, image):
        if len(image.shape) == 3:
            image = imgproc.bgr2gray(image)
        return image[self.srf:-self.srf, self.srf:-self.srf]

    def evaluate_image(self, i):
        if self.verbose:
            print(f'Image {i}')
        hr_image = self.get_hr_image(i)
        self.metrics['Bicubic (Baseline)'] += self.get_metrics(hr_image, self.get_bicubic_image(i), 'Bicubic (Baseline)')
        self.metrics['AGPR'] += self.get_metrics(hr_image, self.get_agpr_image(i), 'AGPR')
        for kernel in [rbf_kernel, exponential_kernel, matern32_kernel, matern52_kernel, spectral_mixture_kernel,
                       linear_kernel]:
            name = get_kernel_name(kernel)
            img = self.get_gpr_image(i, kernel) if (not self.generate_gpr_images or getattr(kernel, "kernel_id", None)
                                                    is not None) else self.get_image(i, f'GPR_{name}')
            if img is not None:
                self.metrics[name] += self.get_metrics(hr_image, img, name)

    def evaluate_method(self, method, metric=PerceptualSimilarityMetric.ALL):
        kernel_name = get_kernel_name(method) if _is_kernel(method) else method
        if self.verbose:
            print(f'Method {kernel_name}')
        result = PerceptualSimilarity(0, 0, 0)
        for i in self.image_nums:
            hr_image = self.get_hr_image(i)
            up = self.get_gpr_image(i, method) if _is_kernel(method) else self.get_image(i, method)
            result += self.get_metrics(hr_image, up, kernel_name)
        result = result.average()
        if self.verbose:
            print(f'Average - PSNR: {result.psnr:.3f}, SSIM: {result.ssim:.3f}', flush=True)
        if metric == PerceptualSimilarityMetric.PSNR:
            return result.psnr
        elif metric == PerceptualSimilarityMetric.SSIM:
            return result.ssim
        return result

    def evaluate(self):
        for i in self.image_nums:
            self.evaluate_image(i)
        for key, value in self.metrics.items():
            self.metrics[key] = value.average()
        print('Averages')
        for key, value in self.metrics.items():
            print(f'{key} - PSNR: {value.psnr:.3f}, SSIM: {value.ssim:.3f}')
        return self.metrics
