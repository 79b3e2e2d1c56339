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


class _Tally:
    """Running PSNR / SSIM sums per label; averages on demand (labels keep their first-seen order)."""

    def __init__(self):
        self.sums = {}

    def add(self, label, score):
        if score.number_of_images:
            p, q, k = self.sums.get(label, (0.0, 0.0, 0))
            self.sums[label] = (p + score.psnr, q + score.ssim, k + score.number_of_images)

    def mean(self, label):
        p, q, k = self.sums.get(label, (0.0, 0.0, 0))
        return PerceptualSimilarity(p / k, q / k, 1) if k else PerceptualSimilarity(0, 0, 0)


# the rows of the reference's per-image comparison (evaluation.py:100-115): label -> where the image comes from
_FILE_ROWS = (("Bicubic (Baseline)", "bicubic", True), ("AGPR", "AGPR", False))
_FILE_LABELS = frozenset(r[0] for r in _FILE_ROWS)
_KERNEL_ROWS = (rbf_kernel, exponential_kernel, matern32_kernel, matern52_kernel, spectral_mixture_kernel, linear_kernel)


class Evaluator:
    def __init__(self, srf, image_nums=range(1, 15), verbose=False, generate_gpr_images=False, dataset_dir="Set14"):
        self.srf = srf
        self.image_nums = image_nums
        self.verbose = verbose
        self.generate_gpr_images = generate_gpr_images
        self.dataset_dir = dataset_dir
        self.clear()

    def clear(self):
        self._tally = _Tally()
        self.metrics = defaultdict(lambda: PerceptualSimilarity(0, 0, 0))

    # ------------------------------------------------------------------ files
    def _path(self, i, method):
        return f'{self.dataset_dir}/image_SRF_{self.srf}/img_{i:03d}_SRF_{self.srf}_{method}.png'

    def get_image(self, i, method, fail_on_nonexistent=False):
        path = self._path(i, method)
        if os.path.exists(path):
            return _imread(path)
        if fail_on_nonexistent:
            raise Exception(f'Image {path} does not exist')
        return None

    def get_hr_image(self, i):
        return self.get_image(i, 'HR', fail_on_nonexistent=True)

    def get_bicubic_image(self, i):
        return self.get_image(i, 'bicubic', fail_on_nonexistent=True)

    def get_agpr_image(self, i):
        return self.get_image(i, 'AGPR')

    def get_gpr_image(self, i, kernel):
        """The stored GPR_<kernel> output; with generate_gpr_images a missing one is produced (and stored) first."""
        tag = f'GPR_{get_kernel_name(kernel)}'
        image = self.get_image(i, tag)
        if image is None and self.generate_gpr_images:
            from .gprsr import GPRSR, USED_COLOR_SPACE
            image = GPRSR(self.srf, kernel, USED_COLOR_SPACE).apply(_imread(self._path(i, 'LR')))
            _imwrite(self._path(i, tag), image)
        return image

    # ------------------------------------------------------------------ metrics (evaluation.py:70-98)
    def preprocess_image(self, image):
        if len(image.shape) == 3:
            image = imgproc.bgr2gray(image)
        return image[self.srf:-self.srf, self.srf:-self.srf]

    def _prepared(self, hr_image, upsampled_image, preprocess):
        return (self.preprocess_image(hr_image), self.preprocess_image(upsampled_image)) if preprocess \
            else (hr_image, upsampled_image)

    def get_psnr(self, hr_image, upsampled_image, preprocess=False):
        return peak_signal_noise_ratio(*self._prepared(hr_image, upsampled_image, preprocess), data_range=255)

    def get_ssim(self, hr_image, upsampled_image, preprocess=False):
        return structural_similarity(*self._prepared(hr_image, upsampled_image, preprocess), data_range=255)

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

    def get_metrics(self, hr_image, upsampled_image, name):
        if upsampled_image is None:
            return PerceptualSimilarity(0, 0, 0)
        score = PerceptualSimilarity(*self._metrics_gpu(hr_image, upsampled_image))
        if self.verbose:
            print(f'{name} - PSNR: {score.psnr:.3f}, SSIM: {score.ssim:.3f}')
        return score

    # ------------------------------------------------------------------ aggregation
    def _rows_of(self, i):
        """(label, image or None) for every method the per-image comparison lists."""
        for label, tag, required in _FILE_ROWS:
            yield label, self.get_image(i, tag, fail_on_nonexistent=required)
        for kernel in _KERNEL_ROWS:
            label = get_kernel_name(kernel)
            fittable = getattr(kernel, "kernel_id", None) is not None   # spectral mixture: stored outputs only
            yield label, (self.get_gpr_image(i, kernel) if fittable or not self.generate_gpr_images
                          else self.get_image(i, f'GPR_{label}'))

    def evaluate_image(self, i):
        if self.verbose:
            print(f'Image {i}')
        hr_image = self.get_hr_image(i)
        for label, image in self._rows_of(i):
            if image is None and label not in _FILE_LABELS:   # a kernel without a stored output has no row
                continue
            score = self.get_metrics(hr_image, image, label)
            self._tally.add(label, score)
            self.metrics[label] += score

    def evaluate_method(self, method, metric=PerceptualSimilarityMetric.ALL):
        by_kernel = _is_kernel(method)
        label = get_kernel_name(method) if by_kernel else method
        if self.verbose:
            print(f'Method {label}')
        tally = _Tally()
        for i in self.image_nums:
            image = self.get_gpr_image(i, method) if by_kernel else self.get_image(i, method)
            tally.add(label, self.get_metrics(self.get_hr_image(i), image, label))
        result = tally.mean(label)
        if self.verbose:
            print(f'Average - PSNR: {result.psnr:.3f}, SSIM: {result.ssim:.3f}', flush=True)
        return {PerceptualSimilarityMetric.PSNR: result.psnr, PerceptualSimilarityMetric.SSIM: result.ssim}.get(metric, result)

    def evaluate(self):
        for i in self.image_nums:
            self.evaluate_image(i)
        for label in list(self.metrics):
            self.metrics[label] = self._tally.mean(label)
        print('Averages')
        for label, value in self.metrics.items():
            print(f'{label} - PSNR: {value.psnr:.3f}, SSIM: {value.ssim:.3f}')
        return self.metrics
