"""AGPR: drop-in for the reference's agpr.py on the MI355X engine.

Same surface (reference agpr.py:17-39, 41-246): the module constants, the
TrainingDataset / ColorSpace enums, AGPRSuperResolution(srf, use_existing_model,
verbose) with .upscale / .upscale_channel / .iterative_back_projection /
.set_training_data / checkpoint helpers, and downscale().  Checkpoints keep the
reference's layout: checkpoints/<DATASET>/SRF_<s>/{train_x,train_y,model}.pt
(torch tensors / a GPyTorch-shaped state_dict).

What runs where:
  * fit (100 Adam steps on the linear-kernel marginal likelihood, agpr.py:75-83):
    the same batched HIP fit as SRGPR with B=1 (gpsr_fit, kernel LINEAR, d=49);
  * inference (agpr.py:165-185): the linear-kernel posterior mean is affine in the
    7x7 window, so the per-pixel GP calls collapse into ONE 49-tap filter
    (gpsr_agpr_weights + gpsr_agpr_filter) over the bicubic image;
  * back-projection (agpr.py:192-209): blur / resize / 3x3 sharpen kernels on the GPU;
  * training-set selection (agpr.py:92-146): the random candidate draw and np.unique stay host numpy as
    in the reference (unseeded np.random); the active sampling proper (agpr.py:120-137: O(N^2 D) distances,
    K-nearest-neighbour scores, 500 greedy rounds) is gpsr_active_sample on the GPU.
"""
import os
from enum import Enum
from math import floor
from time import time

import numpy as np
import torch

from . import imgproc, ops
from .kernels import GeneralModel, linear_kernel

SRF = 2

N = 5000                      # candidate patches
ACTIVE_SAMPLING_RATIO = 0.1   # share actually trained on
K = 10                        # neighbours for the representativeness score
L = 7                         # patch side
assert L % 2 == 1, "L must be odd"
padding = floor(L / 2)


class TrainingDataset(Enum):
    Set14 = 1
    BSD100 = 2


class ColorSpace(Enum):
    YCRCB = 1
    GRAYSCALE = 2


DATASET = TrainingDataset.BSD100
COLOR_SPACE = ColorSpace.YCRCB

SCALE_COEFFICIENT_BANDWIDTH = 0.2
CHARACTERISTIC_TRADEOFF = 0.2
TAU = 0.01   # iterative back projection step

#: where `checkpoints/` and the datasets live (the reference uses the working directory)
BASE_DIR = "."
#: optional folder with BSD100 HR PNGs (the reference fetches 'eugenesiow/BSD100' from the hub; no network here)
BSD100_DIR = None


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("AGPR needs an MI355X (ROCm) device: there is no CPU path in this package")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(_device())


def downscale_device(image):
    """agpr.py:244-246 on a device image.  NB: the reference divides by the MODULE-GLOBAL SRF."""
    image = imgproc.gaussian_blur(image, 3, 1)
    return imgproc.resize_cubic(image, dsize=(image.shape[1] // SRF, image.shape[0] // SRF))


def downscale(image: np.ndarray) -> np.ndarray:
    """agpr.py:244-246.  uint8 images of any channel count (agpr.py:286-287 hands it the uint8 BGR HR image) take
    OpenCV's uint8 paths channel by channel (fixed-point blur, fixed-point bicubic), float images the float64 ones."""
    image = np.asarray(image)
    if image.dtype == np.uint8:
        dev = _device()
        size = (image.shape[1] // SRF, image.shape[0] // SRF)
        planes = [image] if image.ndim == 2 else [image[..., c] for c in range(image.shape[2])]
        out = [imgproc.resize_cubic_u8_dev(imgproc.gaussian_blur_u8_dev(torch.as_tensor(np.ascontiguousarray(p)).to(dev), 3, 1),
                                           dsize=size).cpu().numpy() for p in planes]
        return out[0] if image.ndim == 2 else np.stack(out, -1)
    return downscale_device(_to_dev(image)).cpu().numpy()


class AGPRSuperResolution:
    def __init__(self, srf, use_existing_model=False, verbose=False) -> None:
        self.srf = srf
        self.verbose = verbose
        self.start_time = time()
        if not os.path.exists(self.get_checkpoint_path('model')):
            use_existing_model = False
        if use_existing_model:
            self.train_x = torch.load(self.get_checkpoint_path("train_x"), weights_only=True, map_location="cpu")
            self.train_y = torch.load(self.get_checkpoint_path("train_y"), weights_only=True, map_location="cpu")
        else:
            self.set_training_data()
        self.model = GeneralModel(linear_kernel, self.train_x.double().numpy(), self.train_y.double().numpy())
        if use_existing_model:
            self.model.load_state_dict(torch.load(self.get_checkpoint_path("model"), weights_only=True,
                                                  map_location="cpu"))
        else:
            if self.verbose:
                print("Training model...", flush=True)
            t0 = time()
            self.model.fit(iters=100, lr=0.1)
            os.makedirs(self.get_checkpoint_folder(), exist_ok=True)
            torch.save(self.model.state_dict(), self.get_checkpoint_path("model"))
            if self.verbose:
                print("Model trained in {:.2f} seconds".format(time() - t0))
        self._weights = None

    # ------------------------------------------------------------------ inference
    def filter_weights(self):
        """w[49] = outputscale * variance * X^T alpha and the bias (constant mean), on the device."""
        if self._weights is None:
            alpha = self.model.ensure_alpha()
            self._weights = ops.agpr_weights(self.model.train_x, alpha[0], self.model.raw[0].contiguous())
        return self._weights

    def upscale(self, image: np.ndarray) -> np.ndarray:
        # the colour image goes to the GPU once: conversion, chroma resize and merge are HIP kernels (agpr.py:148-158)
        return self.upscale_device(torch.as_tensor(np.ascontiguousarray(image, dtype=np.uint8)).to(_device())).cpu().numpy()

    def upscale_device(self, bgr):
        """uint8 BGR device image -> device result of upscale() (agpr.py:148-158): uint8 BGR in YCrCb mode, float64
        grey * 255 in GRAYSCALE mode."""
        if COLOR_SPACE == ColorSpace.YCRCB:
            y, cr, cb = ops.bgr2ycrcb(bgr)
            y = ops.unit_to_u8(self.upscale_channel_device(ops.u8_to_unit(y)), 1.0)     # .astype(np.uint8), agpr.py:151
            cr = imgproc.resize_cubic_u8_dev(cr, dsize=(y.shape[1], y.shape[0]))
            cb = imgproc.resize_cubic_u8_dev(cb, dsize=(y.shape[1], y.shape[0]))
            return ops.ycrcb2bgr(y, cr, cb)
        elif COLOR_SPACE == ColorSpace.GRAYSCALE:
            return self.upscale_channel_device(ops.u8_to_unit(ops.bgr2gray(bgr)))

    def upscale_channel(self, image: np.ndarray) -> np.ndarray:
        return self.upscale_channel_device(_to_dev(image / 255.)).cpu().numpy()

    def upscale_channel_device(self, image):
        """device float64 LR image in [0,1] -> device float64 HR image * 255 (agpr.py:160-190)."""
        up = imgproc.resize_cubic(image, dsize=(image.shape[1] * self.srf, image.shape[0] * self.srf))
        pred = ops.agpr_filter(up, L, self.filter_weights(), self.model.raw[0].contiguous())
        pred = self.iterative_back_projection_device(pred, image)
        return torch.clamp(pred, 0.0, 1.0) * 255

    def iterative_back_projection(self, interpolated_image: np.ndarray, original_image: np.ndarray) -> np.ndarray:
        return self.iterative_back_projection_device(_to_dev(interpolated_image), _to_dev(original_image)).cpu().numpy()

    def iterative_back_projection_device(self, interpolated, original):
        max_iterations = 50
        kern = [[-1, -1, -1], [-1, 9, -1], [-1, -1, -1]]
        cur = interpolated
        current_error = np.inf
        for _ in range(max_iterations):
            down = downscale_device(cur)
            diff = original - down
            err = float((diff * diff).sum())            # np.linalg.norm(difference)**2
            if current_error - 0.01 <= err:
                break
            current_error = err
            up = imgproc.resize_cubic(diff.contiguous(), dsize=(cur.shape[1], cur.shape[0]))
            cur = ops.axpy(cur, imgproc.filter2d_3x3(up, kern), TAU)
        return cur

    # ------------------------------------------------------------------ training data (host, as the reference)
    def set_training_data(self) -> None:
        t0 = time()
        hr_images = self.get_training_images("HR")
        hr_images = [h if h.shape[0] % self.srf == 0 else h[:-(h.shape[0] % self.srf)] for h in hr_images]
        hr_images = [h if h.shape[1] % self.srf == 0 else h[:, :-(h.shape[1] % self.srf)] for h in hr_images]
        lr_images = [downscale(h) for h in hr_images]
        bicubic_images = [imgproc.resize_cubic(_to_dev(l), dsize=(l.shape[1] * self.srf, l.shape[0] * self.srf)).cpu().numpy()
                          for l in lr_images]
        rows = []
        for _ in range(N):
            k = np.random.randint(0, len(hr_images))
            hr, bic = hr_images[k], bicubic_images[k]
            x = np.random.randint(padding, hr.shape[1] - padding)
            y = np.random.randint(padding, hr.shape[0] - padding)
            patch = bic[y - padding:y + padding + 1, x - padding:x + padding + 1].reshape(-1)
            rows.append(np.concatenate((patch, [hr[y, x]])))
        self.original_dataset = np.unique(np.array(rows), axis=0)
        if ACTIVE_SAMPLING_RATIO == 1:
            self.dataset = np.copy(self.original_dataset)
        else:
            self.dataset = self._active_sampling(self.original_dataset)
        self.train_x = torch.tensor(np.array([d[:-1] for d in self.dataset]))
        self.train_y = torch.tensor(np.array([d[-1] for d in self.dataset]))
        os.makedirs(self.get_checkpoint_folder(), exist_ok=True)
        torch.save(self.train_x, self.get_checkpoint_path("train_x"))
        torch.save(self.train_y, self.get_checkpoint_path("train_y"))
        if self.verbose:
            print("Training data set in {:.2f} seconds".format(time() - t0))

    def _active_sampling(self, data):
        """agpr.py:122-137 on the GPU (gpsr_active_sample): distances, representativeness scores and the greedy loop
        are three launches; the reference's habits that decide the choice are kept (scores indexed by the position
        in the shrinking pool, numpy's summation order, first maximum wins).  The bandwidth (agpr.py:125) is a
        median of the INDEX arrays np.where returns, not of the distances: every index 0..N-1 occurs 2(N-1)
        times in them (all distances off the diagonal are finite), so it is 0.2 * (N-1)/2 whatever the data."""
        n = len(data)
        self.bandwidth = SCALE_COEFFICIENT_BANDWIDTH * ((n - 1) / 2.0)
        idx, rep = ops.active_sample(_to_dev(data), K, floor(ACTIVE_SAMPLING_RATIO * N), self.bandwidth,
                                     CHARACTERISTIC_TRADEOFF, 1 - CHARACTERISTIC_TRADEOFF)
        self.representativeness_scores = rep.cpu().numpy()
        self.active_indices = idx.cpu().numpy()
        return [data[i] for i in self.active_indices]

    def get_characteristic_score(self, i, datapoint) -> float:
        return (CHARACTERISTIC_TRADEOFF * self.representativeness_scores[i]
                + (1 - CHARACTERISTIC_TRADEOFF) * self.get_diversity(datapoint))

    def get_diversity(self, datapoint) -> float:
        if len(self.dataset) == 0:
            return 0
        d = np.sum((np.asarray(self.dataset) - datapoint) ** 2, axis=1)
        return np.min(-np.exp(-d / (2 * self.bandwidth)))

    def get_training_images(self, image_type):
        from PIL import Image
        if DATASET == TrainingDataset.Set14:
            paths = [os.path.join(BASE_DIR, f"Set14/image_SRF_{self.srf}/img_{i:03d}_SRF_{self.srf}_{image_type}.png")
                     for i in range(1, 15)]
        else:
            if BSD100_DIR is None:
                raise RuntimeError("BSD100 is fetched from the hub by the reference (agpr.py:228); offline, set "
                                   "agpr.BSD100_DIR to a folder of HR PNGs or use the committed checkpoints "
                                   "(use_existing_model=True)")
            paths = sorted(os.path.join(BSD100_DIR, f) for f in os.listdir(BSD100_DIR) if f.lower().endswith(".png"))
        images = [np.ascontiguousarray(np.array(Image.open(p).convert("RGB"))[:, :, ::-1]) for p in paths]
        if COLOR_SPACE == ColorSpace.YCRCB:
            return [imgproc.bgr2ycrcb(im)[0] / 255 for im in images]
        return [imgproc.bgr2gray(im) / 255 for im in images]

    def get_checkpoint_path(self, filename):
        return os.path.join(self.get_checkpoint_folder(), filename + ".pt")

    def get_checkpoint_folder(self):
        return os.path.join(BASE_DIR, f"checkpoints/{DATASET.name}/SRF_{self.srf}/")
