"""MI355X-native batched Gaussian-process super-resolution (SRGPR / AGPR).

Drop-in for the per-patch GP path of fynsta/Super-resolution (gprsr.py,
kernels.py, image_patches.py, agpr.py).  All arithmetic on the path runs in
hand-written HIP kernels for gfx950 behind the C ABI of include/gpsr.h.
"""
__version__ = "0.1"
