"""Drop-in module name of the reference (`from image_patches import PatchHandler, PatchInterpolationMode`)."""
import sys

from super_resolution_amd import image_patches as _impl

sys.modules[__name__] = _impl
