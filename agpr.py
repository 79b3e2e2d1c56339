"""Drop-in module name of the reference (`from agpr import AGPRSuperResolution`); the same
object as super_resolution_amd.agpr, so agpr.SRF / agpr.DATASET set here are the ones used."""
import sys

from super_resolution_amd import agpr as _impl

sys.modules[__name__] = _impl
