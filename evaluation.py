"""Drop-in module name of the reference (`from evaluation import Evaluator`)."""
import sys

from super_resolution_amd import evaluation as _impl

sys.modules[__name__] = _impl
