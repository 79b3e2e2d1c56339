"""Drop-in module name of the reference (`from kernels import matern52_kernel, get_kernel_name, ...`);
the same object as super_resolution_amd.kernels (the kernel singletons carry state)."""
import sys

from super_resolution_amd import kernels as _impl

sys.modules[__name__] = _impl
