"""Drop-in module name of the reference (`from gprsr import GPRSR, ColorSpace, SRF, ...`).
This module IS super_resolution_amd.gprsr (same object), so option constants set on it
(gprsr.STRIDE_PERCENTAGE = 0.8, ...) are the ones the engine reads.  Run as a script it
behaves like the reference's `python gprsr.py` (gprsr.py:217-227): see main.py."""
import sys

from super_resolution_amd import gprsr as _impl

if __name__ == "__main__":
    import main
    main.main(sys.argv[1:])
else:
    sys.modules[__name__] = _impl
