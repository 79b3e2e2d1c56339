"""ctypes binding of libgpsr.so (the C ABI declared in include/gpsr.h).

The HIP library is the product path: there is NO CPU fallback.  If the shared
object is missing or cannot be loaded, every compute entry point raises.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPSR_LIB", os.path.join(_HERE, "libgpsr.so"))   # override: instrumented debug builds (tools/upd_trace.py)
SRC = os.path.join(_HERE, "csrc", "gpsr_api.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-ffp-contract=off"]

_lib = None

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/gpsr.h declares
SIGNATURES = {
    "gpsr_version": (ctypes.c_char_p, []),
    "gpsr_last_error": (ctypes.c_char_p, []),
    "gpsr_padded_n": (c_int, [c_int]),
    "gpsr_create": (c_void_p, [c_int, c_int, c_int, c_int]),
    "gpsr_destroy": (None, [c_void_p]),
    "gpsr_ctx_workspace": (c_void_p, [c_void_p]),
    "gpsr_ctx_workspace_bytes": (c_size_t, [c_void_p]),
    "gpsr_flag_count": (c_int, [c_void_p, c_int, c_void_p]),
    "gpsr_gather": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 4),
    "gpsr_init_hypers": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "gpsr_covariance": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gpsr_covariance_grad": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gpsr_invert_workspace_bytes": (c_size_t, [c_int, c_int]),
    "gpsr_invert": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpsr_fit_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gpsr_fit": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float,
                         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpsr_noalloc_scratch_bytes": (c_size_t, []),
    "gpsr_fit_noalloc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_float,
                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "gpsr_set_dual_stream": (c_int, [c_int]),
    "gpsr_set_split_small": (c_int, [c_int]),
    "gpsr_set_fused": (c_int, [c_int]),
    "gpsr_loss_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpsr_predict": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p, c_void_p]),
    "gpsr_combine": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_int,
                             c_int, c_int, c_void_p, c_void_p]),
    "gpsr_predict_pixels_workspace_bytes": (c_size_t, [c_int] * 5),
    "gpsr_predict_pixels": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int]
                            + [c_int, c_float, c_float, c_int, c_float, c_int, c_int]
                            + [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpsr_agpr_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gpsr_agpr_filter": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gpsr_selftest_mfma": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "gpsr_profile_select": (c_int, [ctypes.c_uint]),
    "gpsr_profile_begin": (c_int, [c_int]),
    "gpsr_profile_end": (c_int, [c_void_p, c_void_p]),
    "gpsr_u8_to_unit": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpsr_unit_to_u8": (c_int, [c_void_p, c_void_p, c_size_t, ctypes.c_double, c_void_p]),
    "gpsr_resample": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 4 + [c_int, c_void_p]),
    "gpsr_decimate2": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "gpsr_sepfilter_sym": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "gpsr_filter3x3": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gpsr_axpy": (c_int, [c_void_p, c_void_p, ctypes.c_double, c_void_p, c_size_t, c_void_p]),
    "gpsr_bgr2gray": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpsr_bgr2yuv": (c_int, [c_void_p] * 4 + [c_size_t, c_void_p]),
    "gpsr_yuv2bgr": (c_int, [c_void_p] * 4 + [c_size_t, c_void_p]),
    "gpsr_bgr2ycrcb": (c_int, [c_void_p] * 4 + [c_size_t, c_void_p]),
    "gpsr_ycrcb2bgr": (c_int, [c_void_p] * 4 + [c_size_t, c_void_p]),
    "gpsr_resample_u8": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int] + [c_void_p] * 5),
    "gpsr_active_sample_workspace_bytes": (c_size_t, [c_int]),
    "gpsr_active_sample": (c_int, [c_void_p, c_int, c_int, c_int, c_int, ctypes.c_double, ctypes.c_double,
                                   ctypes.c_double, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "gpsr_sepfilter_u8": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "gpsr_metrics_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "gpsr_metrics": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
}


def build(force=False):
    """Compile csrc/gpsr_api.hip for gfx950 into libgpsr.so (in-tree)."""
    srcs = [SRC, os.path.join(_HERE, "csrc", "gpsr_kernels.hpp"),
            os.path.join(os.path.dirname(_HERE), "include", "gpsr.h")]
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs):
        return LIB_PATH
    cmd = [HIPCC] + HIPCC_FLAGS + ["-o", LIB_PATH, SRC]
    subprocess.run(cmd, check=True, cwd=_HERE)
    return LIB_PATH


def load():
    """Load libgpsr.so; raises if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is the only compute path "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`)")
    # torch first: the process must hold ONE HIP runtime, the copy torch ships (it owns the device memory and the streams
    # this library is handed).  Loaded before torch, libgpsr.so would bring /opt/rocm's libamdhip64 with it, torch its
    # own afterwards, and the launches of whichever runtime comes second fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


class GpsrError(RuntimeError):
    pass


def check(rc, what):
    """Negative status -> exception; positive = number of flagged GPs (returned)."""
    if rc < 0:
        raise GpsrError(f"{what} failed ({rc}): {load().gpsr_last_error().decode()}")
    return rc
