"""ctypes binding of libstereo_hip.so (the C ABI declared in include/stereo_hip.h).

There is NO CPU fallback: if the shared library is missing, or no HIP device is
visible when a solver is called, the call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# STEREO_HIP_LIB: another build of the same library (development: the message-profile flavour,
# stereo_amd/csrc/build.sh -DSTEREO_HIP_MESSAGE_PROFILE with STEREO_HIP_OUT set)
LIB_PATH = os.environ.get("STEREO_HIP_LIB") or os.path.join(_HERE, "libstereo_hip.so")

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_i64p = C.POINTER(C.c_int64)

ABI_VERSION = 6   # include/stereo_hip.h: STEREO_HIP_ABI_VERSION

_lib = None


class StereoHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StereoHipError(
                "%s not found: build it with stereo_amd/csrc/build.sh "
                "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
        # PyTorch-ROCm ships its own libamdhip64; a process must not end up with two HIP runtimes
        # (torch then reports "No HIP GPUs are available").  Loading torch first makes this
        # library resolve to the same runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        if L.stereo_hip_abi_version() != ABI_VERSION:
            raise StereoHipError("%s reports ABI version %d, this binding needs %d: rebuild it with "
                                 "stereo_amd/csrc/build.sh" % (LIB_PATH, L.stereo_hip_abi_version(), ABI_VERSION))
        L.stereo_hip_last_error.restype = C.c_char_p
        L.stereo_trws_plan_destroy.restype = None
        L.stereo_rd_plan_destroy.restype = None
        L.stereo_fusion_destroy.restype = None
        for name in ("stereo_trws_cache_clear", "stereo_rd_cache_clear"):
            if hasattr(L, name):   # (an older build loaded through STEREO_HIP_LIB for an A/B timing lacks them)
                getattr(L, name).restype = None
        _lib = L
        # one-time runtime initialisation now, not inside the first solver call: it consumes libc
        # rand() values, which QPBO Improve draws its permutation from (see stereo_hip_warm_up)
        # (on the device this rank will use: torchrun exports LOCAL_RANK, and a warm-up on the default
        # device would leave a HIP context of every rank on GPU 0)
        ndev = L.stereo_hip_device_count()
        if ndev > 0:
            local = os.environ.get("LOCAL_RANK")
            if local is not None and local.isdigit() and int(local) < ndev:
                L.stereo_hip_set_device(int(local))
            L.stereo_hip_warm_up()
    return _lib


def check(rc, err):
    if rc != 0:
        msg = err.value.decode("utf-8", "replace") if err is not None else ""
        if not msg:
            msg = lib().stereo_hip_last_error().decode("utf-8", "replace")
        raise StereoHipError(msg or ("libstereo_hip call failed (rc=%d)" % rc))


def errbuf():
    return C.create_string_buffer(1024)


def device_count():
    return int(lib().stereo_hip_device_count())


def require_device():
    if device_count() < 1:
        raise StereoHipError("no HIP device visible: the stereo_amd solvers only run on the GPU")
