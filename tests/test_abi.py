"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/stereo_hip.h declares; the solvers refuse to run without a device (no CPU
fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(ROOT, "include", fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names += re.findall(r"\b(stereo_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported():
    from stereo_amd import _lib
    L = _lib.lib()
    names = declared_functions()
    assert "stereo_trws" in names and "stereo_rd" in names and len(names) >= 12
    for n in names:
        assert hasattr(L, n), "libstereo_hip.so does not export %s" % n


def test_abi_version():
    from stereo_amd import _lib
    # the header's number, the library's and the binding's agree (a stale .so is refused at load)
    text = open(os.path.join(ROOT, "include", "stereo_hip.h")).read()
    declared = int(re.search(r"#define STEREO_HIP_ABI_VERSION (\d+)", text).group(1))
    assert _lib.lib().stereo_hip_abi_version() == declared == _lib.ABI_VERSION >= 3


def test_no_cpu_fallback_without_device():
    import stereo_amd
    if stereo_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(stereo_amd.StereoHipError, match="no HIP device"):
        stereo_amd.trws(1, np.zeros((3, 4)), np.array([[1, 2], [2, 3]]), np.zeros((3, 2)),
                        np.zeros((3, 2)), np.ones(2), 1.0, {})


def test_wrapper_argument_checks():
    """trws.m / trws_mex.cpp argument validation is mirrored before the device is touched."""
    import stereo_amd
    from stereo_amd import StereoHipError
    u = np.zeros((3, 4))
    c = np.array([[1, 2], [2, 3]])
    with pytest.raises(StereoHipError, match="q contains NaN"):
        stereo_amd.trws(1, u, c, np.full((3, 2), np.nan), np.zeros((3, 2)), np.ones(2), 1.0, {})
    with pytest.raises(StereoHipError, match="K x E"):
        stereo_amd.trws(1, u, c, np.zeros((2, 2)), np.zeros((3, 2)), np.ones(2), 1.0, {})
    with pytest.raises(AssertionError):
        stereo_amd.trws(1, u, np.array([[0, 2], [2, 3]]), np.zeros((3, 2)), np.zeros((3, 2)),
                        np.ones(2), 1.0, {})
    with pytest.raises(StereoHipError, match="unknown option"):
        stereo_amd.trws(1, u, c, np.zeros((3, 2)), np.zeros((3, 2)), np.ones(2), 1.0, {"bogus": 1})
