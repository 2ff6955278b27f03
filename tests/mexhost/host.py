"""Test infrastructure: builds mex/<name>.cpp against the in-memory MATLAB-API host of this directory
and calls its mexFunction from Python (ctypes).  See mex.h / mexhost.cpp."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CLASS = {np.dtype("float64"): 6, np.dtype("uint8"): 9, np.dtype("int32"): 12, np.dtype("uint32"): 13, np.dtype("uint64"): 15}
DTYPE = {v: k for k, v in CLASS.items()}


def build(name):
    """g++ of mex/<name>.cpp + mexhost.cpp into _build/<name>.so, linked with libstereo_hip.so."""
    out = os.path.join(HERE, "_build", name + ".so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    src = [os.path.join(ROOT, "mex", name + ".cpp"), os.path.join(HERE, "mexhost.cpp")]
    lib = os.path.join(ROOT, "stereo_amd", "libstereo_hip.so")
    newest = max(os.path.getmtime(f) for f in src + [os.path.join(HERE, "mex.h"), lib])
    if not os.path.exists(out) or os.path.getmtime(out) < newest:
        cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-Werror", "-I" + HERE, "-I" + os.path.join(ROOT, "include")] + src + [
            "-L" + os.path.join(ROOT, "stereo_amd"), "-lstereo_hip", "-Wl,-rpath," + os.path.join(ROOT, "stereo_amd"), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("gateway %s does not compile:\n%s" % (name, r.stderr))
    return out


class MexError(RuntimeError):
    pass


class Gateway:
    def __init__(self, name):
        # In a Python process PyTorch's own HIP runtime must be the first one loaded (stereo_amd/_lib.py
        # sees to that); a gateway loaded before it would pull in the system runtime and the process
        # would end up with two.  (Under MATLAB, or any plain C host, there is only the system one:
        # tools/abi_smoke.c.)
        import stereo_amd._lib as _l
        _l.lib()
        self.lib = C.CDLL(build(name))
        L = self.lib
        for f in ("mh_numeric", "mh_string", "mh_struct"):
            getattr(L, f).restype = C.c_void_p
        L.mh_data.restype = C.c_void_p
        L.mh_numel.restype = C.c_uint64
        L.mxGetM.restype = C.c_size_t
        L.mxGetN.restype = C.c_size_t

    def _to_mx(self, v):
        L = self.lib
        if isinstance(v, str):
            return L.mh_string(v.encode())
        if isinstance(v, dict):
            s = L.mh_struct()
            for k, x in v.items():
                L.mh_set_field(C.c_void_p(s), k.encode(), C.c_void_p(self._to_mx(x)))
            return s
        a = np.asarray(v)
        if a.dtype not in CLASS:
            a = a.astype(np.float64)
        if a.ndim < 2:
            a = a.reshape(a.size if a.ndim else 1, 1) if a.ndim else a.reshape(1, 1)
        f = np.asfortranarray(a)                                    # MATLAB arrays are column major
        dims = (C.c_uint64 * f.ndim)(*f.shape)
        return L.mh_numeric(C.c_int(CLASS[f.dtype]), C.c_int(f.ndim), dims, f.ctypes.data_as(C.c_void_p))

    def call(self, nlhs, *args):
        """mexFunction(nlhs, plhs, len(args), args); returns the outputs as NumPy arrays."""
        L = self.lib
        rhs = [self._to_mx(a) for a in args]
        prhs = (C.c_void_p * max(len(rhs), 1))(*rhs)
        plhs = (C.c_void_p * max(nlhs, 1))()
        err = C.create_string_buffer(1024)
        rc = L.mh_call(C.c_int(nlhs), plhs, C.c_int(len(rhs)), prhs, err, C.c_uint64(len(err)))
        outs = []
        if rc == 0:
            for i in range(nlhs):
                p = C.c_void_p(plhs[i])
                m, n, cls = L.mxGetM(p), L.mxGetN(p), L.mh_class(p)
                buf = (C.c_char * (int(L.mh_numel(p)) * DTYPE[cls].itemsize)).from_address(L.mh_data(p))
                outs.append(np.frombuffer(bytes(buf), dtype=DTYPE[cls]).reshape((m, n), order="F").copy())
                L.mh_free(p)
        for r in rhs:
            L.mh_free(C.c_void_p(r))
        if rc:
            raise MexError(err.value.decode())
        return outs
