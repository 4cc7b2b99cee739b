"""Builds the C part of the oracle (TEST INFRASTRUCTURE): oracle/svml_f32.c -> oracle/_build/libovn_oracle_c.so (gcc).

`__graft_entry__.build()` calls `build()`; `load()` builds on demand (gcc is in the image, here and on the GPU box) and binds the
entry points with ctypes.  Nothing under overlapnet_amd/ imports this."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_build")
SO = os.path.join(OUT_DIR, "libovn_oracle_c.so")
SRCS = [os.path.join(HERE, "svml_f32.c")]
DEPS = SRCS + [os.path.join(HERE, "approx14_tables.h")]
_lib = None


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in DEPS):
        return SO
    # -ffp-contract=off: one IEEE operation per C operation; fmaf() stays a fused operation (libm's software fmaf is exact too)
    tmp = SO + ".tmp.%d" % os.getpid()
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-I", HERE] + SRCS + ["-o", tmp, "-lm"])
    os.replace(tmp, SO)
    return SO


def load():
    global _lib
    if _lib is None:
        lib = ctypes.CDLL(build())
        fp = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
        lib.ovn_svml_atan2f_array.argtypes = [fp, fp, fp, ctypes.c_long]
        lib.ovn_svml_atan2f_array.restype = None
        lib.ovn_svml_asinf_array.argtypes = [fp, fp, ctypes.c_long]
        lib.ovn_svml_asinf_array.restype = None
        for n in ("ovn_rcp14", "ovn_rsqrt14"):
            getattr(lib, n).argtypes = [ctypes.c_float]
            getattr(lib, n).restype = ctypes.c_float
        _lib = lib
    return _lib


def svml_arctan2(y: np.ndarray, x: np.ndarray) -> np.ndarray:
    """np.arctan2 on float32 arrays as NumPy's AVX512_SKX build evaluates it (SVML __svml_atan2f16), on any host."""
    y = np.ascontiguousarray(y, np.float32)
    x = np.ascontiguousarray(x, np.float32)
    assert y.shape == x.shape
    out = np.empty_like(x)
    load().ovn_svml_atan2f_array(y.reshape(-1), x.reshape(-1), out.reshape(-1), x.size)
    return out


def svml_arcsin(x: np.ndarray) -> np.ndarray:
    """np.arcsin on a float32 array as NumPy's AVX512_SKX build evaluates it (SVML __svml_asinf16), on any host."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    load().ovn_svml_asinf_array(x.reshape(-1), out.reshape(-1), x.size)
    return out


if __name__ == "__main__":
    print(build(force=True))
