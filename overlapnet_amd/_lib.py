"""ctypes binding of libovn_hip.so (C ABI declared in include/ovn_hip.h).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# OVN_LIB names another build of the library (same ABI version) -- A/B timing of two builds in one checkout, tools/experiments
LIB_PATH = os.environ.get("OVN_LIB") or os.path.join(_HERE, "libovn_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

ABI_VERSION = 6

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/ovn_hip.h one to one
SIGNATURES = {
    "ovn_abi_version": (C.c_int, []),
    "ovn_last_error": (C.c_char_p, []),
    "ovn_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "ovn_destroy": (C.c_int, [_vp]),
    "ovn_add_leg_layer": (C.c_int, [_vp, C.c_char_p, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ovn_set_head_weights": (C.c_int, [_vp] + [_vp] * 8 + [_vp]),
    "ovn_set_head_geometry": (C.c_int, [_vp, C.c_int]),
    "ovn_finalize": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "ovn_leg": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp]),
    "ovn_heads": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp]),
    "ovn_delta_head": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp]),
    "ovn_corr_head": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp]),
    "ovn_spectrum": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp]),
    "ovn_corr_head_spectral": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp]),
    "ovn_heads_spectral": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp]),
    "ovn_delta_cache": (C.c_int, [_vp, _vp, C.c_int64, _vp, _vp]),
    "ovn_set_head_pipeline": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "ovn_get_head_pipeline": (C.c_int, [_vp, _i64p, _i64p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ovn_best_match": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int64, C.c_float, C.c_int64, _vp, _vp]),
    "ovn_project": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double,
                              C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ovn_normals": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "ovn_projection_angles": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp]),
    "ovn_gt_range_images": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int64, _vp, _vp, C.c_int, C.c_int, C.c_double, C.c_double,
                                      C.c_double, _vp, _vp]),
    "ovn_gt_overlap_counts": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "ovn_set_head_precision": (C.c_int, [_vp, C.c_int]),
    "ovn_set_leg_precision": (C.c_int, [_vp, C.c_int]),
    "ovn_set_projection_trig": (C.c_int, [_vp, C.c_int]),
    "ovn_set_head_compaction": (C.c_int, [_vp, C.c_int]),
    "ovn_profile_begin": (C.c_int, [_vp]),
    "ovn_profile_end": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "ovn_debug_conv": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "ovn_debug_head_activations": (C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    "ovn_head_walk_stats": (C.c_int, [_vp, _i32p, _vp]),
    "ovn_workspace_bytes": (C.c_int64, [_vp]),
    "ovn_comm_unique_id": (C.c_int, [_vp]),
    "ovn_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "ovn_comm_destroy": (C.c_int, [_vp]),
    "ovn_gather_scores": (C.c_int, [_vp, _vp, _vp, _i64p, C.c_int, _vp, _vp, _vp]),
    "ovn_selftest": (C.c_int, [_vp]),
}


class OvnError(Exception):
    pass


def build(force: bool = False, quiet: bool = True) -> str:
    """Compile libovn_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, capture_output=quiet)
    r = subprocess.run(["make", "-C", CSRC_DIR, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise OvnError("building libovn_hip.so failed:\n" + r.stdout + r.stderr)
    if not os.path.isfile(LIB_PATH):
        raise OvnError("make succeeded but %s is missing" % LIB_PATH)
    return LIB_PATH


_lib = None


def load() -> C.CDLL:
    """Load the shared library and bind every symbol of the header. Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64/libhsa-runtime64 under torch/lib.  Two HIP runtimes in one
    # process do not work (the second one finds no device), so torch's copy must be mapped BEFORE
    # libovn_hip.so is: the dynamic loader then binds our DT_NEEDED libamdhip64.so.7 to the loaded one.
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise OvnError("%s not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                       "g.build()'` (or `make -C overlapnet_amd/csrc`). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise OvnError("libovn_hip.so does not export %s" % name) from e
        fn.restype = res
        fn.argtypes = args
    v = lib.ovn_abi_version()
    if v != ABI_VERSION:
        raise OvnError("libovn_hip.so ABI version %d, Python binding expects %d: rebuild" % (v, ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ovn_last_error()
        raise OvnError("%s failed (code %d): %s" % (what, rc, (msg or b"").decode("utf-8", "replace")))
