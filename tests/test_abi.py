"""CPU: the C-ABI library builds, loads and exports every symbol include/ovn_hip.h declares; the
product path refuses to run without a GPU (no CPU fallback, nothing under overlapnet_amd imports the oracle)."""
import ctypes
import os
import re

import pytest
import torch

from overlapnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "ovn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ovn_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header():
    _lib.build()
    assert os.path.isfile(_lib.LIB_PATH)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "libovn_hip.so does not export %s declared in include/ovn_hip.h" % n
    # the Python binding table covers the header one to one
    assert sorted(_lib.SIGNATURES) == names


def test_load_binds_and_versions():
    lib = _lib.load()
    assert lib.ovn_abi_version() == _lib.ABI_VERSION
    assert lib.ovn_last_error() is not None


def test_argument_errors_without_gpu_calls():
    lib = _lib.load()
    # NULL out pointer is rejected before any HIP call
    assert lib.ovn_create(0, 64, 900, 4, None) == 1
    assert b"out is NULL" in lib.ovn_last_error()
    assert lib.ovn_finalize(None, None) == 1
    assert lib.ovn_workspace_bytes(None) == 0
    assert lib.ovn_destroy(None) == 0
    # the optional collective: argument errors before RCCL is even loaded
    assert lib.ovn_comm_unique_id(None) == 1 and b"id_out is NULL" in lib.ovn_last_error()
    assert lib.ovn_comm_init(None, 0, 1, None) == 1
    assert lib.ovn_comm_destroy(None) == 1
    assert lib.ovn_gather_scores(None, None, None, None, 0, None, None, None) != 0 and b"no communicator" in lib.ovn_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_fails_loudly_without_gpu():
    from overlapnet_amd.engine import OvnEngine
    with pytest.raises(_lib.OvnError):
        OvnEngine(64, 900, 4)
    from overlapnet_amd.infer import Infer
    cfg = {"model": {"leg_output_width": 360, "inputShape": [64, 900], "legsType": "360OutputkLegs",
                     "overlap_head": "DeltaLayerConv1NetworkHead", "orientation_head": "CorrelationHead"},
           "infer_seqs": "x", "data_root_folder": "/tmp", "use_depth": True, "use_normals": True,
           "use_class_probabilities": False, "use_class_probabilities_pca": False, "use_intensity": False,
           "batch_size": 16, "pretrained_weightsfilename": ""}
    with pytest.raises(_lib.OvnError):
        Infer(cfg)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "overlapnet_amd")
    for dp, _, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                # nor the test / bench infrastructure (synthetic inputs, golden fixtures)
                assert "from tools" not in txt and "import tools" not in txt and "tests/golden" not in txt, f
