import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with `-m gpu` on the GPU box")


def pytest_collection_modifyitems(config, items):
    """`pytest tests/` on a box without a GPU: every `gpu`-marked test is SKIPPED (the product itself has no CPU fallback and raises
    OvnError -- tests/test_abi.py checks that); on the GPU box nothing is touched."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def fixture_npz():
    from tools import synthetic as S
    return S.load_fixture_images()


@pytest.fixture(scope="session")
def nn_golden():
    import numpy as np
    with np.load(os.path.join(ROOT, "tests", "golden", "nn_oracle.npz")) as z:
        return {k: z[k] for k in z.files}
