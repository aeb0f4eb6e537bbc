"""pytest configuration: registers the ``gpu`` marker and provides the package/golden helpers.

``-m "not gpu"`` runs in the dev container (no GPU, /root/reference may exist but is never read
by the tests); ``-m gpu`` runs on the MI355X box (no /root/reference there)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG_NAME = "task-level-insights-from-eigenvalues-across-sequence-models_amd"
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The product package (its directory name has hyphens, so it is imported via importlib).  The shared library is a
    build artefact (git-ignored): compile it in-tree first if this checkout does not have it yet (hipcc cross-compiles
    without a GPU; the same thing __graft_entry__.build() does)."""
    pkg = importlib.import_module(PKG_NAME)
    if not os.path.exists(pkg._capi.LIB_PATH):
        pkg._capi.build()
    return pkg


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return _load


@pytest.fixture(scope="session")
def gpu_ops(pkg):
    """Device ops; importing/using them on a box without a GPU or without the built HIP library
    raises -- there is no CPU fallback (the oracle is never on the product path)."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test collected on a machine without a GPU")
    return pkg.ops
