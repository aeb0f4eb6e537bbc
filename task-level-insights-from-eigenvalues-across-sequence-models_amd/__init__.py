"""MI355X-native eigenvalue-analysis hot path (drop-in for the reference's ``analysis/eval_eig.py``).

Layout
  csrc/            hand-written HIP kernels for gfx950 + the extern "C" ABI (include/eigspec.h)
  libeigspec_hip.so  built in-tree by ``make`` / ``__graft_entry__.build()``
  _capi.py         ctypes binding of the C ABI (raises if the library is missing -- no CPU fallback)
  ops.py           thin device-tensor wrappers (torch is used for device memory / streams only)
  dist.py          batch sharding across ranks + the single all-gather of spectra
  analysis/        host-side mirror of the reference's ``analysis`` package (same names / signatures)
  ckpt.py          orbax aggregate-file (msgpack) checkpoint decoder, no JAX needed
  models.py        drop-in torch ``models`` (Transformer / Mamba) whose token mixers run in the HIP library

The directory name contains hyphens; import it with
``importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd")``.
"""
from . import _capi  # noqa: F401
from . import ops  # noqa: F401
from . import dist  # noqa: F401
from . import ckpt  # noqa: F401
from . import models  # noqa: F401
from .analysis import eval_eig  # noqa: F401

__all__ = ["ops", "dist", "models", "eval_eig", "build"]


def build(verbose=False):
    """Compile libeigspec_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    return _capi.build(verbose=verbose)
