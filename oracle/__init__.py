"""CPU oracle for the eigenvalue-analysis hot path.

TEST INFRASTRUCTURE ONLY.  This package is a NumPy restatement of the arithmetic in the
reference's ``analysis/eval_eig.py`` (every function cites the reference file:line it follows).
It exists so that ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg can check / time-beside the HIP path.  Nothing in the product package imports it, and the
product path fails loudly when the HIP library is missing -- there is no CPU fallback.

Pinning: the reference ships no tests, fixtures or golden vectors, so the
oracle is pinned against outputs of the reference's *own functions* run in the dev container
(``tests/golden/make_golden.py`` loads ``/root/reference/analysis/eval_eig.py`` by path with
inert stubs for the absent third-party modules and ``jax.numpy`` aliased to NumPy) and against
LAPACK-independent known-answer tests (``tests/test_oracle_*.py``).
"""
from .eig_oracle import *  # noqa: F401,F403
