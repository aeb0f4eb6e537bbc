"""CPU checks of the two-bulge Francis sweep of the n = 64 HIP solver BEFORE it meets a GPU: the scalar prototype
(oracle/hqr2_proto.py) and the lane-level model of the kernel's bookkeeping (tools/qr2_lane_model.py: per-lane pointers,
validity constants, parking, first / last steps; it asserts that no element has two writers in a pass, that every store
lies inside the active window and that no out-of-window operand reaches a stored value) against LAPACK.  The reference
only calls np.linalg.eigvals (analysis/eval_eig.py:296)."""
import importlib.util
import os

import numpy as np
import pytest
import scipy.linalg as sl

import oracle as O
from oracle import hqr2_proto as P2
from oracle import hqr_proto as P1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lane_model():
    spec = importlib.util.spec_from_file_location("qr2_lane_model", os.path.join(ROOT, "tools", "qr2_lane_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _cases(n, rng):
    yield "gaussian", rng.standard_normal((n, n)) / np.sqrt(n)
    a = np.zeros((n, n))
    a[np.arange(n), np.roll(np.arange(n), 1)] = 1.0                 # cyclic permutation: only exceptional shifts converge
    yield "cyclic", a
    d = np.diag(np.exp(-np.arange(n) * 0.5))
    yield "graded", d @ rng.standard_normal((n, n)) @ d


@pytest.mark.parametrize("shifts", ["same", "e44"])
def test_two_bulge_prototype_matches_lapack(shifts):
    rng = np.random.default_rng(11)
    for n in (5, 17, 40):
        for name, a in _cases(n, rng):
            h = sl.hessenberg(a)
            st = {}
            w, info = P2.hqr2_eigvals(h, shifts, st)
            assert info == 0, (name, n)
            ref = np.linalg.eigvals(a)
            scale = max(1.0, float(np.abs(ref).max()))
            w1, _ = P1.hqr_eigvals(h)
            # graded matrices have ill-conditioned small eigenvalues: hold the two-bulge iteration to the single-bulge one
            tol = max(1e-11 * scale, 20 * O.max_abs_dlambda(w1[None], ref[None]))
            assert O.max_abs_dlambda(w[None], ref[None]) <= tol, (name, n, shifts)


def test_two_bulge_iteration_count_on_the_bench_matrices():
    """The reason the kernel chases two bulges: fewer iterations than single-bulge steps (0.66 on the C2 workload)."""
    import torch
    import bench
    a, _ = bench.gen_known_spectrum(2, 64, torch.Generator().manual_seed(5), torch.device("cpu"))
    for k in range(2):
        h = sl.hessenberg(a[k].numpy())
        s1, s2 = {}, {}
        P1.hqr_eigvals(h, s1)
        P2.hqr2_eigvals(h, "same", s2)
        assert s2["iters"] < 0.75 * s1["steps"], (s1, s2)


def test_lane_model_of_the_kernel_bookkeeping():
    lm = _lane_model()
    rng = np.random.default_rng(64)
    n = 64
    for name, a in list(_cases(n, rng))[:2]:
        h = sl.hessenberg(a)
        st = {}
        w, info = lm.hqr_lanes(h, st)                               # raises on overlapping / out-of-window stores
        assert info == 0, name
        ref = np.linalg.eigvals(a)
        assert O.max_abs_dlambda(w[None], ref[None]) < 1e-11, name
