"""Pins the CPU oracle (oracle/eig_oracle.py) against the golden vectors the reference's own functions
produced in the dev container (tests/golden/make_golden.py) and against LAPACK-independent
known-answer tests.  CPU only."""
import os

import numpy as np
import pytest

import oracle as O


# ---- attention ------------------------------------------------------------------------------
@pytest.mark.parametrize("case,rtol", [("a", 2e-5), ("b", 1e-4), ("c", 2e-3)])
def test_softmax_eta_matches_reference(golden, case, rtol):
    g = golden("attention")
    with np.errstate(over="ignore"):
        got = O.eta_softmax(g[f"x_{case}"], g[f"w_{case}"], g[f"b_{case}"], int(g["d_qk"]), int(g["H"]), int(g["D"]))
    want = g[f"softmax_{case}"]
    assert got.shape == want.shape == (int(g["B"]), int(g["L"]) - 1, int(g["H"]), 1)
    assert got.dtype == np.float64
    # eta carries exp(m_t - m_{t+1}) with m in fp32 (eval_eig.py:86-88): its relative noise is
    # |m| * 6e-8, and torch-CPU vs NumPy fp32 GEMMs sum in different orders.  Case c has |scores| of
    # several hundred (and overflows to inf in places), hence the magnitude-scaled tolerances.
    finite = np.isfinite(want)
    assert np.array_equal(finite, np.isfinite(got))
    np.testing.assert_allclose(got[finite], want[finite], rtol=rtol, atol=0)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_linear_eta_matches_reference(golden, case):
    g = golden("attention")
    args = (g[f"x_{case}"], g[f"w_{case}"], g[f"b_{case}"], int(g["d_qk"]), int(g["H"]), int(g["D"]))
    got = O.eta_linear(*args)
    want = g[f"linear_{case}"]
    assert got.shape == want.shape and got.dtype == np.float64
    np.testing.assert_allclose(got, want, rtol=2e-6)
    np.testing.assert_allclose(O.eta_linear_streaming(*args), want, rtol=2e-6)


@pytest.mark.parametrize("fn", ["exp", "elu", "softplus", "sigmoid"])
@pytest.mark.parametrize("use_off", [0, 1])
def test_norm_eta_matches_reference(golden, fn, use_off):
    g = golden("norm_attention")
    got = O.eta_norm(g["x"], g["w"], g["b"], g["offset"] if use_off else None, fn,
                     int(g["d_qk"]), int(g["H"]), int(g["D"]))
    want = g[f"eta_{fn}_{use_off}"]
    assert got.shape == want.shape
    finite = np.isfinite(want)
    assert np.array_equal(finite, np.isfinite(got))
    np.testing.assert_allclose(got[finite], want[finite], rtol=5e-5)


def test_norm_eta_unknown_fn_raises(golden):
    g = golden("norm_attention")
    with pytest.raises(RuntimeError, match="normalization function tanh not implemented!"):
        O.eta_norm(g["x"], g["w"], g["b"], None, "tanh", 8, 2, 16)


# ---- mamba2 ---------------------------------------------------------------------------------
def test_mamba2_lambda_matches_reference(golden):
    g = golden("mamba2")
    got = O.lambda_mamba2(g["x"], g["w"], g["b"], g["dt_bias"], g["A_log"], int(g["d_inner"]),
                          int(g["ngroups"]), int(g["d_state"]), int(g["nheads"]))
    assert got.dtype == np.float32 and got.shape == g["lam"].shape
    np.testing.assert_allclose(got, g["lam"], rtol=3e-6, atol=1e-7)
    got = O.lambda_mamba2_lti(g["x"], g["A"], g["beta"])
    assert got.dtype == np.float32 and got.shape == g["lam_lti"].shape
    np.testing.assert_allclose(got, g["lam_lti"], rtol=1e-6)


# ---- SSM ------------------------------------------------------------------------------------
def _layers(g, fam, names):
    return [{n: g[f"{fam}_{i}_{n}"] for n in names} for i in range(2)]


def test_ssm_closed_forms_match_reference(golden):
    g = golden("ssm")
    s5 = _layers(g, "s5", ["Lambda_re", "Lambda_im", "log_step"])
    lru = _layers(g, "lru", ["nu_log", "theta_log"])
    for i in range(2):
        np.testing.assert_allclose(O.eigvals_ssm("s5", s5, i, 1, 16), g[f"s5_{i}_eig"], rtol=1e-14)
        np.testing.assert_allclose(O.eigvals_ssm("lru", lru, i, 1, 16), g[f"lru_{i}_eig"], rtol=1e-14)
    assert bool(g["unknown_returns_none"]) and O.eigvals_ssm("s6", s5, 0, 1, 16) is None


def test_s4_dplr_and_eigs_match_reference(golden):
    g = golden("ssm")
    s4 = _layers(g, "s4", ["log_step", "Lambda_re", "Lambda_im", "P", "B", "C"])
    for i in range(2):
        Lam, Pv, Bv, Ct, step = O.s4_column_inputs(s4[i], 1)
        Ab, Bb, Cb = O.discrete_dplr(Lam, Pv, Pv, Bv, Ct, step, int(g["SEQ"]))
        np.testing.assert_allclose(Ab, g[f"s4_{i}_Ab"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(Bb, g[f"s4_{i}_Bb"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(Cb, g[f"s4_{i}_Cb"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(O.s4_dplr_ab(Lam, Pv, step), g[f"s4_{i}_Ab"], rtol=1e-11, atol=1e-12)
        w = O.eigvals_ssm("s4", s4, i, 1, int(g["SEQ"]))
        assert w.shape == (int(g["N"]), 1)
        assert O.max_abs_dlambda(w[:, 0], g[f"s4_{i}_eig"][:, 0]) < 1e-11


def test_discrete_dplr_with_separate_p_and_q_matches_reference(golden):
    """discrete_DPLR's general form (analysis/eval_eig.py:254-278): A = diag(Lambda) - P Q^H with Q != P and an unclipped
    Lambda (the clip is the caller's, :288) -- golden from the reference's own function (make_golden_dplr_pq.py)."""
    g = golden("dplr_pq")
    for i in range(2):
        Ab, Bb, Cb = O.discrete_dplr(g[f"c{i}_Lambda"], g[f"c{i}_P"], g[f"c{i}_Q"], g[f"c{i}_B"], g[f"c{i}_C"],
                                     float(g[f"c{i}_step"]), int(g[f"c{i}_L"]))
        assert np.max(g[f"c{i}_Lambda"].real) > -1e-4                      # the case really exercises "no clip here"
        np.testing.assert_allclose(Ab, g[f"c{i}_Ab"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(Bb, g[f"c{i}_Bb"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(Cb, g[f"c{i}_Cb"], rtol=1e-9, atol=1e-10)


def test_dplr_secular_prototype_matches_lapack_and_the_reference_golden(golden):
    """The algorithm of the optional S4 fast path (csrc/dplr_secular.hip) on the CPU first: secular equation of
    diag(Lambda) - P Q^H by Ehrlich-Aberth + Cayley map, against np.linalg.eigvals of the dense Ab (analysis/eval_eig.py:296),
    the reference golden, and degenerate inputs (zero weights, repeated poles)."""
    from oracle import dplr_secular_proto as S
    rng = np.random.default_rng(41)
    for n in (1, 2, 5, 16, 64, 128):
        for pq in (False, True):
            lam = -np.abs(rng.standard_normal(n)) - 0.1 + 1j * rng.standard_normal(n) * 5
            p = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.2
            q = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.2 if pq else None
            a, its, bad = S.rank1_eigvals(lam, p, q)
            assert bad == 0 and its < 120, (n, its)
            ref = np.linalg.eigvals(np.diag(lam) - np.outer(p, np.conj(p if q is None else q)))
            assert O.max_abs_dlambda(a, ref) < 1e-10, n
    g = golden("ssm")
    for i in range(2):
        w, bad = S.s4_eigs(g[f"s4_{i}_Lambda_re"][:, 1], g[f"s4_{i}_Lambda_im"][:, 1], g[f"s4_{i}_P"][:, 1],
                           float(np.exp(g[f"s4_{i}_log_step"][0, 1])))
        assert bad == 0 and O.max_abs_dlambda(w, g[f"s4_{i}_eig"][:, 0]) < 1e-10
    # zero weights and exactly repeated poles are eigenvalues as they stand
    lam = np.array([-1.0 + 1j, -2.0, -1.0 + 1j, -0.5 - 2j, -2.0, -3.0])
    p = np.array([0.3, 0.0, 0.4j, 0.2, 0.1, 0.0])
    a, _, bad = S.rank1_eigvals(lam, p)
    assert bad == 0 and O.max_abs_dlambda(a, np.linalg.eigvals(np.diag(lam) - np.outer(p, np.conj(p)))) < 1e-12
    # raw HiPPO, N = 64: root by root backward stable although the spectrum itself is noise at this order
    lam, p, _, _, _ = O.make_dplr_hippo(64)
    a, its, bad = S.rank1_eigvals(lam, p)
    A = np.diag(lam) - np.outer(p, p.conj())
    nrm = np.linalg.norm(A, 2)
    assert bad == 0 and max(np.linalg.svd(A - z * np.eye(64), compute_uv=False)[-1] for z in a[::4]) / nrm < 1e-13


def test_s4_cayley_known_answer():
    """LAPACK-independent structure check: Ab = (g - A)^{-1} (g + A) with g = 2/step, so
    eig(Ab) = (g + a)/(g - a) for a in eig(A)."""
    rng = np.random.default_rng(5)
    n = 24
    Lam = -np.abs(rng.standard_normal(n)) + 1j * rng.standard_normal(n) * 4
    P = (rng.standard_normal(n) + 1j * rng.standard_normal(n)) * 0.5
    step = 0.037
    Ab = O.s4_dplr_ab(Lam, P, step)
    A = np.diag(Lam) - np.outer(P, P.conj())
    gg = 2.0 / step
    np.testing.assert_allclose(Ab, np.linalg.solve(gg * np.eye(n) - A, gg * np.eye(n) + A), rtol=1e-10, atol=1e-12)
    a = np.linalg.eigvals(A)
    assert O.max_abs_dlambda((gg + a) / (gg - a), np.linalg.eigvals(Ab)) < 1e-10


def test_hippo_dplr_reconstructs_hippo():
    """make_dplr_hippo: V diag(Lambda) V^H - P P^T == HiPPO-LegS (models/common.py:215-241)."""
    n = 16
    Lam, P, B, V, B_orig = O.make_dplr_hippo(n)
    A, p, b = O.make_nplr_hippo(n)
    np.testing.assert_allclose(V @ np.diag(Lam) @ V.conj().T - np.outer(p, p), A, atol=1e-10)
    np.testing.assert_allclose(V.conj().T @ p, P, atol=1e-12)
    assert np.allclose(Lam.real, -0.5)


# ---- thresholds + report --------------------------------------------------------------------
def test_threshold_analysis_matches_reference(golden):
    g = golden("thresholds")
    pct = O.threshold_analysis(g["vals"], O.THRESHOLDS_RADIUS, 3, 2, 8)
    np.testing.assert_array_equal(pct, g["pct"])
    np.testing.assert_array_equal(O.threshold_analysis(g["vals2"], O.THRESHOLDS_RADIUS, 3, 2, 8), g["pct_init"])
    with np.errstate(invalid="ignore"):
        np.testing.assert_array_equal(O.threshold_analysis(0 * g["vals"], O.THRESHOLDS_PHASE, 3, 2, 8), g["pct_phase"])
    # edge values are double counted (Q12): column with values exactly on all six edges
    assert pct[:, 0, 0, 0].sum() > 100.0


def test_threshold_analysis_ssm_matches_reference(golden):
    g = golden("thresholds")
    rad, ph = O.radius(g["cplx"]), O.phase_deg(g["cplx"])
    np.testing.assert_array_equal(O.threshold_analysis_ssm(rad, O.THRESHOLDS_RADIUS, 3), g["pct_s"])
    np.testing.assert_array_equal(O.threshold_analysis_ssm(rad * 0.5, O.THRESHOLDS_RADIUS, 3), g["pct_s_init"])
    np.testing.assert_array_equal(O.threshold_analysis_ssm(ph, O.THRESHOLDS_PHASE, 3), g["pct_sp"])
    np.testing.assert_array_equal(O.threshold_analysis_ssm(ph * 0.5, O.THRESHOLDS_PHASE, 3), g["pct_sp_init"])


def test_report_text_matches_reference(golden, tmp_path):
    g = golden("thresholds")
    p = tmp_path / "percentage_file.txt"
    pct, pct_init = g["pct"], g["pct_init"]
    O.write_percentage_file(str(p), O.THRESHOLDS_RADIUS, pct, pct_init, pct.mean(axis=1), pct_init.mean(axis=1),
                            pct.std(axis=1), pct_init.std(axis=1))
    assert p.read_text() == str(g["report"])
    O.write_percentage_file_ssm(str(p), O.THRESHOLDS_RADIUS, O.THRESHOLDS_PHASE, g["pct_s"], g["pct_s_init"],
                                g["pct_sp"], g["pct_sp_init"])
    assert p.read_text() == str(g["report_ssm"])


# ---- analytic known answers (no LAPACK in the expected value) -----------------------------------
def test_linear_eta_constant_sequence_limit():
    """Constant tokens: nu_t = (t+1) c  =>  eta_t = (t+1)/(t+2)."""
    rng = np.random.default_rng(0)
    D, d_qk, H, L = 8, 4, 2, 9
    x = np.tile(rng.standard_normal((1, 1, D)).astype(np.float32), (2, L, 1))
    w = rng.standard_normal((2 * d_qk + D, D)).astype(np.float32)
    b = rng.standard_normal(2 * d_qk + D).astype(np.float32)
    eta = O.eta_linear(x, w, b, d_qk, H, D)
    t = np.arange(L - 1)
    np.testing.assert_allclose(eta[0, :, 0, 0], (t + 1) / (t + 2), rtol=1e-6)


def test_softmax_eta_constant_sequence_limit():
    """Constant tokens with score s: m_t = max(s, 0) for t < L-1 and s for t = L-1 (Q5);
    nu_t = (t+1) exp(s - m_t) + (L-1-t) (Q4)."""
    rng = np.random.default_rng(1)
    D, d_qk, H, L = 8, 4, 1, 7
    x = np.tile(rng.standard_normal((1, 1, D)).astype(np.float32), (1, L, 1))
    w = rng.standard_normal((2 * d_qk + D, D)).astype(np.float32)
    b = np.zeros(2 * d_qk + D, np.float32)
    qkv = O.linear_f32(x[0, 0], w, b)
    s = float(np.dot(qkv[:d_qk].astype(np.float32), qkv[d_qk:2 * d_qk].astype(np.float32)))
    t = np.arange(L)
    m = np.where(t < L - 1, max(s, 0.0), s)
    nu = (t + 1) * np.exp(s - m) + (L - 1 - t)
    want = nu[:-1] / nu[1:] * np.exp(m[:-1] - m[1:])
    got = O.eta_softmax(x, w, b, d_qk, H, D)[0, :, 0, 0]
    np.testing.assert_allclose(got, want, rtol=1e-5)


def test_dense_eig_similarity_known_spectrum():
    rng = np.random.default_rng(64)
    n = 12
    lam = np.exp((-0.5 + 1j * rng.uniform(0, 20, n // 2)) * 0.05)
    lam = np.concatenate([lam, lam.conj()])
    # real matrix with that spectrum: block diagonal rotations under a well-conditioned similarity
    Dm = np.zeros((n, n))
    for i, z in enumerate(lam[: n // 2]):
        Dm[2 * i:2 * i + 2, 2 * i:2 * i + 2] = [[z.real, z.imag], [-z.imag, z.real]]
    V = np.eye(n) + 0.1 * rng.standard_normal((n, n)) / np.sqrt(n)
    A = V @ Dm @ np.linalg.inv(V)
    assert O.max_abs_dlambda(O.eigvals_dense(A), lam) < 1e-12


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_notebook_softmax_eta_from_qk_golden(golden, case):
    """oracle.eta_softmax_from_qk against the reference notebook's own get_eig_from_qkv_att_softmax (cell 13)."""
    g = golden("notebook")
    got = O.eta_softmax_from_qk(g[f"q_{case}"], g[f"k_{case}"])
    want = g[f"eta_{case}"]
    assert got.shape == want.shape and got.dtype == np.float64
    np.testing.assert_allclose(got, want, rtol=3e-5)


# ---- prototype of the level-3 complex solver (oracle/zms_proto.py) ------------------------------------------------------
def test_zms_prototype_matches_lapack():
    """The NumPy restatement of csrc/eig_zlarge.hip (blocked Hessenberg + windowed multishift QR) against LAPACK:
    exact Hessenberg structure, unitary similarity, spectra -- including a block larger than one window (n = 90) and
    the doubled-shift strategy the kernel uses."""
    from oracle import zms_proto as Z
    rng = np.random.default_rng(9)
    for n, nb in ((24, 4), (90, 16)):
        a = (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(n)
        h = Z.hess_blocked(a, nb)
        assert np.all(np.tril(h, -2) == 0)
        np.testing.assert_allclose(np.linalg.norm(h), np.linalg.norm(a), rtol=1e-13)
        st = {}
        w = Z.ms_qr(h, st, nshift=8, rep=2)
        assert O.max_abs_dlambda(w, np.linalg.eigvals(a)) < 1e-10
        if n > 64:
            assert st["sweeps"] > 0 and st["stages"] > st["sweeps"]
