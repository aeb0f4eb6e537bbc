"""GPU parity tests (run on the MI355X box with ``-m gpu``): every op goes through the C ABI of libeigspec_hip.so
and is compared with the CPU oracle (oracle/) on the same seeded inputs and with the golden vectors the
reference's own functions produced (tests/golden/).  Tolerances: BASELINE.json's |dlambda| < 1e-5 for eigenvalues
(the tests hold the solver to much tighter bounds on well-conditioned inputs); histogram counts are bit-exact."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t.to(dtype) if dtype is not None else t


def known_spectrum_real(rng, nmat, n, nonnormal=0.1):
    """Real matrices Q (D + N) Q^T: D = 2x2 rotation-scaling blocks (+ one real 1x1 if n is odd) with S5-style
    eigenvalues exp((-0.5 + i w) dt), N strictly block-upper-triangular, Q a product of Householder reflectors."""
    npair = n // 2
    om = rng.uniform(0, 40, (nmat, npair))
    dt = np.exp(rng.uniform(np.log(1e-3), np.log(1e-1), (nmat, npair)))
    lam = np.exp((-0.5 + 1j * om) * dt)
    a = np.zeros((nmat, n, n))
    for p in range(npair):
        a[:, 2 * p, 2 * p] = lam[:, p].real
        a[:, 2 * p + 1, 2 * p + 1] = lam[:, p].real
        a[:, 2 * p, 2 * p + 1] = lam[:, p].imag
        a[:, 2 * p + 1, 2 * p] = -lam[:, p].imag
    spec = [lam, lam.conj()]
    if n % 2:
        r = rng.uniform(0.5, 1.0, (nmat, 1))
        a[:, n - 1, n - 1] = r[:, 0]
        spec.append(r.astype(complex))
    up = np.triu(rng.standard_normal((nmat, n, n)), 2) * nonnormal / np.sqrt(n)
    a += up
    for _ in range(3):
        u = rng.standard_normal((nmat, n, 1))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        a = a - 2 * u @ (u.transpose(0, 2, 1) @ a)
        a = a - 2 * (a @ u) @ u.transpose(0, 2, 1)
    return a, np.concatenate(spec, axis=1)


# ---- K5 real ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 8, 16, 31, 33, 64])
def test_eig_real_matches_lapack_and_known_spectrum(gpu_ops, n):
    rng = np.random.default_rng(1000 + n)
    nmat = 37                                       # not a multiple of the 8 matrices per workgroup
    a, lam = known_spectrum_real(rng, nmat, n)
    w, info = gpu_ops.eig_general_batched(dev(a))
    w, info = w.cpu().numpy(), info.cpu().numpy()
    assert w.shape == (nmat, n) and info.shape == (nmat,)
    assert np.all(info == 0)
    # hold the solver to LAPACK's own accuracy on the same matrices (both against the prescribed spectrum)
    err_lapack = O.max_abs_dlambda(O.eigvals_dense(a), lam)
    assert O.max_abs_dlambda(w, lam) < max(1e-10, 10 * err_lapack)
    assert O.max_abs_dlambda(w, O.eigvals_dense(a)) < max(1e-9, 20 * err_lapack)


@pytest.mark.parametrize("n", [6, 16, 64])
def test_eig_real_random_dense(gpu_ops, n):
    rng = np.random.default_rng(7 + n)
    a = rng.standard_normal((64, n, n)) / np.sqrt(n)
    w, info = gpu_ops.eig_general_batched(dev(a))
    assert int(info.abs().sum()) == 0
    assert O.max_abs_dlambda(w.cpu().numpy(), O.eigvals_dense(a)) < 1e-9


def test_eig_real_structured_edge_cases(gpu_ops):
    n = 12
    rng = np.random.default_rng(3)
    mats = [np.zeros((n, n)), np.eye(n), np.diag(np.arange(1.0, n + 1)),
            np.triu(rng.standard_normal((n, n)))]                       # already triangular: immediate deflation
    comp = np.diag(np.ones(n - 1), -1)
    comp[0, :] = -rng.standard_normal(n)                               # companion matrix
    mats.append(comp)
    blk = np.zeros((n, n))
    blk[:6, :6] = rng.standard_normal((6, 6))
    blk[6:, 6:] = rng.standard_normal((6, 6))                           # reducible: exact zero subdiagonal block
    mats.append(blk)
    rot = np.zeros((n, n))
    for p in range(n // 2):
        rot[2 * p, 2 * p + 1] = 1.0
        rot[2 * p + 1, 2 * p] = -1.0                                    # purely imaginary pairs +-i
    mats.append(rot)
    a = np.stack(mats)
    w, info = gpu_ops.eig_general_batched(dev(a))
    assert int(info.abs().sum()) == 0
    assert O.max_abs_dlambda(w.cpu().numpy(), O.eigvals_dense(a)) < 1e-9


def test_eig_empty_batch_and_bad_args(gpu_ops):
    w, info = gpu_ops.eig_general_batched(torch.zeros((0, 8, 8), dtype=torch.float64, device="cuda"))
    assert w.shape == (0, 8) and info.shape == (0,)
    with pytest.raises(TypeError):
        gpu_ops.eig_general_batched(torch.zeros((2, 4, 4), dtype=torch.int64, device="cuda"))
    with pytest.raises(ValueError):
        gpu_ops.eig_general_batched(torch.zeros((2, 4, 5), dtype=torch.float64, device="cuda"))


def test_hessenberg_stage_preserves_spectrum(gpu_ops):
    rng = np.random.default_rng(11)
    a = rng.standard_normal((9, 24, 24))
    h = gpu_ops.hessenberg_batched(dev(a)).cpu().numpy()
    assert np.all(np.tril(h, -2) == 0.0)
    assert O.max_abs_dlambda(O.eigvals_dense(h), O.eigvals_dense(a)) < 1e-10
    np.testing.assert_allclose(np.trace(h, axis1=1, axis2=2), np.trace(a, axis1=1, axis2=2), rtol=1e-12, atol=1e-12)
    w, info = gpu_ops.hseqr_batched(dev(h))
    assert int(info.abs().sum()) == 0
    assert O.max_abs_dlambda(w.cpu().numpy(), O.eigvals_dense(a)) < 1e-10


def test_eig_real_large_batch_properties(gpu_ops):
    """Size-independent checks at a bench-like batch: sum(lambda) = trace, conjugate-pair closure, known spectrum."""
    rng = np.random.default_rng(64)
    a, lam = known_spectrum_real(rng, 4096, 64)
    w, info = gpu_ops.eig_general_batched(dev(a))
    w = w.cpu().numpy()
    assert int(info.abs().sum()) == 0
    np.testing.assert_allclose(w.sum(axis=1).real, np.trace(a, axis1=1, axis2=2), rtol=1e-10, atol=1e-10)
    assert np.max(np.abs(w.sum(axis=1).imag)) < 1e-10
    assert O.max_abs_dlambda(w[:256], lam[:256]) < 1e-9
    # the whole batch against the known spectrum through a permutation-invariant statistic
    np.testing.assert_allclose(np.sort(np.abs(w), axis=1), np.sort(np.abs(lam), axis=1), atol=1e-9)


def test_discrete_dplr_general_p_q_matches_reference_golden(pkg, gpu_ops, golden):
    """The operator-level ``discrete_DPLR(Lambda, P, Q, B, C, step, L)`` with Q != P and Lambda as given (review r4 #5:
    it used to raise NotImplementedError): Ab from the HIP kernel (egs_s4_dplr_build_pq, clip off), Bb / Cb on request."""
    import importlib
    ev = importlib.import_module(pkg.__name__ + ".analysis.eval_eig")
    g = golden("dplr_pq")
    for i in range(2):
        args = (g[f"c{i}_Lambda"], g[f"c{i}_P"], g[f"c{i}_Q"], g[f"c{i}_B"], g[f"c{i}_C"], float(g[f"c{i}_step"]),
                int(g[f"c{i}_L"]))
        ab, bb, cb = ev.discrete_DPLR(*args)
        assert bb is None and cb is None                                    # the reference's caller discards them (Q18)
        np.testing.assert_allclose(ab.cpu().numpy(), g[f"c{i}_Ab"], rtol=1e-11, atol=1e-12)
        ab, bb, cb = ev.discrete_DPLR(*args, compute_bc=True)
        np.testing.assert_allclose(ab.cpu().numpy(), g[f"c{i}_Ab"], rtol=1e-11, atol=1e-12)
        np.testing.assert_allclose(bb.cpu().numpy(), g[f"c{i}_Bb"], rtol=1e-10, atol=1e-11)
        np.testing.assert_allclose(cb.cpu().numpy(), g[f"c{i}_Cb"], rtol=1e-8, atol=1e-9)
    # Q = P through the general entry point, clip on, equals the original entry point bit for bit
    rng = np.random.default_rng(7)
    lam_re, lam_im = -np.abs(rng.standard_normal((3, 16))) + 0.05, rng.standard_normal((3, 16))
    p = rng.standard_normal((3, 16)) + 1j * rng.standard_normal((3, 16))
    ls = np.log(rng.uniform(1e-3, 1e-1, 3))
    a1 = gpu_ops.s4_dplr_build(dev(lam_re), dev(lam_im), dev(p), dev(ls))
    a2 = gpu_ops.s4_dplr_build(dev(lam_re), dev(lam_im), dev(p), dev(ls), q=dev(p), clip=True)
    assert torch.equal(torch.view_as_real(a1), torch.view_as_real(a2))


# ---- K5 complex ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 8, 33, 64, 96, 97, 130])
def test_eig_complex_matches_lapack(gpu_ops, n):
    rng = np.random.default_rng(200 + n)
    nmat = 5
    a = (rng.standard_normal((nmat, n, n)) + 1j * rng.standard_normal((nmat, n, n))) / np.sqrt(2 * n)
    w, info = gpu_ops.eig_general_batched(dev(a))
    assert int(info.abs().sum()) == 0
    assert O.max_abs_dlambda(w.cpu().numpy(), O.eigvals_dense(a)) < 1e-9


def test_eig_real_above_64_uses_workgroup_path(gpu_ops):
    rng = np.random.default_rng(5)
    a, lam = known_spectrum_real(rng, 3, 80)
    w, info = gpu_ops.eig_general_batched(dev(a))
    assert int(info.abs().sum()) == 0
    assert O.max_abs_dlambda(w.cpu().numpy(), lam) < 1e-9


def test_s4_build_and_eigs_match_golden(gpu_ops, golden):
    g = golden("ssm")
    for i in range(2):
        lam_re = g[f"s4_{i}_Lambda_re"][:, 1][None]
        lam_im = g[f"s4_{i}_Lambda_im"][:, 1][None]
        p = g[f"s4_{i}_P"][:, 1][None]
        ls = g[f"s4_{i}_log_step"][0, 1].reshape(1)
        ab = gpu_ops.s4_dplr_build(dev(lam_re), dev(lam_im), dev(p), dev(ls))
        np.testing.assert_allclose(ab.cpu().numpy()[0], g[f"s4_{i}_Ab"], rtol=1e-11, atol=1e-12)
        w, info = gpu_ops.eig_general_batched(ab)
        assert int(info.abs().sum()) == 0
        assert O.max_abs_dlambda(w.cpu().numpy()[0], g[f"s4_{i}_eig"][:, 0]) < 1e-10


def backward_error(a, w):
    """max_i sigma_min(A - w_i I) / ||A||_2: w_i is an exact eigenvalue of A + E with ||E||_2 equal to this."""
    n = a.shape[0]
    nrm = np.linalg.norm(a, 2)
    return max(np.linalg.svd(a - z * np.eye(n), compute_uv=False)[-1] for z in w) / nrm


@pytest.mark.parametrize("n,h", [(16, 3), (64, 4), (128, 2)])
def test_s4_hippo_batched_columns(gpu_ops, n, h):
    """Config C5 in miniature: HiPPO-LegS DPLR parameters, all feature columns in one batched build + solve.

    HiPPO-LegS is unitarily similar to a lower-triangular matrix and so non-normal that its eigenvalues move by 1e-2
    under 1e-15 relative perturbations for N >= 64 (LAPACK's own zgeev output is noise at that level), so elementwise
    |dlambda| < 1e-5 against LAPACK is only meaningful for N = 16.  For every N the solver is held to the criterion
    that *is* meaningful for a backward-stable eigensolver: each computed eigenvalue is an exact eigenvalue of a matrix
    within 1e-13 ||A|| of the input (LAPACK itself achieves ~3e-16), and sum(lambda) = trace."""
    rng = np.random.default_rng(256)
    lam, p, _, _, _ = O.make_dplr_hippo(n)
    ls = rng.uniform(np.log(1e-3), np.log(1e-1), h)
    ab = gpu_ops.s4_dplr_build(dev(np.tile(lam.real, (h, 1))), dev(np.tile(lam.imag, (h, 1))), dev(np.tile(p, (h, 1))),
                               dev(ls))
    w, info = gpu_ops.eig_general_batched(ab)
    assert int(info.abs().sum()) == 0
    w = w.cpu().numpy()
    abh = ab.cpu().numpy()
    for c in range(h):
        ab_ref = O.s4_dplr_ab(np.minimum(lam.real, -1e-4) + 1j * lam.imag, p, np.exp(ls[c]))
        np.testing.assert_allclose(abh[c], ab_ref, rtol=1e-9, atol=1e-11)
        assert backward_error(abh[c], w[c]) < 1e-13
        assert abs(w[c].sum() - np.trace(abh[c])) < 1e-10 * n
        if n <= 16:
            assert O.max_abs_dlambda(w[c], O.eigvals_dense(ab_ref)) < 1e-5      # north_star tolerance


def test_s4_random_dplr_batched_columns(gpu_ops):
    """Well-conditioned DPLR (random Lambda, small P): elementwise parity with LAPACK at N = 64, 48 columns."""
    rng = np.random.default_rng(77)
    n, h = 64, 48
    lam_re = -np.abs(rng.standard_normal((h, n))) - 0.1
    lam_im = rng.standard_normal((h, n)) * 5
    p = (rng.standard_normal((h, n)) + 1j * rng.standard_normal((h, n))) * 0.2
    ls = rng.uniform(np.log(1e-3), np.log(1e-1), h)
    ab = gpu_ops.s4_dplr_build(dev(lam_re), dev(lam_im), dev(p), dev(ls))
    w, info = gpu_ops.eig_general_batched(ab)
    assert int(info.abs().sum()) == 0
    ref = np.stack([O.eigvals_dense(O.s4_dplr_ab(lam_re[c] + 1j * lam_im[c], p[c], np.exp(ls[c]))) for c in range(h)])
    assert O.max_abs_dlambda(w.cpu().numpy(), ref) < 1e-9


def test_diag_ssm_matches_golden(gpu_ops, golden):
    g = golden("ssm")
    for i in range(2):
        w = gpu_ops.diag_ssm_lambda("s5", dev(g[f"s5_{i}_Lambda_re"]), dev(g[f"s5_{i}_Lambda_im"]),
                                    dev(g[f"s5_{i}_log_step"].reshape(-1)))
        np.testing.assert_allclose(w.cpu().numpy(), g[f"s5_{i}_eig"][:, 0], rtol=1e-13, atol=1e-15)
        w = gpu_ops.diag_ssm_lambda("lru", dev(g[f"lru_{i}_nu_log"]), dev(g[f"lru_{i}_theta_log"]))
        np.testing.assert_allclose(w.cpu().numpy(), g[f"lru_{i}_eig"][:, 0], rtol=1e-13, atol=1e-15)


# ---- K2 / norm ------------------------------------------------------------------------------------------------------
def test_mamba2_matches_golden(gpu_ops, golden):
    g = golden("mamba2")
    off = int(g["d_inner"]) + 2 * int(g["ngroups"]) * int(g["d_state"])
    nh = int(g["nheads"])
    lam = gpu_ops.mamba2_lambda(dev(g["x"]), dev(g["w"][off:off + nh]), dev(g["b"][off:off + nh]), dev(g["dt_bias"]),
                                dev(g["A_log"]))
    assert lam.dtype == torch.float32
    np.testing.assert_allclose(lam.cpu().numpy()[..., None], g["lam"], rtol=1e-5, atol=1e-7)
    lti = gpu_ops.mamba2_lti_lambda(dev(g["A"]), dev(g["beta"]), g["x"].shape[0], g["x"].shape[1])
    np.testing.assert_allclose(lti.cpu().numpy()[..., None], g["lam_lti"], rtol=1e-6)


@pytest.mark.parametrize("d,nh,bias", [(128, 4, False), (512, 8, True), (20, 3, True), (130, 12, False), (7, 1, True),
                                       (64, 2, True), (256, 1, False), (1024, 5, True), (128, 8, False), (128, 9, False)])
def test_mamba2_random_vs_oracle(gpu_ops, d, nh, bias):
    rng = np.random.default_rng(d * 31 + nh)
    bsz, seqlen = 3, 171
    x = rng.standard_normal((bsz, seqlen, d)).astype(np.float32)
    w = (rng.uniform(-1, 1, (nh, d)) / np.sqrt(d)).astype(np.float32) * 4
    b = rng.standard_normal(nh).astype(np.float32) if bias else None
    dt = np.exp(rng.uniform(np.log(1e-3), np.log(1e-1), nh))
    dt_bias = (dt + np.log(-np.expm1(-dt))).astype(np.float32)
    a_log = np.log(rng.uniform(1, 16, nh)).astype(np.float32)
    got = gpu_ops.mamba2_lambda(dev(x), dev(w), dev(b) if bias else None, dev(dt_bias), dev(a_log)).cpu().numpy()
    want = O.lambda_mamba2(x, w, b, dt_bias, a_log, 0, 0, 0, nh)[..., 0]
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("fn", ["exp", "elu", "softplus", "sigmoid"])
@pytest.mark.parametrize("use_off", [0, 1])
def test_norm_attention_matches_golden(gpu_ops, golden, fn, use_off):
    g = golden("norm_attention")
    d_model, d_qk, h = int(g["D"]), int(g["d_qk"]), int(g["H"])
    lo = d_model + 2 * d_qk
    eta = gpu_ops.normattn_eta(dev(g["x"]), dev(g["w"][lo:lo + h]), dev(g["b"][lo:lo + h]),
                               dev(g["offset"]) if use_off else None, fn, h).cpu().numpy()[..., None]
    want = g[f"eta_{fn}_{use_off}"]
    finite = np.isfinite(want)
    assert np.array_equal(finite, np.isfinite(eta))
    np.testing.assert_allclose(eta[finite], want[finite], rtol=1e-4)


def test_norm_attention_unknown_fn_raises(gpu_ops):
    x = torch.zeros((1, 4, 8), device="cuda")
    with pytest.raises(RuntimeError, match="normalization function tanh not implemented!"):
        gpu_ops.normattn_eta(x, torch.zeros((2, 8), device="cuda"), None, None, "tanh", 2)


def test_norm_attention_tile_boundaries_vs_oracle(gpu_ops):
    rng = np.random.default_rng(12)
    bsz, seqlen, d, h, d_qk = 3, 1500, 32, 4, 8              # > one 1024-token tile, sequence ends inside tiles
    x = rng.standard_normal((bsz, seqlen, d)).astype(np.float32)
    w = rng.standard_normal((d + 2 * d_qk + h, d)).astype(np.float32) / np.sqrt(d)
    b = rng.standard_normal(d + 2 * d_qk + h).astype(np.float32) * 0.1
    lo = d + 2 * d_qk
    got = gpu_ops.normattn_eta(dev(x), dev(w[lo:]), dev(b[lo:]), None, "softplus", h).cpu().numpy()
    want = O.eta_norm(x, w, b, None, "softplus", d_qk, h, d)[..., 0]
    np.testing.assert_allclose(got, want, rtol=2e-5)


# ---- K3 / K4 ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case,rtol", [("a", 1e-5), ("b", 1e-5), ("c", 1e-5)])
def test_linear_attention_matches_golden(gpu_ops, golden, case, rtol):
    g = golden("attention")
    d_qk, h = int(g["d_qk"]), int(g["H"])
    eta = gpu_ops.linattn_eta(dev(g[f"x_{case}"]), dev(g[f"w_{case}"][:2 * d_qk]), dev(g[f"b_{case}"][:2 * d_qk]),
                              d_qk, h).cpu().numpy()[..., None]
    np.testing.assert_allclose(eta, g[f"linear_{case}"], rtol=rtol)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_softmax_attention_matches_golden(gpu_ops, golden, case):
    """The reference forms q.k in fp32, so its own eta carries the score noise e^(1e-7 |score|) -- 1e-6 in case a,
    2e-4 in case c (|score| ~ 500).  A fixed rtol would either hide real differences in case a or have to be loosened
    to case c's noise.  Instead both results are measured against the float64-exact value of the same formula
    (oracle.eta_softmax_exact): the HIP path may be at most twice as far from it as the reference itself is."""
    g = golden("attention")
    d_qk, h = int(g["d_qk"]), int(g["H"])
    eta = gpu_ops.softmax_eta(dev(g[f"x_{case}"]), dev(g[f"w_{case}"][:2 * d_qk]), dev(g[f"b_{case}"][:2 * d_qk]),
                              d_qk, h).cpu().numpy()[..., None]
    ref = g[f"softmax_{case}"]
    with np.errstate(all="ignore"):
        exact = O.eta_softmax_exact(g[f"x_{case}"], g[f"w_{case}"], g[f"b_{case}"], d_qk, h)
    assert np.array_equal(np.isfinite(ref), np.isfinite(eta)) and np.array_equal(ref == 0, eta == 0)
    ok = np.isfinite(exact) & (exact != 0) & np.isfinite(ref) & (ref != 0)
    err_ref = np.max(np.abs(ref[ok] - exact[ok]) / np.abs(exact[ok]))
    err_hip = np.max(np.abs(eta[ok] - exact[ok]) / np.abs(exact[ok]))
    assert err_ref < 3e-4                                     # the reference's own distance (1.2e-6 / 1.5e-5 / 1.9e-4)
    assert err_hip <= 2.0 * err_ref + 1e-6, (err_hip, err_ref)       # 1e-6: the fp32 exponential's own floor (3e-7)


@pytest.mark.parametrize("seqlen,d,d_qk,h", [(130, 24, 48, 2), (64, 16, 16, 1), (257, 40, 160, 1), (2, 8, 8, 4),
                                              (300, 18, 12, 3), (70, 36, 320, 1), (513, 128, 128, 1),
                                              (320, 32, 64, 2), (768, 64, 128, 1)])
def test_attention_random_vs_oracle(gpu_ops, seqlen, d, d_qk, h):
    rng = np.random.default_rng(seqlen * 7 + d_qk)
    bsz = 3
    x = rng.standard_normal((bsz, seqlen, d)).astype(np.float32)
    w = (rng.standard_normal((2 * d_qk + d, d)) / np.sqrt(d)).astype(np.float32)
    b = (rng.standard_normal(2 * d_qk + d) * 0.1).astype(np.float32)
    lin = gpu_ops.linattn_eta(dev(x), dev(w[:2 * d_qk]), dev(b[:2 * d_qk]), d_qk, h).cpu().numpy()
    np.testing.assert_allclose(lin, O.eta_linear(x, w, b, d_qk, h, d)[..., 0], rtol=1e-5)
    sm = gpu_ops.softmax_eta(dev(x), dev(w[:2 * d_qk]), dev(b[:2 * d_qk]), d_qk, h).cpu().numpy()
    np.testing.assert_allclose(sm, O.eta_softmax(x, w, b, d_qk, h, d)[..., 0], rtol=5e-5)


@pytest.mark.parametrize("heads", [1, 2])
@pytest.mark.parametrize("seqlen", [128, 200, 391])
def test_linear_attention_fused_path(gpu_ops, monkeypatch, heads, seqlen):
    """The fused projection + prefix kernel (batch >= 128, 2 d_qk = 256, H in {1, 2}; q and k never reach HBM) against
    the streaming oracle and against the two-kernel path on the same inputs, with sequence lengths that end inside a
    128-token tile and inside a 32-token round."""
    rng = np.random.default_rng(41 + seqlen + heads)
    b, d, d_qk = 130, 64, 128
    x = rng.standard_normal((b, seqlen, d)).astype(np.float32)
    w = (rng.standard_normal((2 * d_qk + d, d)) / np.sqrt(d)).astype(np.float32)
    bias = (rng.standard_normal(2 * d_qk + d) * 0.1).astype(np.float32)
    got = gpu_ops.linattn_eta(dev(x), dev(w[:2 * d_qk]), dev(bias[:2 * d_qk]), d_qk, heads).cpu().numpy()
    want = O.eta_linear_streaming(x[:6], w, bias, d_qk, heads, d)[..., 0]
    np.testing.assert_allclose(got[:6], want, rtol=1e-5)
    monkeypatch.setenv("EGS_LINATTN_UNFUSED", "1")
    ref = gpu_ops.linattn_eta(dev(x), dev(w[:2 * d_qk]), dev(bias[:2 * d_qk]), d_qk, heads).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-9)          # same fp32 projection bit for bit, fp64 sums in another order


@pytest.mark.parametrize("heads", [1, 2])
def test_linear_attention_fused_kernel_at_the_c4_shard_shape(gpu_ops, monkeypatch, heads):
    """BASELINE config C4's own shape on the kernel that serves it: B = 128 (the fused kernel's threshold), L = 4096,
    D = 128, d_qk = 128 -- 32 tiles of register-carried prefix state per sequence.  First and last three sequences
    against the O(L) float64 oracle, the whole batch against the two-kernel path."""
    g = torch.Generator(device="cuda")
    g.manual_seed(4096 + heads)
    bsz, seqlen, d, d_qk = 128, 4096, 128, 128
    x = torch.randn((bsz, seqlen, d), generator=g, device="cuda")
    w = (torch.rand((2 * d_qk + d, d), generator=g, device="cuda") * 2 - 1) / np.sqrt(d)
    b = (torch.rand((2 * d_qk + d,), generator=g, device="cuda") * 2 - 1) / np.sqrt(d)
    got = gpu_ops.linattn_eta(x, w[:2 * d_qk].contiguous(), b[:2 * d_qk].contiguous(), d_qk, heads)
    assert got.shape == (bsz, seqlen - 1, heads) and bool(torch.isfinite(got).all()) and bool((got > 0).all())
    pick = [0, 1, 2, bsz - 3, bsz - 2, bsz - 1]
    want = O.eta_linear_streaming(x[pick].cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), d_qk, heads, d)[..., 0]
    np.testing.assert_allclose(got[pick].cpu().numpy(), want, rtol=1e-5)
    monkeypatch.setenv("EGS_LINATTN_UNFUSED", "1")
    ref = gpu_ops.linattn_eta(x, w[:2 * d_qk].contiguous(), b[:2 * d_qk].contiguous(), d_qk, heads)
    monkeypatch.delenv("EGS_LINATTN_UNFUSED")
    err = float(((got - ref).abs() / ref.abs()).max().item())
    assert err < 1e-9, err                                   # same fp32 projection bit for bit, fp64 sums in another order


@pytest.mark.parametrize("seqlen,d_qk,h", [(70, 16, 2), (300, 128, 1), (513, 192, 3)])
def test_linear_attention_eta_from_existing_qk(gpu_ops, seqlen, d_qk, h):
    """The (q, k)-input entry point (raw projections in, elu + 1 applied on the device) against the x-input path fed
    the same projections, and against the oracle."""
    rng = np.random.default_rng(seqlen + d_qk)
    bsz, d = 3, 24
    x = rng.standard_normal((bsz, seqlen, d)).astype(np.float32)
    w = (rng.standard_normal((2 * d_qk + d, d)) / np.sqrt(d)).astype(np.float32)
    b = (rng.standard_normal(2 * d_qk + d) * 0.1).astype(np.float32)
    qk = O.linear_f32(x, w[:2 * d_qk], b[:2 * d_qk])
    q = qk[..., :d_qk].reshape(bsz, seqlen, h, d_qk // h)
    k = qk[..., d_qk:].reshape(bsz, seqlen, h, d_qk // h)
    got = gpu_ops.linattn_eta_from_qk(dev(q), dev(k)).cpu().numpy()
    np.testing.assert_allclose(got, O.eta_linear(x, w, b, d_qk, h, d)[..., 0], rtol=1e-5)
    via_x = gpu_ops.linattn_eta(dev(x), dev(w[:2 * d_qk]), dev(b[:2 * d_qk]), d_qk, h).cpu().numpy()
    np.testing.assert_allclose(got, via_x, rtol=2e-6)


def test_linear_attention_long_sequence_streaming_property(gpu_ops):
    """At a length the quadratic reference form cannot reach, check against the O(L) float64 restatement."""
    rng = np.random.default_rng(4096)
    bsz, seqlen, d, d_qk, h = 2, 4096, 128, 128, 1
    x = rng.standard_normal((bsz, seqlen, d)).astype(np.float32)
    w = rng.uniform(-1, 1, (2 * d_qk + d, d)).astype(np.float32) / np.sqrt(d)
    b = rng.uniform(-1, 1, 2 * d_qk + d).astype(np.float32) / np.sqrt(d)
    got = gpu_ops.linattn_eta(dev(x), dev(w[:2 * d_qk]), dev(b[:2 * d_qk]), d_qk, h).cpu().numpy()
    want = O.eta_linear_streaming(x, w, b, d_qk, h, d)[..., 0]
    np.testing.assert_allclose(got, want, rtol=1e-5)
    assert np.all(got > 0)


# ---- histograms -----------------------------------------------------------------------------------------------------
def test_threshold_hist_bit_exact_vs_golden(gpu_ops, golden):
    g = golden("thresholds")
    pct = gpu_ops.threshold_hist(dev(g["vals"]), O.THRESHOLDS_RADIUS, "value").cpu().numpy()
    np.testing.assert_array_equal(pct, g["pct"])
    np.testing.assert_array_equal(gpu_ops.threshold_hist(dev(g["vals"]), O.THRESHOLDS_PHASE, "zero").cpu().numpy(),
                                  g["pct_phase"])
    c = dev(g["cplx"]).unsqueeze(0)
    np.testing.assert_array_equal(gpu_ops.threshold_hist(c, O.THRESHOLDS_RADIUS, "radius").cpu().numpy()[:, 0], g["pct_s"])
    np.testing.assert_array_equal(gpu_ops.threshold_hist(c, O.THRESHOLDS_PHASE, "phase").cpu().numpy()[:, 0], g["pct_sp"])


@pytest.mark.parametrize("shape", [(5, 700, 3), (2, 33, 300), (1, 5000, 1), (70, 9, 2, 3)])
def test_threshold_hist_random_layouts(gpu_ops, shape):
    rng = np.random.default_rng(sum(shape))
    vals = np.abs(rng.standard_normal(shape)) * rng.choice([0.05, 0.7, 1.0, 30.0, 500.0], size=shape)
    vals.flat[::17] = rng.choice([0.1, 0.5, 0.9, 1.0, 10.0, 100.0], size=vals.flat[::17].shape)   # exact edges
    got = gpu_ops.threshold_hist(dev(vals), O.THRESHOLDS_RADIUS, "value").cpu().numpy()
    outer, t_len = shape[0], shape[1]
    flat = vals.reshape(outer, t_len, -1)
    want = np.stack([m.sum(axis=1) / t_len * 100 for m in O.eig_oracle._bin_masks(flat, O.THRESHOLDS_RADIUS)])
    np.testing.assert_array_equal(got.reshape(want.shape), want)
    f32 = gpu_ops.threshold_hist(dev(vals.astype(np.float32)), O.THRESHOLDS_RADIUS, "radius").cpu().numpy()
    want32 = np.stack([m.sum(axis=1) / t_len * 100
                       for m in O.eig_oracle._bin_masks(flat.astype(np.float32).astype(np.float64), O.THRESHOLDS_RADIUS)])
    np.testing.assert_array_equal(f32.reshape(want32.shape), want32)


# ---- robustness / degenerate shapes ---------------------------------------------------------------------------------
@pytest.mark.parametrize("scale", [1e-200, 1e-150, 1e150, 1e250])
def test_eig_real_extreme_scaling(gpu_ops, scale):
    """xGEEV-style scaling: entries far outside the range where squares are representable."""
    rng = np.random.default_rng(21)
    a, lam = known_spectrum_real(rng, 9, 24)
    w, info = gpu_ops.eig_general_batched(dev(a * scale))
    assert int(info.abs().sum()) == 0
    assert O.max_abs_dlambda(w.cpu().numpy() / scale, lam) < 1e-9


def test_eig_nan_matrix_is_isolated(gpu_ops):
    """A NaN / Inf matrix must terminate (info > 0, NaN spectrum) without touching its batch neighbours."""
    rng = np.random.default_rng(22)
    a, lam = known_spectrum_real(rng, 10, 16)
    a[3, 5, 7] = np.nan
    a[6, 0, 0] = np.inf
    w, info = gpu_ops.eig_general_batched(dev(a))
    w, info = w.cpu().numpy(), info.cpu().numpy()
    good = [i for i in range(10) if i not in (3, 6)]
    assert np.all(info[good] == 0) and info[3] == 16 and info[6] == 16      # like LAPACK: all n outstanding
    assert np.all(np.isnan(w[3].real)) and np.all(np.isnan(w[6].real))
    assert O.max_abs_dlambda(w[good], lam[good]) < 1e-9
    ac = a[good].astype(complex)
    ac[1, 2, 2] = complex(np.nan, 0.0)
    wc, infoc = gpu_ops.eig_general_batched(dev(ac))
    infoc = infoc.cpu().numpy()
    assert infoc[1] == 16 and np.all(np.delete(infoc, 1) == 0)
    assert np.all(np.isnan(wc.cpu().numpy()[1].real))


def test_degenerate_shapes(gpu_ops):
    z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device="cuda")
    # sequences of length 1 have no ratio; empty batches are no-ops
    assert gpu_ops.linattn_eta(z(3, 1, 8), z(8, 8), None, 4, 2).shape == (3, 0, 2)
    assert gpu_ops.softmax_eta(z(0, 5, 8), z(8, 8), None, 4, 2).shape == (0, 4, 2)
    assert gpu_ops.normattn_eta(z(2, 1, 8), z(2, 8), None, None, "exp", 2).shape == (2, 0, 2)
    assert gpu_ops.mamba2_lambda(z(0, 7, 8), z(2, 8), None, z(2), z(2)).shape == (0, 7, 2)
    one = gpu_ops.mamba2_lambda(torch.ones((1, 1, 128), device="cuda"), z(4, 128), None, z(4), z(4))
    np.testing.assert_allclose(one.cpu().numpy(), np.exp(-np.log1p(np.exp(np.float32(0.0)))), rtol=1e-6)
    pct = gpu_ops.threshold_hist(z(2, 0, 3, dt=torch.float64), O.THRESHOLDS_RADIUS, "value")
    assert pct.shape == (7, 2, 3)          # T = 0: 0/0 like NumPy
    assert gpu_ops.diag_ssm_lambda("lru", z(0, dt=torch.float64), z(0, dt=torch.float64)).shape == (0,)
    w, info = gpu_ops.eig_general_batched(z(5, 0, 0, dt=torch.float64))
    assert w.shape == (5, 0) and info.shape == (5,)


def test_non_contiguous_and_strided_inputs(gpu_ops):
    rng = np.random.default_rng(23)
    a, lam = known_spectrum_real(rng, 6, 12)
    big = torch.zeros((6, 12, 20), dtype=torch.float64, device="cuda")
    big[:, :, 3:15] = dev(a)
    w, info = gpu_ops.eig_general_batched(big[:, :, 3:15])            # non-contiguous view
    assert int(info.abs().sum()) == 0 and O.max_abs_dlambda(w.cpu().numpy(), lam) < 1e-9
    w, info = gpu_ops.eig_general_batched(dev(a).reshape(2, 3, 12, 12))   # leading batch dims
    assert w.shape == (2, 3, 12) and info.shape == (2, 3)


def test_chunked_execution_paths(gpu_ops, monkeypatch):
    """Batches larger than the scratch space are processed in chunks: dense solver (explicit small workspace) and the
    attention kernels (chunk size forced through the EGS_ATTN_CHUNK test hook)."""
    rng = np.random.default_rng(31)
    a, lam = known_spectrum_real(rng, 11, 16)
    small_ws = torch.empty(3 * 16 * 16 * 8, dtype=torch.uint8, device="cuda")        # room for 3 of the 11 matrices
    w, info = gpu_ops.eig_general_batched(dev(a), workspace=small_ws)
    assert int(info.abs().sum()) == 0 and O.max_abs_dlambda(w.cpu().numpy(), lam) < 1e-9
    bsz, seqlen, d, d_qk, h = 5, 70, 16, 16, 2
    x = rng.standard_normal((bsz, seqlen, d)).astype(np.float32)
    wq = (rng.standard_normal((2 * d_qk + d, d)) / np.sqrt(d)).astype(np.float32)
    bq = (rng.standard_normal(2 * d_qk + d) * 0.1).astype(np.float32)
    monkeypatch.setenv("EGS_ATTN_CHUNK", "2")                                          # 3 chunks: 2 + 2 + 1 sequences
    lin = gpu_ops.linattn_eta(dev(x), dev(wq[:2 * d_qk]), dev(bq[:2 * d_qk]), d_qk, h).cpu().numpy()
    sm = gpu_ops.softmax_eta(dev(x), dev(wq[:2 * d_qk]), dev(bq[:2 * d_qk]), d_qk, h).cpu().numpy()
    np.testing.assert_allclose(lin, O.eta_linear(x, wq, bq, d_qk, h, d)[..., 0], rtol=1e-5)
    np.testing.assert_allclose(sm, O.eta_softmax(x, wq, bq, d_qk, h, d)[..., 0], rtol=5e-5)


@pytest.mark.parametrize("d,h,fn,use_off", [(128, 4, "softplus", True), (64, 2, "exp", False), (512, 8, "sigmoid", True),
                                            (256, 3, "elu", False), (1024, 1, "softplus", False)])
def test_norm_attention_fast_path_vs_oracle(gpu_ops, d, h, fn, use_off):
    """Shapes served by normattn_fast_kernel (overlapped wave iterations); sequence ends fall inside iterations."""
    rng = np.random.default_rng(d + h)
    bsz, seqlen, d_qk = 3, 203, 8
    x = rng.standard_normal((bsz, seqlen, d)).astype(np.float32)
    w = (rng.standard_normal((d + 2 * d_qk + h, d)) / np.sqrt(d)).astype(np.float32)
    b = (rng.standard_normal(d + 2 * d_qk + h) * 0.1).astype(np.float32)
    off = rng.standard_normal(h).astype(np.float32) if use_off else None
    lo = d + 2 * d_qk
    got = gpu_ops.normattn_eta(dev(x), dev(w[lo:]), dev(b[lo:]), dev(off) if use_off else None, fn, h).cpu().numpy()
    want = O.eta_norm(x, w, b, off, fn, d_qk, h, d)[..., 0]
    assert got.shape == want.shape == (bsz, seqlen - 1, h)
    np.testing.assert_allclose(got, want, rtol=3e-5)


# ---- rare data-dependent branches (exceptional shifts, sweeps started below a small subdiagonal) -------------------
def _cyclic(n):
    p = np.zeros((n, n))
    p[np.arange(1, n), np.arange(n - 1)] = 1.0
    p[0, n - 1] = 1.0
    return p


@pytest.mark.parametrize("n", [3, 4, 8, 16, 33, 64])
def test_eig_real_cyclic_permutation_needs_exceptional_shifts(gpu_ops, n):
    """The cyclic shift matrix is an orthogonal Hessenberg matrix on which the standard double shift is stationary
    (trailing 2x2 block [[0,0],[1,0]] gives zero shifts forever): convergence relies on the exceptional shifts taken
    every 10 stalled iterations.  Its spectrum is known in closed form (the n-th roots of unity)."""
    mats = np.stack([_cyclic(n), 2.5 * _cyclic(n), -_cyclic(n).T])
    w, info = gpu_ops.eig_general_batched(dev(mats))
    assert int(info.abs().sum()) == 0
    roots = np.exp(2j * np.pi * np.arange(n) / n)
    w = w.cpu().numpy()
    assert O.max_abs_dlambda(w[0], roots) < 1e-10
    assert O.max_abs_dlambda(w[1], 2.5 * roots) < 1e-10
    assert O.max_abs_dlambda(w[2], -roots) < 1e-10


@pytest.mark.parametrize("n", [4, 16, 64, 100])
def test_eig_complex_cyclic_permutation(gpu_ops, n):
    phase = np.exp(0.3j)
    mats = np.stack([_cyclic(n).astype(complex) * phase, (1 + 2j) * _cyclic(n).T])
    w, info = gpu_ops.eig_general_batched(dev(mats))
    assert int(info.abs().sum()) == 0
    roots = np.exp(2j * np.pi * np.arange(n) / n)
    w = w.cpu().numpy()
    assert O.max_abs_dlambda(w[0], phase * roots) < 1e-10
    assert O.max_abs_dlambda(w[1], (1 + 2j) * roots) < 1e-10


@pytest.mark.parametrize("n", [12, 40, 64])
def test_eig_graded_matrices_small_subdiagonals(gpu_ops, n):
    """Strongly graded Hessenberg-like matrices: many subdiagonal entries are tiny but not negligible, which exercises the
    'two consecutive small subdiagonals' sweep start (m > l) and the conservative deflation criterion."""
    rng = np.random.default_rng(n)
    mats = []
    for decay in (0.5, 0.2, 0.05):
        g = decay ** np.arange(n)
        a = rng.standard_normal((n, n)) * g[:, None] * 1.0            # rows scaled by a geometric grading
        a += np.diag(np.linspace(1.0, 2.0, n))
        mats.append(np.triu(a, -1))                                      # upper Hessenberg, graded subdiagonal
    mats = np.stack(mats)
    w, info = gpu_ops.eig_general_batched(dev(mats))
    assert int(info.abs().sum()) == 0
    ref = O.eigvals_dense(mats)
    assert O.max_abs_dlambda(w.cpu().numpy(), ref) < 1e-8
    wc, infoc = gpu_ops.eig_general_batched(dev(mats.astype(complex) * np.exp(0.7j)))
    assert int(infoc.abs().sum()) == 0
    assert O.max_abs_dlambda(wc.cpu().numpy(), ref * np.exp(0.7j)) < 1e-8


def test_eig_randomised_differential(gpu_ops):
    """Many small random cases of mixed structure against LAPACK (sizes 2..64, real and complex)."""
    rng = np.random.default_rng(2026)
    for n in (2, 3, 5, 7, 11, 17, 23, 32, 47, 64):
        cases = []
        for kind in range(6):
            a = rng.standard_normal((n, n))
            if kind == 1:
                a = np.triu(a, -1)                                        # already Hessenberg
            elif kind == 2:
                a = a + a.T                                               # symmetric: real spectrum
            elif kind == 3:
                a = a - a.T                                               # skew: imaginary spectrum
            elif kind == 4:
                a[n // 2:, : n // 2] = 0.0                                # block triangular
            elif kind == 5:
                a = a * (rng.random((n, n)) < 0.3)                        # sparse
            cases.append(a / max(1.0, np.sqrt(n)))
        a = np.stack(cases)
        w, info = gpu_ops.eig_general_batched(dev(a))
        assert int(info.abs().sum()) == 0, n
        err_scale = 1e-7 if n > 32 else 1e-9
        assert O.max_abs_dlambda(w.cpu().numpy(), O.eigvals_dense(a)) < err_scale, n
        ac = a + 1j * np.roll(a, 1, axis=0)
        wc, infoc = gpu_ops.eig_general_batched(dev(ac))
        assert int(infoc.abs().sum()) == 0, n
        assert O.max_abs_dlambda(wc.cpu().numpy(), O.eigvals_dense(ac)) < err_scale, n


@pytest.mark.parametrize("n", [6, 16, 40, 64, 80, 130])
@pytest.mark.parametrize("cplx", [False, True])
def test_eig_balancing_badly_scaled(gpu_ops, n, cplx):
    """A = D B D^-1 with D spanning 16 decades: without xGEBAL's diagonal scaling the small eigenvalues drown in the
    rounding of the large entries; with it the spectrum of B comes back (numpy.linalg.eigvals balances too)."""
    rng = np.random.default_rng(300 + n)
    b = rng.standard_normal((4, n, n)) / np.sqrt(n)
    if cplx:
        b = b + 1j * rng.standard_normal((4, n, n)) / np.sqrt(n)
    d = 2.0 ** rng.integers(-26, 27, size=(4, n)).astype(np.float64)       # exact similarity
    a = d[:, :, None] * b / d[:, None, :]
    ref = O.eigvals_dense(b)
    w, info = gpu_ops.eig_general_batched(dev(a))
    assert int(info.abs().sum()) == 0
    err = O.max_abs_dlambda(w.cpu().numpy(), ref)
    lap = O.max_abs_dlambda(O.eigvals_dense(a), ref)
    assert err < max(1e-10, 50 * lap), (err, lap)


@pytest.mark.parametrize("n", [5, 12, 33, 64, 96, 150])
@pytest.mark.parametrize("cplx", [False, True])
def test_eig_balancing_isolates_eigenvalues(gpu_ops, n, cplx):
    """Reducible matrices: P T P^T with T block upper triangular (triangular head and tail around a dense block).  The
    permutation step of xGEBAL recovers the diagonal entries of the triangular parts EXACTLY, including defective
    repeated ones that an unpermuted QR iteration can only resolve to eps^(1/k)."""
    rng = np.random.default_rng(400 + n)
    nb = 4
    a = np.zeros((nb, n, n), dtype=np.complex128 if cplx else np.float64)
    exact = []
    for b in range(nb):
        h, t = n // 3, n // 4
        m = rng.standard_normal((n, n))
        if cplx:
            m = m + 1j * rng.standard_normal((n, n))
        m[h:, :h] = 0.0
        m[n - t:, : n - t] = 0.0
        m[:h, :h] = np.triu(m[:h, :h])
        m[n - t:, n - t:] = np.triu(m[n - t:, n - t:])
        m[np.arange(h), np.arange(h)] = 0.5                                 # defective multiple eigenvalue
        perm = rng.permutation(n)
        a[b] = m[np.ix_(perm, perm)]
        exact.append(np.concatenate([np.diag(m)[:h], np.diag(m)[n - t:]]))
    w, info = gpu_ops.eig_general_batched(dev(a))
    assert int(info.abs().sum()) == 0
    w = w.cpu().numpy()
    for b in range(nb):
        for lam in exact[b]:
            assert np.min(np.abs(w[b] - lam)) == 0.0, (b, lam)            # bit-exact isolated eigenvalues
    assert O.max_abs_dlambda(w, O.eigvals_dense(a)) < 1e-9


def test_eig_balancing_triangular_and_diagonal(gpu_ops):
    """Fully reducible inputs (triangular, permuted triangular, diagonal) return the diagonal exactly."""
    rng = np.random.default_rng(77)
    for n in (1, 2, 7, 64, 90, 200):
        t = np.triu(rng.standard_normal((n, n)))
        perm = rng.permutation(n)
        a = np.stack([t, t.T.copy(), t[np.ix_(perm, perm)], np.diag(np.diag(t))])
        w, info = gpu_ops.eig_general_batched(dev(a))
        assert int(info.abs().sum()) == 0
        w = w.cpu().numpy()
        for b in range(4):
            assert np.array_equal(np.sort(w[b].real), np.sort(np.diag(t))), (n, b)
            assert np.all(w[b].imag == 0.0)


@pytest.mark.parametrize("cfg", [
    # B, L, D, d_qk, H
    (3, 70, 24, 16, 2),        # head_dim 8 (MFMA J = 4, zero-padded)
    (2, 200, 32, 48, 2),       # head_dim 24
    (2, 129, 40, 128, 1),      # head_dim 128, ragged last tile
    (1, 90, 16, 20, 2),        # head_dim 10: not a multiple of 4 -> VALU tile kernel
])
def test_softmax_eta_matrix_core_and_valu_kernels(gpu_ops, monkeypatch, cfg):
    """K4: the matrix-core row-statistics kernel and the VALU tile kernel against the oracle and each other."""
    B, L, D, d_qk, H = cfg
    rng = np.random.default_rng(sum(cfg))
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    w = (rng.standard_normal((2 * d_qk + D, D)) / np.sqrt(D)).astype(np.float32)
    b = (rng.standard_normal(2 * d_qk + D) * 0.1).astype(np.float32)
    want = O.eta_softmax(x, w, b, d_qk, H, D)[..., 0]
    got = gpu_ops.softmax_eta(dev(x), dev(w[: 2 * d_qk]), dev(b[: 2 * d_qk]), d_qk, H).cpu().numpy()
    monkeypatch.setenv("EGS_SOFTMAX_VALU", "1")
    valu = gpu_ops.softmax_eta(dev(x), dev(w[: 2 * d_qk]), dev(b[: 2 * d_qk]), d_qk, H).cpu().numpy()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-5 * scale
    assert np.abs(valu - want).max() <= 2e-5 * scale
    assert np.abs(got - valu).max() <= 2e-6 * scale


@pytest.mark.parametrize("n,nmat", [(64, 6000), (40, 9001), (17, 20000)])
def test_eig_real_work_queue_covers_every_matrix(gpu_ops, n, nmat):
    """Batches larger than one resident wave set run through the QR kernel's work queue: every matrix must be solved
    exactly once -- checked for ALL matrices through the trace (sum of eigenvalues) and the squared Frobenius-type
    invariant sum(lambda^2) = trace(A^2), and against LAPACK on a sample."""
    g = torch.Generator(device="cuda")
    g.manual_seed(n * 1000 + 7)
    a = torch.randn((nmat, n, n), generator=g, device="cuda", dtype=torch.float64) / np.sqrt(n)
    w, info = gpu_ops.eig_general_batched(a)
    assert int(info.abs().sum()) == 0
    tr = torch.diagonal(a, dim1=1, dim2=2).sum(dim=1)
    tr2 = torch.einsum("bij,bji->b", a, a)
    assert float((w.sum(dim=1).real - tr).abs().max()) < 1e-10
    assert float((w.sum(dim=1).imag).abs().max()) < 1e-10
    assert float(((w * w).sum(dim=1).real - tr2).abs().max()) < 1e-9
    idx = torch.tensor([0, 1, nmat // 2, nmat - 2, nmat - 1], device="cuda")
    assert O.max_abs_dlambda(w[idx].cpu().numpy(), O.eigvals_dense(a[idx].cpu().numpy())) < 1e-9


@pytest.mark.parametrize("kind", ["rising", "falling", "spiky", "huge_negative", "ragged"])
def test_softmax_one_pass_kernel_on_adversarial_score_profiles(gpu_ops, kind):
    """K4 keeps a running row maximum and rescales its running sum when a key block raises it (round 4: one pass over
    the score tiles instead of two).  Profiles that stress exactly that: maxima that rise with every key block (rescale
    in every block), fall from the first key on (never again), single spikes late in the row, scores so negative that
    every causal term underflows next to the masked zeros of quirk Q5, and a length that ends inside a tile.  Through the
    (q, k) entry, against the oracle's restatement of the notebook cell / eval_eig.py:43-95."""
    rng = np.random.default_rng(len(kind) * 31)
    B, L, H, hd = 2, (391 if kind == "ragged" else 320), 2, 16
    q = rng.standard_normal((B, L, H, hd)).astype(np.float32)
    k = rng.standard_normal((B, L, H, hd)).astype(np.float32)
    ramp = np.linspace(0.0, 1.0, L, dtype=np.float32)[None, :, None, None]
    if kind == "rising":
        k = k * 0.05 + 6.0 * ramp * np.sign(q.mean(axis=1, keepdims=True))       # later keys score higher and higher
        q = np.abs(q) * np.sign(q.mean(axis=1, keepdims=True))
    elif kind == "falling":
        k = k * 0.05 - 6.0 * ramp * np.sign(q.mean(axis=1, keepdims=True))
        q = np.abs(q) * np.sign(q.mean(axis=1, keepdims=True))
    elif kind == "spiky":
        k[:, 250] *= 40.0                                                        # one key far above the rest, late
        k[:, 3] *= 25.0
    elif kind == "huge_negative":
        q = np.abs(q) + 1.0
        k = -(np.abs(k) + 1.0) * 8.0                                             # every score << 0: the masked zeros are the row maximum
    want = O.eta_softmax_from_qk(q, k)[..., 0]
    got = gpu_ops.softmax_eta_from_qk(dev(q), dev(k)).cpu().numpy()
    assert got.shape == want.shape and np.all(np.isfinite(got) == np.isfinite(want))
    # Scores of several hundred carry fp32 rounding of 3e-5 into exp(score - max), in the reference's arithmetic (which
    # the oracle restates) as much as in the kernel's -- but not the same rounding: the kernel subtracts the RUNNING
    # maximum in fp32 and rescales exactly.  So, as in the golden test, both are measured against the float64 value of
    # the same formula; the kernel may be at most twice as far from it as the reference's arithmetic is.
    q64, k64 = q.astype(np.float64), k.astype(np.float64)
    sc = np.einsum("bthd,bshd->btsh", q64, k64)
    tril = np.tril(np.ones((L, L)))
    sc = sc * tril[None, :, :, None]
    m = sc.max(axis=2)
    nu = (np.exp(sc - m[:, :, None, :]) * tril[None, :, :, None]).sum(axis=2) + (L - 1 - np.arange(L))[None, :, None]
    with np.errstate(over="ignore", invalid="ignore"):
        exact = nu[:, :-1] / nu[:, 1:] * np.exp(m[:, :-1] - m[:, 1:])
    ok = np.isfinite(exact) & (exact != 0) & np.isfinite(want) & (want != 0)
    err_ref = np.max(np.abs(want[ok] - exact[ok]) / np.abs(exact[ok]))
    err_hip = np.max(np.abs(got[ok] - exact[ok]) / np.abs(exact[ok]))
    assert err_ref < 1e-3, err_ref
    assert err_hip <= 2.0 * err_ref + 1e-6, (err_hip, err_ref)       # 1e-6: the fp32 exponential's own floor (3e-7)


@pytest.mark.gpu
def test_softmax_eta_more_units_than_resident_workgroups(gpu_ops):
    """Round 4: K4's row-statistics kernel runs a resident grid striding over (sequence-head, tile pair) units.  With
    4,800 units every workgroup takes several: eta of a sequence must not depend on the rest of the batch (bit for bit
    against the two halves run alone), and first / middle / last sequences must match the oracle."""
    rng = np.random.default_rng(45)
    B, L, D, d_qk, H = 600, 512, 32, 32, 2
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    w = (rng.standard_normal((2 * d_qk + D, D)) / np.sqrt(D)).astype(np.float32)
    b = (rng.standard_normal(2 * d_qk + D) * 0.1).astype(np.float32)
    tx, tw, tb = dev(x), dev(w[:2 * d_qk]), dev(b[:2 * d_qk])
    full = gpu_ops.softmax_eta(tx, tw, tb, d_qk, H)
    lo = gpu_ops.softmax_eta(tx[:300], tw, tb, d_qk, H)
    hi = gpu_ops.softmax_eta(tx[300:], tw, tb, d_qk, H)
    assert torch.equal(full[:300], lo) and torch.equal(full[300:], hi)
    pick = [0, 1, 299, 300, 598, 599]
    want = O.eta_softmax(x[pick], w, b, d_qk, H, D)[..., 0]
    got = full[pick].cpu().numpy()
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-5 * scale
