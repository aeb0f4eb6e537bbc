"""The S4 fast path that uses the DPLR structure (analysis key ``s4_solver: dplr``; csrc/dplr_secular.hip):
eig(Ab) = (g + a) / (g - a), a in eig(diag(Lambda) - P Q^H) from the secular equation -- against the DENSE path
(DPLR build + Hessenberg / QR, the default and the checker), the oracle, and the reference's own golden
(tests/golden/ssm.npz: get_eigvals_ssm("s4") = discrete_DPLR + np.linalg.eigvals, analysis/eval_eig.py:282-301).

Criteria as in tests/test_gpu_eig_large.py: elementwise |dlambda| (1e-9; north_star asks 1e-5) on well-conditioned
inputs; on raw HiPPO columns (N >= 64: computed eigenvalues are noise at the 1e-2 level for ANY backward-stable
method) the backward error of every computed eigenvalue with respect to the dense matrix."""
import importlib

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def backward_error(a, w, stride=1):
    n = a.shape[0]
    nrm = np.linalg.norm(a, 2)
    return max(np.linalg.svd(a - z * np.eye(n), compute_uv=False)[-1] for z in w[::stride]) / nrm


def random_dplr(rng, n, h, pq=False):
    lam_re = -np.abs(rng.standard_normal((h, n))) - 0.1
    lam_im = rng.standard_normal((h, n)) * 5
    p = (rng.standard_normal((h, n)) + 1j * rng.standard_normal((h, n))) * 0.2
    q = (rng.standard_normal((h, n)) + 1j * rng.standard_normal((h, n))) * 0.2 if pq else None
    ls = rng.uniform(np.log(1e-3), np.log(1e-1), h)
    return lam_re, lam_im, p, q, ls


@pytest.mark.parametrize("n", [1, 2, 3, 8, 13, 64, 100, 128, 256, 512])
@pytest.mark.parametrize("pq", [False, True])
def test_secular_path_equals_dense_path_on_random_dplr(gpu_ops, n, pq):
    rng = np.random.default_rng(1000 + n + (7 if pq else 0))
    h = 5
    lam_re, lam_im, p, q, ls = random_dplr(rng, n, h, pq)
    args = (dev(lam_re), dev(lam_im), dev(p), dev(ls))
    w, info = gpu_ops.s4_dplr_eigs(*args, q=None if q is None else dev(q))
    assert int(info.abs().sum().item()) == 0 and w.shape == (h, n)
    ab = gpu_ops.s4_dplr_build(*args, q=None if q is None else dev(q))
    wd, infod = gpu_ops.eig_general_batched(ab)
    assert int(infod.abs().sum().item()) == 0
    w, wd = w.cpu().numpy(), wd.cpu().numpy()
    for c in range(h):
        assert O.max_abs_dlambda(w[c], wd[c]) < 1e-9, (n, c)
        qc = p[c] if q is None else q[c]
        lam = np.minimum(lam_re[c], -1e-4) + 1j * lam_im[c]
        ref = O.eigvals_dense(O.discrete_dplr(lam, p[c], qc, np.zeros(n), np.zeros(n), np.exp(ls[c]), 1)[0])
        assert O.max_abs_dlambda(w[c], ref) < 1e-9, (n, c)


def test_secular_path_matches_the_reference_golden(gpu_ops, golden):
    g = golden("ssm")
    for i in range(2):
        for clip_col in (1,):
            lam_re = g[f"s4_{i}_Lambda_re"][:, clip_col][None]
            lam_im = g[f"s4_{i}_Lambda_im"][:, clip_col][None]
            p = g[f"s4_{i}_P"][:, clip_col][None]
            ls = g[f"s4_{i}_log_step"][0, clip_col].reshape(1)
            w, info = gpu_ops.s4_dplr_eigs(dev(lam_re), dev(lam_im), dev(p), dev(ls))
            assert int(info.abs().sum().item()) == 0
            assert O.max_abs_dlambda(w.cpu().numpy()[0], g[f"s4_{i}_eig"][:, 0]) < 1e-10
    # the golden's Lambda_re has entries above the clip level (quirk Q11): the clip is applied inside, like the dense build
    assert np.max(g["s4_0_Lambda_re"]) > -1e-4


@pytest.mark.parametrize("n,h", [(64, 4), (256, 6)])
def test_secular_path_on_hippo_columns_by_backward_error(gpu_ops, n, h):
    """Raw HiPPO-LegS DPLR columns (config C5's input at n = 256): every value the secular path returns is an eigenvalue
    of a matrix within 1e-13 ||Ab|| of the dense Ab -- the bar the dense path is held to on the same input."""
    rng = np.random.default_rng(256)
    lam, p, _, _, _ = O.make_dplr_hippo(n)
    ls = rng.uniform(np.log(1e-3), np.log(1e-1), h)
    args = (dev(np.tile(lam.real, (h, 1))), dev(np.tile(lam.imag, (h, 1))), dev(np.tile(p, (h, 1))), dev(ls))
    w, info = gpu_ops.s4_dplr_eigs(*args)
    assert int(info.abs().sum().item()) == 0
    abh = gpu_ops.s4_dplr_build(*args).cpu().numpy()
    w = w.cpu().numpy()
    for c in range(0, h, 2):
        assert backward_error(abh[c], w[c], stride=1 if n <= 64 else 4) < 1e-13
    # one solve served all columns (same Lambda, P): the spectra differ only through the step's Cayley map
    g0, g1 = 2.0 / np.exp(ls[0]), 2.0 / np.exp(ls[1])
    a0 = g0 * (w[0] - 1.0) / (w[0] + 1.0)
    a1 = g1 * (w[1] - 1.0) / (w[1] + 1.0)
    np.testing.assert_allclose(a0, a1, rtol=1e-9, atol=1e-9)


def test_secular_path_full_c5_batch(gpu_ops):
    """C5 at full size: 6 layers x 128 feature columns of order 256 with per-column steps.  All flags clear, every
    spectrum finite, deduplicated and non-deduplicated runs agree bit for bit."""
    rng = np.random.default_rng(256)
    n, h = 256, 768
    lam, p, _, _, _ = O.make_dplr_hippo(n)
    ls = rng.uniform(np.log(1e-3), np.log(1e-1), h)
    args = (dev(np.tile(lam.real, (h, 1))), dev(np.tile(lam.imag, (h, 1))), dev(np.tile(p, (h, 1))), dev(ls))
    w, info = gpu_ops.s4_dplr_eigs(*args)
    w2, info2 = gpu_ops.s4_dplr_eigs(*args, dedupe=False)
    assert int(info.abs().sum().item()) == 0 and int(info2.abs().sum().item()) == 0
    assert torch.equal(torch.view_as_real(w), torch.view_as_real(w2))
    assert bool(torch.isfinite(torch.view_as_real(w)).all().item())


def test_secular_path_grouping_of_partly_repeated_columns(gpu_ops):
    """Columns are grouped by the BITS of (Lambda, P, Q): all equal (one solve), some equal (hashed groups, verified word
    for word), none equal -- always the same spectra as without grouping, bit for bit."""
    rng = np.random.default_rng(21)
    n, h = 48, 9
    lam_re, lam_im, p, _, ls = random_dplr(rng, n, h)
    for rep in ([0] * h, [0, 1, 0, 1, 2, 2, 0, 3, 3], list(range(h))):
        lr, li, pp = lam_re[rep], lam_im[rep], p[rep]
        a = gpu_ops.s4_dplr_eigs(dev(lr), dev(li), dev(pp), dev(ls))
        b = gpu_ops.s4_dplr_eigs(dev(lr), dev(li), dev(pp), dev(ls), dedupe=False)
        assert int(a[1].abs().sum().item()) == 0 and int(b[1].abs().sum().item()) == 0
        assert torch.equal(torch.view_as_real(a[0]), torch.view_as_real(b[0]))


def test_secular_path_degenerate_inputs(gpu_ops):
    """Zero weights (P_k = 0: Lambda_k is an eigenvalue), exactly repeated poles (weights combine, the repeats are
    eigenvalues), a fully diagonal matrix (P = 0), and Q orthogonal to P elementwise-in-sum."""
    rng = np.random.default_rng(3)
    n = 24
    lam_re = -np.abs(rng.standard_normal((4, n))) - 0.1
    lam_im = rng.standard_normal((4, n)) * 3
    p = (rng.standard_normal((4, n)) + 1j * rng.standard_normal((4, n))) * 0.5
    p[0, ::3] = 0.0                                  # zero weights
    lam_re[1, 5] = lam_re[1, 2]; lam_im[1, 5] = lam_im[1, 2]          # a repeated pole
    lam_re[1, 9] = lam_re[1, 2]; lam_im[1, 9] = lam_im[1, 2]          # ... three times
    p[2] = 0.0                                       # diagonal matrix
    lam_re[3, :] = lam_re[3, 0]; lam_im[3, :] = lam_im[3, 0]          # Lambda = c I: A = c I - P P^H, n - 1 eigenvalues c
    ls = np.log(rng.uniform(1e-3, 1e-1, 4))
    w, info = gpu_ops.s4_dplr_eigs(dev(lam_re), dev(lam_im), dev(p), dev(ls))
    assert int(info.abs().sum().item()) == 0
    w = w.cpu().numpy()
    for c in range(4):
        lam = lam_re[c] + 1j * lam_im[c]
        ref = O.eigvals_dense(O.s4_dplr_ab(lam, p[c], np.exp(ls[c])))
        assert O.max_abs_dlambda(w[c], ref) < 1e-8, c


def test_eval_eig_s4_with_the_dplr_solver(pkg, gpu_ops):
    """Operator level: get_eigvals_ssm / get_eigvals_s4_all with s4_solver="dplr" against the dense default."""
    ev = importlib.import_module(pkg.__name__ + ".analysis.eval_eig")
    rng = np.random.default_rng(12)
    N, H = 32, 5
    layers = [{"log_step": rng.uniform(np.log(1e-3), np.log(1e-1), (1, H)),
               "Lambda_re": -np.abs(rng.standard_normal((N, H))) * 0.5 + 0.05,
               "Lambda_im": rng.standard_normal((N, H)) * 3.0,
               "P": (rng.standard_normal((N, H)) + 1j * rng.standard_normal((N, H))) * 0.7} for _ in range(2)]
    for i in range(2):
        d = ev.get_eigvals_ssm("s4", layers, i, 1, 16)
        s = ev.get_eigvals_ssm("s4", layers, i, 1, 16, s4_solver="dplr")
        assert d.shape == s.shape == (N, 1) and O.max_abs_dlambda(s[:, 0], d[:, 0]) < 1e-10
    da = ev.get_eigvals_s4_all(layers)
    sa = ev.get_eigvals_s4_all(layers, s4_solver="dplr")
    assert da.shape == sa.shape == (N, H, 2)
    for i in range(2):
        for hcol in range(H):
            assert O.max_abs_dlambda(sa[:, hcol, i], da[:, hcol, i]) < 1e-10
    with pytest.raises(ValueError):
        ev.get_eigvals_ssm("s4", layers, 0, 1, 16, s4_solver="qr")


def test_secular_path_adversarial_poles_and_weights(gpu_ops):
    """Inputs at the edges of the iteration: poles one ulp apart, weights of 1e-160 (the root IS the pole to working
    precision), weights spread over ten decades, a single non-zero weight, Hermitian input (real poles, real weights:
    interlacing roots), poles on a circle with equal weights.  Criterion: every flag clear and every returned value an
    eigenvalue of a matrix within 1e-12 ||Ab|| of the dense Ab the reference would solve, and close to LAPACK's values where
    those are well conditioned."""
    rng = np.random.default_rng(99)
    n = 24
    cases = []
    lam = -1.0 + 1j * (0.5 + np.arange(n) * 2.220446049250313e-16 * 4)
    cases.append(("poles a few ulp apart", lam, rng.standard_normal(n) * 0.3 + 0j, None))
    lam = -np.abs(rng.standard_normal(n)) - 0.1 + 1j * rng.standard_normal(n) * 3
    cases.append(("weights of 1e-160", lam, np.full(n, 1e-80) + 0j, None))
    # (six decades: beyond that gI - A is so ill-conditioned that the dense Ab -- the reference's own route -- carries
    # errors of 1e-9 and is no yardstick any more; measured 4e-11 against it with weights up to 1e8)
    cases.append(("weights over six decades", lam, 10.0 ** rng.uniform(-5, 1, n) * np.exp(1j * rng.uniform(0, 6.28, n)), None))
    one = np.zeros(n, complex); one[7] = 2.5 - 1j
    cases.append(("a single non-zero weight", lam, one, None))
    cases.append(("hermitian", -np.sort(rng.uniform(0.1, 5, n)) + 0j, rng.standard_normal(n) + 0j, None))
    cases.append(("poles on a circle", 3.0 * np.exp(2j * np.pi * np.arange(n) / n) - 4.0, np.full(n, 0.7 + 0j), None))
    cases.append(("general q", lam, rng.standard_normal(n) + 1j * rng.standard_normal(n),
                  rng.standard_normal(n) + 1j * rng.standard_normal(n)))
    step = 0.05
    g = 2.0 / step
    for name, lam, p, q in cases:
        args = (dev(lam.real[None]), dev(lam.imag[None]), dev(p[None]), dev(np.log(np.array([step]))))
        w, info = gpu_ops.s4_dplr_eigs(*args, q=None if q is None else dev(q[None]), clip=False)
        assert int(info.abs().sum().item()) == 0, name
        mu = w.cpu().numpy()[0]
        # against the matrix the reference would hand to eigvals: Ab = (gI - A)^-1 (gI + A), built by the oracle
        A = np.diag(lam) - np.outer(p, np.conj(p if q is None else q))
        Ab = np.linalg.solve(g * np.eye(n) - A, g * np.eye(n) + A)
        nrm = np.linalg.norm(Ab, 2)
        be = max(np.linalg.svd(Ab - z * np.eye(n), compute_uv=False)[-1] for z in mu) / nrm
        assert be < 1e-12, (name, be)
        assert O.max_abs_dlambda(mu, np.linalg.eigvals(Ab)) < 1e-6 * max(1.0, np.linalg.cond(np.linalg.eig(Ab)[1]) * 1e-8), name
        # and the kernel is the prototype (oracle/dplr_secular_proto.py, checked against LAPACK on the CPU) to rounding
        from oracle import dplr_secular_proto as S
        mu_cpu, bad = S.s4_eigs(lam.real, lam.imag, p, step, q=q, clip=False)
        assert bad == 0 and O.max_abs_dlambda(mu, mu_cpu) < 1e-9, name
