"""GPU tests of the register-resident Hessenberg kernel (csrc/eig_real_reg.hip: n = 64, the first stage behind
np.linalg.eigvals at analysis/eval_eig.py:296 for BASELINE config C2), through the C ABI: the reduced matrix is upper
Hessenberg and orthogonally similar to its input, it agrees with the LDS-resident kernel behind ``EGS_HESS_LDS=1``, and
the whole solver matches LAPACK on the inputs that exercise xGEBAL's permutation and scaling inside the kernel."""
import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_reduced_matrix_is_hessenberg_and_similar(gpu_ops):
    rng = np.random.default_rng(640)
    a = rng.standard_normal((48, 64, 64))
    h = gpu_ops.hessenberg_batched(dev(a)).cpu().numpy()
    assert np.all(np.tril(h, -2) == 0.0)
    # orthogonal similarity: trace, Frobenius norm, trace of the square and the spectrum itself
    np.testing.assert_allclose(np.trace(h, axis1=1, axis2=2), np.trace(a, axis1=1, axis2=2), rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.linalg.norm(h, axis=(1, 2)), np.linalg.norm(a, axis=(1, 2)), rtol=1e-13)
    np.testing.assert_allclose(np.einsum("bij,bji->b", h, h), np.einsum("bij,bji->b", a, a), rtol=0, atol=1e-10)
    assert O.max_abs_dlambda(O.eigvals_dense(h), O.eigvals_dense(a)) < 1e-11


def test_matches_the_lds_resident_kernel(gpu_ops, monkeypatch):
    """Same algorithm (xGEHD2 with the same reflector convention), different summation order: the two reductions agree
    element by element up to rounding."""
    rng = np.random.default_rng(641)
    a = rng.standard_normal((16, 64, 64))
    h = gpu_ops.hessenberg_batched(dev(a)).cpu().numpy()
    monkeypatch.setenv("EGS_HESS_LDS", "1")
    h_lds = gpu_ops.hessenberg_batched(dev(a)).cpu().numpy()
    np.testing.assert_allclose(h, h_lds, rtol=0, atol=5e-12)


def test_strided_input(gpu_ops):
    """lda > n through the C ABI itself: a 64 x 64 window of a wider row-major array (the Python wrapper would copy)."""
    import importlib
    capi = importlib.import_module(gpu_ops.__name__.rsplit(".", 1)[0] + "._capi")
    rng = np.random.default_rng(642)
    big = rng.standard_normal((5, 64, 96))
    t = dev(big)
    nmat, n, lda = 5, 64, 96
    w = torch.empty((nmat, n), dtype=torch.complex128, device="cuda")
    info = torch.empty((nmat,), dtype=torch.int32, device="cuda")
    need = capi.lib().egs_eig_workspace_bytes(capi.EGS_F64, nmat, n)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    st = capi.lib().egs_eig_general_batched(t.data_ptr() + 16 * 8, capi.EGS_F64, nmat, n, lda, n * lda, w.data_ptr(),
                                            info.data_ptr(), ws.data_ptr(), need,
                                            torch.cuda.current_stream().cuda_stream)
    assert st == 0
    torch.cuda.synchronize()
    assert int(info.abs().sum().item()) == 0
    assert O.max_abs_dlambda(w.cpu().numpy(), np.linalg.eigvals(big[:, :, 16:80])) < 1e-11


def test_balancing_inside_the_kernel(gpu_ops):
    """Rows / columns that isolate an eigenvalue (permutation through the global-memory copy), triangular and block
    triangular matrices, graded scaling, zero / identity, entries near the ends of the exponent range."""
    rng = np.random.default_rng(643)
    cases = []
    b = rng.standard_normal((64, 64)); b[5, :] = 0.0; b[5, 5] = 2.0; cases.append(b)
    b = rng.standard_normal((64, 64)); b[:, 9] = 0.0; b[9, 9] = -1.0; b[40, :] = 0.0; b[40, 40] = 0.5; cases.append(b)
    cases.append(np.triu(rng.standard_normal((64, 64))))
    cases.append(np.tril(rng.standard_normal((64, 64))))
    d = 2.0 ** rng.integers(-20, 20, 64)
    cases.append(rng.standard_normal((64, 64)) * d[:, None] / d[None, :])
    cases.append(np.zeros((64, 64)))
    cases.append(np.eye(64))
    cases.append(rng.standard_normal((64, 64)) * 1e-200)
    cases.append(rng.standard_normal((64, 64)) * 1e200)
    b = rng.standard_normal((64, 64)); b[32:, :32] = 0.0; cases.append(b)
    p = rng.permutation(64)                                          # a permuted triangular matrix: everything isolates
    cases.append(np.triu(rng.standard_normal((64, 64)))[np.ix_(p, p)])
    c = np.stack(cases)
    w, info = gpu_ops.eig_general_batched(dev(c))
    assert int(info.abs().sum().item()) == 0
    w = w.cpu().numpy()
    ref = np.linalg.eigvals(c)
    for m in range(len(cases)):
        scale = max(1e-300, np.max(np.abs(ref[m])))
        assert O.max_abs_dlambda(w[m] / scale, ref[m] / scale) < 1e-12, m
    # triangular inputs: the eigenvalues are the diagonal, exactly (isolated one by one, never touched by arithmetic)
    for m in (2, 3, 10):
        assert np.array_equal(np.sort(w[m].real), np.sort(np.diag(c[m]))) and not np.any(w[m].imag), m


def test_nonfinite_input_is_flagged(gpu_ops):
    rng = np.random.default_rng(644)
    bad = rng.standard_normal((3, 64, 64))
    bad[1, 3, 4] = np.nan
    bad[2, 0, 0] = np.inf
    w, info = gpu_ops.eig_general_batched(dev(bad))
    assert info.cpu().tolist() == [0, 64, 64]
    w = w.cpu().numpy()
    assert np.all(np.isnan(w[1:].real)) and O.max_abs_dlambda(w[0], np.linalg.eigvals(bad[0])) < 1e-11


def test_large_batch_against_known_spectra(gpu_ops):
    """A bench-like batch with prescribed spectra (ragged last workgroup: the count is not a multiple of 4)."""
    from test_gpu_parity import known_spectrum_real
    rng = np.random.default_rng(645)
    a, lam = known_spectrum_real(rng, 4099, 64)
    w, info = gpu_ops.eig_general_batched(dev(a))
    assert int(info.abs().sum().item()) == 0
    w = w.cpu().numpy()
    np.testing.assert_allclose(np.sort(np.abs(w), axis=1), np.sort(np.abs(lam), axis=1), atol=1e-9)
    np.testing.assert_allclose(w.sum(axis=1).real, np.trace(a, axis1=1, axis2=2), rtol=1e-10, atol=1e-10)


def test_fused_kernel_equals_the_two_kernel_path(gpu_ops, monkeypatch):
    """eig_real64_kernel (both stages in one kernel, work queue over the whole batch) performs the same arithmetic as
    hess_reg_kernel + hqr_real_kernel: the spectra are bit-identical, also when the grid is smaller than the batch
    (queue in use) and when a small caller-provided workspace limits the number of resident waves."""
    rng = np.random.default_rng(646)
    a = dev(rng.standard_normal((3001, 64, 64)))
    w, info = gpu_ops.eig_general_batched(a)
    assert int(info.abs().sum().item()) == 0
    small = torch.empty(40 * 64 * 64 * 8 + 256, dtype=torch.uint8, device="cuda")      # 40 scratch slots -> 10 workgroups
    w_small, info_small = gpu_ops.eig_general_batched(a, workspace=small)
    monkeypatch.setenv("EGS_EIG_UNFUSED", "1")
    w2, info2 = gpu_ops.eig_general_batched(a)
    assert torch.equal(torch.view_as_real(w), torch.view_as_real(w2)) and torch.equal(info, info2)
    assert torch.equal(torch.view_as_real(w_small), torch.view_as_real(w2)) and torch.equal(info_small, info2)


def test_two_phase_solver_equals_the_one_kernel_iteration(gpu_ops, monkeypatch):
    """The n = 64 solver hands a matrix over to hqr_tail_kernel once its active block ends below row 44 (twice the
    occupancy for the second half of the iteration).  The continuation is the same arithmetic: spectra and info are
    bit-identical to the one-kernel form (EGS_EIG_ONE_PHASE) on Gaussian, prescribed-spectrum, triangular (every
    eigenvalue deflates at once: the hand-over happens in the middle of a run of deflations), zero, reducible and
    non-finite matrices; also with a workspace that only holds records for part of the batch (several passes) and at
    the float32 entry."""
    from test_gpu_parity import known_spectrum_real
    rng = np.random.default_rng(647)
    a = rng.standard_normal((2500, 64, 64))
    a[100:600] = known_spectrum_real(rng, 500, 64)[0]
    a[600:700] = np.triu(a[600:700])
    a[700:720] = 0.0
    a[720:760, 50:, :50] = 0.0                     # block triangular: the trailing 14 x 14 block is solved first
    a[760:800, 30:, :30] = 0.0                     # ... a 34 x 34 block: the leading 30 x 30 one goes to the tail kernel
    a[800] = np.nan
    a[801, 3, 5] = np.inf
    a[802:840] *= 1e-170
    a[840:880] *= 1e150
    ad = dev(a)
    w, info = gpu_ops.eig_general_batched(ad)
    w32, info32 = gpu_ops.eig_general_batched(ad[:500].to(torch.float32))
    # a workspace with hand-off records for 33000 of 40000 matrices: two passes (a pass is at least 32768 matrices)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(648)
    big = torch.randn((40000, 64, 64), generator=gen, device="cuda", dtype=torch.float64)
    slots = gpu_ops.device_props(0)["compute_units"] * 8 * 64 * 64 * 8
    part = torch.empty(slots + 33000 * (44 * 44 + 2) * 8 + 256, dtype=torch.uint8, device="cuda")
    w_part, info_part = gpu_ops.eig_general_batched(big, workspace=part)
    monkeypatch.setenv("EGS_EIG_ONE_PHASE", "1")
    w1, info1 = gpu_ops.eig_general_batched(ad)
    w1_32, info1_32 = gpu_ops.eig_general_batched(ad[:500].to(torch.float32))
    w1_big, info1_big = gpu_ops.eig_general_batched(big)
    monkeypatch.delenv("EGS_EIG_ONE_PHASE")
    same = lambda x, y: torch.equal(torch.view_as_real(x).view(torch.int64), torch.view_as_real(y).view(torch.int64))
    assert same(w, w1) and torch.equal(info, info1)
    assert same(w_part, w1_big) and torch.equal(info_part, info1_big) and int(info_part.abs().sum().item()) == 0
    assert same(w32, w1_32) and torch.equal(info32, info1_32)
    assert int(info[800].item()) == 64 and int(info[801].item()) == 64 and int(info.abs().sum().item()) == 128
    assert O.max_abs_dlambda(w.cpu().numpy()[:100], np.linalg.eigvals(a[:100])) < 1e-11
    assert O.max_abs_dlambda(w.cpu().numpy()[720:800], np.linalg.eigvals(a[720:800])) < 1e-11


@pytest.mark.parametrize("nmat", [1, 2, 3, 5, 9])
def test_tiny_batches(gpu_ops, nmat):
    """Fewer matrices than a workgroup has waves, and ragged last workgroups (waves without a matrix leave at once)."""
    rng = np.random.default_rng(650 + nmat)
    a = rng.standard_normal((nmat, 64, 64))
    w, info = gpu_ops.eig_general_batched(dev(a))
    assert int(info.abs().sum().item()) == 0
    assert O.max_abs_dlambda(w.cpu().numpy(), np.linalg.eigvals(a)) < 1e-11
    h = gpu_ops.hessenberg_batched(dev(a)).cpu().numpy()
    assert np.all(np.tril(h, -2) == 0.0) and O.max_abs_dlambda(O.eigvals_dense(h), O.eigvals_dense(a)) < 1e-11


def test_two_streams_do_not_share_scratch(gpu_ops):
    """Two batches solved concurrently on two torch streams (separate workspaces per stream in ops._workspace): each
    kernel has its own queue counter and scratch slots."""
    rng = np.random.default_rng(660)
    a1 = dev(rng.standard_normal((4096, 64, 64)))
    a2 = dev(rng.standard_normal((4096, 64, 64)))
    w1_ref, _ = gpu_ops.eig_general_batched(a1)
    w2_ref, _ = gpu_ops.eig_general_batched(a2)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(s1):
        w1, i1 = gpu_ops.eig_general_batched(a1)
    with torch.cuda.stream(s2):
        w2, i2 = gpu_ops.eig_general_batched(a2)
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(w1), torch.view_as_real(w1_ref)) and int(i1.abs().sum().item()) == 0
    assert torch.equal(torch.view_as_real(w2), torch.view_as_real(w2_ref)) and int(i2.abs().sum().item()) == 0


def test_queue_and_scratch_slots_under_load(gpu_ops, monkeypatch):
    """70,000 matrices through the work queue, every 53rd with rows / columns that isolate eigenvalues (so that many
    waves use their global-memory scratch slot for the permutation at the same time): every matrix is solved exactly
    once and the spectra are bit-identical to the two-kernel path, which keeps its scratch per matrix."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(670)
    nmat = 70000
    a = torch.randn((nmat, 64, 64), generator=gen, device="cuda", dtype=torch.float64)
    idx = torch.arange(0, nmat, 53, device="cuda")
    a[idx, 7, :] = 0.0
    a[idx, 7, 7] = 1.5
    a[idx, :, 40] = 0.0
    a[idx, 40, 40] = -0.5
    tri = torch.arange(11, nmat, 211, device="cuda")
    a[tri] = torch.triu(a[tri])
    w, info = gpu_ops.eig_general_batched(a)
    assert int(info.abs().sum().item()) == 0
    tr = a.diagonal(dim1=1, dim2=2).sum(dim=1)
    assert float((w.sum(dim=1).real - tr).abs().max().item()) < 1e-10
    assert float(w.sum(dim=1).imag.abs().max().item()) < 1e-10
    # the isolated eigenvalues come out exactly
    wi = w[idx]
    assert bool(((wi.real == 1.5) & (wi.imag == 0)).any(dim=1).all()) and bool(((wi.real == -0.5) & (wi.imag == 0)).any(dim=1).all())
    monkeypatch.setenv("EGS_EIG_UNFUSED", "1")
    w2, info2 = gpu_ops.eig_general_batched(a)
    assert torch.equal(torch.view_as_real(w), torch.view_as_real(w2)) and torch.equal(info, info2)


def test_two_bulge_sweep_is_backward_stable_on_hard_structures(gpu_ops):
    """The n = 64 solver (two bulges per sweep, rounding-level entries below the sub-diagonal instead of exact zeros,
    single-bulge fallback when the sweep starts below the top of the window) on structures that stress exactly those
    parts: nearly reducible Hessenberg matrices (tiny interior subdiagonals: starts at m > l), companion matrices,
    orthogonal matrices (every |lambda| = 1), Jordan-like defective blocks, and matrices with exact zero subdiagonals.
    Criterion independent of conditioning: every computed eigenvalue is an exact eigenvalue of a matrix within
    1e-13 n ||A|| of the input (sigma_min(A - lambda I)), and the spectrum sums to the trace."""
    rng = np.random.default_rng(6464)
    n = 64
    mats = []
    for rep in range(12):
        h = np.triu(rng.standard_normal((n, n)), -1)
        h[np.arange(1, n), np.arange(n - 1)] *= 10.0 ** rng.uniform(-9, 0, n - 1)      # graded / nearly reducible
        mats.append(h)
        c = np.zeros((n, n))
        c[0, :] = rng.standard_normal(n) * 0.5
        c[np.arange(1, n), np.arange(n - 1)] = 1.0                                     # companion matrix
        mats.append(c)
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        mats.append(q)                                                                  # orthogonal
        j = np.diag(np.repeat(rng.standard_normal(n // 4), 4)) + np.diag((np.arange(n - 1) % 4 != 3).astype(float), 1)
        p, _ = np.linalg.qr(rng.standard_normal((n, n)))
        mats.append(p @ j @ p.T)                                                        # 4 x 4 Jordan blocks, rotated
        z = np.triu(rng.standard_normal((n, n)), -1)
        z[np.arange(8, n, 8), np.arange(7, n - 1, 8)] = 0.0                             # exact splits every 8 rows
        mats.append(z)
    a = np.stack(mats)
    w, info = gpu_ops.eig_general_batched(dev(a))
    assert int(info.abs().sum().item()) == 0
    w = w.cpu().numpy()
    nrm = np.linalg.norm(a, axis=(1, 2))
    assert np.max(np.abs(w.sum(axis=1) - np.trace(a, axis1=1, axis2=2)) / nrm) < 1e-12
    worst = 0.0
    for k in range(a.shape[0]):
        lam = w[k, :: 7]                                       # every 7th eigenvalue: 10 SVDs per matrix
        m = a[k][None] - lam[:, None, None] * np.eye(n)[None]
        smin = np.linalg.svd(m, compute_uv=False)[:, -1]
        worst = max(worst, float(np.max(smin) / nrm[k]))
    assert worst < 1e-13 * n, worst


def test_skew_symmetric_and_zero_diagonal_matrices(gpu_ops):
    """Regression (round 3): a skew-symmetric matrix reduces to a tridiagonal form with an exactly / nearly zero diagonal,
    for which LAPACK's deflation test waits until the sub-diagonal itself underflows; the bulge column then passes through
    magnitudes whose squares underflow.  A reflector built from such a column produced eigenvalues of 1e+125 for 1.4 %
    of random skew-symmetric 64 x 64 matrices (no failure flag).  2048 of them, plus zero-diagonal non-normal ones and
    globally tiny / huge scalings of both."""
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    n, nmat = 64, 2048
    s = torch.randn((nmat, n, n), generator=g, device="cuda", dtype=torch.float64) / 16
    skew = s - s.transpose(1, 2)
    zd = s.clone()
    zd.diagonal(dim1=1, dim2=2).zero_()
    for name, a in (("skew", skew), ("zero-diagonal", zd), ("skew * 1e-150", skew * 1e-150), ("skew * 1e150", skew * 1e150)):
        w, info = gpu_ops.eig_general_batched(a)
        assert int(info.abs().sum().item()) == 0, name
        nrm = torch.linalg.matrix_norm(a)
        assert bool(torch.isfinite(torch.view_as_real(w)).all()), name
        assert float((w.abs().max(dim=1).values / nrm).max().item()) < 1.0 + 1e-9, name        # |lambda| <= ||A||_F
        tr2 = (a * a.transpose(1, 2)).sum(dim=(1, 2))                                          # tr(A^2) = sum lambda^2
        assert float((((w * w).sum(dim=1) - tr2).abs() / nrm ** 2).max().item()) < 1e-12, name
        if name.startswith("skew"):
            assert float((w.real.abs().max(dim=1).values / nrm).max().item()) < 1e-13, name   # purely imaginary spectrum
    ref = np.linalg.eigvals(skew[:64].cpu().numpy())
    w, _ = gpu_ops.eig_general_batched(skew[:64].contiguous())
    assert O.max_abs_dlambda(w.cpu().numpy(), ref) < 1e-12


def test_constant_rank_one_and_nilpotent_matrices(gpu_ops):
    """Regression (round 3): c * ones(64, 64) reduces to a Hessenberg form whose entries cascade down to a trailing block
    of 1e-190 -- bulge columns whose squares underflow.  A reflector that ignored such columns never converged (166 of
    256 matrices flagged); they are rescaled now, and sub-diagonal entries below 1e-200 of the normalised matrix deflate."""
    n = 64
    c = torch.linspace(0.5, 2.0, 256, dtype=torch.float64, device="cuda")
    ones = torch.ones((256, n, n), dtype=torch.float64, device="cuda") * c[:, None, None]
    w, info = gpu_ops.eig_general_batched(ones)
    assert int(info.abs().sum().item()) == 0
    w = w.cpu().numpy()
    top = np.abs(w).argmax(axis=1)
    np.testing.assert_allclose(w[np.arange(256), top].real, 64 * c.cpu().numpy(), rtol=1e-13)
    rest = np.sort(np.abs(w), axis=1)[:, :-1]
    assert rest.max() < 1e-12
    g = torch.Generator(device="cuda")
    g.manual_seed(3)
    u = torch.randn((128, n, 1), generator=g, device="cuda", dtype=torch.float64)
    v = torch.randn((128, 1, n), generator=g, device="cuda", dtype=torch.float64)
    w, info = gpu_ops.eig_general_batched((u * v).contiguous())             # rank one: spectrum {v.u, 0 x 63}
    assert int(info.abs().sum().item()) == 0
    w = w.cpu().numpy()
    dot = (v.squeeze(1) * u.squeeze(2)).sum(dim=1).cpu().numpy()
    top = np.abs(w).argmax(axis=1)
    scale = (u.norm(dim=(1, 2)) * v.norm(dim=(1, 2))).cpu().numpy()
    assert np.max(np.abs(w[np.arange(128), top] - dot) / scale) < 1e-13
    assert np.max(np.sort(np.abs(w), axis=1)[:, :-1] / scale[:, None]) < 1e-7      # 63-fold defective zero eigenvalue
    nil = torch.triu(torch.randn((64, n, n), generator=g, device="cuda", dtype=torch.float64), 1)
    w, info = gpu_ops.eig_general_batched(nil)
    assert int(info.abs().sum().item()) == 0 and float(w.abs().max().item()) == 0.0   # exact zeros stay exact


@pytest.mark.parametrize("scale", [1e-220, 1e-150, 1.0, 1e150, 1e220])
def test_two_stage_api_on_unscaled_hessenberg_input(gpu_ops, scale):
    """egs_hessenberg_batched -> egs_hseqr_batched hands the n = 64 QR kernel a matrix nobody scaled: its absolute
    deflation / tiny-column thresholds assume max |entry| in [1, 2), so the kernel normalises such input itself (entries
    of 1e-220 used to deflate wholesale: the diagonal came back as the spectrum with info = 0)."""
    rng = np.random.default_rng(670)
    a = rng.standard_normal((24, 64, 64)) / 8
    h = gpu_ops.hessenberg_batched(dev(a * scale))
    w, info = gpu_ops.hseqr_batched(h)
    assert int(info.abs().sum().item()) == 0
    ref = np.linalg.eigvals(a)
    assert O.max_abs_dlambda(w.cpu().numpy() / scale, ref) < 1e-11
    # a hand-made Hessenberg matrix (no reduction in front), also through the raw second stage
    hh = np.triu(rng.standard_normal((8, 64, 64)), -1) * scale
    w2, info2 = gpu_ops.hseqr_batched(dev(hh))
    assert int(info2.abs().sum().item()) == 0
    assert O.max_abs_dlambda(w2.cpu().numpy() / scale, np.linalg.eigvals(hh / scale)) < 1e-10


def test_workspace_falls_back_to_passes_under_memory_pressure(gpu_ops, monkeypatch):
    """ops._workspace halves its request down to the documented floor when the allocation fails (the n = 64 solver asks
    for 8.9 KB per matrix of a pass): same spectra through the smaller workspace."""
    rng = np.random.default_rng(680)
    a = dev(rng.standard_normal((40000, 64, 64)) / 8)
    w_ref, info_ref = gpu_ops.eig_general_batched(a)
    gpu_ops.release_workspaces()
    real_empty = torch.empty
    need = gpu_ops.lib().egs_eig_workspace_bytes(1, 40000, 64)
    floor = gpu_ops.lib().egs_eig_workspace_bytes(1, 4096, 64)      # what ops.eig_general_batched goes down to
    assert floor < need // 4
    calls = []

    def stingy(size, *args, **kw):
        if kw.get("dtype") == torch.uint8 and isinstance(size, int) and size > need // 4:
            calls.append(size)
            raise torch.cuda.OutOfMemoryError("simulated")
        return real_empty(size, *args, **kw)

    monkeypatch.setattr(torch, "empty", stingy)
    with pytest.warns(RuntimeWarning, match="several passes"):               # the throughput loss is made visible, once
        w, info = gpu_ops.eig_general_batched(a)
    monkeypatch.undo()
    assert len(calls) >= 2 and int(info.abs().sum().item()) == 0
    assert torch.equal(torch.view_as_real(w), torch.view_as_real(w_ref))      # passes do not change a bit
    # the cap was recorded only after the smaller allocation succeeded, for THIS request size ...
    caps = list(gpu_ops._workspace_caps.values())
    assert len(caps) == 1 and caps[0]["fitted"] <= need // 4 < caps[0]["asked"] == need
    # ... and it is not for ever: with memory back, the full request is tried again within _CAP_RETRY_CALLS calls
    for _ in range(gpu_ops._CAP_RETRY_CALLS):
        w2, _ = gpu_ops.eig_general_batched(a)
    assert not gpu_ops._workspace_caps
    assert max(b.numel() for b in gpu_ops._workspaces.values()) >= need
    assert torch.equal(torch.view_as_real(w2), torch.view_as_real(w_ref))
    gpu_ops.release_workspaces()


def test_merged_two_pass_sweep_is_bit_identical_to_the_three_pass_sweep(gpu_ops, monkeypatch):
    """Round 4: for windows of 33 .. 59 rows an iteration of the n = 64 sweep runs two sub-passes (moving column /
    bottom-row primaries) instead of three.  Every element receives the same operations in the same order, so spectra and
    info agree with the three-pass form (EGS_QR2_NO_MERGE=1) to the last bit -- on Gaussian and benchmark-style matrices,
    Hessenberg input, reducible matrices (sweeps that start below the top of the window: one bulge) and skew-symmetric
    ones (the tiny-column path), in the two-phase, one-kernel and two-kernel forms of the solver."""
    g = torch.Generator(device="cuda")
    g.manual_seed(2026)
    rnd = lambda n: torch.randn((n, 64, 64), generator=g, device="cuda", dtype=torch.float64)
    blk = rnd(1500) / 8
    blk[:, 40:, :40] = 0.0
    sk = rnd(1500)
    a = torch.cat([rnd(3000) / 8, torch.triu(rnd(1500), -1), blk, sk - sk.transpose(1, 2),
                   rnd(1500) * torch.logspace(-3, 3, 64, device="cuda", dtype=torch.float64)[None, :, None]])
    for form in ({}, {"EGS_EIG_ONE_PHASE": "1"}, {"EGS_EIG_UNFUSED": "1"}):
        for k, v in form.items():
            monkeypatch.setenv(k, v)
        w, info = gpu_ops.eig_general_batched(a)
        monkeypatch.setenv("EGS_QR2_NO_MERGE", "1")
        w3, info3 = gpu_ops.eig_general_batched(a)
        monkeypatch.delenv("EGS_QR2_NO_MERGE")
        for k in form:
            monkeypatch.delenv(k)
        assert torch.equal(info, info3) and int(info.abs().sum().item()) == 0, form
        assert torch.equal(torch.view_as_real(w), torch.view_as_real(w3)), form
    assert O.max_abs_dlambda(w[:64].cpu().numpy(), np.linalg.eigvals(a[:64].cpu().numpy())) < 1e-11
