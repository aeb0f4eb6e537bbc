"""BASELINE.json's configs C2 and C3 at their FULL sizes (review r4 #4: the suite only ran them at reduced batch; the
full sizes were exercised inside bench.py alone).  The oracle cannot finish these sizes in seconds, so the checks are
the size-independent properties of the domain -- every matrix converged, sum(lambda) = trace, the modulus order
statistics of every spectrum against the KNOWN spectrum the matrices were built from -- plus the oracle itself on the
first and the last units of the batch.  All through the C ABI (ops -> ctypes -> libeigspec_hip.so)."""
import os
import sys

import numpy as np
import pytest
import torch

import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c2_full_batch_524288_matrices(gpu_ops):
    """C2: S5 d_state = 64, seq_len = 1024, batch = 512 -> 524288 dense 64 x 64 fp64 matrices (17.2 GB), bench.py's own
    generator (known S5-style spectrum under a non-normal orthogonal similarity), ONE call."""
    sys.path.insert(0, ROOT)
    import bench
    n, nmat, chunk = 64, 512 * 1024, 16384
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(64)
    a = torch.empty((nmat, n, n), dtype=torch.float64, device=dev)
    lam = torch.empty((nmat, n), dtype=torch.complex128, device=dev)
    for m0 in range(0, nmat, chunk):
        a[m0:m0 + chunk], lam[m0:m0 + chunk] = bench.gen_known_spectrum(chunk, n, gen, dev)
    w, info = gpu_ops.eig_general_batched(a)
    torch.cuda.synchronize()
    assert w.shape == (nmat, n) and info.shape == (nmat,)
    assert int((info != 0).sum().item()) == 0                                   # every matrix converged
    assert bool(torch.isfinite(torch.view_as_real(w)).all().item())
    # sum(lambda) = trace(A): real part to rounding, imaginary parts cancel in conjugate pairs
    tr = torch.diagonal(a, dim1=1, dim2=2).sum(dim=1)
    s = w.sum(dim=1)
    assert float((s.real - tr).abs().max().item()) < 1e-10
    assert float(s.imag.abs().max().item()) < 1e-10
    # |lambda| and Re(lambda) order statistics of EVERY spectrum against the spectrum the matrix was built from
    # (permutation-invariant; the real parts pin the phases up to conjugation, which a real matrix cannot tell apart).
    # 33.5 M eigenvalues: all of them inside north_star's 1e-5, all but a handful of matrices inside 1e-9 -- the
    # stragglers are matrices with two nearly coincident eigenvalue pairs (condition ~ 1 / gap), and for the worst of
    # them the solver is held to LAPACK's own error on the same matrix
    err_mod = (torch.sort(w.abs(), dim=1).values - torch.sort(lam.abs(), dim=1).values).abs().amax(dim=1)
    err_re = (torch.sort(w.real, dim=1).values - torch.sort(lam.real, dim=1).values).abs().amax(dim=1)
    err = torch.maximum(err_mod, err_re)
    assert float(err.max().item()) < 1e-5
    assert int((err > 1e-9).sum().item()) <= nmat // 1000
    worst = torch.topk(err, 8).indices
    aw = a[worst].cpu().numpy()
    lw = lam[worst].cpu().numpy()
    d_gpu = O.max_abs_dlambda(w[worst].cpu().numpy(), lw)
    d_lapack = O.max_abs_dlambda(O.eigvals_dense(aw), lw)
    assert d_gpu <= 10.0 * d_lapack + 1e-12, (d_gpu, d_lapack)
    # first and last matrices of the batch against the oracle's dense path (LAPACK), north_star's tolerance and 1e4 tighter
    for sl in (slice(0, 64), slice(nmat - 64, nmat)):
        d = O.max_abs_dlambda(w[sl].cpu().numpy(), O.eigvals_dense(a[sl].cpu().numpy()))
        assert d < 1e-9, d
    del a, lam, w
    gpu_ops.release_workspaces()
    torch.cuda.empty_cache()


def test_c3_full_batch_mamba2_lambda(gpu_ops):
    """C3: Mamba-2 lambda_t, batch = 1024, seq_len = 2048, D = 128, 4 heads (d_state = 16 only moves the unused B / C
    column offset): 2.1 M tokens, 1.07 GB in.  Whole tensor finite and in (0, 1] (exp of a non-positive number); first
    and last sequences against the oracle."""
    B, L, D, nh = 1024, 2048, 128, 4
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(16)
    x = torch.randn((B, L, D), generator=g, device=dev)
    w = (torch.rand((nh, D), generator=g, device=dev) * 2 - 1) / np.sqrt(D)
    # dt_bias = inverse softplus of logU[1e-3, 1e-1], A_log = log U[1, 16]  (models/mamba.py:71-78, :85-86)
    dt = torch.exp(torch.rand(nh, generator=g, device=dev) * (np.log(1e-1) - np.log(1e-3)) + np.log(1e-3))
    dtb = dt + torch.log(-torch.expm1(-dt))
    alog = torch.log(torch.rand(nh, generator=g, device=dev) * 15 + 1)
    lam = gpu_ops.mamba2_lambda(x, w, None, dtb, alog)
    torch.cuda.synchronize()
    assert lam.shape == (B, L, nh) and lam.dtype == torch.float32
    assert bool(torch.isfinite(lam).all().item())
    assert float(lam.min().item()) > 0.0 and float(lam.max().item()) <= 1.0
    xs = torch.cat([x[:2], x[-2:]]).cpu().numpy()
    want = O.lambda_mamba2(xs, w.cpu().numpy(), None, dtb.cpu().numpy(), alog.cpu().numpy(), 0, 0, 0, nh)[..., 0]
    got = torch.cat([lam[:2], lam[-2:]]).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-7)
    # the device histogram of the full tensor (the next step of the path) accounts for every value exactly once per edge set
    pct = gpu_ops.threshold_hist(lam.unsqueeze(-1), O.THRESHOLDS_RADIUS, "radius").cpu().numpy()
    assert pct.shape[0] == 7 and np.all(pct >= 0) and np.all(pct.sum(axis=0) >= 100.0 - 1e-9)   # edges double-count (Q12)
