"""NumPy prototype of the S4 fast path of the HIP library (csrc/dplr_secular.hip): eigenvalues of the bilinear-discretised
DPLR state matrix from its structure.  TEST INFRASTRUCTURE (never imported by the product).

discrete_DPLR's Ab (analysis/eval_eig.py:254-273) is the Cayley transform Ab = (gI - A)^-1 (gI + A) of
A = diag(Lambda) - P Q^H, g = 2 / step, so eig(Ab) = (g + a) / (g - a) with a in eig(A); and A, diagonal plus rank one,
has its eigenvalues at the roots of the secular equation 1 - sum_k conj(Q_k) P_k / (Lambda_k - a) = 0, found by the
Ehrlich-Aberth iteration on p(a) = f(a) prod_k (Lambda_k - a) exactly as the kernel runs it: Jacobi style (all roots from
the previous iterate), start Lambda_k - W_k, poles of weight zero and exact repeats of a pole split off as trivial
eigenvalues, a root frozen once |f| <= tol (1 + sum |W_k / (Lambda_k - a)|) or its correction is below tol |a|.
The reference itself calls np.linalg.eigvals on the dense Ab (analysis/eval_eig.py:296); this restatement exists so that the
kernel's control flow meets LAPACK on the CPU (tests/test_oracle_golden.py)."""
from __future__ import annotations

import numpy as np

TOL = 8.0 * np.finfo(np.float64).eps
MAX_ITER = 400


def rank1_eigvals(lam, p, q=None, clip=False):
    """eig(diag(lam) - p q^H).  Returns (a complex128 [n], iterations, unconverged roots)."""
    lam = np.asarray(lam, np.complex128).copy()
    if clip:
        lam = np.minimum(lam.real, -1e-4) + 1j * lam.imag                   # the caller's clip, analysis/eval_eig.py:288
    p = np.asarray(p, np.complex128)
    q = p if q is None else np.asarray(q, np.complex128)
    n = lam.size
    w_raw = np.conj(q) * p
    # exact repeats of a pole: the first occurrence carries the combined weight, the others are eigenvalues as they stand
    same = lam[:, None] == lam[None, :]
    leader = np.argmax(same, axis=1)                                       # first index with the same pole
    w = np.where(leader == np.arange(n), same.astype(np.complex128) @ w_raw, 0.0)
    active = (leader == np.arange(n)) & (w != 0.0)
    w = np.where(active, w, 0.0)
    a = np.where(active, lam - w, lam)
    done = ~active
    for it in range(MAX_ITER):
        if done.all():
            return a, it, 0
        d = lam[None, :] - a[:, None]                                      # [root, pole]
        on_pole = ((np.abs(d) ** 2 < 1e-290) & active[None, :]).any(axis=1)
        r = np.where(active[None, :], 1.0 / np.where(d == 0, 1.0, d), 0.0)
        t = w[None, :] * r
        s0, s1, s2 = r.sum(1), t.sum(1), (t * r).sum(1)
        mag = 1.0 + (np.abs(t.real) + np.abs(t.imag)).sum(1)
        e = a[:, None] - a[None, :]
        other = active[None, :] & (np.abs(e) > 0) & ~np.eye(n, dtype=bool)
        rr = np.where(other, 1.0 / np.where(e == 0, 1.0, e), 0.0).sum(1)
        f = 1.0 - s1
        with np.errstate(all="ignore"):
            corr = f / (-s2 - f * (s0 + rr))
        fin = np.isfinite(corr)
        conv = on_pole | (np.abs(f.real) + np.abs(f.imag) <= TOL * mag) | \
            (fin & (np.abs(corr.real) + np.abs(corr.imag) <= TOL * (np.abs(a.real) + np.abs(a.imag))))
        step = ~done & ~conv
        a = np.where(step & fin, a - corr, a)
        done = done | conv
    return a, MAX_ITER, int((~done).sum())


def s4_eigs(lam_re, lam_im, p, step, q=None, clip=True):
    """eig(Ab) of one feature column: the Cayley map of the secular roots."""
    a, _, bad = rank1_eigvals(np.asarray(lam_re) + 1j * np.asarray(lam_im), p, q, clip)
    g = 2.0 / step
    return (g + a) / (g - a), bad
