"""NumPy prototype of the large-n complex eigenvalue solver of csrc/eig_zlarge.hip (TEST INFRASTRUCTURE).

The reference's only dense eigendecomposition is ``np.linalg.eigvals(Ab)`` on the S4 DPLR matrix
(analysis/eval_eig.py:296) at state_dim 128 / 512 (configs/iclr2026/{mqar/mqar-s4,wikitext/wikitext-103-s4}.yaml)
and BASELINE config C5 asks for n = 256.  This file restates, step for step, the algorithm the HIP kernels run
for n > 64 so that their control flow (window placement, chain timing, deferred updates) can be validated on the
CPU against LAPACK before it meets a GPU:

* ``hess_blocked``  -- blocked Householder reduction (the published xGEHRD / xLAHR2 algorithm: panel of nb
  reflectors with Y = A V T built column by column from one matrix-vector product each, then level-3 updates).
* ``ms_qr``         -- small-bulge multishift QR (Braman, Byers, Mathias 2002, without aggressive early deflation):
  chains of M single-shift bulges two rows apart, chased through W x W diagonal windows in which the reflectors are
  accumulated into U; rows above / columns right of the window are updated by U afterwards (matrix products).

Nothing here is on the product path."""
from __future__ import annotations

import numpy as np

ULP = np.finfo(np.float64).eps
SAFMIN = np.finfo(np.float64).tiny


def abs1(z):
    return abs(z.real) + abs(z.imag)


def zlarfg2(a0, a1):
    """2-element reflector H = I - tau v v^H, v = [1, v2], with H^H [a0, a1] = [beta, 0] (xLARFG, n = 2)."""
    xnorm = abs(a1)
    if xnorm == 0.0 and a0.imag == 0.0:
        return a0, 0.0 + 0.0j, 0.0 + 0.0j
    beta = -np.copysign(np.sqrt(a0.real ** 2 + a0.imag ** 2 + xnorm ** 2), a0.real)
    tau = complex((beta - a0.real) / beta, -a0.imag / beta)
    v2 = a1 / (a0 - beta)
    return complex(beta, 0.0), v2, tau


# --------------------------------------------------------------------------------------------------------------
# blocked Hessenberg reduction
# --------------------------------------------------------------------------------------------------------------
def hess_blocked(a, nb=16):
    """Blocked reduction to upper-Hessenberg form; returns H (exact zeros below the subdiagonal)."""
    A = np.array(a, dtype=np.complex128)
    n = A.shape[0]
    k = 0
    while k < n - 2:
        jb = min(nb, n - 2 - k)
        V = np.zeros((n, jb), dtype=np.complex128)
        Y = np.zeros((n, jb), dtype=np.complex128)        # rows k+1.. filled in the panel, rows 0..k afterwards
        T = np.zeros((jb, jb), dtype=np.complex128)
        taus = np.zeros(jb, dtype=np.complex128)
        for j in range(jb):
            c = k + j
            b = A[k + 1:, c].copy()                         # rows k+1..n-1 of column c
            if j > 0:
                b -= Y[k + 1:, :j] @ np.conj(V[c, :j])      # right update by the earlier reflectors
                w = np.conj(V[k + 1:, :j]).T @ b            # left update: b -= V T^H V^H b
                w = np.conj(T[:j, :j]).T @ w
                b -= V[k + 1:, :j] @ w
            # reflector annihilating rows c+2.. of column c
            x = b[j:]                                       # row c+1 is x[0]
            alpha = x[0]
            xnorm2 = float(np.sum(x[1:].real ** 2 + x[1:].imag ** 2))
            v = np.zeros(n, dtype=np.complex128)
            if xnorm2 == 0.0 and alpha.imag == 0.0:
                tau = 0.0 + 0.0j
                beta = alpha
                v[c + 1] = 1.0
            else:
                beta = -np.copysign(np.sqrt(alpha.real ** 2 + alpha.imag ** 2 + xnorm2), alpha.real)
                tau = complex((beta - alpha.real) / beta, -alpha.imag / beta)
                v[c + 1] = 1.0
                v[c + 2:] = x[1:] / (alpha - beta)
            A[k + 1:c + 1, c] = b[:j]
            A[c + 1, c] = beta
            A[c + 2:, c] = 0.0
            V[:, j] = v
            taus[j] = tau
            # y = A v (rows k+1..), then the corrections that make Y = A V T
            y = A[k + 1:, c + 1:] @ v[c + 1:]
            z = np.conj(V[:, :j]).T @ v
            y -= Y[k + 1:, :j] @ z
            Y[k + 1:, j] = tau * y
            T[:j, j] = -tau * (T[:j, :j] @ z)
            T[j, j] = tau
        # top rows of Y:  Y(0:k+1, :) = A(0:k+1, k+1:) V T
        Y[:k + 1, :] = (A[:k + 1, k + 1:] @ V[k + 1:, :]) @ T
        # panel columns, rows 0..k: right update only (these rows see no left reflector of this panel)
        for j in range(1, jb):
            c = k + j
            A[:k + 1, c] -= Y[:k + 1, :j] @ np.conj(V[c, :j])
        # trailing columns: right update, then left update
        c0 = k + jb
        if c0 < n:
            A[:, c0:] -= Y @ np.conj(V[c0:, :]).T
            Wm = np.conj(V[k + 1:, :]).T @ A[k + 1:, c0:]
            A[k + 1:, c0:] -= V[k + 1:, :] @ (np.conj(T).T @ Wm)
        k += jb
    return A


# --------------------------------------------------------------------------------------------------------------
# plain single-shift QR on a small Hessenberg block (the shift generator; xLAHQR control flow)
# --------------------------------------------------------------------------------------------------------------
def small_zqr(hin, stats=None):
    H = np.array(hin, dtype=np.complex128)
    n = H.shape[0]
    w = np.zeros(n, dtype=np.complex128)
    smlnum = SAFMIN * (n / ULP)
    i = n - 1
    itmax = 30 * max(10, n)
    kdefl = 0
    while i >= 0:
        l = 0
        done = False
        for _ in range(itmax + 1):
            l = _find_small(H, 0, i, smlnum)
            if l > 0:
                H[l, l - 1] = 0.0
            if l >= i:
                done = True
                break
            kdefl += 1
            t = _wilkinson(H, l, i, kdefl)
            # start at l (no search for two consecutive small subdiagonals in the prototype)
            a0, a1 = H[l, l] - t, H[l + 1, l]
            for k in range(l, i):
                if k > l:
                    a0, a1 = H[k, k - 1], H[k + 1, k - 1]
                beta, v2, tau = zlarfg2(a0, a1)
                if k > l:
                    H[k, k - 1] = beta
                    H[k + 1, k - 1] = 0.0
                s = np.conj(tau) * (H[k, k:i + 1] + np.conj(v2) * H[k + 1, k:i + 1])
                H[k, k:i + 1] -= s
                H[k + 1, k:i + 1] -= s * v2
                r1 = min(k + 2, i) + 1
                s = tau * (H[l:r1, k] + H[l:r1, k + 1] * v2)
                H[l:r1, k] -= s
                H[l:r1, k + 1] -= s * np.conj(v2)
                if stats is not None:
                    stats["small_steps"] = stats.get("small_steps", 0) + 1
        if not done:
            raise RuntimeError("small_zqr failed to converge")
        w[i] = H[i, i]
        kdefl = 0
        i = l - 1
    return w


def _find_small(H, l, i, smlnum):
    """Largest k in (l, i] whose subdiagonal H[k, k-1] passes the Ahues-Tisseur test; l if none."""
    n = H.shape[0]
    for k in range(i, l, -1):
        hkk1 = abs1(H[k, k - 1])
        if hkk1 <= smlnum:
            return k
        tst = abs1(H[k - 1, k - 1]) + abs1(H[k, k])
        if tst == 0.0:
            if k - 2 >= 0:
                tst += abs1(H[k - 1, k - 2])
            if k + 1 <= n - 1:
                tst += abs1(H[k + 1, k])
        if hkk1 <= ULP * tst:
            hk1k = abs1(H[k - 1, k])
            ab, ba = max(hkk1, hk1k), min(hkk1, hk1k)
            df = abs1(H[k - 1, k - 1] - H[k, k])
            aa, bb = max(abs1(H[k, k]), df), min(abs1(H[k, k]), df)
            s = aa + ab
            if ba * (ab / s) <= max(smlnum, ULP * (bb * (aa / s))):
                return k
    return l


def _wilkinson(H, l, i, kdefl):
    if kdefl % 20 == 0:
        return 0.75 * abs1(H[i, i - 1]) + H[i, i]
    if kdefl % 10 == 0:
        return 0.75 * abs1(H[l + 1, l]) + H[l, l]
    t = H[i, i]
    u = np.sqrt(H[i - 1, i]) * np.sqrt(H[i, i - 1])
    s = abs1(u)
    if s != 0.0:
        x = 0.5 * (H[i - 1, i - 1] - t)
        sx = abs1(x)
        s = max(s, sx)
        y = s * np.sqrt((x / s) ** 2 + (u / s) ** 2)
        if sx > 0.0:
            xn = x / sx
            if xn.real * y.real + xn.imag * y.imag < 0.0:
                y = -y
        t = t - u * (u / (x + y))
    return t


# --------------------------------------------------------------------------------------------------------------
# multishift QR with chains of single-shift bulges
# --------------------------------------------------------------------------------------------------------------
W_WIN = 64          # window order (LDS: two W x W complex arrays)
M_BULGES = 16       # bulges per chain = shifts per sweep
NB_STEPS = W_WIN - 2 * M_BULGES - 1   # time steps per window


def _chase_window(Wm, U, lo, hi, s, l, i, shifts, t0, t1):
    """All bulges advance through time steps t0..t1 inside the window Wm = H[s:s+w, s:s+w] (indices below are
    global).  lo / hi: first / last global index covered by the window.  U (or None) accumulates the reflectors."""
    m = len(shifts)
    for t in range(t0, t1 + 1):
        refl = []
        for b in range(m):
            p = l + t - 2 * b
            if t - 2 * b < 0 or p > i - 1:
                continue
            if p == l:
                a0, a1 = Wm[l - s, l - s] - shifts[b], Wm[l + 1 - s, l - s]
            else:
                a0, a1 = Wm[p - s, p - 1 - s], Wm[p + 1 - s, p - 1 - s]
            refl.append((p, *zlarfg2(a0, a1)))
        # row phase
        for p, beta, v2, tau in refl:
            if p > l:
                Wm[p - s, p - 1 - s] = beta
                Wm[p + 1 - s, p - 1 - s] = 0.0
            c0, c1 = p - s, hi - s + 1
            sm = np.conj(tau) * (Wm[p - s, c0:c1] + np.conj(v2) * Wm[p + 1 - s, c0:c1])
            Wm[p - s, c0:c1] -= sm
            Wm[p + 1 - s, c0:c1] -= sm * v2
        # column phase
        for p, beta, v2, tau in refl:
            r0, r1 = lo - s, min(p + 2, i) - s + 1
            sm = tau * (Wm[r0:r1, p - s] + Wm[r0:r1, p + 1 - s] * v2)
            Wm[r0:r1, p - s] -= sm
            Wm[r0:r1, p + 1 - s] -= sm * np.conj(v2)
            if U is not None:
                sm = tau * (U[:, p - s] + U[:, p + 1 - s] * v2)
                U[:, p - s] -= sm
                U[:, p + 1 - s] -= sm * np.conj(v2)


def ms_qr(hin, stats=None, wwin=W_WIN, mb=M_BULGES, nsmall=16, nshift=None, rep=1):
    """Eigenvalues of an upper-Hessenberg complex matrix by windowed multishift QR."""
    H = np.array(hin, dtype=np.complex128)
    n = H.shape[0]
    w = np.zeros(n, dtype=np.complex128)
    smlnum = SAFMIN * (n / ULP)
    st = stats if stats is not None else {}
    for key in ("sweeps", "stages", "steps", "base_sweeps", "base_steps", "small_calls", "gemm_cols"):
        st.setdefault(key, 0)
    i = n - 1
    nsweep_here = 0
    while i >= 0:
        l = _find_small(H, 0, i, smlnum)
        if l > 0:
            H[l, l - 1] = 0.0
        if l == i:
            w[i] = H[i, i]
            i -= 1
            nsweep_here = 0
            continue
        na = i - l + 1
        if na <= nsmall:
            st["small_calls"] += 1
            w[l:i + 1] = small_zqr(H[l:i + 1, l:i + 1], st)
            i = l - 1
            nsweep_here = 0
            continue
        nsweep_here += 1
        if nsweep_here > 30 * max(10, n):
            raise RuntimeError("ms_qr failed to converge")
        ns = min(mb, nsmall) if nshift is None else nshift
        st["small_calls"] += 1
        if nsweep_here % 6 == 0:                                  # exceptional shifts
            shifts = np.array([H[j, j] + 0.75 * abs1(H[j, j - 1]) for j in range(i - ns + 1, i + 1)])
        else:
            shifts = small_zqr(H[i - ns + 1:i + 1, i - ns + 1:i + 1], st)
        if rep > 1:
            shifts = np.repeat(shifts, rep)
        m = len(shifts)
        tmax = (i - 1 - l) + 2 * (m - 1)
        if na <= wwin:                                            # whole active block in one window, no U
            Wm = H[l:i + 1, l:i + 1].copy()
            _chase_window(Wm, None, l, i, l, l, i, shifts, 0, tmax)
            H[l:i + 1, l:i + 1] = Wm
            st["base_sweeps"] += 1
            st["base_steps"] += tmax + 1
            continue
        nbs = wwin - 2 * m - 1
        st["sweeps"] += 1
        k = 0
        while k * nbs <= tmax:
            t0, t1 = k * nbs, min((k + 1) * nbs - 1, tmax)
            s = max(l, l + t0 - 2 * (m - 1) - 1)
            e = min(i, l + t1 + 2)
            wd = e - s + 1
            assert wd <= wwin, (wd, wwin)
            Wm = H[s:e + 1, s:e + 1].copy()
            U = np.eye(wd, dtype=np.complex128)
            _chase_window(Wm, U, s, e, s, l, i, shifts, t0, t1)
            H[s:e + 1, s:e + 1] = Wm
            if s > l:
                H[l:s, s:e + 1] = H[l:s, s:e + 1] @ U
            if e < i:
                H[s:e + 1, e + 1:i + 1] = np.conj(U).T @ H[s:e + 1, e + 1:i + 1]
            st["stages"] += 1
            st["steps"] += t1 - t0 + 1
            st["gemm_cols"] += (s - l) + (i - e)
            k += 1
    return w


def eigvals(a, nb=16, stats=None):
    return ms_qr(hess_blocked(a, nb), stats)
