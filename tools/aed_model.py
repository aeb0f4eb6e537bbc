#!/usr/bin/env python3
"""What aggressive early deflation (Braman, Byers, Mathias 2002, part II; LAPACK zlaqr2) would buy the level-3 QR of
csrc/eig_zlarge.hip: the NumPy prototype of that kernel (oracle/zms_proto.py: small-bulge multishift QR, chains of 16
bulges, 64 x 64 windows) with and without an AED step in front of every sweep, on config C5's matrices (HiPPO-DPLR S4
`Ab`, n = 256) and on dense complex Gaussian matrices.  Counts sweeps, chain time steps (what the chase kernel's time
is proportional to) and the columns the deferred U updates touch; the spectra of both variants are checked against
LAPACK.  Development tool (imports the oracle package; nothing here is on the product path).

    python tools/aed_model.py [n] [window]"""
import os
import sys

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O
from oracle import zms_proto as Z

ULP, SAFMIN = Z.ULP, Z.SAFMIN


def swap_adjacent(T, V, k):
    """Exchange T[k,k] and T[k+1,k+1] of an upper-triangular T by a unitary similarity (zlartg + zrot of ztrexc)."""
    t11, t22, t12 = T[k, k], T[k + 1, k + 1], T[k, k + 1]
    f, g = t12, t22 - t11
    r = np.hypot(abs(f), abs(g))
    if r == 0.0:
        return
    c, s = f / r, g / r                                   # rotation G = [[c, s], [-conj(s), conj(c)]] acting on columns
    n = T.shape[0]
    G = np.array([[c, -np.conj(s)], [s, np.conj(c)]])     # columns: new_k = c col_k + s col_k+1
    T[:, k:k + 2] = T[:, k:k + 2] @ G
    T[k:k + 2, :] = np.conj(G).T @ T[k:k + 2, :]
    V[:, k:k + 2] = V[:, k:k + 2] @ G
    T[k + 1, k] = 0.0


def aed(H, l, i, nw, smlnum, st):
    """One AED step on the trailing window of the active block [l, i].  Returns (nd, shifts): nd eigenvalues deflated
    (written to the diagonal of H, H stays upper Hessenberg), shifts = the undeflatable eigenvalues of the window."""
    kwtop = max(l, i - nw + 1)
    jw = i - kwtop + 1
    s = H[kwtop, kwtop - 1] if kwtop > l else 0.0
    T, V = sl.schur(H[kwtop:i + 1, kwtop:i + 1], output="complex")
    st["aed_calls"] += 1
    st["aed_schur_n3"] += jw ** 3
    ns, ilst = jw, 0
    for _ in range(jw):
        foo = Z.abs1(T[ns - 1, ns - 1]) or Z.abs1(s)
        if Z.abs1(s) * Z.abs1(V[0, ns - 1]) <= max(smlnum, ULP * foo):
            ns -= 1                                       # converged: stays at the bottom
        else:
            for k in range(ns - 2, ilst - 1, -1):         # move it up to position ilst
                swap_adjacent(T, V, k)
            ilst += 1
    nd = jw - ns
    shifts = np.diag(T)[:ns].copy()
    if nd == 0:
        return 0, shifts
    if ns == 0:
        s = 0.0
    # restore Hessenberg form: spike s * conj(V[0, :ns]) -> (beta, 0, ..) by a reflector, then re-reduce T[:ns, :ns]
    if ns > 1 and s != 0.0:
        spike = s * np.conj(V[0, :ns])
        x = spike.copy()
        alpha = x[0]
        nrm = np.linalg.norm(x)
        beta = -np.copysign(nrm, alpha.real) if alpha.real != 0 else -nrm
        v = x.copy()
        v[0] = alpha - beta
        v /= v[0]
        tau = (beta - alpha) / beta
        Hh = np.eye(ns, dtype=complex) - tau * np.outer(v, np.conj(v))      # Hh^H spike = beta e1
        T[:ns, :] = np.conj(Hh).T @ T[:ns, :]
        T[:, :ns] = T[:, :ns] @ Hh
        V[:, :ns] = V[:, :ns] @ Hh
        Hs, Q = sl.hessenberg(T[:ns, :ns], calc_q=True)
        T[:ns, :ns] = Hs
        T[:ns, ns:] = np.conj(Q).T @ T[:ns, ns:]
        V[:, :ns] = V[:, :ns] @ Q
        st["aed_rehess_n3"] += ns ** 3
    if kwtop > l:
        H[kwtop, kwtop - 1] = s * np.conj(V[0, 0])
    H[kwtop:i + 1, kwtop:i + 1] = np.triu(T, -1)
    if kwtop > l:
        H[l:kwtop, kwtop:i + 1] = H[l:kwtop, kwtop:i + 1] @ V       # columns above the window (eigenvalues only)
        st["gemm_cols"] += kwtop - l
    return nd, shifts


def ms_qr_aed(hin, st, nw, mb=16, wwin=64, nsmall=16, nibble=14):
    """zms_proto.ms_qr with an AED step before every sweep (LAPACK zlaqr0's control flow, simplified)."""
    H = np.array(hin, dtype=np.complex128)
    n = H.shape[0]
    w = np.zeros(n, dtype=np.complex128)
    smlnum = SAFMIN * (n / ULP)
    for key in ("sweeps", "stages", "steps", "base_sweeps", "base_steps", "small_calls", "gemm_cols", "aed_calls",
                "aed_schur_n3", "aed_rehess_n3", "aed_deflated"):
        st.setdefault(key, 0)
    i = n - 1
    nsweep_here = 0
    while i >= 0:
        l = Z._find_small(H, 0, i, smlnum)
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
            w[l:i + 1] = Z.small_zqr(H[l:i + 1, l:i + 1], st)
            i = l - 1
            nsweep_here = 0
            continue
        nd, shifts = aed(H, l, i, min(nw, na), smlnum, st)
        if nd > 0:
            st["aed_deflated"] += nd
            for j in range(i - nd + 1, i + 1):
                w[j] = H[j, j]
            i -= nd
            nsweep_here = 0
            if i < l or 100 * nd > min(nw, na) * nibble:      # enough deflated: look again before sweeping
                continue
            na = i - l + 1
            if na <= nsmall:
                continue
        nsweep_here += 1
        if nsweep_here > 300:
            raise RuntimeError("ms_qr_aed failed to converge")
        if len(shifts) >= 2:
            shifts = shifts[-min(mb, len(shifts)):]
        else:
            shifts = Z.small_zqr(H[i - mb + 1:i + 1, i - mb + 1:i + 1], st)
        if len(shifts) < mb:                                  # the chain kernel runs 16 bulges: repeat what there is
            shifts = np.resize(shifts, mb)
        m = len(shifts)
        tmax = (i - 1 - l) + 2 * (m - 1)
        if na <= wwin:
            Wm = H[l:i + 1, l:i + 1].copy()
            Z._chase_window(Wm, None, l, i, l, l, i, shifts, 0, tmax)
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
            Wm = H[s:e + 1, s:e + 1].copy()
            U = np.eye(wd, dtype=np.complex128)
            Z._chase_window(Wm, U, s, e, s, l, i, shifts, t0, t1)
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


def c5_matrix(n, rng):
    lam, p, _ = O.hippo_dplr(n) if hasattr(O, "hippo_dplr") else (None, None, None)
    if lam is None:
        q = np.sqrt(1 + 2 * np.arange(n))
        a = -(np.tril(q[:, None] * q[None, :]) - np.diag(np.arange(n)))
        pp = np.sqrt(np.arange(n) + 0.5)
        s = a + pp[:, None] * pp[None, :]
        lam_im, v = np.linalg.eigh(s * -1j)
        lam = np.mean(np.diagonal(s)) + 1j * lam_im
        p = v.conj().T @ pp
    step = np.exp(rng.uniform(np.log(1e-3), np.log(1e-1)))
    return O.s4_dplr_ab(np.minimum(lam.real, -1e-4) + 1j * lam.imag, p, step)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    nw = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    rng = np.random.default_rng(3)
    for kind in ("c5_hippo", "randn"):
        tot = {"plain": {}, "aed": {}}
        for rep in range(2):
            a = c5_matrix(n, rng) if kind == "c5_hippo" else (rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))) / np.sqrt(2 * n)
            h = Z.hess_blocked(a, 16)
            ref = np.linalg.eigvals(a)
            for tag in ("plain", "aed"):
                st = {}
                w = Z.ms_qr(h, st) if tag == "plain" else ms_qr_aed(h, st, nw)
                err = O.max_abs_dlambda(w, ref)
                for k2, v2 in st.items():
                    tot[tag][k2] = tot[tag].get(k2, 0) + v2
                tot[tag]["max_err"] = max(tot[tag].get("max_err", 0.0), err)
        print(kind, "n", n, "window", nw)
        for tag in ("plain", "aed"):
            t = tot[tag]
            print("  %-5s sweeps %4d  chain steps %6d  single-window steps %5d  U-update columns %7d  small-QR steps %6d"
                  "  aed calls %3d deflated %3d schur n^3 %8d  max err %.1e" % (
                      tag, t.get("sweeps", 0), t.get("steps", 0), t.get("base_steps", 0), t.get("gemm_cols", 0),
                      t.get("small_steps", 0), t.get("aed_calls", 0), t.get("aed_deflated", 0), t.get("aed_schur_n3", 0),
                      t["max_err"]))


if __name__ == "__main__":
    main()
