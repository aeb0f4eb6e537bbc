#!/usr/bin/env python3
"""Lane-level model of qr2_sweep / qr2_iter (csrc/eig_real_lds.hip): 64 lanes with the kernel's per-lane pointers,
validity constants and parking rules, run on a NaN-padded array so that any use of an out-of-window operand shows up.
Plugged into oracle/hqr2_proto's driver in place of its sweep and compared with LAPACK.  Development tool (CPU)."""
import math
import sys
import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from oracle import hqr2_proto as P2, hqr_proto as P1  # noqa: E402
import oracle as O  # noqa: E402

NEVER = 1 << 20
DIST = int(__import__("os").environ.get("QR2_DIST", 4))
PAD = 80


def coeffs(a0, a1, a2):
    """larfg3_coeffs of the kernel: a column with |a|^2 < 2^-800 is rescaled by a power of two first (xLARFG's rescaling;
    u and t are scale-invariant), an exactly zero column gets the reflection diag(-1, 1, 1)."""
    nrm2 = a0 * a0 + (a2 * a2 + a1 * a1)
    if nrm2 < 2.0 ** -800:
        big = max(abs(a0), abs(a1), abs(a2))
        if big > 0.0:
            e = math.frexp(big)[1] - 1
            a0, a1, a2 = math.ldexp(a0, -e), math.ldexp(a1, -e), math.ldexp(a2, -e)
        else:
            a0, a1, a2 = 1.0, 0.0, 0.0
        nrm2 = a0 * a0 + (a2 * a2 + a1 * a1)
    rn = 1.0 / math.sqrt(nrm2)
    apn = abs(a0) + nrm2 * rn
    scal = math.copysign(1.0 / apn, a0)
    t1 = apn * rn
    u1, u2 = a1 * scal, a2 * scal
    return t1, t1 * u1, t1 * u2, u1, u2


def sweep(hp, l, i, m, v, rts, stats):
    """hp: padded array (n+PAD)^2 whose [0:n,0:n] block is H, NaN elsewhere."""
    two = (m == l)
    nA = i - m
    T = nA + (DIST if two else 0)
    lanes = []
    for lane in range(64):
        g, j = lane >> 5, lane & 31
        S = dict(g=g, j=j, c1=max(i - 31 + j, 0), c0=max(i - 63 + j, 0), r0=min(l + j, 63), r1=min(l + 32 + j, 63),
                 krow=m, kcol=m, bpr=m + 1, bpc=m)
        # the kernel folds the window thresholds L = Lbase - t, M = Mbase + t into the per-lane constants, so every
        # validity test compares a constant with the iteration counter t itself
        Lbase, Mbase = nA + 1, m + 3 - l
        S["er_live"] = Lbase - (31 - j - DIST * g)
        S["ec0_live"] = j + DIST * g - Mbase if l + j <= i else NEVER
        S["ec1_live"] = j + 32 + DIST * g - Mbase if l + 32 + j <= i else NEVER
        live = g == 0
        S["er"] = S["er_live"] if live else -NEVER
        S["ec0"] = S["ec0_live"] if live else NEVER
        S["ec1"] = S["ec1_live"] if live else NEVER
        S["inc"] = 1 if live else 0
        lanes.append(S)
    v0, v1, v2 = v
    rt1r, rt1i, rt2r, rt2i = rts
    for t in range(T):
        stats["iters"] = stats.get("iters", 0) + 1
        L = nA - t + 1
        M = m + t + 3 - l
        fA, lA = t == 0, t == nA - 1
        fB, lB = two and t == DIST, two and t == nA - 1 + DIST
        if fB:
            hmm, hm1m, hm1m1 = hp[m, m], hp[m + 1, m], hp[m + 1, m + 1]
            s = abs(hmm - rt2r) + abs(rt2i) + abs(hm1m)
            h21s = hm1m / s
            v0 = h21s * hp[m, m + 1] + (hmm - rt1r) * ((hmm - rt2r) / s) - rt1i * (rt2i / s)
            v1 = h21s * (hmm + hm1m1 - rt1r - rt2r)
            v2 = h21s * hp[m + 2, m + 1]
            s = abs(v0) + abs(v1) + abs(v2)
            v0, v1, v2 = v0 / s, v1 / s, v2 / s
            for S in lanes:
                if S["g"] == 1:
                    S["er"], S["ec0"], S["ec1"], S["inc"] = S["er_live"], S["ec0_live"], S["ec1_live"], 1
        gen = fA or fB or lA or lB
        # ---- loads of the bulge column + reflectors (per lane) ----
        for S in lanes:
            first = (fA if S["g"] == 0 else fB) if gen else False
            last = (lA if S["g"] == 0 else lB) if gen else False
            S["first"], S["last"] = first, last
            a0, a1, a2 = hp[S["bpr"], S["bpc"]], hp[S["bpr"] + 1, S["bpc"]], hp[S["bpr"] + 2, S["bpc"]]
            if first:
                a0, a1, a2 = v0, v1, v2
            elif last:
                a2 = 0.0
            S["co"] = coeffs(a0, a1, a2)
            S["tg"] = t + 1 if (first and two) else t
        # ---- row passes ----
        for sub in range(2):
            if sub == 1 and not (L >= 28):
                continue
            writes = {}
            for S in lanes:
                c = S["c1"] if sub == 0 else S["c0"]
                k = S["krow"]
                x = [hp[k, c], hp[k + 1, c], hp[k + 2, c]]
                if S["last"]:
                    x[2] = 0.0
                t1, t2, t3, u1, u2 = S["co"]
                sm = x[0] + u1 * x[1] + u2 * x[2]
                n = [x[0] - sm * t1, x[1] - sm * t2, x[2] - sm * t3]
                if S["er"] - 32 * sub >= S["tg"]:
                    for q in range(2 if S["last"] else 3):
                        assert (k + q, c) not in writes, ("row overlap", t, k + q, c)
                        assert l - (0 if two else 1) <= c <= i and m <= k + q <= i, ("row range", t, k + q, c, l, i, m)
                        writes[(k + q, c)] = n[q]
            for (r, c), val in writes.items():
                assert math.isfinite(val), ("row nan", t, r, c)
                hp[r, c] = val
        # ---- column passes ----
        for sub in range(2):
            if sub == 1 and not (M >= 32):
                continue
            writes = {}
            for S in lanes:
                r = S["r0"] if sub == 0 else S["r1"]
                k = S["kcol"]
                x = [hp[r, k], hp[r, k + 1], hp[r, k + 2]]
                if S["last"]:
                    x[2] = 0.0
                t1, t2, t3, u1, u2 = S["co"]
                sm = x[0] + u1 * x[1] + u2 * x[2]
                n = [x[0] - sm * t1, x[1] - sm * t2, x[2] - sm * t3]
                if (S["ec0"] if sub == 0 else S["ec1"]) <= t:
                    for q in range(2 if S["last"] else 3):
                        assert (r, k + q) not in writes, ("col overlap", t, r, k + q)
                        assert l <= r <= min(k + 3, i) and k + q <= i, ("col range", t, r, k + q)
                        writes[(r, k + q)] = n[q]
            for (r, c), val in writes.items():
                assert math.isfinite(val), ("col nan", t, r, c)
                hp[r, c] = val
        # ---- advance / park ----
        for S in lanes:
            if not S["last"]:
                S["krow"] += S["inc"]
                S["kcol"] += S["inc"]
            if not (S["first"] or S["last"]):
                S["bpr"] += S["inc"]
                S["bpc"] += S["inc"]
            if S["last"]:
                S["ec0"] = S["ec1"] = NEVER
                S["er"] = -NEVER
                S["inc"] = 0


def hqr_lanes(hin, stats):
    """hqr2_proto's driver (deflation, shifts, start search) around the lane-level sweep."""
    n = hin.shape[0]
    hp = np.full((n + PAD, n + PAD), np.nan)
    hp[:n, :n] = np.triu(hin, -1)
    hp[:n, :n][np.tril_indices(n, -2)] = 0.0
    h = hp[:n, :n]
    ULP, SAFMIN = P1.ULP, P1.SAFMIN
    wr, wi = np.zeros(n), np.zeros(n)
    smlnum = 1e-200        # the kernel's n = 64 drivers scale the matrix to max |entry| in [1, 2)
    i = n - 1
    kdefl = 0
    while i >= 0:
        l = 0
        converged = False
        for _ in range(30 * max(10, n) + 1):
            k = i
            while k > l:
                hkk1 = abs(h[k, k - 1])
                if hkk1 <= smlnum:
                    break
                tst = abs(h[k - 1, k - 1]) + abs(h[k, k])
                if tst == 0.0:
                    if k - 2 >= 0:
                        tst += abs(h[k - 1, k - 2])
                    if k + 1 <= n - 1:
                        tst += abs(h[k + 1, k])
                if hkk1 <= ULP * tst:
                    ab, ba = max(hkk1, abs(h[k - 1, k])), min(hkk1, abs(h[k - 1, k]))
                    aa = max(abs(h[k, k]), abs(h[k - 1, k - 1] - h[k, k]))
                    bb = min(abs(h[k, k]), abs(h[k - 1, k - 1] - h[k, k]))
                    s = aa + ab
                    if ba * (ab / s) <= max(smlnum, ULP * (bb * (aa / s))):
                        break
                k -= 1
            l = k
            if l > 0:
                h[l, l - 1] = 0.0
            if l >= i - 1:
                converged = True
                break
            kdefl += 1
            if kdefl % 20 == 0:
                s = abs(h[i, i - 1]) + abs(h[i - 1, i - 2])
                h11, h12, h21 = 0.75 * s + h[i, i], -0.4375 * s, s
                h22 = h11
            elif kdefl % 10 == 0:
                s = abs(h[l + 1, l]) + abs(h[l + 2, l + 1])
                h11, h12, h21 = 0.75 * s + h[l, l], -0.4375 * s, s
                h22 = h11
            else:
                h11, h21, h12, h22 = h[i - 1, i - 1], h[i, i - 1], h[i - 1, i], h[i, i]
            s = abs(h11) + abs(h12) + abs(h21) + abs(h22)
            if s == 0.0:
                rt1r = rt1i = rt2r = rt2i = 0.0
            else:
                h11, h21, h12, h22 = h11 / s, h21 / s, h12 / s, h22 / s
                tr = (h11 + h22) / 2.0
                det = (h11 - tr) * (h22 - tr) - h12 * h21
                rtdisc = math.sqrt(abs(det))
                if det >= 0.0:
                    rt1r = rt2r = tr * s
                    rt1i, rt2i = rtdisc * s, -rtdisc * s
                else:
                    rt1r, rt2r = tr + rtdisc, tr - rtdisc
                    if abs(rt1r - h22) <= abs(rt2r - h22):
                        rt1r = rt2r = rt1r * s
                    else:
                        rt1r = rt2r = rt2r * s
                    rt1i = rt2i = 0.0
            m = i - 2
            while True:
                h21s = abs(h[m + 1, m])
                s = abs(h[m, m] - rt2r) + abs(rt2i) + h21s
                h21s = h[m + 1, m] / s
                v0 = h21s * h[m, m + 1] + (h[m, m] - rt1r) * ((h[m, m] - rt2r) / s) - rt1i * (rt2i / s)
                v1 = h21s * (h[m, m] + h[m + 1, m + 1] - rt1r - rt2r)
                v2 = h21s * h[m + 2, m + 1]
                s = abs(v0) + abs(v1) + abs(v2)
                v0, v1, v2 = v0 / s, v1 / s, v2 / s
                if m == l:
                    break
                if abs(h[m, m - 1]) * (abs(v1) + abs(v2)) <= ULP * abs(v0) * (abs(h[m - 1, m - 1]) + abs(h[m, m]) + abs(h[m + 1, m + 1])):
                    break
                m -= 1
            stats["sweeps"] = stats.get("sweeps", 0) + 1
            sweep(hp, l, i, m, (v0, v1, v2), (rt1r, rt1i, rt2r, rt2i), stats)
        if not converged:
            return wr + 1j * wi, i + 1
        if l == i:
            wr[i], wi[i] = h[i, i], 0.0
        else:
            wr[i - 1], wi[i - 1], wr[i], wi[i] = P1.lanv2(h[i - 1, i - 1], h[i - 1, i], h[i, i - 1], h[i, i])
        kdefl = 0
        i = l - 1
    return wr + 1j * wi, 0


if __name__ == "__main__":
    import scipy.linalg as sl
    rng = np.random.default_rng(3)
    worst = 0.0
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        n = 64
        kind = trial % 3
        if kind == 0:
            a = rng.standard_normal((n, n)) / 8
        elif kind == 1:
            a = np.diag(rng.standard_normal(n)) + np.triu(rng.standard_normal((n, n)), 1) * 0.3
            q, _ = np.linalg.qr(rng.standard_normal((n, n)))
            a = q @ a @ q.T
        else:
            a = np.zeros((n, n))
            perm = rng.permutation(n)
            a[np.arange(n), perm] = 1.0           # permutation matrix: exceptional shifts
        hh = sl.hessenberg(a)
        st = {}
        w, info = hqr_lanes(hh, st)
        ref = np.linalg.eigvals(a)
        d = O.max_abs_dlambda(w[None], ref[None])
        worst = max(worst, d)
        w1, _ = P1.hqr_eigvals(hh)
        w2, _ = P2.hqr2_eigvals(hh, "same")
        d1 = O.max_abs_dlambda(w1[None], ref[None])
        d2 = O.max_abs_dlambda(w2[None], ref[None])
        print(trial, kind, "info", info, "dlambda %.2e (single-bulge proto %.2e, two-bulge proto %.2e)" % (d, d1, d2), st)
    print("worst", worst)
