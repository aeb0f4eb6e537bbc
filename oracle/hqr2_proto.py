"""Scalar NumPy prototype of the TWO-BULGE Francis sweep of the n = 64 HIP solver (csrc/eig_real_lds.hip, round 3).
TEST INFRASTRUCTURE (never imported by the product).

One sweep chases two double-shift bulges through the active window [l, i], the second one `DIST` rows behind the first,
exactly in the order the wavefront executes them: per iteration both reflectors are formed from the current bulge
columns, then both row operations are applied, then both column operations (lanes 0-31 of the wavefront work for the
leading bulge, lanes 32-63 for the trailing one, same instructions).  With DIST >= 4 the two row operations touch
disjoint rows, the two column operations disjoint columns and no element receives a row operation of one bulge and a
column operation of the other in the same iteration, so the result equals two consecutive single-bulge sweeps
(LAPACK dlahqr's deflation tests, exceptional shifts and start search are kept).  Shift strategies:
  "same" : both bulges use the Wilkinson double shift of the trailing 2x2 block;
  "e44"  : the two real quadratic factors of the characteristic polynomial of the trailing 4x4 block (Bairstow's
           iteration started from the trailing 2x2 block's quadratic; shifts need no more than a few digits).
The reference itself only calls np.linalg.eigvals (analysis/eval_eig.py:296)."""
from __future__ import annotations

import math

import numpy as np

from .hqr_proto import ULP, SAFMIN, lanv2

DIST = 4
EXACT_ZEROS = False
E44_MIN_WINDOW = 44


def charpoly4(h):
    """Monic characteristic polynomial x^4 + c3 x^3 + c2 x^2 + c1 x + c0 of a 4x4 upper-Hessenberg block (Hyman's
    recurrence p_k = (x - h_kk) p_{k-1} - sum_j h_jk (prod of subdiagonals j+1..k) p_{j-1}), coefficients low -> high."""
    p = [np.array([1.0])]
    for k in range(4):
        q = np.convolve(p[k], np.array([-h[k, k], 1.0]))
        sub = 1.0
        for j in range(k - 1, -1, -1):
            sub *= h[j + 1, j]
            t = -h[j, k] * sub * p[j]
            q[:t.size] += t
        p.append(q)
    return p[4]          # length 5, p[4][4] == 1


def bairstow_split(c, s0, p0, iters=6):
    """Factor the monic quartic with coefficients c (low -> high) into (x^2 - s x + p)(x^2 - s' x + p'), starting from
    (s0, p0) = (sum, product) of a shift pair.  Returns (s, p, s', p').  A fixed number of iterations; a failed step
    (singular Jacobian) keeps the current estimate -- the result only has to be a usable pair of double shifts."""
    a3, a2, a1, a0 = c[3], c[2], c[1], c[0]
    u, v = -s0, p0                                   # divisor x^2 + u x + v
    for _ in range(iters):
        # quartic / (x^2 + u x + v): quotient x^2 + b1 x + b0, remainder r1 x + r0
        b1 = a3 - u
        b0 = a2 - u * b1 - v
        r1 = a1 - u * b0 - v * b1
        r0 = a0 - v * b0
        # quotient / divisor again (partial derivatives of the remainder)
        c1 = b1 - u
        c0 = b0 - u * c1 - v
        # d r1/du = -c0 ... standard Bairstow: solve [c0' c1'; ...]
        # With g = quotient: dr/du = -(x g) mod d, dr/dv = -g mod d.
        # g mod d = (b1 - u) x + (b0 - v) =: g1 x + g0
        g1, g0 = b1 - u, b0 - v
        # x g mod d = g1 x^2 + g0 x = g1 (-u x - v) + g0 x = (g0 - u g1) x - v g1
        xg1, xg0 = g0 - u * g1, -v * g1
        # Newton: [xg1 g1; xg0 g0] [du dv]^T = [r1 r0]^T
        det = xg1 * g0 - g1 * xg0
        if det == 0.0 or not math.isfinite(det):
            break
        du = (r1 * g0 - g1 * r0) / det
        dv = (xg1 * r0 - xg0 * r1) / det
        if not (math.isfinite(du) and math.isfinite(dv)):
            break
        u += du
        v += dv
    b1 = a3 - u
    b0 = a2 - u * b1 - v
    return -u, v, -b1, b0


def _shift_pair_to_roots(s, p):
    """(rt1r, rt1i, rt2r, rt2i) of x^2 - s x + p, real roots kept distinct (a double shift only needs s and p)."""
    tr = 0.5 * s
    disc = tr * tr - p
    if disc < 0.0:
        r = math.sqrt(-disc)
        return tr, r, tr, -r
    r = math.sqrt(disc)
    return tr + r, 0.0, tr - r, 0.0


def _start(h, m, s, p):
    """Start vector of a double-shift sweep at row m for the shift pair with sum s and product p (scaled to unit
    1-norm): first column of (H - r1)(H - r2) restricted to rows m..m+2."""
    h00, h10, h01, h11, h21 = h[m, m], h[m + 1, m], h[m, m + 1], h[m + 1, m + 1], h[m + 2, m + 1]
    sc = abs(h00) + abs(h10) + abs(h01) + abs(h11) + abs(s) + math.sqrt(abs(p))
    if sc == 0.0:
        return 0.0, 0.0, 0.0
    v0 = (h00 * (h00 - s) + p) / sc + h01 * (h10 / sc)
    v1 = (h10 / sc) * (h00 + h11 - s)
    v2 = (h10 / sc) * h21
    t = abs(v0) + abs(v1) + abs(v2)
    if t == 0.0:
        return 0.0, 0.0, 0.0
    return v0 / t, v1 / t, v2 / t


def _larfg3(v0, v1, v2):
    xn = math.hypot(v1, v2)
    if xn == 0.0:
        return v0, 0.0, 0.0, 0.0
    beta = -math.copysign(math.hypot(v0, xn), v0)
    t1 = (beta - v0) / beta
    sc = 1.0 / (v0 - beta)
    return beta, t1, v1 * sc, v2 * sc


def _aed_real(h, l, i, w, smlnum, stats, deflate=True):
    """Aggressive early deflation on the trailing window of the active block [l, i] (Braman, Byers, Mathias 2002 II;
    LAPACK dlaqr2's flow): real Schur form of the window with vectors, spike test from the bottom, undeflatable blocks
    moved to the top (dtrexc), Hessenberg form restored by a reflector on the spike and a re-reduction, the columns of
    the rows above the window multiplied by the accumulated V (eigenvalues only: nothing right of the window).
    Returns (eigenvalues deflated [complex list, bottom first], shift pairs of the best-converged undeflated blocks).
    Model code: SciPy's LAPACK does the window's Schur form, the swaps and the re-reduction -- what is counted is how
    often and on which sizes (stats["aed_log"]), the kernel-side price is set by the caller (tools/aed64_model.py)."""
    import scipy.linalg as sl
    from scipy.linalg import lapack
    jw = min(w, i - l + 1)
    kwtop = i - jw + 1
    s = h[kwtop, kwtop - 1] if kwtop > l else 0.0
    T, V = sl.schur(np.triu(h[kwtop:i + 1, kwtop:i + 1], -1), output="real")
    T = np.asfortranarray(T)
    V = np.asfortranarray(V)
    ns, ilst, nswap = jw, 1, 0
    while ilst <= ns:
        bulge = ns > 1 and T[ns - 1, ns - 2] != 0.0
        if not bulge:
            foo = abs(T[ns - 1, ns - 1]) or abs(s)
            if abs(s * V[0, ns - 1]) <= max(smlnum, ULP * foo):
                ns -= 1
            else:
                T, V, info = lapack.dtrexc(T, V, ns, ilst)
                nswap += ns - ilst
                ilst += 1
        else:
            foo = abs(T[ns - 1, ns - 1]) + math.sqrt(abs(T[ns - 1, ns - 2])) * math.sqrt(abs(T[ns - 2, ns - 1]))
            if foo == 0.0:
                foo = abs(s)
            if max(abs(s * V[0, ns - 1]), abs(s * V[0, ns - 2])) <= max(smlnum, ULP * foo):
                ns -= 2
            else:
                T, V, info = lapack.dtrexc(T, V, ns - 1, ilst)
                nswap += 2 * (ns - 1 - ilst)
                ilst += 2
    nd = jw - ns
    # eigenvalues of the window's diagonal blocks: [0, ns) undeflated (candidate shifts), [ns, jw) deflated
    blocks = []
    k = 0
    while k < jw:
        if k + 1 < jw and T[k + 1, k] != 0.0:
            blocks.append((k, 2, lanv2(T[k, k], T[k, k + 1], T[k + 1, k], T[k + 1, k + 1])))
            k += 2
        else:
            blocks.append((k, 1, (T[k, k], 0.0, T[k, k], 0.0)))
            k += 1
    deflated = []
    for (k0, sz, (r1, i1, r2, i2)) in blocks:
        if k0 >= ns:
            deflated.append((k0, complex(r1, i1)))
            if sz == 2:
                deflated.append((k0 + 1, complex(r2, i2)))
    # shift pairs (sum, product): the undeflated blocks with the smallest spike entries first; two real eigenvalues
    # make one pair
    cand = sorted((b for b in blocks if b[0] < ns), key=lambda b: max(abs(V[0, b[0]]), abs(V[0, b[0] + b[1] - 1])))
    pairs, reals = [], []
    for (k0, sz, (r1, i1, r2, i2)) in cand:
        if sz == 2:
            pairs.append((r1 + r2, r1 * r2 - i1 * i2))
        else:
            reals.append(r1)
            if len(reals) == 2:
                pairs.append((reals[0] + reals[1], reals[0] * reals[1]))
                reals = []
    if stats is not None:
        stats.setdefault("aed_log", []).append((l, i, jw, ns if deflate else jw, nswap))
        stats.setdefault("aed_windows", []).append(np.triu(h[kwtop:i + 1, kwtop:i + 1], -1).copy())
    if nd == 0 or not deflate:
        return [], pairs
    if ns == 0:
        s = 0.0
    if ns > 1 and s != 0.0:
        spike = s * V[0, :ns]
        beta, _t1, _u1, _u2 = 0.0, 0.0, 0.0, 0.0
        alpha = spike[0]
        nrm = float(np.linalg.norm(spike))
        beta = -math.copysign(nrm, alpha) if alpha != 0.0 else -nrm
        v = spike.copy()
        v[0] = alpha - beta
        v /= v[0]
        tau = (beta - alpha) / beta
        Hh = np.eye(ns) - tau * np.outer(v, v)
        T[:ns, :] = Hh @ T[:ns, :]
        T[:, :ns] = T[:, :ns] @ Hh
        V[:, :ns] = V[:, :ns] @ Hh
        Hs, Q = sl.hessenberg(T[:ns, :ns], calc_q=True)
        T[:ns, :ns] = Hs
        T[:ns, ns:] = Q.T @ T[:ns, ns:]
        V[:, :ns] = V[:, :ns] @ Q
    if kwtop > l:
        h[kwtop, kwtop - 1] = s * V[0, 0]
    h[kwtop:i + 1, kwtop:i + 1] = np.triu(T, -1)
    if kwtop > l:
        h[l:kwtop, kwtop:i + 1] = h[l:kwtop, kwtop:i + 1] @ V
    return deflated, pairs


class _Bulge:
    __slots__ = ("k", "vec", "first", "s", "p")


def hqr2_eigvals(hin, shifts="e44", stats=None, nb=2, aed=0, aed_min=0, aed_shifts=False, nibble=14, aed_deflate=True):
    """aed = w > 0: an early-deflation step on the trailing w rows before every sweep of a window of more than aed_min
    rows (model runs, tools/aed64_model.py); aed_shifts: the two best-converged undeflated blocks of the window give the
    two bulges' shift pairs; nibble: skip the sweep when at least that percentage of the window deflated (dlaqr0)."""
    h = np.array(hin, dtype=np.float64)
    n = h.shape[0]
    wr = np.zeros(n)
    wi = np.zeros(n)
    if n == 0:
        return wr + 1j * wi, 0
    smlnum = SAFMIN * (n / ULP)
    itmax = 30 * max(10, n)
    kexsh = 10
    sweeps = steps = iters_total = 0
    i = n - 1
    kdefl = 0
    while i >= 0:
        l = 0
        converged = False
        for _its in range(itmax + 1):
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
                    ab = max(hkk1, abs(h[k - 1, k]))
                    ba = min(hkk1, abs(h[k - 1, k]))
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
            aed_pairs = None
            if aed > 0 and i - l + 1 > max(aed_min, 3):
                defl, aed_pairs = _aed_real(h, l, i, aed, smlnum, stats, aed_deflate)
                if defl:
                    jw_ = min(aed, i - l + 1)
                    for (k0, z) in defl:
                        wr[i - jw_ + 1 + k0], wi[i - jw_ + 1 + k0] = z.real, z.imag
                    i -= len(defl)
                    kdefl = 0
                    if i < l:
                        converged = None          # the window took the rest of the active block: start over above it
                        break
                    if len(defl) * 100 >= nibble * jw_ or i - l + 1 < 3:
                        continue                  # re-scan (and try the window again) before spending a sweep
                    if not aed_shifts:
                        aed_pairs = None
                if l >= i - 1:
                    continue
            kdefl += 1
            # ---- shift pairs (sum, product) of the two bulges ----
            if kdefl % (2 * kexsh) == 0:
                s = abs(h[i, i - 1]) + abs(h[i - 1, i - 2])
                h11 = 0.75 * s + h[i, i]
                sA, pA = 2.0 * h11, h11 * h11 + 0.4375 * s * s
                sB, pB = sA, pA
            elif kdefl % kexsh == 0:
                s = abs(h[l + 1, l]) + abs(h[l + 2, l + 1])
                h11 = 0.75 * s + h[l, l]
                sA, pA = 2.0 * h11, h11 * h11 + 0.4375 * s * s
                sB, pB = sA, pA
            else:
                a, b, c, d = h[i - 1, i - 1], h[i - 1, i], h[i, i - 1], h[i, i]
                sA, pA = a + d, a * d - b * c
                # Wilkinson's choice for real roots: use the one closer to h[i,i] twice
                disc = 0.25 * (a - d) ** 2 + b * c
                if disc >= 0.0:
                    r = math.sqrt(disc)
                    r1, r2 = 0.5 * (a + d) + r, 0.5 * (a + d) - r
                    rr = r1 if abs(r1 - d) <= abs(r2 - d) else r2
                    sA, pA = 2.0 * rr, rr * rr
                sB, pB = sA, pA
                # "e44hi" (model runs): the 4 x 4 shifts only in windows of more than E44_MIN_WINDOW rows
                if (shifts == "e44" or (shifts == "e44hi" and i - l >= E44_MIN_WINDOW)) and i - l + 1 >= 6:
                    blk = h[i - 3:i + 1, i - 3:i + 1]
                    sc = float(np.abs(blk).sum())
                    if sc > 0.0:
                        cp = charpoly4(blk / sc)
                        a_, b_, c_, d_ = a / sc, b / sc, c / sc, d / sc
                        s1, p1, s2, p2 = bairstow_split(cp, a_ + d_, a_ * d_ - b_ * c_)
                        if all(math.isfinite(x) for x in (s1, p1, s2, p2)):
                            sA, pA, sB, pB = s1 * sc, p1 * sc * sc, s2 * sc, p2 * sc * sc
            if aed_shifts and aed_pairs and kdefl % kexsh != 0:
                sA, pA = aed_pairs[0]
                sB, pB = aed_pairs[1] if len(aed_pairs) > 1 else aed_pairs[0]
            # ---- common start row: the largest m >= l at which BOTH start vectors may be launched ----
            m = i - 2
            while m > l:
                ok = True
                for (s_, p_) in ((sA, pA), (sB, pB)):
                    v0, v1, v2 = _start(h, m, s_, p_)
                    h00 = abs(h[m - 1, m - 1])
                    if not (abs(h[m, m - 1]) * (abs(v1) + abs(v2)) <= ULP * abs(v0) * (h00 + abs(h[m, m]) + abs(h[m + 1, m + 1]))):
                        ok = False
                if ok:
                    break
                m -= 1
            sweeps += 1
            # ---- the pipelined chase ----
            # A start below the top of the window (m > l) was validated for the matrix as it is NOW; the trailing bulge
            # would start there three steps later on a changed matrix, so such sweeps (rare) chase one bulge only.
            single = m > l
            if stats is not None and single:
                stats["single_sweeps"] = stats.get("single_sweeps", 0) + 1
            bl = []
            # nb > 2 (model runs only): further bulges repeat the two shift pairs alternately
            for (s_, p_) in (((sA, pA),) if single else tuple(((sA, pA), (sB, pB))[q & 1] for q in range(nb))):
                b = _Bulge()
                b.s, b.p = s_, p_
                bl.append(b)
            nit = (i - 1 - m) + 1 + (0 if single else DIST * (nb - 1))
            if stats is not None and "log" in stats:
                stats["log"].append((l, i, m, nit))          # one record per sweep (cost models weight by window length)
            for t in range(nit):
                iters_total += 1
                act = []
                for bi, b in enumerate(bl):
                    k = m + t - bi * DIST
                    if k < m or k > i - 1:
                        continue
                    nr = min(3, i - k + 1)
                    if k == m:
                        v0, v1, v2 = _start(h, m, b.s, b.p)
                    else:
                        v0, v1 = h[k, k - 1], h[k + 1, k - 1]
                        v2 = h[k + 2, k - 1] if nr == 3 else 0.0
                    if nr == 2:
                        v2 = 0.0
                    beta, t1, u1, u2 = _larfg3(v0, v1, v2)
                    act.append((k, nr, beta, t1, u1, u2, k == m))
                    steps += 1
                # The bulge column k-1 is NOT overwritten with (beta, 0, 0): it goes through the row operation like
                # every other column (lane 0 of the group), which leaves (beta~, eps, eps) -- rounding-level entries
                # below the subdiagonal instead of exact zeros (EXACT_ZEROS restores LAPACK's form for comparison).
                c0 = {}
                for (k, nr, beta, t1, u1, u2, first) in act:
                    c0[k] = k - 1 if (k - 1 >= l and not EXACT_ZEROS) else k
                    if EXACT_ZEROS:
                        if not first:
                            h[k, k - 1] = beta
                            h[k + 1, k - 1] = 0.0
                            if nr == 3:
                                h[k + 2, k - 1] = 0.0
                        elif m > l:
                            h[k, k - 1] *= (1.0 - t1)
                # row operations of both bulges (disjoint rows)
                for (k, nr, beta, t1, u1, u2, first) in act:
                    t2, t3 = t1 * u1, t1 * u2
                    ca = c0[k]
                    if nr == 3:
                        sm = h[k, ca:i + 1] + u1 * h[k + 1, ca:i + 1] + u2 * h[k + 2, ca:i + 1]
                        h[k, ca:i + 1] -= sm * t1
                        h[k + 1, ca:i + 1] -= sm * t2
                        h[k + 2, ca:i + 1] -= sm * t3
                    else:
                        sm = h[k, ca:i + 1] + u1 * h[k + 1, ca:i + 1]
                        h[k, ca:i + 1] -= sm * t1
                        h[k + 1, ca:i + 1] -= sm * t2
                # column operations of both bulges (disjoint columns)
                for (k, nr, beta, t1, u1, u2, first) in act:
                    t2, t3 = t1 * u1, t1 * u2
                    if nr == 3:
                        hi = min(k + 3, i) + 1
                        sm = h[l:hi, k] + u1 * h[l:hi, k + 1] + u2 * h[l:hi, k + 2]
                        h[l:hi, k] -= sm * t1
                        h[l:hi, k + 1] -= sm * t2
                        h[l:hi, k + 2] -= sm * t3
                    else:
                        sm = h[l:i + 1, k] + u1 * h[l:i + 1, k + 1]
                        h[l:i + 1, k] -= sm * t1
                        h[l:i + 1, k + 1] -= sm * t2
        if converged is None:                     # the early-deflation window took the rest of the active block
            continue
        if not converged:
            if stats is not None:
                stats.update(sweeps=sweeps, steps=steps, iters=iters_total)
            return wr + 1j * wi, i + 1
        if l == i:
            wr[i] = h[i, i]
            wi[i] = 0.0
        else:
            wr[i - 1], wi[i - 1], wr[i], wi[i] = lanv2(h[i - 1, i - 1], h[i - 1, i], h[i, i - 1], h[i, i])
        kdefl = 0
        i = l - 1
    if stats is not None:
        stats.update(sweeps=sweeps, steps=steps, iters=iters_total)
    return wr + 1j * wi, 0
