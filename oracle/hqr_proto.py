"""Scalar NumPy prototype of the algorithm the HIP dense solver (csrc/eig_real_lds.hip) runs:
Householder reduction to Hessenberg form followed by the Francis double-shift QR iteration with
LAPACK-style deflation tests (the published xLAHQR algorithm: Golub & Van Loan 7.5; Ahues & Tisseur
LAWN 122 deflation criterion).  TEST INFRASTRUCTURE: used to cross-check the GPU kernel's control flow
(iteration / bulge-step counts) and as a LAPACK-independent second opinion on small matrices.  The
reference itself only calls ``np.linalg.eigvals`` (analysis/eval_eig.py:296)."""
from __future__ import annotations

import math

import numpy as np

ULP = np.finfo(np.float64).eps          # LAPACK dlamch('P') = eps*base = 2.22e-16
SAFMIN = np.finfo(np.float64).tiny


def hessenberg(a):
    """Unblocked Householder reduction (dgehd2).  Returns the Hessenberg matrix (lower part zeroed)."""
    h = np.array(a, dtype=np.float64)
    n = h.shape[0]
    for k in range(n - 2):
        alpha = h[k + 1, k]
        x = h[k + 2:, k]
        xnorm2 = float(x @ x)
        if xnorm2 == 0.0:
            continue
        beta = -math.copysign(math.sqrt(alpha * alpha + xnorm2), alpha)
        tau = (beta - alpha) / beta
        v = np.empty(n - k - 1)
        v[0] = 1.0
        v[1:] = x / (alpha - beta)
        h[k + 1, k] = beta
        h[k + 2:, k] = 0.0
        w = v @ h[k + 1:, k + 1:]
        h[k + 1:, k + 1:] -= tau * np.outer(v, w)
        w = h[:, k + 1:] @ v
        h[:, k + 1:] -= tau * np.outer(w, v)
    return h


def lanv2(a, b, c, d):
    """Eigenvalues of the real 2x2 block [[a,b],[c,d]] via the standardising rotation (dlanv2).
    Returns (rt1r, rt1i, rt2r, rt2i)."""
    multpl = 4.0
    eps = ULP
    if c == 0.0:
        pass
    elif b == 0.0:
        a, d = d, a
        b, c = -c, 0.0
    elif (a - d) == 0.0 and (b > 0) != (c > 0):
        pass
    else:
        temp = a - d
        p = 0.5 * temp
        bcmax = max(abs(b), abs(c))
        bcmis = min(abs(b), abs(c)) * math.copysign(1.0, b) * math.copysign(1.0, c)
        scale = max(abs(p), bcmax)
        z = (p / scale) * p + (bcmax / scale) * bcmis
        if z >= multpl * eps:
            z = p + math.copysign(math.sqrt(scale) * math.sqrt(z), p)
            a = d + z
            d = d - (bcmax / z) * bcmis
            b = b - c
            c = 0.0
        else:
            sigma = b + c
            tau = math.hypot(sigma, temp)
            cs = math.sqrt(0.5 * (1.0 + abs(sigma) / tau))
            sn = -(p / (tau * cs)) * math.copysign(1.0, sigma)
            aa = a * cs + b * sn
            bb = -a * sn + b * cs
            cc = c * cs + d * sn
            dd = -c * sn + d * cs
            a = aa * cs + cc * sn
            b = bb * cs + dd * sn
            c = -aa * sn + cc * cs
            d = -bb * sn + dd * cs
            temp = 0.5 * (a + d)
            a = temp
            d = temp
            if c != 0.0:
                if b != 0.0:
                    if (b > 0) == (c > 0):
                        sab = math.sqrt(abs(b))
                        sac = math.sqrt(abs(c))
                        p = math.copysign(sab * sac, c)
                        a = temp + p
                        d = temp - p
                        b = b - c
                        c = 0.0
                else:
                    b, c = -c, 0.0
    rt1r, rt2r = a, d
    if c == 0.0:
        rt1i = rt2i = 0.0
    else:
        rt1i = math.sqrt(abs(b)) * math.sqrt(abs(c))
        rt2i = -rt1i
    return rt1r, rt1i, rt2r, rt2i


def hqr_eigvals(hin, stats=None):
    """Eigenvalues of an upper-Hessenberg matrix by double-shift QR (dlahqr, wantt = wantz = False).
    Returns (w complex[n], info).  ``stats`` (dict) receives ``sweeps`` and ``steps`` counters."""
    h = np.array(hin, dtype=np.float64)
    n = h.shape[0]
    wr = np.zeros(n)
    wi = np.zeros(n)
    if n == 0:
        return wr + 1j * wi, 0
    smlnum = SAFMIN * (n / ULP)
    itmax = 30 * max(10, n)
    kexsh = 10
    sweeps = steps = 0
    i = n - 1
    kdefl = 0
    while i >= 0:
        l = 0
        converged = False
        for _its in range(itmax + 1):
            # small-subdiagonal search
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
            kdefl += 1
            if kdefl % (2 * kexsh) == 0:
                s = abs(h[i, i - 1]) + abs(h[i - 1, i - 2])
                h11 = 0.75 * s + h[i, i]
                h12 = -0.4375 * s
                h21 = s
                h22 = h11
            elif kdefl % kexsh == 0:
                s = abs(h[l + 1, l]) + abs(h[l + 2, l + 1])
                h11 = 0.75 * s + h[l, l]
                h12 = -0.4375 * s
                h21 = s
                h22 = h11
            else:
                h11, h21, h12, h22 = h[i - 1, i - 1], h[i, i - 1], h[i - 1, i], h[i, i]
            s = abs(h11) + abs(h12) + abs(h21) + abs(h22)
            if s == 0.0:
                rt1r = rt1i = rt2r = rt2i = 0.0
            else:
                h11 /= s
                h21 /= s
                h12 /= s
                h22 /= s
                tr = (h11 + h22) / 2.0
                det = (h11 - tr) * (h22 - tr) - h12 * h21
                rtdisc = math.sqrt(abs(det))
                if det >= 0.0:
                    rt1r = tr * s
                    rt2r = rt1r
                    rt1i = rtdisc * s
                    rt2i = -rt1i
                else:
                    rt1r = tr + rtdisc
                    rt2r = tr - rtdisc
                    if abs(rt1r - h22) <= abs(rt2r - h22):
                        rt1r = rt1r * s
                        rt2r = rt1r
                    else:
                        rt2r = rt2r * s
                        rt1r = rt2r
                    rt1i = rt2i = 0.0
            # two consecutive small subdiagonals
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
                h00 = abs(h[m - 1, m - 1])
                h11a = abs(h[m, m])
                h22a = abs(h[m + 1, m + 1])
                if abs(h[m, m - 1]) * (abs(v1) + abs(v2)) <= ULP * abs(v0) * (h00 + h11a + h22a):
                    break
                m -= 1
            sweeps += 1
            for k in range(m, i):
                steps += 1
                nr = min(3, i - k + 1)
                if k > m:
                    v0 = h[k, k - 1]
                    v1 = h[k + 1, k - 1]
                    v2 = h[k + 2, k - 1] if nr == 3 else 0.0
                # dlarfg on (v0; v1, v2)
                xnorm = math.hypot(v1, v2) if nr == 3 else abs(v1)
                if xnorm == 0.0:
                    t1 = 0.0
                    beta = v0
                else:
                    beta = -math.copysign(math.hypot(v0, xnorm), v0)
                    t1 = (beta - v0) / beta
                    sc = 1.0 / (v0 - beta)
                    v1 *= sc
                    v2 *= sc
                if k > m:
                    h[k, k - 1] = beta
                    h[k + 1, k - 1] = 0.0
                    if k < i - 1:
                        h[k + 2, k - 1] = 0.0
                elif m > l:
                    h[k, k - 1] *= (1.0 - t1)
                t2 = t1 * v1
                if nr == 3:
                    t3 = t1 * v2
                    sm = h[k, k:i + 1] + v1 * h[k + 1, k:i + 1] + v2 * h[k + 2, k:i + 1]
                    h[k, k:i + 1] -= sm * t1
                    h[k + 1, k:i + 1] -= sm * t2
                    h[k + 2, k:i + 1] -= sm * t3
                    hi = min(k + 3, i) + 1
                    sm = h[l:hi, k] + v1 * h[l:hi, k + 1] + v2 * h[l:hi, k + 2]
                    h[l:hi, k] -= sm * t1
                    h[l:hi, k + 1] -= sm * t2
                    h[l:hi, k + 2] -= sm * t3
                else:
                    sm = h[k, k:i + 1] + v1 * h[k + 1, k:i + 1]
                    h[k, k:i + 1] -= sm * t1
                    h[k + 1, k:i + 1] -= sm * t2
                    sm = h[l:i + 1, k] + v1 * h[l:i + 1, k + 1]
                    h[l:i + 1, k] -= sm * t1
                    h[l:i + 1, k + 1] -= sm * t2
        if not converged:
            if stats is not None:
                stats.update(sweeps=sweeps, steps=steps)
            return wr + 1j * wi, i + 1
        if l == i:
            wr[i] = h[i, i]
            wi[i] = 0.0
        else:
            wr[i - 1], wi[i - 1], wr[i], wi[i] = lanv2(h[i - 1, i - 1], h[i - 1, i], h[i, i - 1], h[i, i])
        kdefl = 0
        i = l - 1
    if stats is not None:
        stats.update(sweeps=sweeps, steps=steps)
    return wr + 1j * wi, 0


def eigvals(a, stats=None):
    return hqr_eigvals(hessenberg(a), stats)
