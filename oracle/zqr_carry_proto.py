"""Prototype (NumPy, test infrastructure only) of the REGISTER-CARRIED single-shift QR sweep for the
workgroup-per-matrix complex solver (n >= 128) -- DESIGN.md section 4.  (The HIP version built from it was correct but
slower than the plain sweep and was not adopted; the prototype stays as the executable description of the data flow.)

The kernel of this round (csrc/eig_cplx.hip, phase 2) reads and writes two rows and two columns of the active window
in global memory at every bulge step, i.e. every row is moved twice (as the lower row of step k-1 and the upper row of
step k), and so is every column.  The carried form keeps the second row / column of a step in the owning thread's
registers for the next step, loads only the genuinely new row / column (which the current step does not touch, so it
can be prefetched a step ahead) and exchanges the handful of elements around the bulge through small LDS slots:

    BULGE   (H(k+1,k), H(k+2,k))      written by the column phase of step k, read by every thread at step k+1
    DIAG    (H(k+1,k+1), H(k+2,k+1))  written by the column phase of step k, read by the row phase of step k+1 (j = k+1)
    ROWK    (H(k,k), H(k,k+1)), ROWK1 (H(k+1,k), H(k+1,k+1))   row phase -> column phase of the same step (r = k, k+1)
    NEXTCB  H(k,k+2)                  row phase of step k (j = k+2) -> column phase of step k+1 (r = k), double-buffered

`sweep_carried` below executes exactly that data flow on explicit "global" (the matrix), "register" (per-index carry
arrays) and "LDS" (the slots) state, counting global element loads / stores, and `sweep_plain` is the step-by-step form
the kernel implements today; tests/test_oracle_golden.py checks that both give the same matrix and that the carried form
moves about half the elements.  Nothing here is imported by the product.
"""
import numpy as np


def reflector2(a0, a1):
    """2-element Householder reflector as csrc/eig_cplx.hip::zlarfg2: returns (beta, v2, tau)."""
    xnorm = abs(a1)
    if xnorm == 0.0 and a0.imag == 0.0:
        return a0, 0.0 + 0.0j, 0.0 + 0.0j
    beta = -np.copysign(np.sqrt(a0.real ** 2 + a0.imag ** 2 + xnorm ** 2), a0.real)
    tau = complex((beta - a0.real) / beta, -a0.imag / beta)
    v2 = a1 / (a0 - beta)
    return complex(beta), v2, tau


def _row_pair(ra, rb, tau, v2):
    s = np.conj(tau) * (ra + np.conj(v2) * rb)
    return ra - s, rb - s * v2


def _col_pair(ca, cb, tau, v2):
    s = tau * (ca + cb * v2)
    return ca - s, cb - s * np.conj(v2)


def start_vector(h, m, t):
    h11s, h21 = h[m, m] - t, h[m + 1, m]
    s = abs(h11s.real) + abs(h11s.imag) + abs(h21.real) + abs(h21.imag)
    return h11s / s, h21 / s


def sweep_plain(h, l, i, m, t):
    """One single-shift sweep from row m to i on the window [l, i], as the kernel does it today.  Returns (H, traffic)
    with traffic = global elements loaded + stored by the row / column updates."""
    h = h.copy()
    traffic = 0
    v0, v1 = start_vector(h, m, t)
    for k in range(m, i):
        a0, a1 = (h[k, k - 1], h[k + 1, k - 1]) if k > m else (v0, v1)
        beta, v2, tau = reflector2(complex(a0), complex(a1))
        ra, rb = _row_pair(h[k, k:i + 1].copy(), h[k + 1, k:i + 1].copy(), tau, v2)
        h[k, k:i + 1], h[k + 1, k:i + 1] = ra, rb
        traffic += 4 * (i + 1 - k)
        if k > m:
            h[k, k - 1], h[k + 1, k - 1] = beta, 0.0
        elif m > l:
            h[k, k - 1] *= 1.0 - np.conj(tau)
        rmax = min(k + 2, i)
        ca, cb = _col_pair(h[l:rmax + 1, k].copy(), h[l:rmax + 1, k + 1].copy(), tau, v2)
        h[l:rmax + 1, k], h[l:rmax + 1, k + 1] = ca, cb
        traffic += 4 * (rmax + 1 - l)
    return h, traffic


def sweep_carried(h, l, i, m, t):
    """The same sweep with the second row / column of every step carried in registers.  Global memory (`g`) is only
    touched through `load` / `store`; everything else is registers (RA, CA, RB, CB: one entry per column / row index,
    i.e. per owning thread) or the LDS slots named in the module docstring."""
    g = h.copy()
    n = g.shape[0]
    traffic = [0]

    def load(r, c):
        traffic[0] += 1
        return complex(g[r, c])

    def store(r, c, v):
        traffic[0] += 1
        g[r, c] = v

    v0, v1 = start_vector(g, m, t)
    # sweep start: everything comes from global once
    RA = {j: load(m, j) for j in range(m, i + 1)}                    # row k = m
    CA = {r: load(r, m) for r in range(l, m)}                         # column k = m, rows above the bulge block
    RB = {j: load(m + 1, j) for j in range(m, i + 1)}                 # row k+1 (prefetched one step ahead later on)
    CB = {r: load(r, m + 1) for r in range(l, m)}                     # column k+1
    lds = {"BULGE": None, "DIAG": (RA[m], RB[m]), "NEXTCB": [None, None]}
    sub = load(m + 2, m + 1) if m + 2 <= i else 0.0                   # H(k+2, k+1): the subdiagonal entering the block
    for k in range(m, i):
        par = (k - m) & 1
        a0, a1 = lds["BULGE"] if k > m else (v0, v1)
        beta, v2, tau = reflector2(complex(a0), complex(a1))
        # ---- row phase: rows k, k+1, columns k..i ----
        new_ra, new_rb = {}, {}
        for j in range(k, i + 1):
            if j == k:
                ra, rb = lds["DIAG"]                                  # (H(k,k), H(k+1,k)) from the previous column phase
            else:
                ra, rb = RA[j], RB[j]
            new_ra[j], new_rb[j] = _row_pair(ra, rb, tau, v2)
        for j in range(k + 2, i + 1):
            store(k, j, new_ra[j])                                    # row k is final for this sweep
        lds["ROWK"] = (new_ra[k], new_ra.get(k + 1))
        lds["ROWK1"] = (new_rb[k], new_rb.get(k + 1))
        if k + 2 <= i:
            lds["NEXTCB"][par] = new_ra[k + 2]                        # H(k, k+2) for thread r = k at the next step
        # prefetch for step k+1 (untouched by step k): row k+2, columns >= k+2; column k+2, rows < k
        RBn = {j: load(k + 2, j) for j in range(k + 2, i + 1)} if k + 2 <= i else {}
        CBn = {r: load(r, k + 2) for r in range(l, k)} if k + 2 <= i else {}
        sub_next = load(k + 3, k + 2) if k + 3 <= i else 0.0
        # ---- bookkeeping of the annihilated bulge (thread 0) ----
        if k > m:
            store(k, k - 1, beta)
            store(k + 1, k - 1, 0.0)
        elif m > l:
            store(k, k - 1, load(k, k - 1) * (1.0 - np.conj(tau)))
        # ---- column phase: columns k, k+1, rows l..min(k+2, i) ----
        rmax = min(k + 2, i)
        new_ca, new_cb = {}, {}
        for r in range(l, rmax + 1):
            if r == k:
                ca, cb = lds["ROWK"]
            elif r == k + 1:
                ca, cb = lds["ROWK1"]
            elif r == k + 2:
                ca, cb = 0.0 + 0.0j, sub                              # H(k+2,k) = 0 before the step; the subdiagonal
            elif r == k - 1 and k > m:
                ca, cb = CA[r], lds["NEXTCB"][par ^ 1]                # H(k-1,k+1) was rewritten by the previous row phase
            else:
                ca, cb = CA[r], CB[r]
            new_ca[r], new_cb[r] = _col_pair(ca, cb, tau, v2)
        for r in range(l, min(k, rmax) + 1):
            store(r, k, new_ca[r])                                    # column k is final (rows <= k)
        if k + 1 <= rmax:
            lds["BULGE"] = (new_ca[k + 1], new_ca[k + 2] if k + 2 <= rmax else 0.0 + 0.0j)
            lds["DIAG"] = (new_cb[k + 1], new_cb[k + 2] if k + 2 <= rmax else 0.0 + 0.0j)
        # ---- carries for step k+1 ----
        RA = {j: new_rb[j] for j in range(k + 2, i + 1)}              # row k+1, columns beyond the block
        RA[k + 1] = None                                              # column k+1 takes DIAG at the next step
        RB = RBn
        CA = {r: new_cb[r] for r in range(l, k + 1)}                  # column k+1, rows <= k
        CB = CBn
        sub = sub_next
        if k + 1 == i:                                                # last step: flush what is still in flight
            store(i, i - 1, lds["BULGE"][0])                          # H(i, i-1)
            store(i, i, lds["DIAG"][0])                               # H(i, i)
            for r in range(l, i):
                store(r, i, CA[r])                                    # column i, rows above the diagonal
    return g, traffic[0]
