#!/usr/bin/env python3
"""Would aggressive early deflation pay in the n = 64 solver (csrc/eig_real_lds.hip)?  CPU model run (review r4 #1a).

oracle/hqr2_proto.py (the kernel's two-bulge iteration in NumPy) with an early-deflation step on the trailing w rows in
front of every sweep, on BOTH C2 matrix sets (bench.py's S5-style known-spectrum matrices, and N(0,1)/sqrt(n)).
Every variant's spectrum is checked against LAPACK.

Currency: one SUB-PASS of the kernel's steady-state iteration (3 LDS reads, 5 multiply-adds, a compare, 3 stores for
every lane; ~ 12 instructions, ~ 20 % of a three-pass iteration: profiles/r03_four_bulge_timing_mock.patch).
    iteration of the two-bulge sweep   3 sub-passes in windows of 60+ rows, 2 below (merged passes, round 4)
    per sweep                          14 (deflation scan, shifts, start search, lane records, the four special
                                       iterations' surplus: tools/shift44_model.py's figure)
An early-deflation call is priced with the same machinery, because that is what a wavefront would run:
    Schur form of the w x w window     the prototype itself on the window: every iteration 1 sub-pass for the window
                                       (R + C <= w + 6 <= 32 triples per bulge) + 1 for the accumulated vectors V,
                                       14 per sweep
    reordering                         2 sub-passes per adjacent swap (dtrexc: rotation on T and on V)
    spike reflector + re-reduction     2 sub-passes per undeflated row (only when something deflated)
    V applied to the rows above        w (w + 2) / 12 sub-passes (w multiply-adds + load + store per output column)
The "free" column prices the call at ZERO: the upper bound of what any implementation of the window could give.

    python tools/aed64_model.py [nmat]"""
import os
import sys

import numpy as np
import scipy.linalg
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.hqr2_proto as P
import bench

P.DIST = 3
nmat = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SWEEP_OVERHEAD = 14.0


def sweep_cost(log):
    return sum(nit * (3.0 if (i - l) >= 59 else 2.0) + SWEEP_OVERHEAD for (l, i, m, nit) in log)


def window_cost(win):
    """The window's own Schur form by the same two-bulge iteration: sub-pass units (window + vectors)."""
    st = {"log": []}
    P.hqr2_eigvals(win, shifts="same", stats=st, nb=2)
    return sum(nit * 2.0 + SWEEP_OVERHEAD for (_l, _i, _m, nit) in st["log"])


def aed_cost(st):
    c = 0.0
    for (l, i, jw, ns, nswap), win in zip(st.get("aed_log", []), st.get("aed_windows", [])):
        c += window_cost(win) + 2.0 * nswap
        if ns < jw:
            c += 2.0 * ns + jw * (jw + 2) / 12.0
    return c


def matrices(kind):
    if kind == "s5":
        gen = torch.Generator(device="cpu")
        gen.manual_seed(64)
        a, _ = bench.gen_known_spectrum(nmat, 64, gen, torch.device("cpu"))
        return [x for x in a.numpy()]
    rng = np.random.default_rng(5)
    return [rng.standard_normal((64, 64)) / 8.0 for _ in range(nmat)]


VARIANTS = [("kernel today (no window)", dict())]
for w in (2, 3, 4, 6, 8, 12, 16):
    VARIANTS.append((f"AED w={w}, kernel's shifts", dict(aed=w)))
    if w >= 6:
        VARIANTS.append((f"AED w={w}, window's shifts", dict(aed=w, aed_shifts=True)))
    if w in (6, 8):
        VARIANTS.append((f"no deflation, window's shifts w={w}", dict(aed=w, aed_shifts=True, aed_deflate=False)))

print("| set | variant | sweeps | iterations | window calls | sweep cost | + priced window | window free |")
print("|---|---|---|---|---|---|---|---|")
for kind in ("s5", "randn"):
    mats = matrices(kind)
    hs = [scipy.linalg.hessenberg(a) for a in mats]
    base = None
    for name, kw in VARIANTS:
        sw = it = calls = 0
        cs = ca = 0.0
        err = 0.0
        for a, h in zip(mats, hs):
            st = {"log": []}
            w, info = P.hqr2_eigvals(h, shifts="same", stats=st, nb=2, **kw)
            if info != 0:
                err = float("nan")
                break
            ref = np.linalg.eigvals(a)
            err = max(err, max(np.min(np.abs(ref - z)) for z in w), max(np.min(np.abs(w - z)) for z in ref))
            sw += st["sweeps"]
            it += st["iters"]
            calls += len(st.get("aed_log", []))
            cs += sweep_cost(st["log"])
            ca += aed_cost(st)
        if not err < 1e-9:
            print(f"| {kind} | {name} | did not converge / wrong (err {err}) | | | | | |", flush=True)
            continue
        if base is None:
            base = cs
        print(f"| {kind} | {name} | {sw / nmat:.0f} | {it / nmat:.0f} | {calls / nmat:.0f} | {cs / base:.3f} | "
              f"{(cs + ca) / base:.3f} | {cs / base:.3f} |", flush=True)
