#!/usr/bin/env python3
"""What would a second, different double shift for the trailing bulge buy in the long windows of the n = 64 iteration?
oracle/hqr2_proto.py with shifts="same" (the kernel's), "e44" (the two quadratic factors of the trailing 4 x 4 block in
every sweep) and "e44hi" (only in windows longer than a threshold).  Iterations are weighted with the kernel's pass
counts (3 sub-passes per iteration in windows of 60+ rows, 2 below: profiles/r04_variants.md) and every sweep with an
overhead in units of one sub-pass; the 4 x 4 shifts are charged EXTRA sub-pass units per sweep that uses them.
CPU model run.
    python tools/shift44_model.py [nmat]"""
import os, sys
import numpy as np
import scipy.linalg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.hqr2_proto as P

P.DIST = 3
nmat = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(5)
SWEEP_OVERHEAD = 14.0      # scan + shifts + start search + lane records: ~ 500 instructions ~ 14 sub-passes of ~ 35
EXTRA = 6.0                # ~ 200 uniform instructions for charpoly + Bairstow


def s5_like(n):
    lam = np.exp(-rng.uniform(0.001, 0.1, n // 2) + 1j * rng.uniform(0, np.pi, n // 2))
    d = np.zeros((n, n))
    for k, z in enumerate(lam):
        d[2 * k:2 * k + 2, 2 * k:2 * k + 2] = [[z.real, z.imag], [-z.imag, z.real]]
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    t = np.triu(rng.standard_normal((n, n)), 1) * 0.3
    s = q @ (np.eye(n) + t)
    return s @ d @ np.linalg.inv(s)


def cost(log, uses44):
    c = 0.0
    for (l, i, m, nit) in log:
        w = i - l
        c += nit * (3.0 if w >= 59 else 2.0) + SWEEP_OVERHEAD + (EXTRA if uses44(l, i) else 0.0)
    return c


for kind in ("s5", "randn"):
    mats = [s5_like(64) if kind == "s5" else rng.standard_normal((64, 64)) / 8 for _ in range(nmat)]
    hs = [scipy.linalg.hessenberg(a) for a in mats]
    base = None
    for shifts, thr in (("same", 0), ("e44", 0), ("e44hi", 58), ("e44hi", 50), ("e44hi", 43), ("e44hi", 31)):
        P.E44_MIN_WINDOW = thr
        it = sw = 0
        c = 0.0
        err = 0.0
        long_it = 0
        for a, h in zip(mats, hs):
            st = {"log": []}
            w, info = P.hqr2_eigvals(h, shifts=shifts, stats=st, nb=2)
            assert info == 0
            it += st["iters"]; sw += st["sweeps"]
            long_it += sum(r[3] for r in st["log"] if r[1] - r[0] >= 59)
            use = (lambda l, i: False) if shifts == "same" else (lambda l, i: i - l + 1 >= 6) if shifts == "e44" \
                else (lambda l, i, thr=thr: i - l >= thr)
            c += cost(st["log"], use)
            ref = np.linalg.eigvals(a)
            err = max(err, max(np.min(np.abs(ref - z)) for z in w))
        if base is None:
            base = c
        print(f"{kind:6s} {shifts:6s} thr {thr:2d}  iters/matrix {it // nmat:5d} (in windows >= 60 rows: {long_it // nmat:4d})"
              f"  sweeps {sw // nmat:3d}  cost {c / base:.3f}  max err {err:.1e}", flush=True)
