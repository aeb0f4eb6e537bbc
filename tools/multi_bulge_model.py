#!/usr/bin/env python3
"""Iterations / sweeps of the n = 64 QR iteration with nb bulges per sweep (same double shift for all, or the two quadratic
factors of the trailing 4 x 4 block alternately), oracle/hqr2_proto.py.  CPU model run.
    python tools/multi_bulge_model.py [nmat]"""
import os, sys
import numpy as np
import scipy.linalg
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.hqr2_proto as P
P.DIST = 3
nmat = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(5)


def s5_like(n):
    # the benchmark's kind: prescribed spectrum inside the unit disc under a non-normal similarity
    lam = np.exp(-rng.uniform(0.001, 0.1, n // 2) + 1j * rng.uniform(0, np.pi, n // 2))
    d = np.zeros((n, n))
    for k, z in enumerate(lam):
        d[2 * k:2 * k + 2, 2 * k:2 * k + 2] = [[z.real, z.imag], [-z.imag, z.real]]
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    t = np.triu(rng.standard_normal((n, n)), 1) * 0.3
    s = q @ (np.eye(n) + t)
    return s @ d @ np.linalg.inv(s)


for kind in ("s5", "randn"):
    mats = [s5_like(64) if kind == "s5" else rng.standard_normal((64, 64)) / 8 for _ in range(nmat)]
    hs = [scipy.linalg.hessenberg(a) for a in mats]
    for shifts in ("same", "e44"):
        for nb in (1, 2, 3, 4):
            it = sw = 0
            err = 0.0
            for a, h in zip(mats, hs):
                st = {}
                w, info = P.hqr2_eigvals(h, shifts=shifts, stats=st, nb=nb)
                assert info == 0
                it += st["iters"]; sw += st["sweeps"]
                ref = np.linalg.eigvals(a)
                err = max(err, max(np.min(np.abs(ref - z)) for z in w))
            print(kind, shifts, "nb", nb, "iters/matrix", it // nmat, "sweeps", sw // nmat, "max err %.1e" % err, flush=True)
