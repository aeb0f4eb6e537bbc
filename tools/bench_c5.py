#!/usr/bin/env python3
"""Config C5 alone (768 HiPPO-DPLR complex128 256x256 matrices; --n / --nmat change it): the command the rocprofv3
profiles of the level-3 eigenvalue path are taken on.  Prints one JSON line."""
import argparse, importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--nmat", type=int, default=768)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
n, h = a.n, a.nmat
lam, p, _, _, _ = O.make_dplr_hippo(n)
ls = np.random.default_rng(256).uniform(np.log(1e-3), np.log(1e-1), h)
d = lambda z: torch.from_numpy(np.ascontiguousarray(z)).cuda()
ab = ops.s4_dplr_build(d(np.tile(lam.real, (h, 1))), d(np.tile(lam.imag, (h, 1))), d(np.tile(p, (h, 1))), d(ls))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
hs, qr = [], []
for rep in range(a.reps + 1):
    ev[0].record(); hh, ww, ii = ops.hessenberg_general_batched(ab)
    ev[1].record(); ops.hseqr_general_batched(hh, ww, ii)
    ev[2].record(); ev[2].synchronize()
    if rep:
        hs.append(ev[0].elapsed_time(ev[1])); qr.append(ev[1].elapsed_time(ev[2]))
print(json.dumps({"n": n, "nmat": h, "hess_ms": float(np.median(hs)), "qr_ms": float(np.median(qr)),
                  "unconverged": int((ii != 0).sum().item())}))
