#!/usr/bin/env python3
"""In-kernel phase cycles of the multishift QR on config C5 (mean per matrix, Mcycles): scan, shifts / small blocks,
window load + store, chase, deferred updates, single-window chase; sweeps, stages, chain steps of the steady body; and the
chase as wave 0 sees it (row phase, barrier, column phase, barrier).  EGS_ZCHASE_GENERAL=1: general chase body only."""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle as O
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
n, h = 256, 768
lam, p, _, _, _ = O.make_dplr_hippo(n)
ls = np.random.default_rng(256).uniform(np.log(1e-3), np.log(1e-1), h)
d = lambda z: torch.from_numpy(np.ascontiguousarray(z)).cuda()
ab = ops.s4_dplr_build(d(np.tile(lam.real, (h, 1))), d(np.tile(lam.imag, (h, 1))), d(np.tile(p, (h, 1))), d(ls))
hh, ww, ii = ops.hessenberg_general_batched(ab)
w, info, prof = ops.hseqr_general_batched(hh, ww, ii, profile=True)
m = prof.cpu().numpy().mean(axis=0)
names = ["scan", "shifts+small", "window ld/st", "chase", "updates", "single-window chase"]
out = {k: round(float(v) / 1e6, 2) for k, v in zip(names, m[:6])}
out.update(sweeps=round(float(m[6]), 1), stages=int(m[7]), steady_steps=round((float(m[7]) % 1) * 1e6),
           chase_wave0={"row": round(m[8] / 1e6, 2), "barrier1": round(m[9] / 1e6, 2), "col": round(m[10] / 1e6, 2),
                        "barrier2": round(m[11] / 1e6, 2)}, general_only=bool(os.environ.get("EGS_ZCHASE_GENERAL")),
           unconverged=int((info != 0).sum().item()))
print(json.dumps(out))
