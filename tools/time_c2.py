#!/usr/bin/env python3
"""Times the fused n = 64 eigenvalue kernel on the benchmark's own matrices (bench.py's S5-style generator) or on
Gaussian ones, at a reduced batch.  One JSON line: ms per launch, spectra/s, parity on a sample.
    NMAT=131072 KIND=s5|randn REPS=5 python tools/time_c2.py"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import oracle as O
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
nmat = int(os.environ.get("NMAT", 131072)); kind = os.environ.get("KIND", "s5"); reps = int(os.environ.get("REPS", 5))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(64)
a = torch.empty((nmat, 64, 64), dtype=torch.float64, device=dev)
lam0 = None
for m0 in range(0, nmat, 16384):
    cnt = min(16384, nmat - m0)
    if kind == "s5":
        blk, lam = bench.gen_known_spectrum(cnt, 64, gen, dev)
        if m0 == 0: lam0 = lam[:256].cpu().numpy()
    else:
        blk = torch.randn((cnt, 64, 64), generator=gen, device=dev, dtype=torch.float64) / 8
    a[m0:m0 + cnt] = blk
w, info = ops.eig_general_batched(a)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
ev[0].record()
for r in range(reps):
    w, info = ops.eig_general_batched(a)
    ev[r + 1].record()
torch.cuda.synchronize()
ms = [ev[r].elapsed_time(ev[r + 1]) for r in range(reps)]
out = {"kind": kind, "nmat": nmat, "ms": float(np.median(ms)), "ms_all": ms, "spectra_per_s": nmat / (np.median(ms) * 1e-3),
       "unconverged": int((info != 0).sum().item()), "lib": os.environ.get("EGS_LIB_PATH", "tree")}
ws = w[:256].cpu().numpy()
out["dlambda_lapack"] = float(O.max_abs_dlambda(ws, np.linalg.eigvals(a[:256].cpu().numpy())))
if lam0 is not None: out["dlambda_known"] = float(O.max_abs_dlambda(ws, lam0))
print(json.dumps(out))
