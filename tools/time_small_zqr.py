#!/usr/bin/env python3
"""How long does ONE wavefront need for the eigenvalues of a complex Hessenberg block of order w (the early-deflation
window of the level-3 QR)?  768 such blocks, one wave each, through the existing n <= 64 complex path; prints ms per
call for several w.  (Eigenvalues only: a Schur form with vectors costs more -- this is the lower bound.)"""
import importlib, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
rng = np.random.default_rng(0)
for w in (8, 12, 16, 24, 32, 48):
    a = np.triu(rng.standard_normal((768, w, w)) + 1j * rng.standard_normal((768, w, w)), -1) / np.sqrt(w)
    t = torch.from_numpy(a).cuda()
    ops.eig_general_batched(t); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.eig_general_batched(t); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(json.dumps({"w": w, "ms_768_blocks": float(np.median(ts))}), flush=True)
