#!/usr/bin/env python3
"""One-off stress of the n = 64 solver: many matrices of mixed structure against LAPACK on the host (per-matrix optimal
matching of the spectra).  Prints the worst matrices per kind.   NPER=2048 python tools/stress_c2.py"""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oracle as O
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
nper = int(os.environ.get("NPER", 2048)); n = 64
rng = np.random.default_rng(int(os.environ.get("SEED", 7)))
kinds = {}
a = rng.standard_normal((nper, n, n)) / 8
kinds["gaussian"] = a
kinds["hessenberg_graded"] = np.triu(rng.standard_normal((nper, n, n)), -1) * (10.0 ** rng.uniform(-6, 0, (nper, n, 1)))
s = rng.standard_normal((nper, n, n)); kinds["symmetric"] = (s + s.transpose(0, 2, 1)) / 16
kinds["skew"] = (s - s.transpose(0, 2, 1)) / 16
b = rng.standard_normal((nper, n, n)) / 8; b[:, n // 2:, : n // 2] = 0; kinds["block_triangular"] = b
kinds["sparse"] = rng.standard_normal((nper, n, n)) * (rng.random((nper, n, n)) < 0.15)
sh = np.zeros((nper, n, n)); sh[:, np.arange(1, n), np.arange(n - 1)] = 1; sh[:, 0, :] = rng.standard_normal((nper, n)) * 0.3
kinds["companion"] = sh
out = {}
for name, m in kinds.items():
    w, info = ops.eig_general_batched(torch.from_numpy(np.ascontiguousarray(m)).cuda())
    w = w.cpu().numpy(); bad = int((info != 0).sum().item())
    t0 = time.time(); ref = np.linalg.eigvals(m); t1 = time.time()
    nrm = np.linalg.norm(m, axis=(1, 2))
    # cheap symmetric-function checks on everything, optimal matching on a sample
    tr = np.max(np.abs(w.sum(1) - ref.sum(1)) / nrm)
    d = max(O.max_abs_dlambda(w[k:k + 1], ref[k:k + 1]) / max(nrm[k], 1e-300) for k in range(0, nper, max(1, nper // 256)))
    out[name] = {"unconverged": bad, "max_trace_mismatch_rel": float(tr), "max_dlambda_over_norm_sample": float(d), "lapack_s": round(t1 - t0, 1)}
    print(name, out[name], flush=True)
print(json.dumps(out))
