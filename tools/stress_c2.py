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
# further structures (round 3, after the skew-symmetric failure): exact zeros, defective and rank-deficient matrices
idx = np.arange(n)
kinds["nilpotent_strict_upper"] = np.triu(rng.standard_normal((nper, n, n)), 1)
kinds["rank_one"] = rng.standard_normal((nper, n, 1)) * rng.standard_normal((nper, 1, n)) / n
kinds["identity_plus_tiny"] = np.eye(n)[None] + 1e-12 * rng.standard_normal((nper, n, n))
kinds["ones"] = np.ones((nper, n, n)) * rng.uniform(0.5, 2, (nper, 1, 1))
kinds["zero_matrix"] = np.zeros((nper, n, n))
pm = np.zeros((nper, n, n)); perm = np.array([rng.permutation(n) for _ in range(nper)]); pm[np.arange(nper)[:, None], idx[None, :], perm] = 1.0
kinds["permutation"] = pm
tz = np.zeros((nper, n, n)); c = rng.standard_normal((nper, 2 * n - 1))
for d in range(-n + 1, n):
    tz[:, idx[max(0, -d):n - max(0, d)], idx[max(0, -d):n - max(0, d)] + d] = c[:, d + n - 1][:, None]
kinds["toeplitz"] = tz / 8
kinds["row_scaled_1e10"] = (rng.standard_normal((nper, n, n)) / 8) * (10.0 ** rng.uniform(-10, 10, (nper, n, 1)))
q = np.linalg.qr(rng.standard_normal((nper, n, n)))[0]
kinds["orthogonal"] = q
jb = np.zeros((nper, n, n)); jb[:, idx, idx] = np.repeat(rng.standard_normal((nper, n // 8)), 8, axis=1); jb[:, idx[:-1], idx[:-1] + 1] = (idx[:-1] % 8 != 7)
kinds["jordan8_rotated"] = q @ jb @ q.transpose(0, 2, 1)
out = {}
for name, m in kinds.items():
    w, info = ops.eig_general_batched(torch.from_numpy(np.ascontiguousarray(m)).cuda())
    w = w.cpu().numpy(); bad = int((info != 0).sum().item())
    t0 = time.time(); ref = np.linalg.eigvals(m); t1 = time.time()
    nrm = np.linalg.norm(m, axis=(1, 2))
    nrm_safe = np.maximum(nrm, 1e-300)
    # cheap symmetric-function checks on everything, optimal matching on a sample
    tr = np.max(np.abs(w.sum(1) - np.trace(m, axis1=1, axis2=2)) / nrm_safe)
    d = max(O.max_abs_dlambda(w[k:k + 1], ref[k:k + 1]) / max(nrm[k], 1e-300) for k in range(0, nper, max(1, nper // 256)))
    be = 0.0
    for k in range(0, nper, max(1, nper // 32)):
        if nrm[k] == 0.0 or not np.isfinite(w[k]).all():
            continue
        lam = w[k, ::9]
        sm = np.linalg.svd(m[k][None] - lam[:, None, None] * np.eye(n)[None], compute_uv=False)[:, -1]
        be = max(be, float(np.max(sm) / nrm[k]))
    fin = bool(np.isfinite(w).all())
    rad = float(np.max(np.abs(w).max(axis=1) / np.maximum(nrm, 1e-300))) if fin else float("inf")
    out[name] = {"unconverged": bad, "finite": fin, "max_radius_over_fro": rad, "max_trace_mismatch_rel": float(tr),
                 "max_dlambda_over_norm_sample": float(d), "backward_error_sample": be, "lapack_s": round(t1 - t0, 1)}
    nbadfin = int((~np.isfinite(w).all(axis=1)).sum())
    out[name]["matrices_with_nonfinite_spectrum"] = nbadfin
    print(name, out[name], "   <<<<<< CHECK" if (bad or nbadfin or not (rad < 1 + 1e-9) or be > 1e-12) else "", flush=True)
print(json.dumps(out))
