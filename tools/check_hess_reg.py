#!/usr/bin/env python3
"""Development check of the register-resident Hessenberg kernel (csrc/eig_real_reg.hip, n = 64) on one GPU: structure
and spectrum of H, eigenvalues through the whole solver against LAPACK, permuted / badly scaled inputs, and the stage
timing against the LDS-resident kernel behind EGS_HESS_LDS=1."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oracle as O
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
rng = np.random.default_rng(5)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
out = {}

a = rng.standard_normal((32, 64, 64))
h = ops.hessenberg_batched(dev(a)).cpu().numpy()
out["tril_zero"] = bool(np.all(np.tril(h, -2) == 0.0))
out["dlambda_h_vs_a"] = float(O.max_abs_dlambda(O.eigvals_dense(h), O.eigvals_dense(a)))
out["trace_err"] = float(np.max(np.abs(np.trace(h, axis1=1, axis2=2) - np.trace(a, axis1=1, axis2=2))))
out["fro_err"] = float(np.max(np.abs(np.linalg.norm(h, axis=(1, 2)) - np.linalg.norm(a, axis=(1, 2)))))
os.environ["EGS_HESS_LDS"] = "1"
h_old = ops.hessenberg_batched(dev(a)).cpu().numpy()
del os.environ["EGS_HESS_LDS"]
out["max_abs_h_minus_lds_kernel"] = float(np.max(np.abs(np.abs(h) - np.abs(h_old))))

w, info = ops.eig_general_batched(dev(a))
out["eig_info"] = int(info.abs().sum().item())
out["eig_dlambda"] = float(O.max_abs_dlambda(w.cpu().numpy(), np.linalg.eigvals(a)))

# isolated eigenvalues (permutation path), graded scaling, zero matrix, identity, tiny / huge entries
cases = []
b = rng.standard_normal((64, 64)); b[5, :] = 0.0; b[5, 5] = 2.0; cases.append(b)
b = rng.standard_normal((64, 64)); b[:, 9] = 0.0; b[9, 9] = -1.0; b[40, :] = 0.0; b[40, 40] = 0.5; cases.append(b)
b = np.triu(rng.standard_normal((64, 64))); cases.append(b)
b = np.tril(rng.standard_normal((64, 64))); cases.append(b)
d = 2.0 ** rng.integers(-20, 20, 64); cases.append(rng.standard_normal((64, 64)) * d[:, None] / d[None, :])
cases.append(np.zeros((64, 64))); cases.append(np.eye(64))
cases.append(rng.standard_normal((64, 64)) * 1e-200); cases.append(rng.standard_normal((64, 64)) * 1e200)
b = rng.standard_normal((64, 64)); b[32:, :32] = 0.0; cases.append(b)
c = np.stack(cases)
w, info = ops.eig_general_batched(dev(c))
w = w.cpu().numpy()
out["special_info"] = int(info.abs().sum().item())
ref = np.linalg.eigvals(c)
out["special_rel_dlambda"] = [float(O.max_abs_dlambda(w[m], ref[m]) / max(1e-300, np.max(np.abs(ref[m])))) for m in range(len(cases))]
os.environ["EGS_HESS_LDS"] = "1"
w2, _ = ops.eig_general_batched(dev(c))
del os.environ["EGS_HESS_LDS"]
w2 = w2.cpu().numpy()
out["special_rel_dlambda_lds_kernel"] = [float(O.max_abs_dlambda(w2[m], ref[m]) / max(1e-300, np.max(np.abs(ref[m])))) for m in range(len(cases))]
bad = rng.standard_normal((3, 64, 64)); bad[1, 3, 4] = np.nan; bad[2, 0, 0] = np.inf
w, info = ops.eig_general_batched(dev(bad))
out["nonfinite_info"] = info.cpu().tolist()

# stage timing, 65536 matrices
nmat = 65536
A = torch.randn((nmat, 64, 64), dtype=torch.float64, device="cuda")
for tag, env in (("reg", None), ("lds", "1")):
    if env: os.environ["EGS_HESS_LDS"] = env
    for _ in range(2): ops.hessenberg_batched(A)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ops.hessenberg_batched(A)
    torch.cuda.synchronize(); out["hess_ms_" + tag] = (time.perf_counter() - t0) / 5 * 1e3
    for _ in range(1): ops.eig_general_batched(A)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): ops.eig_general_batched(A)
    torch.cuda.synchronize(); out["eig_ms_" + tag] = (time.perf_counter() - t0) / 3 * 1e3
    if env: del os.environ["EGS_HESS_LDS"]
print(json.dumps(out, indent=1))
