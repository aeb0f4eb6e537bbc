#!/usr/bin/env python3
"""Development check of the two-bulge sweep (EGS_QR_PAIR=1) in the fused n = 64 kernel: spectra against LAPACK and
against the single-bulge kernel, timings on the bench's matrices and on Gaussian matrices."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle as O
from test_gpu_parity import known_spectrum_real
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
rng = np.random.default_rng(8)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
out = {}
a = rng.standard_normal((64, 64, 64))
ak, lam = known_spectrum_real(rng, 2048, 64)
for tag, env in (("single", None), ("pair", "1")):
    if env: os.environ["EGS_QR_PAIR"] = env
    w, info = ops.eig_general_batched(dev(a))
    out["info_" + tag] = int(info.abs().sum().item())
    out["dlambda_randn_" + tag] = float(O.max_abs_dlambda(w.cpu().numpy(), np.linalg.eigvals(a)))
    w, info = ops.eig_general_batched(dev(ak))
    out["info_known_" + tag] = int(info.abs().sum().item())
    out["dlambda_known_" + tag] = float(O.max_abs_dlambda(w.cpu().numpy(), lam))
    for kind in ("s5", "randn"):
        nmat = 131072
        if kind == "randn":
            A = torch.randn((nmat, 64, 64), dtype=torch.float64, device="cuda")
        else:
            A = dev(np.tile(ak, (nmat // ak.shape[0], 1, 1)))
        ops.eig_general_batched(A)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): w, info = ops.eig_general_batched(A)
        torch.cuda.synchronize(); out[f"ms_{kind}_{tag}"] = (time.perf_counter() - t0) / 3 * 1e3
        out[f"info_{kind}_{tag}"] = int(info.abs().sum().item())
    if env: del os.environ["EGS_QR_PAIR"]
print(json.dumps(out, indent=1))
