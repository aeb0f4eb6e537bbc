#!/usr/bin/env python3
"""Stress of the OTHER solver paths (real n < 64, complex n <= 64, level-3 n > 64) on structures that broke the n = 64
real path in round 3 (zero diagonals, skew / skew-Hermitian, companion, graded): |lambda| bounded by the norm, finite,
trace identities, and agreement with LAPACK on a sample."""
import importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import oracle as O
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
g = torch.Generator(device="cuda"); g.manual_seed(11)


def kinds(nmat, n, cplx):
    def rnd():
        x = torch.randn((nmat, n, n), generator=g, device="cuda", dtype=torch.float64)
        if cplx:
            x = torch.complex(x, torch.randn((nmat, n, n), generator=g, device="cuda", dtype=torch.float64))
        return x / (2 * np.sqrt(n))
    s = rnd()
    yield "skew", s - s.transpose(1, 2).conj()
    z = rnd(); z.diagonal(dim1=1, dim2=2).zero_()
    yield "zero-diagonal", z
    c = torch.zeros_like(s); idx = torch.arange(1, n, device="cuda")
    c[:, idx, idx - 1] = 1.0; c[:, 0, :] = rnd()[:, 0, :]
    yield "companion", c
    gr = torch.triu(rnd(), -1) * (10.0 ** (-torch.rand((nmat, n, 1), generator=g, device="cuda", dtype=torch.float64) * 8))
    yield "graded-hessenberg", gr
    yield "skew*1e-160", (s - s.transpose(1, 2).conj()) * 1e-160
    yield "constant", torch.ones_like(s) * (0.5 + torch.rand((nmat, 1, 1), generator=g, device="cuda", dtype=torch.float64))
    r1 = rnd()
    yield "rank-one", r1[:, :, :1] * r1[:, :1, :]


out = {}
for n, cplx, nmat in ((32, False, 1024), (47, False, 512), (64, True, 512), (24, True, 512), (128, True, 48), (96, False, 48)):
    for name, a in kinds(nmat, n, cplx):
        w, info = ops.eig_general_batched(a.contiguous())
        nrm = torch.linalg.matrix_norm(a)
        fin = bool(torch.isfinite(torch.view_as_real(w)).all())
        rad = float((w.abs().max(dim=1).values / nrm).max().item()) if fin else float("inf")
        tr = float(((w.sum(dim=1) - a.diagonal(dim1=1, dim2=2).sum(dim=1)).abs() / nrm).max().item()) if fin else float("inf")
        k = min(nmat, 32)
        ref = np.linalg.eigvals(a[:k].cpu().numpy())
        dl = float(O.max_abs_dlambda(w[:k].cpu().numpy(), ref) / float(nrm[:k].max().item())) if fin else float("inf")
        key = "n=%d %s %s" % (n, "complex" if cplx else "real", name)
        out[key] = {"unconverged": int((info != 0).sum().item()), "finite": fin, "max_radius_over_norm": rad, "trace_err": tr, "dlambda_over_norm_sample": dl}
        flag = "" if (fin and rad < 1 + 1e-9 and tr < 1e-11 and out[key]["unconverged"] == 0) else "   <<<<<< CHECK"
        print(key, out[key], flag, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stress_paths.json"), "w"), indent=1)
