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
EXTRA = bool(os.environ.get("STRESS_EXTRA"))     # more structures, more orders (tiny, and both sides of the n = 64 / level-3 switch)


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
    if not EXTRA:
        return
    u = lambda *shape: torch.rand(shape, generator=g, device="cuda", dtype=torch.float64)
    yield "wide-range 1e+-40", rnd() * 10.0 ** (80.0 * u(nmat, n, n) - 40.0)
    yield "row-col-scaled 1e+-60", rnd() * 10.0 ** (120.0 * u(nmat, n, 1) - 60.0) * 10.0 ** -(120.0 * u(nmat, 1, n) - 60.0)
    t = torch.triu(rnd()); t[:, idx, idx - 1] = 1e-18
    yield "triangular + 1e-18 sub-diagonal", t
    j = torch.zeros_like(s); j[:, idx - 1, idx] = 1.0
    j = j + torch.eye(n, device="cuda", dtype=s.dtype) * 0.7
    yield "one Jordan block", j
    q, _ = torch.linalg.qr(rnd())
    yield "rotated Jordan block", q @ j @ q.transpose(1, 2).conj()
    yield "denormal", rnd() * 1e-310
    yield "huge 1e300", rnd() * 1e300
    sp = rnd() * (u(nmat, n, n) < 0.1)
    yield "sparse 10%", sp
    perm = torch.stack([torch.eye(n, device="cuda", dtype=s.dtype)[torch.randperm(n, generator=g, device="cuda")] for _ in range(min(nmat, 64))])
    yield "permutation", perm
    red = torch.triu(rnd(), -1); red[:, idx[::3], idx[::3] - 1] = 0.0
    yield "reducible hessenberg", red
    yield "zero", torch.zeros_like(s)
    yield "identity", torch.eye(n, device="cuda", dtype=s.dtype).expand(min(nmat, 64), n, n).contiguous()
    blk = rnd()[:, :2, :2]
    rep = torch.zeros_like(s)
    for b0 in range(0, n - 1, 2):
        rep[:, b0:b0 + 2, b0:b0 + 2] = blk
    yield "repeated 2x2 blocks", rep


out = {}
CASES = ((32, False, 1024), (47, False, 512), (64, True, 512), (24, True, 512), (128, True, 48), (96, False, 48))
if EXTRA:
    CASES = ((64, False, 512), (2, False, 64), (3, True, 64), (5, False, 64), (16, True, 256)) + CASES + ((65, False, 32), (200, True, 24))
for n, cplx, nmat in CASES:
    for name, a in kinds(nmat, n, cplx):
        nmat_k = a.shape[0]
        w, info = ops.eig_general_batched(a.contiguous())
        amax = a.abs().amax(dim=(1, 2)).clamp_min(1e-320)            # Frobenius norm without squaring tiny entries
        nrm = (torch.linalg.matrix_norm(a / amax[:, None, None]) * amax).clamp_min(1e-320)
        fin = bool(torch.isfinite(torch.view_as_real(w)).all())
        rad = float((w.abs().max(dim=1).values / nrm).max().item()) if fin else float("inf")
        tr = float(((w.sum(dim=1) - a.diagonal(dim1=1, dim2=2).sum(dim=1)).abs() / nrm).max().item()) if fin else float("inf")
        k = min(nmat_k, 32)
        ref = np.linalg.eigvals(a[:k].cpu().numpy())
        dl = float(O.max_abs_dlambda(w[:k].cpu().numpy(), ref) / float(nrm[:k].max().item())) if fin else float("inf")
        key = "n=%d %s %s" % (n, "complex" if cplx else "real", name)
        out[key] = {"unconverged": int((info != 0).sum().item()), "finite": fin, "max_radius_over_norm": rad, "trace_err": tr, "dlambda_over_norm_sample": dl}
        flag = "" if (fin and rad < 1 + 1e-9 and tr < 1e-11 and out[key]["unconverged"] == 0) else "   <<<<<< CHECK"
        print(key, out[key], flag, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "stress_paths.json"), "w"), indent=1)
