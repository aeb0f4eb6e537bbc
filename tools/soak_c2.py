#!/usr/bin/env python3
"""Soak of the n = 64 real solver: many full-size batches of random matrices of several kinds; every spectrum must be
flagged converged, sum to the trace and lie inside the Frobenius ball.  One line per kind.
    BATCHES=8 python tools/soak_c2.py
CMP_NO_MERGE=1 (n = 64): every batch is also solved with the three-pass sweep (EGS_QR2_NO_MERGE=1) and the two spectra
must agree bit for bit (round 4: merged two-pass iterations)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
nb = int(os.environ.get("BATCHES", 8)); nmat = int(os.environ.get("NMAT", 262144))
N = int(os.environ.get("N", 64)); CPLX = bool(os.environ.get("CPLX"))        # other orders / complex input: the other solver paths
g = torch.Generator(device="cuda"); g.manual_seed(int(os.environ.get("SEED", 1)))
def r(*s):
    x = torch.randn(s, generator=g, device="cuda", dtype=torch.float64)
    return torch.complex(x, torch.randn(s, generator=g, device="cuda", dtype=torch.float64)) if CPLX else x

kinds = {
    "gaussian": lambda: r(nmat, N, N) / 8,
    "skew + 1e-3 gaussian": lambda: (lambda x: (x - x.transpose(1, 2).conj()) / 8 + 1e-3 * r(nmat, N, N))(r(nmat, N, N)),
    "symmetric": lambda: (lambda x: (x + x.transpose(1, 2).conj()) / 16)(r(nmat, N, N)),
    "rank-4 + 1e-8 gaussian": lambda: r(nmat, N, 4) @ r(nmat, 4, N) / 8 + 1e-8 * r(nmat, N, N),
    "row-scaled 1e+-6": lambda: r(nmat, N, N) / 8 * 10.0 ** (12 * torch.rand((nmat, N, 1), generator=g, device="cuda", dtype=torch.float64) - 6),
    "upper hessenberg": lambda: torch.triu(r(nmat, N, N), -1) / 8,
}
for name, make in kinds.items():
    bad = 0; tr = 0.0; rad = 0.0; t0 = time.time()
    for b in range(nb):
        a = make()
        w, info = ops.eig_general_batched(a)
        if os.environ.get("CMP_NO_MERGE"):
            os.environ["EGS_QR2_NO_MERGE"] = "1"
            w3, info3 = ops.eig_general_batched(a)
            del os.environ["EGS_QR2_NO_MERGE"]
            if not (torch.equal(torch.view_as_real(w), torch.view_as_real(w3)) and torch.equal(info, info3)):
                bad += int((torch.view_as_real(w) != torch.view_as_real(w3)).any(dim=-1).any(dim=-1).sum().item()) + 1
            del w3, info3
        bad += int((info != 0).sum().item())
        fro = torch.linalg.matrix_norm(a)
        tr = max(tr, float(((w.sum(dim=1) - a.diagonal(dim1=1, dim2=2).sum(dim=1)).abs() / fro).max().item()))
        rad = max(rad, float((w.abs().amax(dim=1) / fro).max().item()))
        nanc = int(torch.isnan(w.real).any(dim=1).sum().item())
        bad += nanc
        del a, w, info
    flag = "" if (bad == 0 and tr < 1e-10 and rad <= 1 + 1e-12) else "   <<<<<< CHECK"
    print(json.dumps({"kind": name, "n": N, "complex": CPLX, "matrices": nb * nmat, "unconverged_or_nan": bad, "max_trace_err_over_fro": tr,
                      "max_radius_over_fro": rad, "s": round(time.time() - t0, 1)}), flag, flush=True)
