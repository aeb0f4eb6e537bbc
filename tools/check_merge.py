#!/usr/bin/env python3
"""Bit-identity of the merged two-pass n = 64 sweep (round 4) with the three-pass one (EGS_QR2_NO_MERGE=1 at launch): every
element receives the same operations in the same order, so spectra and info must agree to the last bit.
    python tools/check_merge.py dump out.pt     (EGS_LIB_PATH / EGS_QR2_NO_MERGE select library and form)
    python tools/check_merge.py cmp a.pt b.pt"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def batch():
    g = torch.Generator(device="cuda"); g.manual_seed(2026)
    import bench
    parts = [torch.randn((6000, 64, 64), generator=g, device="cuda", dtype=torch.float64) / 8,
             bench.gen_known_spectrum(6000, 64, g, torch.device("cuda", 0))[0]]
    tri = torch.triu(torch.randn((2000, 64, 64), generator=g, device="cuda", dtype=torch.float64), -1)   # Hessenberg already
    blk = torch.randn((2000, 64, 64), generator=g, device="cuda", dtype=torch.float64) / 8
    blk[:, 40:, :40] = 0.0                                                                                   # reducible: sweeps below the top
    sk = torch.randn((2000, 64, 64), generator=g, device="cuda", dtype=torch.float64)
    return torch.cat(parts + [tri, blk, sk - sk.transpose(1, 2)])


if sys.argv[1] == "dump":
    ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
    a = batch()
    w, info = ops.eig_general_batched(a)
    os.environ["EGS_EIG_ONE_PHASE"] = "1"
    w1, info1 = ops.eig_general_batched(a)
    torch.save({"w": w.cpu(), "info": info.cpu(), "w1": w1.cpu(), "info1": info1.cpu()}, sys.argv[2])
    ref = np.linalg.eigvals(a[:64].cpu().numpy())
    import oracle as O
    print("lib", os.environ.get("EGS_LIB_PATH", "tree"), "unconverged", int((info != 0).sum()), "dlambda", O.max_abs_dlambda(w[:64].cpu().numpy(), ref))
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in ("w", "w1"):
        same = torch.equal(torch.view_as_real(a[k]), torch.view_as_real(b[k]))
        nd = int((torch.view_as_real(a[k]) != torch.view_as_real(b[k])).any(dim=-1).any(dim=-1).sum())
        print(k, "bit-identical" if same else "DIFFERENT in %d matrices" % nd)
    print("info equal", torch.equal(a["info"], b["info"]), torch.equal(a["info1"], b["info1"]))
