import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
n = 128
c = torch.linspace(0.5, 2.0, 8, dtype=torch.float64, device="cuda")
a = torch.ones((8, n, n), dtype=torch.float64, device="cuda") * c[:, None, None]
a = torch.complex(a, 0.5 * a).contiguous()
h, w, info = ops.hessenberg_general_batched(a)
hh = h.cpu().numpy()
print("stage1 info", info.tolist(), "finite", np.isfinite(hh).all(), "exp", w[:, 0].real.tolist())
m = hh[0].T
print("subdiag", ["%.1e" % abs(m[k, k - 1]) for k in range(1, 14)])
print("diag", ["%.1e" % abs(m[k, k]) for k in range(0, 14)])
print("min nonzero", np.min(np.abs(hh[hh != 0])))
w2, info2, prof = ops.hseqr_general_batched(h.clone(), w.clone(), info.clone(), profile=True)
print("stage2 info", info2.tolist())
print("sweeps", prof[:, 6].tolist())
np.save(os.path.join(ROOT, "gpurun_out", "const128_H.npy"), hh[:2])
