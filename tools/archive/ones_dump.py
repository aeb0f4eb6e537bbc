import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
n = 64
c = torch.linspace(0.5, 2.0, 256, dtype=torch.float64, device="cuda")
a = torch.ones((256, n, n), dtype=torch.float64, device="cuda") * c[:, None, None]
w, info = ops.eig_general_batched(a)
bad = torch.nonzero(info != 0).flatten()[:3]
print("bad idx", bad.tolist(), "info", info[bad].tolist(), "c", c[bad].tolist())
h = ops.hessenberg_batched(a[bad].contiguous())
np.save(os.path.join(ROOT, "gpurun_out", "ones_bad_H.npy"), h.cpu().numpy())
np.save(os.path.join(ROOT, "gpurun_out", "ones_bad_c.npy"), c[bad].cpu().numpy())
w2, info2 = ops.hseqr_batched(h)
print("hseqr on unscaled H: info", info2.tolist(), "finite", torch.isfinite(torch.view_as_real(w2)).all(dim=(1,2)).tolist())
