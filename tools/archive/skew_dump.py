import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
m = np.load(os.path.join(ROOT, "tools", "probe", "skew_bad.npy"))
h = ops.hessenberg_batched(torch.from_numpy(m).cuda())
w, info = ops.hseqr_batched(h)
np.save(os.path.join(ROOT, "gpurun_out", "skew_bad_H.npy"), h.cpu().numpy())
np.save(os.path.join(ROOT, "gpurun_out", "skew_bad_w.npy"), w.cpu().numpy())
print("max|w| per matrix", np.abs(w.cpu().numpy()).max(axis=1), info.cpu().numpy())
print("diag of H[0][:6]", np.diag(h[0].cpu().numpy())[:6])
