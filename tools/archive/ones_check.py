import importlib, os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
n = 64
c = torch.linspace(0.5, 2.0, 256, dtype=torch.float64, device="cuda")
a = torch.ones((256, n, n), dtype=torch.float64, device="cuda") * c[:, None, None]
res = {}
for tag, env in (("fused", None), ("unfused", "1")):
    if env: os.environ["EGS_EIG_UNFUSED"] = env
    w, info = ops.eig_general_batched(a)
    if env: del os.environ["EGS_EIG_UNFUSED"]
    w = w.cpu().numpy()
    res[tag] = {"bad_info": int((info != 0).sum().item()), "nonfinite": int((~np.isfinite(w).all(axis=1)).sum())}
h = ops.hessenberg_batched(a[:1])
hh = h[0].cpu().numpy()
res["lib"] = os.environ.get("EGS_LIB_PATH", "tree")
res["H_block"] = hh[:4, :4].tolist()
res["H_absmax_rest"] = float(np.abs(hh[2:, 2:]).max())
print(json.dumps(res))
