import importlib, json, os, sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
import oracle as O
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
d = lambda z: torch.from_numpy(np.ascontiguousarray(z)).cuda()
for n in (64, 48):
  for h in (16, 128, 512, 2048):
    lam, p, _, _, _ = O.make_dplr_hippo(n)
    ls = np.random.default_rng(n).uniform(np.log(1e-3), np.log(1e-1), h)
    ab = ops.s4_dplr_build(d(np.tile(lam.real,(h,1))), d(np.tile(lam.imag,(h,1))), d(np.tile(p,(h,1))), d(ls))
    res={}
    for mode in ("wave","level3"):
        if mode=="level3": os.environ["EGS_ZLARGE_MIN"]=str(n)
        else: os.environ.pop("EGS_ZLARGE_MIN",None)
        w, info = ops.eig_general_batched(ab); torch.cuda.synchronize()
        ts=[]
        for _ in range(3):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(); w, info = ops.eig_general_batched(ab); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ref=np.linalg.eigvals(ab[:4].cpu().numpy())
        res[mode]=(round(float(np.median(ts)),2), int((info!=0).sum().item()), float(O.max_abs_dlambda(w[:4].cpu().numpy(), ref)))
    print(n,h,res,flush=True)
