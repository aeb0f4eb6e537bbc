import importlib, sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
import oracle as O
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng=np.random.default_rng(9)
def cmp(name, got, want):
    got=np.asarray(got,np.float64); want=np.asarray(want,np.float64)
    fin=np.isfinite(want)
    bad_nonfinite = int((~np.isfinite(got[fin])).sum())
    rel=np.abs(got[fin]-want[fin])/np.maximum(np.abs(want[fin]),1e-300)
    rel=rel[np.isfinite(rel)]
    same_nf = bool(np.array_equal(np.isnan(got[~fin]), np.isnan(want[~fin]))) and bool(np.array_equal(np.isposinf(got[~fin]), np.isposinf(want[~fin]))) if (~fin).any() else True
    print('%-44s max rel %.2e  got-nonfinite-where-want-finite %d  want-nonfinite %d same-kind %s  range [%.2e, %.2e]' % (name, rel.max() if rel.size else 0, bad_nonfinite, int((~fin).sum()), same_nf, np.nanmin(want), np.nanmax(want)), flush=True)
for scale in (1.0, 8.0, 30.0, 100.0, 1000.0):
    for (B,L,D,dqk,H) in ((3,70,32,16,2),(2,130,64,64,1)):
        x=(rng.standard_normal((B,L,D))*scale).astype(np.float32)
        w=(rng.standard_normal((2*dqk+D,D))/np.sqrt(D)).astype(np.float32)
        b=(rng.standard_normal(2*dqk+D)*0.1).astype(np.float32)
        with np.errstate(all='ignore'):
            wl=O.eta_linear(x,w,b,dqk,H,D)[...,0]; ws=O.eta_softmax(x,w,b,dqk,H,D)[...,0]
        gl=ops.linattn_eta(dev(x),dev(w[:2*dqk]),dev(b[:2*dqk]),dqk,H).cpu().numpy()
        gs=ops.softmax_eta(dev(x),dev(w[:2*dqk]),dev(b[:2*dqk]),dqk,H).cpu().numpy()
        cmp('linear  scale %g L %d H %d'%(scale,L,H), gl, wl)
        cmp('softmax scale %g L %d H %d'%(scale,L,H), gs, ws)
