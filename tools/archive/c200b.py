import importlib, sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
np.set_printoptions(linewidth=200, precision=3)
ops = importlib.import_module("task-level-insights-from-eigenvalues-across-sequence-models_amd").ops
n=200
cs=[0.95441435, 0.91700353, 0.7]
a = torch.ones((3,n,n), device="cuda", dtype=torch.float64)*torch.tensor(cs,device="cuda",dtype=torch.float64)[:,None,None]
hh, ww, ii = ops.hessenberg_general_batched(a.contiguous())
h = hh.cpu().numpy().copy()
e = ww.cpu().numpy()[:,0].real
print('exp', e, 'shape', h.shape, h.dtype)
w, info, prof = ops.hseqr_general_batched(hh, ww, ii, profile=True)
print('info', info.cpu().numpy(), 'sweeps', prof.cpu().numpy()[:,6])
for m in range(3):
    H = h[m].T if False else h[m]
    # try both orientations for the sub-diagonal
    sd = np.abs(np.diagonal(H, -1)); sd2=np.abs(np.diagonal(H, 1))
    low = np.abs(np.tril(H,-2)).max(); up=np.abs(np.triu(H,2)).max()
    print(m, 'below-subdiag max', low, 'above max', up)
    print(' subdiag[:12]', sd[:12]); print(' superdiag[:12]', sd2[:12])
    print(' diag[:8]', np.diagonal(H)[:8])
    print(' |subdiag| around 150..170', sd[150:170])
    ev = np.linalg.eigvals(H if low < up else H.T)
    print(' numpy eig of h: max', np.abs(ev).max(), 'second', np.sort(np.abs(ev))[-2])
