#!/usr/bin/env python3
"""Cost model behind DESIGN.md's "two bulges per wave" paragraph: Francis double-shift QR on 64 x 64 real Hessenberg
matrices in NumPy, counting bulge steps for (single) one bulge per sweep, (same) the same double shift chased twice,
(d22) second shift pair from the 2x2 block above the trailing one, (e44) the four eigenvalues of the trailing 4x4
block.  Two bulges chased one behind the other are mathematically two consecutive sweeps, which is how they are
modelled; `cost` charges a paired iteration F (argv[2], default 1.4) single steps plus the unpaired head / tail.
    python tools/qr_two_bulge_model.py randn 1.4 ;  python tools/qr_two_bulge_model.py bench 1.4"""
import numpy as np, scipy.linalg as sl, sys
ulp=2.2e-16; smlnum=1e-290
def refl(x):
    a0=x[0]; nrm=np.linalg.norm(x)
    if nrm==0: return np.zeros_like(x),0.0
    beta=-np.copysign(nrm,a0); v=x.copy(); v[0]=a0-beta; v/= v[0]; tau=(beta-a0)/beta
    return v,tau
def shifts2(H,i):
    a,b,c,d=H[i-1,i-1],H[i-1,i],H[i,i-1],H[i,i]
    tr=a+d; det=a*d-b*c
    return tr,det  # s = sum, p = product
def startvec(H,m,s,p):
    h00,h10,h01,h11,h21=H[m,m],H[m+1,m],H[m,m+1],H[m+1,m+1],H[m+2,m+1]
    v=np.array([h00*h00+h01*h10-s*h00+p, h10*(h00+h11-s), h10*h21])
    return v/ (np.abs(v).sum() or 1)
def sweep_single(H,l,i,s,p,cnt):
    v=startvec(H,l,s,p)
    for k in range(l,i):
        nr=min(3,i-k+1)
        if k>l: v=H[k:k+nr,k-1].copy()
        else: v=v[:nr]
        u,tau=refl(v)
        H[k:k+nr,max(k-1,l):]-= tau*np.outer(u,u@H[k:k+nr,max(k-1,l):])
        r=min(k+3,i)
        H[:r+1,k:k+nr]-= tau*np.outer(H[:r+1,k:k+nr]@u,u)
        if k>l: H[k+1:k+nr,k-1]=0
        cnt[0]+=1
def hqr(H0,mode):
    H=H0.copy(); n=H.shape[0]; i=n-1; cnt=[0]; iters=0; cost=0.0
    F=float(sys.argv[2]) if len(sys.argv)>2 else 1.4
    while i>=0:
        l=0; its=0
        while True:
            # deflation search
            l=0
            for k in range(i,0,-1):
                if abs(H[k,k-1])<= ulp*(abs(H[k,k])+abs(H[k-1,k-1])):
                    H[k,k-1]=0; l=k; break
            if l>=i-1: break
            its+=1
            L=i-l+1
            if its%10==0:
                sA=1.5*(abs(H[i,i-1])+abs(H[i-1,i-2])) ; pA=(abs(H[i,i-1])+abs(H[i-1,i-2]))**2
            else: sA,pA=shifts2(H,i)
            if mode=='single' or L<10:
                sweep_single(H,l,i,sA,pA,cnt); cost+=L-1
            else:
                if mode=='d22':
                    sB,pB=shifts2(H,i-2)
                elif mode=='same':
                    sB,pB=sA,pA
                elif mode=='e44':
                    ev=np.linalg.eigvals(H[i-3:i+1,i-3:i+1])
                    # pair A = the two closest to trailing 2x2's eigs? use standard: sort by closeness to H[i,i]
                    ev=sorted(ev,key=lambda z:abs(z-H[i,i]))
                    # make conjugate-closed pairs
                    def pair(z1,z2): return (z1+z2).real,(z1*z2).real
                    evs=list(ev)
                    # group complex conj pairs
                    used=[False]*4; pairs=[]
                    for a_ in range(4):
                        if used[a_]: continue
                        if abs(evs[a_].imag)>0:
                            for b_ in range(a_+1,4):
                                if not used[b_] and abs(evs[b_]-np.conj(evs[a_]))<1e-12*max(1,abs(evs[a_])):
                                    pairs.append((evs[a_],evs[b_])); used[a_]=used[b_]=True; break
                    reals=[evs[a_] for a_ in range(4) if not used[a_]]
                    while len(reals)>=2: pairs.append((reals[0],reals[1])); reals=reals[2:]
                    pairs=sorted(pairs,key=lambda pr:min(abs(pr[0]-H[i,i]),abs(pr[1]-H[i,i])))
                    sA,pA=pair(*pairs[0]); sB,pB=pair(*pairs[1])
                # sequential application models simultaneous chase exactly (math identical up to rounding)
                sweep_single(H,l,i,sA,pA,cnt)
                sweep_single(H,l,i,sB,pB,cnt)
                cost+= 4 + (L-1-4)*F + 4 + 2   # head + paired middle + tail
            if its>300: raise RuntimeError('noconv')
        i=l-1
    return cnt[0],cost
rng=np.random.default_rng(0)
kind=sys.argv[1] if len(sys.argv)>1 else 'randn'
res={m:[] for m in ('single','same','d22','e44')}
for t in range(6):
    if kind=='randn':
        A=rng.standard_normal((64,64))
    else:
        # bench-like: known spectrum similarity
        om=rng.uniform(0,40,32); dt=np.exp(rng.uniform(np.log(1e-3),np.log(1e-1),32)); lam=np.exp((-0.5+1j*om)*dt)
        A=np.zeros((64,64))
        for p_ in range(32):
            A[2*p_,2*p_]=A[2*p_+1,2*p_+1]=lam[p_].real; A[2*p_,2*p_+1]=lam[p_].imag; A[2*p_+1,2*p_]=-lam[p_].imag
        A+=np.triu(rng.standard_normal((64,64)),2)*0.1/8
        for _ in range(3):
            u=rng.standard_normal((64,1)); u/=np.linalg.norm(u); A=A-2*u@(u.T@A); A=A-2*(A@u)@u.T
    H=sl.hessenberg(A)
    for m in res:
        res[m].append(hqr(H,m))
for m in res:
    a=np.array(res[m]); print(m,'steps',a[:,0].mean(),'cost',a[:,1].mean())
