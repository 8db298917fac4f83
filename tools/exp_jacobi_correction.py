#!/usr/bin/env python3
"""CPU experiment (numpy, float64 emulation of cyclic Jacobi sweeps) behind DESIGN 5: the positive part V max(0, L) V^T from an eigensolver that stops early,
plain and with the first-order correction V (E o Phi) V^T, on the matrices C - N of synthetic frames (oracle patch traces); and the per-pair stopping rule
that was measured out on the chip."""
import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bcd_amd.core as core, oracle_lib as ol
def mats(pattern, sigma, spp, n=120, seed=3):
    W,H=96,72
    col, ns, hist, cov = core.synthetic_scene(W,H,spp,1234,sigma,0.01 if sigma>0.2 else 0.0, pattern=pattern)
    prm = ol.params()
    rng=np.random.default_rng(seed); out=[]
    tries=0
    while len(out)<n and tries<4000:
        tries+=1
        l=int(rng.integers(8,H-8)); c=int(rng.integers(8,W-8))
        t=ol.patch_trace(col,ns,hist,cov,prm,l,c)
        if len(t["members"])>=28: out.append(t["cov1_minus_noise"].astype(np.float64))
    return out
def jacobi_sweeps(A, nsweeps):
    A=A.copy(); n=A.shape[0]; V=np.eye(n); res=[]
    for s in range(nsweeps):
        for p in range(n-1):
            for q in range(p+1,n):
                if A[p,q]==0: continue
                th=(A[q,q]-A[p,p])/(2*A[p,q]); t=np.sign(th)/(abs(th)+np.sqrt(th*th+1)) if th!=0 else 1.0
                c=1/np.sqrt(t*t+1); s_=t*c
                J=np.eye(n); J[p,p]=c; J[q,q]=c; J[p,q]=s_; J[q,p]=-s_
                A=J.T@A@J; V=V@J
        off=np.sqrt((A**2).sum()-(np.diag(A)**2).sum()); dg=np.sqrt((np.diag(A)**2).sum())
        res.append((off/dg, A.copy(), V.copy()))
    return res
def pos_exact(A):
    w,v=np.linalg.eigh(A); return (v*np.maximum(w,0))@v.T
def corrected(Ak,V):
    d=np.diag(Ak); f=np.maximum(d,0)
    di=d[:,None]; dj=d[None,:]
    with np.errstate(divide='ignore',invalid='ignore'):
        phi=np.where((di>0)&(dj>0),1.0,np.where((di<=0)&(dj<=0),0.0,(np.maximum(di,0)-np.maximum(dj,0))/(di-dj)))
    M=Ak*phi; np.fill_diagonal(M,f)
    return V@M@V.T
for name,(pat,sig,spp) in {"checker-noisy":(0,0.35,32),"textured":(1,0.35,32),"tex-8spp":(1,0.15,8)}.items():
    Ms=mats(pat,sig,spp)
    errs={k:[[],[]] for k in range(1,7)}; offs={k:[] for k in range(1,7)}
    for A in Ms:
        ex=pos_exact(A); nrm=np.linalg.norm(A,2)
        for k,(off,Ak,V) in enumerate(jacobi_sweeps(A,6),1):
            plain=(V*np.maximum(np.diag(Ak),0))@V.T
            errs[k][0].append(np.abs(plain-ex).max()/nrm); errs[k][1].append(np.abs(corrected(Ak,V)-ex).max()/nrm); offs[k].append(off)
    print(name, len(Ms))
    for k in range(1,7):
        print("  sweep %d: off/diag med %.1e max %.1e | plain err med %.1e max %.1e | corrected med %.1e max %.1e" % (k,np.median(offs[k]),max(offs[k]),np.median(errs[k][0]),max(errs[k][0]),np.median(errs[k][1]),max(errs[k][1])))

print("---- pairwise criterion")
def crit(Ak):
    d=np.diag(Ak); E=Ak-np.diag(d)
    di=d[:,None]; dj=d[None,:]
    strad=(di*dj<0)
    with np.errstate(divide='ignore',invalid='ignore'):
        q=np.minimum(np.abs(E), E*E/(np.abs(di)+np.abs(dj)))
    q=np.where(strad,q,0.0)
    return q.max()/np.linalg.norm(d)
for name,(pat,sig,spp) in {"checker-noisy":(0,0.35,32),"textured":(1,0.35,32),"tex-8spp":(1,0.15,8)}.items():
    Ms=mats(pat,sig,spp,n=80)
    for tol in (1e-7,3e-8):
        stops=[]; errs=[]; plain6=[]
        for A in Ms:
            ex=pos_exact(A); nrm=np.linalg.norm(A,2)
            res=jacobi_sweeps(A,7)
            done=None
            for k,(off,Ak,V) in enumerate(res,1):
                if crit(Ak)<=tol and off<=3e-3:
                    done=k; errs.append(np.abs(corrected(Ak,V)-ex).max()/nrm); break
            stops.append(done if done else 8)
            # the current rule: off/diag <= 1e-6
            for k,(off,Ak,V) in enumerate(res,1):
                if off<=1e-6 or k==7:
                    plain6.append((k,np.abs((V*np.maximum(np.diag(Ak),0))@V.T-ex).max()/nrm)); break
        print(name,"tol",tol,"sweeps hist",np.bincount(stops),"corrected err med %.1e max %.1e"%(np.median(errs),max(errs)),"| current rule sweeps hist",np.bincount([k for k,_ in plain6]),"err max %.1e"%max(e for _,e in plain6))
