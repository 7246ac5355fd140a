#!/usr/bin/env python3
"""Study (CPU, numpy fp32): how many evaluations the secular solver of eig_lean.h (secular_root_reg, restated line by line in fp32) needs per
root on the merges of a real tridiagonal matrix -- b = S - (S + I)^-1 of the bench's synthetic covariances, Householder-tridiagonalised --
at every merge size 4 ... 128, and why the slow ones are slow.  `--old-guess` restates the starting point as it was up to round 3.

    python scripts/secular_study.py [--old-guess] > profiles/r04_secular_study.txt

Findings (round 4): (1) for K = i + 1 (the root nearer to its right pole: 40 % of all roots) the starting point took the wrong root of the
model's quadratic, never inside the bracket, and fell back to the bracket's midpoint; fixed: mean 3.07 -> 2.90 evaluations; (2) the tail --
5 % of the roots need 5 or more evaluations, and a wave waits for its slowest -- belongs to poles of small weight rho z_K^2, where the model
(two nearest poles exact, the rest frozen) is poor; neither fix changes the MAXIMUM per wave, which is what the kernel's time follows."""
import sys
OLD_GUESS = "--old-guess" in sys.argv
import numpy as np
sys.path.insert(0,'/root/repo')
from scipy.linalg import hessenberg
from uglad_amd.utils.prepare_data import synthetic_covariance_batch
f=np.float32
kEps=f(5.96e-8)
def rcp(x): return f(1.0)/f(x)
def secular_root(ds, rz, rho, nb, i, maxit=48, trace=None):
    ds=ds.astype(f); rz=rz.astype(f)
    last = i==nb-1
    ia = nb-2 if last else i
    hi_last = f(rho*f(1.00001)+f(1e-30))
    dorg=ds[i]
    test=f(0.5)*(ds[i if last else i+1]-dorg)
    if last:
        d1l=ds[nb-2]-dorg; pl=rz[nb-2]; ql=rz[nb-1]
        bl=d1l+pl+ql; cl=ql*d1l
        x0=f(0.5)*(bl+np.sqrt(max(bl*bl-4*cl,f(0))))
        test = x0 if (x0>0 and x0<hi_last) else f(0.5)*hi_last
    wsum=f(0)
    for j in range(nb): wsum += rz[j]*rcp((ds[j]-dorg)-test)
    wt=f(1)+wsum
    K = i if (last or wt>0) else i+1
    dK=ds[K]
    pd=(ds-dK).astype(f)
    d1=ds[ia]-dK; d2=ds[ia+1]-dK
    p=rz[ia]; q=rz[ia+1]
    xt=(dorg-dK)+test
    rest=wt-p*rcp(d1-xt)-q*rcp(d2-xt)
    if last:
        lo = test if wt<0 else f(0); hi = hi_last if wt<0 else test
    else:
        lo = f(0) if K==i else -test; hi = test if K==i else f(0)
    bq=rest*(d1+d2)+p+q
    cq=rest*d1*d2+p*d2+q*d1
    sq0=np.sqrt(max(bq*bq-4*rest*cq,f(0)))
    if K==ia: mu = 2*cq*rcp(bq+sq0) if bq>0 else (bq-sq0)*rcp(2*rest)
    elif last or OLD_GUESS: mu = 2*cq*rcp(bq-sq0) if bq<0 else (bq+sq0)*rcp(2*rest)
    else: mu = 2*cq*rcp(bq+sq0) if bq>0 else (bq-sq0)*rcp(2*rest)
    guess_ok = (mu>lo and mu<hi)
    if not guess_ok: mu=f(0.5)*(lo+hi)
    jr=i+1
    dl1=ds[i]-dK; dl2=(ds[jr]-dK) if jr<nb else f(0)
    it=0; bis=0; newt=0
    while it<maxit:
        r=f(1)/(pd-mu); term=rz*r
        ws=term.sum(dtype=f); as_=np.abs(term).sum(dtype=f); dsum=(term*r).sum(dtype=f); dpsi=(np.minimum(term,0)*r).sum(dtype=f)
        dphi=dsum-dpsi
        D1=dl1-mu; D2=dl2-mu
        w=f(1)+ws
        if abs(w)<=8*kEps*(1+as_): break
        if w<0: lo=mu
        else: hi=mu
        a=w-D1*dpsi-D2*dphi
        b=(D1+D2)*w-D1*D2*dsum
        g=D1*D2*w
        sq=np.sqrt(abs(b*b-4*a*g))
        bneg=b<=0; a0=a==0
        num=(g if a0 else b-sq) if bneg else 2*g
        den=(b if a0 else 2*a) if bneg else b+sq
        add=f(0)
        if jr>=nb:
            c=w-dpsi*D1
            num=dpsi*D1*D1 if c!=0 else f(0); den=c if c!=0 else f(1); add=D1 if c!=0 else f(0)
        with np.errstate(all='ignore'):
            eta=f(num*rcp(den)+add)
            newton=-w*rcp(dsum)
        if (not abs(eta)<3e38) or w*eta>=0: eta=newton; newt+=1
        nw=mu+eta
        if not (nw>lo and nw<hi): nw=f(0.5)*(lo+hi); bis+=1
        if nw==mu: break
        mu=nw; it+=1
    return it+1, K, mu, dict(bis=bis,newt=newt,guess_ok=guess_ok,last=last)

def study(n, seed):
    S=synthetic_covariance_batch(1,n,seed=seed)[0].astype(np.float64)
    Z=np.linalg.inv(S+np.eye(n))
    b=S/1.0-Z
    H,Q=hessenberg(b,calc_q=True)
    d=np.diag(H).copy(); e=np.diag(H,1).copy()
    res=[]
    # one merge at each of several split levels: merge [lo,mid) and [mid,hi)
    for bs in (4,8,16,32,64,128):
        if bs>n: break
        h=bs//2
        for lo in range(0,n,bs):
            mid=lo+h; hi=min(lo+bs,n)
            if mid>=n: continue
            dd=d[lo:hi].copy(); ee=e[lo:hi-1].copy()
            ec=ee[h-1]; rho=2*abs(ec)
            # torn blocks (as in the kernel: subtract |e| at the boundaries of the tear)
            d1=dd[:h].copy(); d2=dd[h:].copy()
            d1[-1]-=abs(ec); d2[0]-=abs(ec)
            T1=np.diag(d1)+np.diag(ee[:h-1],1)+np.diag(ee[:h-1],-1)
            T2=np.diag(d2)+np.diag(ee[h:],1)+np.diag(ee[h:],-1)
            w1,Q1=np.linalg.eigh(T1); w2,Q2=np.linalg.eigh(T2)
            z=np.concatenate([Q1[-1,:], (1 if ec>=0 else -1)*Q2[0,:]])*0.70710678
            dsv=np.concatenate([w1,w2]); o=np.argsort(dsv,kind='stable')
            dsv=dsv[o].astype(f); z=z[o].astype(f)
            z=np.where(np.abs(z)<1e-10, np.where(z<0,-1e-10,1e-10), z).astype(f)  # (the kernel's floor on |z|)
            rz=(f(rho)*z*z).astype(f)
            for i in range(hi-lo):
                ev,K,mu,info=secular_root(dsv,rz,f(rho),hi-lo,i)
                res.append((bs,ev,info['bis'],info['newt'],info['guess_ok'],info['last'], rz[i], ))
    return res
allr=[]
for seed in range(6):
    allr+=study(128,seed)
allr=np.array([(r[0],r[1],r[2],r[3],int(r[4]),int(r[5]),r[6]) for r in allr],dtype=np.float64)
for bs in (4,8,16,32,64,128):
    m=allr[:,0]==bs
    ev=allr[m,1]
    print(f"bs={bs:3d} roots={m.sum():5d} mean evals {ev.mean():.2f}  hist", np.bincount(ev.astype(int))[1:10], " with bisection steps:", (allr[m,2]>0).sum(), " newton fallback:", (allr[m,3]>0).sum(), " bad guess:", (allr[m,4]==0).sum())
    slow=m & (allr[:,1]>=5)
    if slow.sum():
        print("    slow roots: bis>0:", (allr[slow,2]>0).sum(), "newton>0:", (allr[slow,3]>0).sum(), "last:", (allr[slow,5]>0).sum(), "of", slow.sum(), " median rz of slow", np.median(allr[slow,6]), "vs all", np.median(allr[m,6]))
