#!/usr/bin/env python3
"""Study (CPU, numpy fp32): how many evaluations the secular solver of eig_lean.h (secular_root_reg, restated line by line in fp32) needs per
root on the merges of a real tridiagonal matrix -- b = S - (S + I)^-1 of the bench's synthetic covariances, Householder-tridiagonalised --
at every merge size 4 ... 128, and why the slow ones are slow.  `--old-guess` restates the starting point as it was up to round 3.

    python scripts/secular_study.py > profiles/r04_secular_study.txt      (four tables: the schemes below)

Findings (round 4): (1) for K = i + 1 (the root nearer to its right pole: 40 % of all roots) the starting point took the wrong root of the
model's quadratic, never inside the bracket, and fell back to the bracket's midpoint; fixed: mean 3.07 -> 2.90 evaluations; (2) the tail --
5 % of the roots need 5 or more evaluations, and a wave waits for its slowest -- belongs to poles of small weight rho z_K^2, where the model
(two nearest poles exact, the rest frozen) is poor; the starting point does not change the MAXIMUM per wave, which is what the kernel's time follows; (3) Gragg's scheme (the model matches the
function and its first TWO derivatives: cubic) with a step below 2^-12 |mu| accepted without the confirming evaluation does: mean 2.9 -> 2.3,
roots with five or more evaluations 6.6 % -> < 0.5 %, eigenvalues and the relative accuracy of mu unchanged -- the kernels' scheme since round 4."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
OLD_GUESS = False  # (set by the tables below: the starting point as up to round 3)
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

def secular_root_gragg(ds, rz, rho, nb, i, maxit=48, trace=None, relstep=0.0):
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
    if not last: mu = 2*cq*rcp(bq+sq0) if bq>0 else (bq-sq0)*rcp(2*rest)
    else: mu = 2*cq*rcp(bq-sq0) if bq<0 else (bq+sq0)*rcp(2*rest)
    if not (mu>lo and mu<hi): mu=f(0.5)*(lo+hi)
    jr=i+1
    dl1=ds[i]-dK; dl2=(ds[jr]-dK) if jr<nb else f(0)
    it=0
    while it<maxit:
        r=f(1)/(pd-mu); term=rz*r; tr=term*r
        ws=term.sum(dtype=f); as_=np.abs(term).sum(dtype=f); dsum=tr.sum(dtype=f); d3=(tr*r).sum(dtype=f)
        D1=dl1-mu; D2=dl2-mu
        w=f(1)+ws
        if trace is not None: trace.append((float(mu),float(w)))
        if abs(w)<=8*kEps*(1+as_): break
        if w<0: lo=mu
        else: hi=mu
        with np.errstate(all='ignore'):
            if jr>=nb:
                # last root: one-pole model c + s/(D1 - eta), s from f' (all poles to the left)
                c=w-dsum*D1
                eta = (dsum*D1*D1/c + D1) if c!=0 else f(0)   # same as the kernel's formula with dpsi = dsum
                eta=f(eta)
            else:
                den=D2-D1
                s_=D1*(D1*(D1*((D2*d3-dsum)/den)))
                S_=D2*(D2*(D2*((dsum-D1*d3)/den)))
                a=w-s_/D1-S_/D2
                b=a*(D1+D2)+s_+S_
                g=D1*D2*w
                sq=np.sqrt(abs(b*b-4*a*g))
                if b<=0: num=(g if a==0 else b-sq); dn=(b if a==0 else 2*a)
                else: num=2*g; dn=b+sq
                eta=f(num*rcp(dn))
            newton=-w*rcp(dsum)
        if (not abs(eta)<3e38) or w*eta>=0: eta=newton
        nw=mu+eta
        outside = not (nw>lo and nw<hi)
        if outside: nw=f(0.5)*(lo+hi)
        if nw==mu: break
        small = (not outside) and abs(nw-mu) <= relstep*abs(nw)  # (a midpoint is never accepted unseen: eig_lean.h)
        mu=nw; it+=1
        if small:
            it-=1; break  # accepted without the confirming evaluation
    return it+1, K, mu, {}

def merges(n=128, seeds=6, sizes=(4, 8, 16, 32, 64, 128)):
    for seed in range(seeds):
        S=synthetic_covariance_batch(1,n,seed=seed)[0].astype(np.float64)
        Z=np.linalg.inv(S+np.eye(n)); b=S-Z
        H,Q=hessenberg(b,calc_q=True); d=np.diag(H).copy(); e=np.diag(H,1).copy()
        for bs in sizes:
            h=bs//2
            for lo_ in range(0,n,bs):
                mid=lo_+h; hi_=min(lo_+bs,n)
                dd=d[lo_:hi_].copy(); ee=e[lo_:hi_-1].copy(); ec=ee[h-1]; rho=2*abs(ec)
                d1=dd[:h].copy(); d2=dd[h:].copy(); d1[-1]-=abs(ec); d2[0]-=abs(ec)
                T1=np.diag(d1)+np.diag(ee[:h-1],1)+np.diag(ee[:h-1],-1); T2=np.diag(d2)+np.diag(ee[h:],1)+np.diag(ee[h:],-1)
                w1,Q1=np.linalg.eigh(T1); w2,Q2=np.linalg.eigh(T2)
                z=np.concatenate([Q1[-1,:],(1 if ec>=0 else -1)*Q2[0,:]])*0.70710678
                dsv=np.concatenate([w1,w2]); o=np.argsort(dsv,kind='stable'); dsv=dsv[o].astype(f); z=z[o].astype(f)
                z=np.where(np.abs(z)<1e-6, np.where(z<0,-1e-6,1e-6), z).astype(f)  # (the kernel's floor on |z|, kZFloor)
                yield bs, dsv, (f(rho)*z*z).astype(f), f(rho), z


def table(name, fn):
    rows=[]; errs=[]
    for bs, dsv, rz, rho, z in merges():
        scale=max(np.abs(dsv).max(), rho)
        exact=np.linalg.eigvalsh(np.diag(dsv.astype(np.float64))+float(rho)*np.outer(z.astype(np.float64),z.astype(np.float64)))
        for i in range(len(dsv)):
            ev,K,mu,_=fn(dsv,rz,rho,len(dsv),i)
            rows.append((bs,ev)); errs.append(abs((float(dsv[K])+float(mu))-exact[i])/scale)
    rows=np.array(rows)
    print(f"## {name}")
    for bs in (4,8,16,32,64,128):
        ev=rows[rows[:,0]==bs,1]
        print(f"   merge to {bs:3d}: {ev.size} roots, mean {ev.mean():.2f} evaluations, max {ev.max()}, histogram 1.. {np.bincount(ev)[1:9]}, five or more: {100.0*(ev>=5).mean():.2f} %")
    print(f"   eigenvalue error / max(|d|, rho): max {max(errs):.2e} mean {np.mean(errs):.2e}\n")


if __name__ == "__main__":
    import warnings; warnings.filterwarnings("ignore")
    OLD_GUESS = True
    table("middle way, starting point as up to round 3", lambda *a: secular_root(*a))
    OLD_GUESS = False
    table("middle way, starting point fixed (first half of round 4)", lambda *a: secular_root(*a))
    table("Gragg's scheme, same stopping rule", lambda *a: secular_root_gragg(*a))
    table("Gragg's scheme, a step below 2^-12 |mu| accepted without the confirming evaluation (the kernels since round 4)", lambda *a: secular_root_gragg(*a, relstep=2.44140625e-4))
