#!/usr/bin/env python3
"""Weighted minimax fits behind the polynomial GELU forms of csrc/common.h (gelu_scaled, gelu_scaled_dgrad): Lawson iteration on
the linearised problem (error of exp2(P) weighted to the absolute error of the GELU value / of 2 GELU'), then the true fp32
error of the Horner + exp2 + fused-step evaluation, per degree; monotonicity of P beyond the fitted range.  CPU only."""
import numpy as np
from scipy.special import ndtr, log_ndtr
K=0.84932180028801904272
def fit(deg, A=6.4, iters=400, n=20001):
    x=np.linspace(1e-6,A,n)      # |x|
    a=K*x
    Pstar=(log_ndtr(-x)+np.log(2.0))/np.log(2.0)   # log2(2 Phi(-x))
    w=(x/2)*2*ndtr(-x)*np.log(2.0)  # d(GELU err)/dP
    V=np.vander(a,deg+1)            # highest power first
    lam=np.ones(n)
    best=None
    for it in range(iters):
        W=np.sqrt(lam)*w
        c,*_=np.linalg.lstsq(V*W[:,None], Pstar*W, rcond=None)
        e=np.abs(w*(V@c-Pstar))
        lam=lam*(e/e.max()+1e-3); lam/=lam.sum()/n
        if best is None or e.max()<best[0]: best=(e.max(),c.copy())
    return best
def true_err(c, lo=-7, hi=7, n=400001):
    xs=np.linspace(lo,hi,n)
    xp=(xs*K).astype(np.float32); a=np.abs(xp)
    p=np.full_like(a,np.float32(c[0]))
    for ci in c[1:]:
        p=(p.astype(np.float64)*a+np.float64(np.float32(ci))).astype(np.float32)
    q2=np.exp2(p.astype(np.float64)).astype(np.float32)
    y2=(-(a.astype(np.float64))*q2+(a+xp).astype(np.float32)).astype(np.float32)
    y=y2.astype(np.float64)/(2*K)
    err=np.abs(y-xs*ndtr(xs))
    return err.max(), np.sqrt(np.mean(err**2))
for deg in (7,6,5,4):
    e,c=fit(deg)
    print(deg, "linearised minimax err %.3g"%e, "lead %.3e"%c[0], "true fp32 max/rms on [-7,7]: %.3g %.3g"%true_err(c), "on[-3,3]: %.3g %.3g"%true_err(c,-3,3))
    print("   ", ", ".join("%.9e"%v for v in c))
print("---- value form: monotone check of degree 5")
e,c5=fit(5)
a=np.linspace(0,2000,2000001)
P=np.polyval(c5,a)
print("P5 decreasing everywhere:", bool(np.all(np.diff(P)<0)), "P(5.4)=%.2f"%np.polyval(c5,5.4))
print("---- derivative form")
from scipy.stats import norm
a0x=0.7517915246935645
def fit_d(deg, A=6.6, iters=600, n=40001):
    x=np.linspace(1e-5,A,n)
    x=x[np.abs(x-a0x)>2e-4]
    a=K*x
    H=2*ndtr(-x)-2*x*norm.pdf(x)
    R=H/(K*(a0x-x))          # H/(a0 - a), a0 = K*a0x
    Pstar=np.log2(R)
    w=np.abs(K*(a0x-x))*R*np.log(2.0)
    V=np.vander(a,deg+1)
    lam=np.ones(len(x)); best=None
    for it in range(iters):
        W=np.sqrt(lam)*w
        c,*_=np.linalg.lstsq(V*W[:,None], Pstar*W, rcond=None)
        e=np.abs(w*(V@c-Pstar))
        lam=lam*(e/e.max()+1e-3); lam/=lam.mean()
        if best is None or e.max()<best[0]: best=(e.max(),c.copy())
    return best
def true_err_d(c, lo=-7, hi=7, n=400001):
    xs=np.linspace(lo,hi,n)
    xp=(xs*K).astype(np.float32); a=np.abs(xp)
    p=np.full_like(a,np.float32(c[0]))
    for ci in c[1:]:
        p=(p.astype(np.float64)*a+np.float64(np.float32(ci))).astype(np.float32)
    e=np.exp2(p.astype(np.float64)).astype(np.float32)
    g=((a-np.float32(K*a0x)).astype(np.float32).astype(np.float64)*e+1.0).astype(np.float32)
    d=(np.sign(xs)*g.astype(np.float64)+1.0).astype(np.float32).astype(np.float64)
    exact=2.0*(ndtr(xs)+xs*norm.pdf(xs))
    err=np.abs(d-exact)
    return err.max(), np.sqrt(np.mean(err**2))
for deg in (7,6,5):
    e,c=fit_d(deg)
    P=np.polyval(c,np.linspace(0,2000,2000001))
    print(deg,"lin err %.3g"%e,"lead %.3e"%c[0],"true max/rms [-7,7]: %.3g %.3g"%true_err_d(c),"[-3,3]: %.3g %.3g"%true_err_d(c,-3,3),"decreasing:",bool(np.all(np.diff(P)<0)))
    print("   ", ", ".join("%.9e"%v for v in c))
