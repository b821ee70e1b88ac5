#!/usr/bin/env python3
"""CPU simulation behind the T16 row format (common.h): T, U and grad_T of a realistic layer (N = 1536, k ~ 32, F = 2, C = 64,
K = 32) from the fp64 oracle, quantised to 16-bit mantissas with one power-of-two scale per block for several block shapes;
printed: the relative error each format puts into the tensor's consumers (out, dW | dX | gphi).  eq = with a per-basis
equalisation in front (no help).  Output kept as profiles/r04_t16_format_simulation.txt."""
import sys, math, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import se3conv_oracle as O
torch.set_num_threads(8)
g = torch.Generator().manual_seed(0)
n, f, c, kb, deg = 1536, 2, 64, 32, 32
pts = torch.rand(n,3,generator=g); bid = torch.zeros(n,dtype=torch.int32)
fr = O.random_frames(n,f,g)
r = O.radius_for_degree(n,deg)
nb,_ = O.ball_query(pts,pts,bid,bid,r)
a,b,w = O.init_parameters(9,c,c,kb,g)
x = torch.randn(n*f,c,generator=g); go = torch.randn(n*f,c,generator=g)
D = torch.float64
rt = O.get_rot_tensors(pts.to(D),pts.to(D),fr.to(D),fr.to(D),nb,torch.tensor(1.0/r,dtype=D),n_rows=n*f)
phi = O.kernel_mlp(rt["rel_pts_rel_orient"], a.to(D), b.to(D))
seg, src = rt["neighbs"][:,0], rt["neighbs"][:,1]
rows = n*f
T = torch.zeros((rows,c,kb),dtype=D).index_add(0,seg, x.to(D)[src][:,:,None]*phi[:,None,:])
W = w.to(D)
out = torch.einsum("nik,iko->no",T,W)
dW = torch.einsum("nik,no->iko",T,go.to(D))
gT = torch.einsum("no,iko->nik",go.to(D),W)
# consumers of gT: gphi[e,k] = sum_i gT[m,i,k] f[p,i]; then dA = desc^T (gphi*gelu')   -> use gphi error as proxy and dA
gphi = torch.einsum("eik,ei->ek", gT[seg], x.to(D)[src])
# U (transposed): U[p,o,k] = sum_e phi[e,k] g[m,o] ; dX = U W'
U = torch.zeros((rows,c,kb),dtype=D).index_add(0,src, go.to(D)[seg][:,:,None]*phi[:,None,:])
dX = torch.einsum("pok,iko->pi",U,W)
rel = lambda a_,b_: float((a_-b_).norm()/b_.norm())

def q_block(Tn, block_fn, bits=16, pow2=True):
    """quantise with a scale per block; block_fn maps tensor [rows,c,k] -> view where last dim is the block"""
    v = block_fn(Tn)
    mx = v.abs().amax(-1,keepdim=True).clamp_min(1e-300)
    if pow2:
        sc = torch.exp2(torch.ceil(torch.log2(mx)))  # power of two >= max
    else:
        sc = mx
    lv = 2**(bits-1)-1
    q = torch.round(v/sc*lv).clamp(-lv,lv)
    return (q*sc/lv)

def inv(fn_view, shape):
    return fn_view

blocks = {
 "row (2048)":            (lambda t: t.reshape(t.shape[0],-1), lambda v,s: v.reshape(s)),
 "row x chan-half (1024)":(lambda t: t.reshape(t.shape[0],2,-1), lambda v,s: v.reshape(s)),
 "(row,k) over 64 ch":    (lambda t: t.permute(0,2,1), lambda v,s: v.permute(0,2,1)),
 "(row,ch) over 32 k":    (lambda t: t, lambda v,s: v),
 "(row,chpair) 64":       (lambda t: t.reshape(t.shape[0],c//2,2*kb), lambda v,s: v.reshape(s)),
 "(row,k,4ch)":           (lambda t: t.permute(0,2,1).reshape(t.shape[0],kb,c//4,4), lambda v,s: v.reshape(s[0],kb,c).permute(0,2,1)),
 "(row,k,8ch)":           (lambda t: t.permute(0,2,1).reshape(t.shape[0],kb,c//8,8), lambda v,s: v.reshape(s[0],kb,c).permute(0,2,1)),
 "(row,k,16ch)":          (lambda t: t.permute(0,2,1).reshape(t.shape[0],kb,c//16,16), lambda v,s: v.reshape(s[0],kb,c).permute(0,2,1)),
 "(row,k,32ch)":          (lambda t: t.permute(0,2,1).reshape(t.shape[0],kb,c//32,32), lambda v,s: v.reshape(s[0],kb,c).permute(0,2,1)),
}
def colscale(t):  # per-k equalisation (power of two), from the rms over rows and channels
    rms = t.pow(2).mean((0,1)).sqrt()
    return torch.exp2(torch.round(torch.log2(rms)))
print("k-profile rms spread of T:", (T.pow(2).mean((0,1)).sqrt().max()/T.pow(2).mean((0,1)).sqrt().min()).item())
print(f"{'format':28s} {'out':>9s} {'dW':>9s} | {'dX(U)':>9s} | {'gphi(gT)':>9s}")
for eq in (False, True):
  for name,(fv,bv) in blocks.items():
    for pow2 in (True,):
        res=[]
        for X, cons in ((T,[lambda q: rel(torch.einsum("nik,iko->no",q,W),out), lambda q: rel(torch.einsum("nik,no->iko",q,go.to(D)),dW)]),
                        (U,[lambda q: rel(torch.einsum("pok,iko->pi",q,W),dX)]),
                        (gT,[lambda q: rel(torch.einsum("eik,ei->ek", q[seg], x.to(D)[src]),gphi)])):
            d = colscale(X) if eq else torch.ones(kb,dtype=D)
            Xn = X/d
            q = bv(q_block(Xn, fv, 16, pow2), X.shape)*d
            for cfn in cons: res.append(cfn(q))
        print(f"{('eq ' if eq else '   ')+name:28s} {res[0]:9.2e} {res[1]:9.2e} | {res[2]:9.2e} | {res[3]:9.2e}")
# reference: 24-bit float (T24) and bf16
def fl(t, mant):
    m,e = torch.frexp(t); return torch.ldexp(torch.round(m*2**mant)/2**mant, e)
for mant,name in ((16,"T24 (16 significant bits)"),(8,"bf16")):
    print(f"{name:28s} {rel(torch.einsum('nik,iko->no',fl(T,mant),W),out):9.2e}")
