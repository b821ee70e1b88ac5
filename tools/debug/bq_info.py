import sys, torch
sys.path.insert(0, '.')
import se3conv3d_amd as amd
from oracle import se3conv_oracle as O
DEV="cuda:0"
n_src,n_dst,batches,r = 5000,5000,1,0.0007
g = torch.Generator().manual_seed(n_src)
ps, pd = torch.rand(n_src, 3, generator=g), torch.rand(n_dst, 3, generator=g)
bs = torch.zeros(n_src, dtype=torch.int32); bd = torch.zeros(n_dst, dtype=torch.int32)
nb_r, ends_r = O.ball_query(ps, pd, bs, bd, r)
e = nb_r.shape[0]
print("oracle e", e)
args = (ps.to(DEV), pd.to(DEV), bs.to(DEV), bd.to(DEV), r)
for cap in (e + 100, e, e // 2, 1, 0):
    nb, ends, info = amd.ops.ball_query_bounded(*args, capacity=cap, n_batches=batches)
    print("cap", cap, "info", info.tolist(), "ends max", int(ends.max()) if ends.numel() else None)
