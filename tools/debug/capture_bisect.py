import sys, torch
sys.path.insert(0, '.')
import se3conv3d_amd as amd
from se3conv3d_amd.workloads import radius_for_degree
DEV = "cuda:0"
variant = sys.argv[1]
torch.manual_seed(0)
n, f, c = 6000, 2, 64
pc = amd.pc.PointcloudRotEquiv(torch.rand(n, 3, device=DEV), torch.zeros(n, dtype=torch.int32, device=DEV), {"pca": False, "n_frames": f, "fixed_axis": False})
pc.num_batches()
r = radius_for_degree(n, 24)
ref_nbh = amd.pc.BQNeighborhood(pc, pc, r); e = ref_nbh.num_edges()
conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(c, c).to(DEV)
conv.norm_neigh_dist_.fill_(1.0 / r), conv.norm_num_neighs_.fill_(n / e)
x = torch.randn(n * f, c, device=DEV, requires_grad=True); g = torch.randn(n * f, c, device=DEV)
if variant == "fwd_only":
    with torch.no_grad():
        conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=ref_nbh)
elif variant == "fwd_bwd":
    conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=ref_nbh).backward(g)
elif variant == "fwd_bwd_side":
    s0 = torch.cuda.Stream(); s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=ref_nbh).backward(g)
    torch.cuda.current_stream().wait_stream(s0)
elif variant == "torch_bwd":
    w = torch.randn(c, c, device=DEV, requires_grad=True); (x @ w).sum().backward()
elif variant == "retain_out":
    out_ref = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=ref_nbh); out_ref.backward(g)
elif variant == "retain_clones":
    conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=ref_nbh).backward(g)
    dx_ref, dw_ref = x.grad.clone(), conv.conv_weights_.grad.clone()
elif variant == "retain_all":
    out_ref = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=ref_nbh); out_ref.backward(g)
    dx_ref, dw_ref = x.grad.clone(), conv.conv_weights_.grad.clone()
elif variant == "fwd_bwd_keepgrad":
    conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=ref_nbh).backward(g)
torch.cuda.synchronize()
keep_grads = variant == "fwd_bwd_keepgrad"
holder = {}
def step():
    if not keep_grads:
        x.grad = None; conv.zero_grad(set_to_none=True)
    nbh = amd.pc.BQNeighborhood(pc, pc, r, p_capacity=int(e * 1.25)) if "nobq" not in sys.argv else ref_nbh
    out = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=nbh); out.backward(g)
    holder.update(nbh=nbh, out=out)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s): step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr): step()
gr.replay(); torch.cuda.synchronize()
print("OK", sys.argv[1:], flush=True)
