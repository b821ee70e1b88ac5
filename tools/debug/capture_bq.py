import sys, torch
sys.path.insert(0, '.')
import se3conv3d_amd as amd
from se3conv3d_amd.workloads import radius_for_degree
DEV = "cuda:0"
n, mode = int(sys.argv[1]), sys.argv[2]
torch.manual_seed(0)
pc = amd.pc.PointcloudRotEquiv(torch.rand(n, 3, device=DEV), torch.zeros(n, dtype=torch.int32, device=DEV), {"pca": False, "n_frames": 2, "fixed_axis": False})
pc.num_batches()
r = radius_for_degree(n, 24)
e = amd.pc.BQNeighborhood(pc, pc, r).num_edges()
conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(64, 64).to(DEV)
x = torch.randn(n * 2, 64, device=DEV, requires_grad=True); g = torch.randn(n * 2, 64, device=DEV)
keep = {}
def step():
    if mode in ("bq", "both", "bqonly_ops"):
        if mode == "bqonly_ops":
            keep["r"] = amd.ops.ball_query_bounded(pc.pts_, pc.pts_, pc.batch_ids_, pc.batch_ids_, r, int(e * 1.25), 1)
            return
        nbh = amd.pc.BQNeighborhood(pc, pc, r, p_capacity=int(e * 1.25))
    else:
        nbh = keep["nbh0"]
    keep["nbh"] = nbh
    if mode in ("conv", "both"):
        x.grad = None; conv.zero_grad(set_to_none=True)
        out = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=nbh); out.backward(g); keep["out"] = out
keep["nbh0"] = amd.pc.BQNeighborhood(pc, pc, r)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
gr.replay(); torch.cuda.synchronize()
print("OK", n, mode, flush=True)
