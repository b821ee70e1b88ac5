import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import se3conv3d_amd as amd
import test_gpu_bounded_query as T
variant = sys.argv[1]
if variant == "plain":
    T.test_neighbourhood_and_conv_step_in_one_graph(amd)
elif variant == "noeager":
    # the test body without the eager reference run
    from se3conv3d_amd.workloads import radius_for_degree
    DEV = "cuda:0"
    torch.manual_seed(0)
    n, f, c = 6000, 2, 64
    pc = amd.pc.PointcloudRotEquiv(torch.rand(n, 3, device=DEV), torch.zeros(n, dtype=torch.int32, device=DEV), {"pca": False, "n_frames": f, "fixed_axis": False})
    pc.num_batches()
    r = radius_for_degree(n, 24)
    ref_nbh = amd.pc.BQNeighborhood(pc, pc, r); e = ref_nbh.num_edges()
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(c, c).to(DEV)
    conv.norm_neigh_dist_.fill_(1.0 / r), conv.norm_num_neighs_.fill_(n / e)
    x = torch.randn(n * f, c, device=DEV, requires_grad=True); g = torch.randn(n * f, c, device=DEV)
    holder = {}
    def step():
        x.grad = None; conv.zero_grad(set_to_none=True)
        nbh = amd.pc.BQNeighborhood(pc, pc, r, p_capacity=int(e * 1.25))
        out = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=nbh); out.backward(g)
        holder.update(nbh=nbh, out=out)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr): step()
    gr.replay(); torch.cuda.synchronize()
print("OK", variant, flush=True)
