"""GPU: the row-wise glue kernels of a block (scope row f-3: csrc/glue.hip through ops.BatchNormTrain /
SkipDropPath / BiasGelu) against the plain torch fp32 formulation of the same steps -- what the reference's
BatchNormPC / SkipConnection / DropPathPC / ResNetFormer run (layers/ResNetFormer.py:64-88).  The block as a whole is
pinned by the reference fixture in test_gpu_network.py::test_reference_resnetformer_block_runs_unchanged (fused path);
here: odd channel counts, single rows, empty tensors, the drop-path gate with frame-aware batch ids, and fused ==
unfused on a whole block with drop path on.  Tolerance 2e-6 relative (fp32 element-wise maths, fp64 channel sums)."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 2e-6


@pytest.fixture(scope="module")
def amd(built_library):
    import se3conv3d_amd as amd
    amd.set_precision("bf16x3")
    return amd


@pytest.mark.parametrize("rows,c", [(1, 32), (7, 1), (1000, 3), (1000, 48), (70000, 64), (513, 260), (0, 16)])
def test_batch_norm_train_matches_torch(amd, rows, c):
    torch.manual_seed(rows + c)
    x = (torch.randn(rows, c, device=DEV) * 3 + 5).requires_grad_(True)   # mean >> 0: the variance must not cancel
    w = torch.randn(c, device=DEV).requires_grad_(True)
    b = torch.randn(c, device=DEV).requires_grad_(True)
    g = torch.randn(rows, c, device=DEV)
    rm, rv = torch.randn(c, device=DEV), torch.rand(c, device=DEV) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    tracked = torch.full((), 41, dtype=torch.int64, device=DEV)   # BatchNorm1d.num_batches_tracked: += 1 inside the op
    y = amd.ops.BatchNormTrain.apply(x, w, b, rm, rv, 0.2, 1e-5, tracked)
    assert int(tracked) == 42
    if rows == 0:
        assert y.shape == (0, c)
        return
    y.backward(g)
    x2, w2, b2 = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    if rows == 1:   # torch refuses one value per channel in training mode; the formula still holds (var = 0)
        y_ref = (x2 - x2) * w2 + b2
        assert torch.allclose(y, y_ref, atol=1e-5)
        return
    y_ref = torch.nn.functional.batch_norm(x2, rm_ref, rv_ref, w2, b2, True, 0.2, 1e-5)
    y_ref.backward(g)
    assert rel_err(y, y_ref) < TOL
    assert rel_err(x.grad, x2.grad) < 2e-5          # dx cancels: dy - mean(dy) - xhat mean(dy xhat)
    assert rel_err(w.grad, w2.grad) < 1e-5 and rel_err(b.grad, b2.grad) < 1e-5
    assert torch.allclose(rm, rm_ref, rtol=1e-5, atol=1e-6) and torch.allclose(rv, rv_ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("rows,c,frames,batches", [(1000, 64, 2, 3), (999, 48, 1, 4), (4096, 3, 4, 2), (0, 32, 2, 1)])
def test_skip_with_drop_path_gate_matches_torch(amd, rows, c, frames, batches):
    torch.manual_seed(3)
    rows = rows // frames * frames
    x = torch.randn(rows, c, device=DEV, requires_grad=True)
    y = torch.randn(rows, c, device=DEV, requires_grad=True)
    gamma = torch.randn(1, c, device=DEV, requires_grad=True)
    g = torch.randn(rows, c, device=DEV)
    pt_batch = torch.sort(torch.randint(0, batches, (rows // frames,), device=DEV)).values.to(torch.int32)
    row_batch = pt_batch.repeat_interleave(frames)            # batch_ids_considering_frames_
    keep = 0.7
    u = torch.rand(batches, device=DEV)
    gate = torch.floor(keep + u) / keep
    for use_gate in (True, "in-kernel", False):   # "in-kernel": the op gets the uniform draws and keep_prob
        for t in (x, y, gamma):
            t.grad = None
        if use_gate == "in-kernel":
            out = amd.ops.SkipDropPath.apply(x, y, gamma, u, row_batch, keep)
        else:
            out = amd.ops.SkipDropPath.apply(x, y, gamma, gate if use_gate else None, row_batch if use_gate else None)
        x2, y2, ga2 = (t.detach().clone().requires_grad_(True) for t in (x, y, gamma))
        ref = x2 * ga2
        if use_gate:
            ref = ref * gate.index_select(0, row_batch.to(torch.int64)).reshape(-1, 1)
        ref = ref + y2
        if rows == 0:
            assert out.shape == ref.shape
            continue
        out.backward(g)
        ref.backward(g)
        assert rel_err(out, ref) < TOL and rel_err(x.grad, x2.grad) < TOL and torch.equal(y.grad, y2.grad)
        assert rel_err(gamma.grad, ga2.grad) < 1e-5 and gamma.grad.shape == (1, c)


@pytest.mark.parametrize("rows,c", [(1000, 128), (33, 6), (70000, 64)])
def test_bias_gelu_matches_torch(amd, rows, c):
    torch.manual_seed(4)
    z = (torch.randn(rows, c, device=DEV) * 2).requires_grad_(True)
    b = torch.randn(c, device=DEV, requires_grad=True)
    g = torch.randn(rows, c, device=DEV)
    out = amd.ops.BiasGelu.apply(z, b)
    out.backward(g)
    z2, b2 = z.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.gelu(z2 + b2)
    ref.backward(g)
    assert rel_err(out, ref) < TOL and rel_err(z.grad, z2.grad) < TOL and rel_err(b.grad, b2.grad) < 1e-5


@pytest.mark.parametrize("rows,n_in,n_out,bias", [(131072, 64, 128, True), (18414, 128, 64, True), (2652, 64, 64, False),
                                                  (1, 32, 5, True), (777, 13, 130, True), (0, 16, 8, True)])
def test_linear_weight_gradient_on_the_row_split_gemm_matches_torch(amd, rows, n_in, n_out, bias):
    """ops.Linear: forward / input gradient are the BLAS GEMMs, the weight gradient is se3_linear_wgrad (fp32 MFMA, rows
    split over the chip, fixed-order reduction) -- against torch.nn.functional.linear's own backward in fp64."""
    from se3conv3d_amd import ops
    torch.manual_seed(rows + n_in)
    x = torch.randn(rows, n_in, device=DEV, requires_grad=True)
    w = torch.randn(n_out, n_in, device=DEV, requires_grad=True)
    b = torch.randn(n_out, device=DEV, requires_grad=True) if bias else None
    g = torch.randn(rows, n_out, device=DEV)
    y = ops.Linear.apply(x, w, b)
    y.backward(g)
    xd, wd, gd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True), g.double()
    bd = b.detach().double().requires_grad_(True) if bias else None
    yd = torch.nn.functional.linear(xd, wd, bd)
    yd.backward(gd)
    assert y.shape == (rows, n_out) and w.grad.shape == (n_out, n_in)
    if rows == 0:
        assert float(w.grad.abs().max()) == 0.0
        return
    assert rel_err(y, yd.float()) < 1e-5
    assert rel_err(w.grad, wd.grad.float()) < 1e-5
    assert rel_err(x.grad, xd.grad.float()) < 1e-5
    if bias:
        assert rel_err(b.grad, bd.grad.float()) < 1e-5


def test_fused_block_equals_torch_block_with_drop_path(amd):
    """A whole ResNetFormer in training mode with drop path on: the fused formulation and the plain torch one draw the
    same gates from the same seed and agree on output, gradients and batch-norm statistics."""
    from se3conv3d_amd import blocks
    from se3conv3d_amd.workloads import radius_for_degree

    torch.manual_seed(0)
    n_el, nb, f, c_in, c_out = 1500, 3, 2, 32, 48
    pts = torch.rand(n_el * nb, 3, device=DEV)
    bid = torch.arange(nb, device=DEV, dtype=torch.int32).repeat_interleave(n_el)
    pc = amd.pc.PointcloudRotEquiv(pts, bid, {"pca": False, "n_frames": f, "fixed_axis": False})
    nbh = amd.pc.BQNeighborhood(pc, pc, radius_for_degree(n_el, 16))
    blk = amd.ResNetFormer(c_in, c_out, amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu"), amd.BatchNormPC, 0.4).to(DEV)
    blk.spatial_conv_.norm_neigh_dist_.fill_(1.0 / radius_for_degree(n_el, 16))
    blk.spatial_conv_.norm_num_neighs_.fill_(1.0 / 16)
    with torch.no_grad():
        blk.skip_path_1_.gamma_.fill_(0.5), blk.skip_path_2_.gamma_.fill_(0.7)   # 1e-6 would hide the residual branches
    state0 = {k: v.clone() for k, v in blk.state_dict().items()}
    x0 = torch.randn(n_el * nb * f, c_in, device=DEV)
    g = torch.randn(n_el * nb * f, c_out, device=DEV)
    results = []
    for fused in (True, False):
        blocks.FUSED = fused
        try:
            blk.load_state_dict(state0)
            blk.train()
            blk.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            torch.manual_seed(123)                      # same drop-path draws
            out = blk(pc, x, nbh)
            out.backward(g)
            results.append((out.detach(), x.grad, {k: p.grad.clone() for k, p in blk.named_parameters()},
                            {k: v.clone() for k, v in blk.state_dict().items() if "running" in k or "tracked" in k}))
        finally:
            blocks.FUSED = True
    (o_a, dx_a, gp_a, st_a), (o_b, dx_b, gp_b, st_b) = results
    assert rel_err(o_a, o_b) < 1e-5 and rel_err(dx_a, dx_b) < 1e-4
    for k in gp_a:
        assert rel_err(gp_a[k], gp_b[k]) < 1e-4, k
    for k in st_a:
        assert torch.allclose(st_a[k].float(), st_b[k].float(), rtol=1e-5, atol=1e-6), k
