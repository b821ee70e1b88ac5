"""Operands prepared once per step (round 6, include/se3conv.h `struct se3conv_prepared`): the host-side bookkeeping on the
CPU, the results on the GPU.  The packed geometry records of a cloud are kept on the cloud object and shared by every
convolution that touches it; they must follow the cloud's points and frames (storage AND version), never outlive them."""
from types import SimpleNamespace

import pytest
import torch

from conftest import rel_err


def _cloud(n=50, f=2, device="cpu"):
    g = torch.Generator().manual_seed(3)
    return SimpleNamespace(pts_=torch.rand(n, 3, generator=g).to(device),
                           local_frames_=torch.rand(n, f, 9, generator=g).to(device), n_frames_=f)


def test_holder_follows_storage_and_version():
    from se3conv3d_amd import ops

    pc = _cloud()
    h = ops.prepared_records(pc)
    assert h is ops.prepared_records(pc), "one holder per cloud object"
    h.bind(pc.pts_, pc.local_frames_)
    assert h.tensor.shape == (100, 16) and not h.valid
    h.valid = True
    assert h.bind(pc.pts_, pc.local_frames_).valid, "nothing changed: still valid"
    buf = h.tensor
    pc.pts_.add_(1.0)  # in place: same storage, another version
    assert not h.bind(pc.pts_, pc.local_frames_).valid and h.tensor is buf, "an in-place update invalidates, the buffer stays"
    h.valid = True
    pc.local_frames_ = pc.local_frames_.clone()  # another storage
    assert not h.bind(pc.pts_, pc.local_frames_).valid
    h.valid = True
    ops.invalidate_prepared(pc)
    assert not h.valid
    ops.invalidate_prepared(SimpleNamespace())  # a cloud that never had records: nothing to do


def test_objects_that_take_no_attribute_get_no_holder():
    from se3conv3d_amd import ops

    class Slotted:
        __slots__ = ("pts_", "local_frames_")

    assert ops.prepared_records(Slotted()) is None


DEV = "cuda:0"


@pytest.fixture()
def layer_case(built_library):
    import se3conv3d_amd as amd
    from oracle import se3conv_oracle as O

    amd.set_precision("bf16x3")
    g = torch.Generator().manual_seed(11)
    n, f, c = 700, 2, 64
    pts = torch.rand(n, 3, generator=g)
    bid = torch.zeros(n, dtype=torch.int32)
    frames = O.random_frames(n, f, g)
    r = O.radius_for_degree(n, 18)
    pc = amd.pc.PointcloudRotEquiv.from_frames(pts.to(DEV), bid.to(DEV), frames.to(DEV))
    nbh = amd.pc.BQNeighborhood(pc, pc, r)
    fac = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu")
    convs = [fac.create_conv_layer(c, c).to(DEV) for _ in range(2)]
    for cv in convs:
        cv.norm_neigh_dist_.fill_(1.0 / r)
        cv.norm_num_neighs_.fill_(n / nbh.num_edges())
    x = torch.randn(n * f, c, generator=g).to(DEV)
    go = torch.randn(n * f, c, generator=g).to(DEV)
    return amd, pc, nbh, convs, x, go


def _run(conv, pc, nbh, x, go):
    xg = x.clone().requires_grad_(True)
    for p in conv.parameters():
        p.grad = None
    out = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=xg, p_neighborhood=nbh)
    out.backward(go)
    return [out.detach().clone(), xg.grad.clone()] + [p.grad.clone() for p in conv.parameters()]


@pytest.mark.gpu
def test_shared_records_give_the_results_of_private_ones(layer_case):
    """Two layers on one cloud: the second one reads the records the first one built (same buffer, still valid), and both give
    bit for bit what they give when every call builds its own records in its workspace."""
    amd, pc, nbh, convs, x, go = layer_case
    from se3conv3d_amd import layers, ops

    holder = ops.prepared_records(pc)
    first = _run(convs[0], pc, nbh, x, go)
    assert holder.valid and layers._geometry_of(pc, pc, nbh).records_in is holder
    image = holder.tensor.clone()
    second = _run(convs[1], pc, nbh, x, go)
    assert holder.valid and torch.equal(holder.tensor, image), "the second layer did not rebuild the records"
    geom = layers._geometry_of(pc, pc, nbh)
    geom.records_in = geom.records_out = None  # private records: what se3conv_fwd / se3conv_bwd do
    try:
        for conv, want in ((convs[0], first), (convs[1], second)):
            for a, b in zip(_run(conv, pc, nbh, x, go), want):
                assert torch.equal(a, b)
    finally:
        geom.records_in = geom.records_out = holder


@pytest.mark.gpu
def test_records_follow_an_in_place_update_of_the_cloud(layer_case):
    amd, pc, nbh, convs, x, go = layer_case
    from se3conv3d_amd import ops

    base = _run(convs[0], pc, nbh, x, go)
    pc.pts_.mul_(1.0).add_(0.0)  # a no-op in value, a new version: the records are rebuilt and the results stay
    assert all(torch.equal(a, b) for a, b in zip(_run(convs[0], pc, nbh, x, go), base))
    # a rigid rotation of points and frames together leaves the operator's output unchanged (the
    # reference's random_rotate) -- only if the records are rebuilt from the rotated cloud
    q = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(5)))[0].to(DEV)
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    pc.pts_.copy_(pc.pts_ @ q.t())
    fr = pc.local_frames_.reshape(-1, 3, 3)
    pc.local_frames_.copy_((q @ fr).reshape(pc.local_frames_.shape))
    rotated = _run(convs[0], pc, nbh, x, go)
    assert rel_err(rotated[0], base[0]) < 5e-5 and rel_err(rotated[1], base[1]) < 5e-5
    assert ops.prepared_records(pc).valid


@pytest.mark.gpu
def test_raw_entry_points_with_and_without_prepared_buffers(layer_case):
    """se3conv_fwd_prepared / se3conv_bwd_prepared through the Python wrappers: feature words written by forward and read by
    backward, against the calls without any caller-kept buffer."""
    amd, pc, nbh, convs, x, go = layer_case
    from se3conv3d_amd import layers, ops

    conv = convs[0]
    geom = layers._geometry_of(pc, pc, nbh)
    a, b, w = conv.proj_axes_.detach(), conv.proj_biases_.detach(), conv.conv_weights_.detach()
    rho, nu = conv.norm_neigh_dist_, conv.norm_num_neighs_
    fw = torch.empty(x.numel(), dtype=torch.int32, device=DEV)
    out_p, _ = ops.se3conv_forward(geom, x, a, b, w, rho, nu, save_t=False, feat_words=fw)
    grads_p = ops.se3conv_backward(geom, x, a, b, w, rho, nu, None, go, feat_words=fw)
    holder = geom.records_in
    geom.records_in = geom.records_out = None
    try:
        out_0, _ = ops.se3conv_forward(geom, x, a, b, w, rho, nu, save_t=False)
        grads_0 = ops.se3conv_backward(geom, x, a, b, w, rho, nu, None, go)
    finally:
        geom.records_in = geom.records_out = holder
    assert torch.equal(out_p, out_0)
    for u, v in zip(grads_p, grads_0):
        assert torch.equal(u, v)
