"""GPU: clouds beyond what one GEMM launch can address with 32-bit offsets (> 262 144 rows of 64 x 32 values in the
3-byte format's GEMMs, > 524 288 rows at 4 bytes): the library walks such products row block by row block
(csrc/gemm_bf16.hip).  No oracle finishes at this size; the property used instead: a batch made of two copies of one
body, far apart, must give each copy the rows the body gets alone, and parameter gradients twice the body's."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _body(amd, n, frames, seed):
    torch.manual_seed(seed)
    # coordinates on a 2^-20 grid: shifting the copy by 8 then is exact (8 + k 2^-20 is a float), so both copies have the same
    # differences, the same neighbour sets and the same descriptors bit for bit
    pts = torch.round(torch.rand(n, 3, device=DEV) * 2**20) / 2**20
    pc = amd.pc.PointcloudRotEquiv(pts, torch.zeros(n, dtype=torch.int32, device=DEV),
                                   {"pca": False, "n_frames": frames, "fixed_axis": False})
    return pts, pc.local_frames_.clone()


def _run(amd, pts, batch, frames_t, conv, x, g, radius):
    pc = amd.pc.PointcloudRotEquiv(pts, batch, {"pca": False, "n_frames": frames_t.shape[1], "fixed_axis": False})
    pc.local_frames_ = frames_t
    nbh = amd.pc.BQNeighborhood(pc, pc, radius)
    x = x.detach().clone().requires_grad_(True)
    conv.zero_grad(set_to_none=True)
    out = conv(p_pc_in=pc, p_pc_out=pc, p_in_features=x, p_neighborhood=nbh)
    out.backward(g)
    return out.detach(), x.grad.detach(), [p.grad.detach().clone() for p in conv.parameters()]


@pytest.mark.parametrize("precision", ["bf16x3", "fp32", "bf16x3_t16"])   # fp32: beyond the reach of its buffer-load forms too
@pytest.mark.parametrize("n,frames", [(140_000, 2)])         # 2 x 280 000 = 560 000 rows: above both row limits
def test_two_copies_of_a_body_equal_the_body_alone(n, frames, precision):
    import se3conv3d_amd as amd
    from se3conv3d_amd.workloads import radius_for_degree
    amd.set_precision(precision)
    c = 64
    pts, fr = _body(amd, n, frames, 5)
    radius = radius_for_degree(n, 24)
    conv = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu").create_conv_layer(c, c).to(DEV)
    conv.norm_neigh_dist_.fill_(1.0 / radius)
    conv.norm_num_neighs_.fill_(1.0 / 24)
    conv.eval()   # no EMA update of the two normalisers: both runs see the same values
    torch.manual_seed(6)
    x = torch.randn(n * frames, c, device=DEV)
    g = torch.randn(n * frames, c, device=DEV)
    zeros = torch.zeros(n, dtype=torch.int32, device=DEV)
    out1, dx1, gp1 = _run(amd, pts, zeros, fr, conv, x, g, radius)
    pts2 = torch.cat([pts, pts + 8.0])
    out2, dx2, gp2 = _run(amd, pts2, torch.cat([zeros, zeros + 1]), torch.cat([fr, fr]), conv, torch.cat([x, x]),
                          torch.cat([g, g]), radius)
    rows = n * frames
    assert out2.shape[0] == 2 * rows
    for part in (out2[:rows], out2[rows:]):
        assert rel_err(part, out1) < 2e-6
    for part in (dx2[:rows], dx2[rows:]):
        assert rel_err(part, dx1) < 2e-6
    for a, b in zip(gp2, gp1):
        assert rel_err(a, 2.0 * b) < 2e-5
    amd.set_precision("bf16x3")
