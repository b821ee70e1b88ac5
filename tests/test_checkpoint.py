"""CPU: checkpoint round trip in the layout the reference's task scripts write and read (scope row f-4):
``save_checkpoint`` stores ``model.state_dict()`` under ``params_dict`` next to the optimizer / scheduler / epoch
entries and ``torch.save``s the dictionary (tasks/SemSeg/train_dfaust_rot.py:411-432); the test scripts ``torch.load``
it with ``map_location="cpu"`` and ``load_state_dict`` the entry (test_dfaust_rot.py:75,257).  No kernel runs here --
only parameter / buffer names, shapes and values have to survive."""
import torch

import se3conv3d_amd as amd
from se3conv3d_amd import blocks


class TinyModel(torch.nn.Module):
    """An equivariant and a plain convolution plus a ResNetFormer block, named like members of the reference's models."""

    def __init__(self):
        super().__init__()
        eq = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu")
        std = amd.PNEConvLayerFactory(3, 32, "mlp_gelu")
        self.first_conv_ = eq.create_conv_layer(1, 32)
        self.plain_conv_ = std.create_conv_layer(3, 16)
        self.block_ = blocks.ResNetFormer(32, 64, eq, blocks.BatchNormPC, 0.1)


def test_reference_style_checkpoint_round_trip(tmp_path):
    torch.manual_seed(0)
    model = TinyModel()
    with torch.no_grad():                       # a "trained" state: EMA buffers and every parameter away from their init
        for p in model.parameters():
            p.add_(torch.randn_like(p) * 0.1)
        for name, b in model.named_buffers():
            if b.dtype.is_floating_point:
                b.add_(torch.rand_like(b) + 0.5)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    save_dict = {"train_dict": {"lr": 1e-3}, "dataset_dict": {}, "model_dict": {"name": "tiny"},
                 "params_dict": model.state_dict(), "optimizer_dict": opt.state_dict(), "scheduler_dict": {},
                 "best_mIoU": 0.5, "epoch": 7}
    path = tmp_path / "ckpt.pth"
    torch.save(save_dict, path)

    dictionary = torch.load(path, map_location="cpu")
    fresh = TinyModel()
    missing, unexpected = fresh.load_state_dict(dictionary["params_dict"], strict=True)
    assert not missing and not unexpected
    assert dictionary["epoch"] == 7
    want = model.state_dict()
    got = fresh.state_dict()
    assert list(got) == list(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k
    # the names a reference checkpoint carries for the convolutions (PNEConvLayer.py:79-88, IConvLayer.py:33-36)
    for prefix in ("first_conv_", "plain_conv_", "block_.spatial_conv_"):
        for leaf in ("proj_axes_", "proj_biases_", "conv_weights_", "norm_neigh_dist_", "norm_num_neighs_"):
            assert f"{prefix}.{leaf}" in want, f"{prefix}.{leaf}"


def test_state_dict_layout_equals_the_reference_modules():
    """Names AND shapes of every parameter / buffer against a key list generated from the reference's own modules
    (tools/gen_golden.py `keys`: PNEConvLayerRotEquiv, PNEConvLayer, BatchNormPC, SkipConnection, ResNetFormer with and
    without its `skip_conv_`), then a strict load of a state dict carrying exactly those keys."""
    import json
    import os

    from conftest import GOLDEN

    with open(os.path.join(GOLDEN, "state_dict_keys.json")) as fh:
        ref = json.load(fh)
    eq = amd.PNEConvLayerRotEquivFactory(9, 32, "mlp_gelu")
    std = amd.PNEConvLayerFactory(3, 32, "mlp_gelu")
    ours = {
        "PNEConvLayerRotEquiv(9,16,24,32)": eq.create_conv_layer(16, 24),
        "PNEConvLayer(3,16,24,32)": std.create_conv_layer(16, 24),
        "BatchNormPC(24)": blocks.BatchNormPC(24),
        "SkipConnection(24)": blocks.SkipConnection(0.1, 24),
        "ResNetFormer(16,24)": blocks.ResNetFormer(16, 24, eq, blocks.BatchNormPC, 0.1),
        "ResNetFormer(24,24)": blocks.ResNetFormer(24, 24, eq, blocks.BatchNormPC, 0.0),
    }
    assert sorted(ours) == sorted(ref)
    for name, module in ours.items():
        got = {k: list(v.shape) for k, v in module.state_dict().items()}
        assert got == ref[name], name
        # a checkpoint with the reference's keys (values arbitrary) loads strictly
        fake = {k: torch.full(shape, 0.25) if k.split(".")[-1] != "num_batches_tracked" else torch.tensor(3)
                for k, shape in ref[name].items()}
        missing, unexpected = module.load_state_dict(fake, strict=True)
        assert not missing and not unexpected, name
