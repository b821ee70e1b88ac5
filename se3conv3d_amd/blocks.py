"""Block-level glue of the reference's networks around the convolution (scope row f-3): drop path with frame-aware
batch ids, the residual connection with its learned gain, the point-cloud batch norm and the ResNetFormer block, with
the reference's attribute names so that a reference ``state_dict`` loads key for key.  On GPU tensors the row-wise
passes run as the library's fused kernels (csrc/glue.hip through ops.BatchNormTrain / SkipDropPath / BiasGelu): training
batch norm in 3 launches each way, skip + layer scale + drop-path gate in one, bias + GELU behind a bias-free GEMM in
one, the linear layers' weight gradients on the library's row-split TN GEMM (timings per block:
profiles/r03_block_glue_timing.txt); the forward and input-gradient C x C products stay BLAS GEMMs.
``FUSED = False`` (or SE3_BLOCKS_FUSED=0) keeps the plain torch formulation (the A/B of tools/time_block.py).

  DropPathPC      layers/DropPathPC.py:30-46      one keep / drop decision per batch element, scaled by 1 / keep_prob
  SkipConnection  layers/SkipConnection.py        drop_path(x * gamma_) + y, gamma_ initialised to 1e-6
  BatchNormPC     layers/BatchNormPC.py:22-32     BatchNorm1d(momentum=0.2) on the feature rows (the cloud is unused)
  ResNetFormer    layers/ResNetFormer.py:33-88    norm -> conv -> skip; norm -> linear (x2) -> GELU -> linear -> skip
"""
import os

import torch

from . import ops
from .layers import PreProcessModule

FUSED = os.environ.get("SE3_BLOCKS_FUSED", "1") != "0"


def _num_batches(p_pc) -> int:
    """Batch count as a host integer: the clouds of this package cache it (no device read-back per call, and none
    inside a graph capture); a foreign cloud object is asked once per call like the reference does."""
    return p_pc.num_batches() if hasattr(p_pc, "num_batches") else int(p_pc.batch_size_)


def _fused(t: torch.Tensor) -> bool:
    return FUSED and t.is_cuda and t.dim() == 2 and t.dtype == torch.float32


class DropPathPC(torch.nn.Module):
    def __init__(self, p_drop_prob):
        super().__init__()
        self.drop_prob_ = p_drop_prob

    def forward(self, p_x, p_pc):
        if self.drop_prob_ == 0.0 or not self.training:
            return p_x
        keep = 1.0 - self.drop_prob_
        # one uniform draw per batch element, keep where keep + u >= 1; rows of a cloud with frames carry the batch id
        # of their point (batch_ids_considering_frames_)
        gate = torch.floor(keep + torch.rand((_num_batches(p_pc),), dtype=p_x.dtype, device=p_x.device))
        ids = getattr(p_pc, "batch_ids_considering_frames_", None)
        if ids is None:
            ids = p_pc.batch_ids_
        return p_x.div(keep) * gate.index_select(0, ids.to(torch.int64)).reshape(-1, 1)


class SkipConnection(torch.nn.Module):
    def __init__(self, p_drop_prob, p_num_features, p_init_gamma=1e-6):
        super().__init__()
        self.drop_path_ = DropPathPC(p_drop_prob)
        self.gamma_ = torch.nn.Parameter(torch.full((1, p_num_features), float(p_init_gamma)))

    def forward(self, p_x, p_y, p_pc):
        # drop_prob >= 1 (keep <= 0): the reference divides by zero (DropPathPC.py:45) -- the plain formulation reproduces
        # that instead of handing the kernel keep = 0, which it reads as "the gate is the factor itself"
        if not _fused(p_x) or p_x.shape != p_y.shape or (self.training and self.drop_path_.drop_prob_ >= 1.0):
            return self.drop_path_(p_x * self.gamma_, p_pc) + p_y
        gate = ids = None
        keep = 0.0
        if self.drop_path_.drop_prob_ != 0.0 and self.training:
            keep = 1.0 - self.drop_path_.drop_prob_
            # the same draw as DropPathPC (one uniform per batch element); floor(keep + u) / keep is evaluated in the kernel
            gate = torch.rand((_num_batches(p_pc),), dtype=p_x.dtype, device=p_x.device)
            ids = getattr(p_pc, "batch_ids_considering_frames_", None)
            if ids is None:
                ids = p_pc.batch_ids_
        return ops.SkipDropPath.apply(p_x, p_y, self.gamma_, gate, ids, keep)


class NormLayerPC(torch.nn.Module):
    def __init__(self, p_num_features):
        super().__init__()
        self.num_feats_ = p_num_features


class BatchNormPC(NormLayerPC):
    def __init__(self, p_num_features):
        super().__init__(p_num_features)
        self.layer_ = torch.nn.BatchNorm1d(p_num_features, momentum=0.2)

    def forward(self, p_x, p_pc):
        bn = self.layer_
        if not (_fused(p_x) and bn.training and bn.track_running_stats and bn.momentum is not None):
            return bn(p_x)
        return ops.BatchNormTrain.apply(p_x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                        bn.num_batches_tracked)


class Block(PreProcessModule):
    def __init__(self, p_in_features, p_out_features, p_conv_fact, p_norm_layer, p_path_drop_prob):
        super().__init__()
        self.feat_input_size_ = p_in_features
        self.feat_output_size_ = p_out_features


class ResNetFormer(Block):
    """Same-level residual block: the convolution keeps the width, the point-wise MLP widens by 2 and maps to the
    output width; a linear skip appears only when the widths differ."""

    def __init__(self, p_in_features, p_out_features, p_conv_fact, p_norm_layer, p_path_drop_prob):
        super().__init__(p_in_features, p_out_features, p_conv_fact, p_norm_layer, p_path_drop_prob)
        c_in, c_out = self.feat_input_size_, self.feat_output_size_
        self.act_func_ = torch.nn.GELU()
        self.feat_scale_factor_ = 2
        self.spatial_conv_ = p_conv_fact.create_conv_layer(c_in, c_in)
        self.norm_1_ = p_norm_layer(c_in)
        self.norm_2_ = p_norm_layer(c_in)
        self.linear_1_ = torch.nn.Linear(c_in, c_in * self.feat_scale_factor_)
        self.linear_2_ = torch.nn.Linear(c_in * self.feat_scale_factor_, c_out)
        self.skip_path_1_ = SkipConnection(p_path_drop_prob, c_in)
        self.skip_path_2_ = SkipConnection(p_path_drop_prob, c_out)
        if c_in != c_out:
            self.skip_conv_ = torch.nn.Linear(c_in, c_out)

    def forward(self, p_pc_in, p_in_features, p_neighborhood):
        x = self.spatial_conv_(p_pc_in=p_pc_in, p_pc_out=p_pc_in, p_in_features=self.norm_1_(p_in_features, p_pc_in),
                               p_neighborhood=p_neighborhood)
        x = self.skip_path_1_(x, p_in_features, p_pc_in)
        h = self.norm_2_(x, p_pc_in)
        if _fused(h):
            # bias + GELU in one pass behind a bias-free GEMM; the linear layers' weight gradients on the row-split GEMM
            h = ops.BiasGelu.apply(ops.Linear.apply(h, self.linear_1_.weight, None), self.linear_1_.bias)
            y = ops.Linear.apply(h, self.linear_2_.weight, self.linear_2_.bias)
            skip = x
            if self.feat_input_size_ != self.feat_output_size_:
                skip = ops.Linear.apply(x, self.skip_conv_.weight, self.skip_conv_.bias)
        else:
            h = self.act_func_(self.linear_1_(h))
            y = self.linear_2_(h)
            skip = self.skip_conv_(x) if self.feat_input_size_ != self.feat_output_size_ else x
        return self.skip_path_2_(y, skip, p_pc_in)
