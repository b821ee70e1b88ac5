"""se3conv3d_amd -- MI355X (gfx950) implementation of the PNEConvLayerRotEquiv hot path.

Layout mirrors the slice of ``point_cloud_lib`` that the path touches:

  se3conv3d_amd.layers   PNEConvLayerRotEquiv(+Factory), PNEConvLayer(+Factory), IConvLayer(+Factory), PreProcessModule
  se3conv3d_amd.blocks   ResNetFormer, SkipConnection, DropPathPC, BatchNormPC (torch glue around the conv)
  se3conv3d_amd.pc       Pointcloud(RotEquiv), BQNeighborhood, PointHierarchy(RotEquiv), frame sampling
  se3conv3d_amd.ops      FeatBasisProj, BallQuery, ComputeKeys, SE3ConvFunction (ctypes -> C ABI)
  se3conv3d_amd.csrc     HIP kernels + the extern "C" boundary (include/se3conv.h)

Importing the package does not load the HIP library; the first op call does and raises if it has
not been built (``python -m se3conv3d_amd.build``).
"""
from . import blocks, layers, ops, pc  # noqa: F401
from .blocks import BatchNormPC, DropPathPC, ResNetFormer, SkipConnection  # noqa: F401
from .layers import (IConvLayer, IConvLayerFactory, PNEConvLayer, PNEConvLayerFactory,  # noqa: F401
                     PNEConvLayerRotEquiv, PNEConvLayerRotEquivFactory, PreProcessModule)
from .ops import (BallQuery, ComputeKeys, FeatBasisProj, KNNQuery, SE3ConvFunction,  # noqa: F401
                  get_precision, set_precision)
from .pc import (BQNeighborhood, KnnNeighborhood, Pointcloud, PointcloudRotEquiv, PointHierarchy,  # noqa: F401
                 PointHierarchyRotEquiv)

__version__ = "0.1.0"
