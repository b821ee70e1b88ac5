"""Tensor-level wrappers and autograd Functions over the C ABI (include/se3conv.h).

PyTorch is used for device memory, the current stream and autograd bookkeeping only; every
computation below runs in libse3conv_hip.so.  The class names / call signatures mirror the
reference's ``point_cloud_lib.custom_ops`` (FeatBasisProj.py, BallQuery.py, ComputeKeys.py).
"""
from __future__ import annotations

import ctypes as C
import functools
from dataclasses import dataclass, field
from typing import Callable, Optional, Tuple

import torch

from . import _lib
from ._lib import Se3Shape


# ------------------------------------------------------------------------------------- precision
# Arithmetic of the operator's contractions (include/se3conv.h, SE3_PRECISION_*): "bf16x3" = split-bf16
# MFMA with fp32 accumulate (~1e-5 rel. to the fp32 reference, the default), "fp32" = exact-fp32 MFMA.
import os as _os

_precision = _os.environ.get("SE3CONV_PRECISION", "bf16x3")
if _precision not in _lib.PRECISIONS:
    raise ValueError(f"SE3CONV_PRECISION={_precision!r}; expected one of {sorted(_lib.PRECISIONS)}")


def set_precision(name: str) -> None:
    global _precision
    if name not in _lib.PRECISIONS:
        raise ValueError(f"precision {name!r}; expected one of {sorted(_lib.PRECISIONS)}")
    _precision = name


def get_precision() -> str:
    return _precision


# --------------------------------------------------------------------------------------- helpers
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device) -> C.c_void_p:
    """The current HIP stream of ``device``, which must be the current device: the library launches on the stream it
    is handed and never switches devices itself, so tensors on another GPU than the current one would be launched
    on the wrong device's stream.  (The raw query: ``torch.cuda.current_stream()`` builds a Python object through
    several device-index lookups, ~20 us per call.)"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx != torch.cuda.current_device():
        raise ValueError(f"tensors live on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}: "
                         "call inside `with torch.cuda.device(tensor.device):` (one process per GPU is the intended use)")
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(idx))
    return C.c_void_p(torch.cuda.current_stream(idx).cuda_stream)


def _ptr(t: Optional[torch.Tensor], dtype: torch.dtype, name: str, device=None) -> C.c_void_p:
    """Device pointer of a contiguous CUDA tensor of the given dtype (ValueError otherwise)."""
    if t is None:
        return C.c_void_p(0)
    # (the good case first: an eager neighbourhood build passes ~11 tensors per query and is bound by the host,
    # tools/host_split_neighbourhoods.py)
    if isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous() and (device is None or t.device == device):
        return C.c_void_p(t.data_ptr())
    if not isinstance(t, torch.Tensor):
        raise ValueError(f"{name}: expected a tensor, got {type(t)}")
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback), got {t.device}")
    if device is not None and t.device != device:
        raise ValueError(f"{name}: on {t.device}, expected {device}")
    if t.dtype != dtype:
        raise ValueError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _as(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """Contiguous copy-if-needed in ``dtype`` (what the reference wrappers do with ``.to``)."""
    if t.dtype == dtype and not t.requires_grad and t.is_contiguous():
        return t  # (nothing to convert or to detach: three tensor objects less per argument)
    return t.detach().to(dtype).contiguous()


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------- grid keys (a10, f-2)
def compute_keys(pts, batch_ids, aabb_min, num_cells, cell_size) -> torch.Tensor:
    """``point_cloud_lib_ops.compute_keys`` (reference ComputeKeys.py / compute_keys.cu:75-124)."""
    lib = _lib.load()
    pts = _as(pts, torch.float32)
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("compute_keys: only 3-D points are supported")
    dev = pts.device
    n = pts.shape[0]
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    b = _as(batch_ids, torch.int32)
    mn = _as(aabb_min, torch.float32)
    nc = _as(num_cells, torch.int32).to(dev)
    cs = _as(cell_size, torch.float32).to(dev)
    _lib.check(lib.se3_compute_keys(_ptr(pts, torch.float32, "pts"), _ptr(b, torch.int32, "batch_ids", dev),
                                    _ptr(mn, torch.float32, "aabb_min", dev), _ptr(nc, torch.int32, "num_cells"),
                                    _ptr(cs, torch.float32, "cell_size"), n, _ptr(keys, torch.int64, "keys"),
                                    _stream(dev)), "se3_compute_keys")
    return keys


class ComputeKeys(torch.autograd.Function):
    """Drop-in for ``point_cloud_lib.custom_ops.ComputeKeys``."""

    @staticmethod
    def forward(ctx, p_pts, p_batch_ids, p_aabb_min, p_grid_size, p_cell_size):
        return compute_keys(p_pts, p_batch_ids, p_aabb_min, p_grid_size, p_cell_size)

    @staticmethod
    def backward(ctx, grad):
        return None, None, None, None, None


# ------------------------------------------------------------------------------- ball query (a10)
def batch_aabb(pts, batch_ids, n_batches: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-batch bounding boxes ``(min [B,3], max [B,3])`` (the reference's scatter_min / scatter_max calls)."""
    lib = _lib.load()
    pts = _as(pts, torch.float32)
    b = _as(batch_ids, torch.int32)
    if n_batches is None:
        # one tiny sync, the same one the reference pays with `torch::amax(...).item()` (ball_query.cu:46)
        n_batches = int(b.max().item()) + 1 if b.numel() else 1
    mn = torch.empty((n_batches, 3), dtype=torch.float32, device=pts.device)
    mx = torch.empty((n_batches, 3), dtype=torch.float32, device=pts.device)
    _lib.check(lib.se3_batch_aabb(_ptr(pts, torch.float32, "pts"), _ptr(b, torch.int32, "batch_ids", pts.device),
                                  pts.shape[0], n_batches, _ptr(mn, torch.float32, "aabb_min"),
                                  _ptr(mx, torch.float32, "aabb_max"), _stream(pts.device)), "se3_batch_aabb")
    return mn, mx


@functools.lru_cache(maxsize=4096)
def _ball_query_sizes(n_src: int, n_dst: int) -> Tuple[bool, int, int]:
    """(needs a cell grid, workspace bytes, grid bytes) of a query of this size: three library calls, asked once per size."""
    lib = _lib.load()
    return (bool(lib.se3_ball_query_needs_grid(n_src)), int(lib.se3_ball_query_workspace_bytes(n_src, n_dst)),
            int(lib.se3_ball_query_grid_bytes(n_src)))


def ball_query_needs_grid(n_src: int) -> bool:
    """Whether a query against ``n_src`` source points uses the cell grid (small source sets are scanned whole)."""
    return bool(_lib.load().se3_ball_query_needs_grid(int(n_src)))


def _batch_aabb_min_and_cells(pts_src, batch_src, radius: float, n_batches: Optional[int] = None, box=None):
    """Grid parameters exactly as BallQuery.forward builds them (BallQuery.py:34-38), on device, one library call.
    ``box`` = the source cloud's ``batch_aabb`` result when the caller has it (``Pointcloud.aabb()``: computed once per
    cloud, a cloud is the source of several queries per step): one tiny launch instead of a pass over the points."""
    lib = _lib.load()
    if n_batches is None:
        # one tiny sync, the same one the reference pays with `torch::amax(...).item()` (ball_query.cu:46)
        n_batches = int(batch_src.max().item()) + 1 if batch_src.numel() else 1
    dev = pts_src.device
    if box is not None and box[0].shape == (n_batches, 3) and box[0].device == dev:
        mn = torch.empty((n_batches, 3), dtype=torch.float32, device=dev)
        num_cells = torch.empty(3, dtype=torch.int32, device=dev)
        _lib.check(lib.se3_ball_query_grid_from_box(_ptr(box[0], torch.float32, "box_min", dev), _ptr(box[1], torch.float32, "box_max", dev),
                                                    n_batches, float(radius), _ptr(mn, torch.float32, "aabb_min"),
                                                    _ptr(num_cells, torch.int32, "num_cells"), _stream(dev)),
                   "se3_ball_query_grid_from_box")
        return mn, num_cells
    box = torch.empty((2, n_batches, 3), dtype=torch.float32, device=dev)  # [0] = shifted minimum, [1] = scratch
    num_cells = torch.empty(3, dtype=torch.int32, device=dev)
    _lib.check(lib.se3_ball_query_grid(
        _ptr(pts_src, torch.float32, "pts_src"), _ptr(batch_src, torch.int32, "batch_src", dev), pts_src.shape[0],
        n_batches, float(radius), C.c_void_p(box[0].data_ptr()), C.c_void_p(box[1].data_ptr()),
        _ptr(num_cells, torch.int32, "num_cells"), _stream(dev)), "se3_ball_query_grid")
    return box[0], num_cells


def ball_query(pts_src, pts_dst, batch_src, batch_dst, radius: float,
               n_batches: Optional[int] = None, src_box=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Radius neighbours: ``neighbors [E,2] int32`` (col0 sample, col1 source; grouped by sample)
    and ``ends [M] int32`` (inclusive end offsets).  Two-phase C ABI, one host sync for E."""
    lib = _lib.load()
    pts_src = _as(pts_src, torch.float32)
    pts_dst = _as(pts_dst, torch.float32)
    if pts_src.dim() != 2 or pts_src.shape[1] != 3 or pts_dst.dim() != 2 or pts_dst.shape[1] != 3:
        raise ValueError("ball_query: only [N,3] point sets are supported")
    if not (radius > 0):
        raise ValueError("ball_query: radius must be positive")
    dev = pts_src.device
    bs = _as(batch_src, torch.int32)
    bd = _as(batch_dst, torch.int32)
    n_src, n_dst = pts_src.shape[0], pts_dst.shape[0]
    if n_dst == 0 or n_src == 0:
        return torch.zeros((0, 2), dtype=torch.int32, device=dev), torch.zeros(n_dst, dtype=torch.int32, device=dev)
    ends = torch.empty(n_dst, dtype=torch.int32, device=dev)  # every entry is written by the count phase
    # small source sets are searched all-pairs by the library: no boxes / cell grid to prepare
    mn, nc = _batch_aabb_min_and_cells(pts_src, bs, radius, n_batches, src_box) if lib.se3_ball_query_needs_grid(n_src) else (None, None)
    nbytes = lib.se3_ball_query_workspace_bytes(n_src, n_dst)
    ws = _workspace(nbytes, dev)
    f32, i32 = torch.float32, torch.int32
    _lib.check(lib.se3_ball_query_count(
        _ptr(pts_src, f32, "pts_src"), _ptr(pts_dst, f32, "pts_dst", dev), _ptr(bs, i32, "batch_src", dev),
        _ptr(bd, i32, "batch_dst", dev), _ptr(mn, f32, "aabb_min"), _ptr(nc, i32, "num_cells"), float(radius),
        n_src, n_dst, C.c_void_p(ws.data_ptr()), ws.numel(), _ptr(ends, i32, "ends"), _stream(dev)),
        "se3_ball_query_count")
    n_edges = int(ends[-1].item())
    neighbors = torch.empty((n_edges, 2), dtype=torch.int32, device=dev)
    _lib.check(lib.se3_ball_query_store(
        _ptr(pts_dst, f32, "pts_dst"), _ptr(bd, i32, "batch_dst"), float(radius), n_src, n_dst,
        C.c_void_p(ws.data_ptr()), ws.numel(), _ptr(ends, i32, "ends"), n_edges,
        _ptr(neighbors, i32, "neighbors"), _stream(dev)), "se3_ball_query_store")
    return neighbors, ends


SHARED_GRIDS = _os.environ.get("SE3_SHARED_GRIDS", "1") != "0"  # (A/B switch, tools/time_faust_geometry.py)


class SourceGrids:
    """The cell grids of one source cloud, one per search radius (``se3_ball_query_bounded_shared``): the queries of a step
    that search the cloud with the same radius sort it once.  Kept on the cloud object (``source_grids``); a grid is
    rebuilt when the cloud's points, batch ids or batch count are no longer the ones it was built from."""

    __slots__ = ("grids", "params")

    def __init__(self):
        self.grids = {}
        self.params = {}  # per radius: (shifted box minima, cell counts) of the grid (se3_ball_query_grid_from_box)

    def slot(self, pts, batch_ids, radius, n_batches, nbytes):
        """``(buffer, valid)`` for this radius; ``valid`` says the buffer already holds the grid (the call that gets
        ``False`` builds it)."""
        key = (pts.data_ptr(), pts._version, batch_ids.data_ptr(), batch_ids._version, int(pts.shape[0]), int(n_batches or 0),
               str(pts.device))
        hit = self.grids.get(float(radius))
        if hit is not None and hit[0] == key and hit[1].numel() >= nbytes:
            return hit[1], True
        if len(self.grids) >= 8:  # (a cloud is searched with two or three radii; a sweep over many drops the oldest)
            oldest = next(iter(self.grids))
            self.grids.pop(oldest)
            self.params.pop(oldest, None)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=pts.device)
        self.grids[float(radius)] = (key, buf)
        self.params.pop(float(radius), None)
        return buf, False


def source_grids(cloud) -> Optional["SourceGrids"]:
    """The holder of ``cloud``'s source grids, created on first use and kept on the object; None when the object takes
    no attribute."""
    holder = getattr(cloud, "_se3_grids_", None)
    if holder is None:
        holder = SourceGrids()
        try:
            cloud._se3_grids_ = holder
        except (AttributeError, TypeError):
            return None
    return holder


def forget_source_grids(cloud) -> None:
    """Drop ``cloud``'s source grids (what a new step's cloud starts without; timing loops that rebuild a step's
    neighbourhoods on the same cloud objects call it per repetition)."""
    holder = getattr(cloud, "_se3_grids_", None)
    if holder is not None:
        holder.grids.clear()
        holder.params.clear()


def ball_query_bounded(pts_src, pts_dst, batch_src, batch_dst, radius: float, capacity: int,
                       n_batches: Optional[int] = None, want_sources: bool = False,
                       neighbors_out: Optional[torch.Tensor] = None, src_box=None, grids: Optional[SourceGrids] = None):
    """The same query without the host round trip for the edge count (``se3_ball_query_bounded``): the caller sizes the
    edge buffer (``capacity`` rows, e.g. 1.25 x the previous step's count).  Returns ``(neighbors [capacity,2] int32,
    ends [M] int32, info [2] int32 on the device)`` with ``info[0]`` = true edge count and ``info[1]`` = 1 when it did
    not fit (the list is then truncated and ``ends`` clamped: rerun with a larger buffer).  Capturable in a HIP graph
    when ``n_batches`` is given.  ``want_sources``: a fourth result, the source ids as a dense ``[capacity]`` array
    (the source-major edge list of a cloud against itself).  ``neighbors_out``: a caller-owned contiguous
    ``[capacity, 2]`` int32 buffer to write into (e.g. a slice of a larger arena) instead of a fresh allocation.
    ``grids``: the source cloud's ``SourceGrids`` -- its cell grid for this radius is then built once and shared by every
    query that passes the holder (not while a HIP graph is being captured: a replay must not depend on what ran before)."""
    lib = _lib.load()
    pts_src = _as(pts_src, torch.float32)
    pts_dst = _as(pts_dst, torch.float32)
    if pts_src.dim() != 2 or pts_src.shape[1] != 3 or pts_dst.dim() != 2 or pts_dst.shape[1] != 3:
        raise ValueError("ball_query: only [N,3] point sets are supported")
    if not (radius > 0) or capacity < 0:
        raise ValueError("ball_query_bounded: radius must be positive and capacity non-negative")
    dev = pts_src.device
    bs, bd = _as(batch_src, torch.int32), _as(batch_dst, torch.int32)
    n_src, n_dst = pts_src.shape[0], pts_dst.shape[0]
    f32, i32 = torch.float32, torch.int32
    if neighbors_out is not None:
        if neighbors_out.shape != (int(capacity), 2) or neighbors_out.dtype != i32 or not neighbors_out.is_contiguous():
            raise ValueError(f"neighbors_out must be a contiguous int32 [{int(capacity)}, 2] tensor")
        neighbors = neighbors_out
    else:
        neighbors = torch.empty((int(capacity), 2), dtype=i32, device=dev)
    sources = torch.empty(int(capacity), dtype=i32, device=dev) if want_sources else None
    ends = torch.empty(n_dst, dtype=i32, device=dev)
    if n_dst == 0:
        info = torch.zeros(2, dtype=i32, device=dev)
        return (neighbors, ends, info, sources) if want_sources else (neighbors, ends, info)
    info = torch.empty(2, dtype=i32, device=dev)  # both words are written by the store pass
    needs_grid, ws_bytes, grid_bytes = _ball_query_sizes(n_src, n_dst)
    ws = _workspace(ws_bytes, dev)
    if grids is not None and needs_grid and src_box is not None and SHARED_GRIDS and not torch.cuda.is_current_stream_capturing():
        # (src_box: the grid parameters are then a pure function of the cloud's cached boxes and the radius, so two calls
        # with the same key search the same cells -- and the parameters themselves are kept with the grid)
        grid, valid = grids.slot(pts_src, bs, radius, n_batches, grid_bytes)
        params = grids.params.get(float(radius)) if valid else None
        if params is None:
            params = grids.params[float(radius)] = _batch_aabb_min_and_cells(pts_src, bs, radius, n_batches, src_box)
        mn, nc = params
        _lib.check(lib.se3_ball_query_bounded_shared(
            _ptr(pts_src, f32, "pts_src"), _ptr(pts_dst, f32, "pts_dst", dev), _ptr(bs, i32, "batch_src", dev),
            _ptr(bd, i32, "batch_dst", dev), _ptr(mn, f32, "aabb_min"), _ptr(nc, i32, "num_cells"), float(radius), n_src,
            n_dst, int(n_batches or 0), C.c_void_p(grid.data_ptr()), grid.numel(), int(valid), C.c_void_p(ws.data_ptr()),
            ws.numel(), int(capacity), _ptr(neighbors, i32, "neighbors"), _ptr(sources, i32, "sources"), _ptr(ends, i32, "ends"),
            _ptr(info, i32, "info"), _stream(dev)), "se3_ball_query_bounded_shared")
        return (neighbors, ends, info, sources) if want_sources else (neighbors, ends, info)
    mn, nc = _batch_aabb_min_and_cells(pts_src, bs, radius, n_batches, src_box) if needs_grid else (None, None)
    _lib.check(lib.se3_ball_query_bounded(
        _ptr(pts_src, f32, "pts_src"), _ptr(pts_dst, f32, "pts_dst", dev), _ptr(bs, i32, "batch_src", dev),
        _ptr(bd, i32, "batch_dst", dev), _ptr(mn, f32, "aabb_min"), _ptr(nc, i32, "num_cells"), float(radius), n_src,
        n_dst, int(n_batches or 0), C.c_void_p(ws.data_ptr()), ws.numel(), int(capacity), _ptr(neighbors, i32, "neighbors"),
        _ptr(sources, i32, "sources"), _ptr(ends, i32, "ends"), _ptr(info, i32, "info"), _stream(dev)),
        "se3_ball_query_bounded")
    return (neighbors, ends, info, sources) if want_sources else (neighbors, ends, info)


class BallQuery(torch.autograd.Function):
    """Drop-in for ``point_cloud_lib.custom_ops.BallQuery`` (BallQuery.py:11-53).  Like the
    reference it returns ``neighbors`` as int64 (ball_query.cu:99-101 promotes through ``cat``)
    and ``start_ids`` (inclusive ends) as int32; ``max_neighbors`` must be 0."""

    @staticmethod
    def forward(ctx, p_pt_src, p_pt_sample, p_batch_id_src, p_batch_id_sample, radius, max_neighbors, n_batches=None):
        """``n_batches`` (extension; the reference reads it back from the device, ball_query.cu:46): batch count
        when the caller knows it -- saves a host sync."""
        if max_neighbors != 0:
            raise NotImplementedError("max_neighbors > 0 (random sub-sampling) is not used by any model path")
        nb, ends = ball_query(p_pt_src, p_pt_sample, p_batch_id_src, p_batch_id_sample, radius, n_batches)
        return nb.to(torch.int64), ends

    @staticmethod
    def backward(ctx, *grads):
        return None, None, None, None, None, None, None


def csr_transpose(neighbors_i32: torch.Tensor, n_src: int, n_valid: Optional[torch.Tensor] = None, want_edge_ids: bool = False):
    """Source-major copy of an edge list: ``t_samples [E]``, ``t_ends [n_src]`` (inclusive).  ``n_valid`` (device int32
    tensor, e.g. ``info`` of ``ball_query_bounded``): only the first ``n_valid[0]`` rows are edges -- the unset tail of a
    capacity-sized buffer is ignored, without a host round trip.  ``want_edge_ids``: a third result ``t_edge_ids [E]``,
    the row of ``neighbors_i32`` every entry came from."""
    lib = _lib.load()
    dev = neighbors_i32.device
    e = neighbors_i32.shape[0]
    t_samples = torch.empty(e, dtype=torch.int32, device=dev)
    t_ends = torch.empty(n_src, dtype=torch.int32, device=dev)  # every entry is written by the library
    t_ids = torch.empty(e, dtype=torch.int32, device=dev) if want_edge_ids else None
    ws = _workspace(lib.se3_csr_transpose_workspace_bytes(e), dev)
    _lib.check(lib.se3_csr_transpose_bounded(_ptr(neighbors_i32, torch.int32, "neighbors"), e,
                                             _ptr(n_valid, torch.int32, "n_valid", dev), n_src, C.c_void_p(ws.data_ptr()),
                                             ws.numel(), _ptr(t_samples, torch.int32, "t_samples"),
                                             _ptr(t_ends, torch.int32, "t_ends"), _ptr(t_ids, torch.int32, "t_edge_ids"),
                                             _stream(dev)), "se3_csr_transpose_bounded")
    return (t_samples, t_ends, t_ids) if want_edge_ids else (t_samples, t_ends)


# ------------------------------------------------------------------------- hierarchy build (row f-2)
POOL_MODES = {"avg": 0, "max": 1, "min": 2, "sum": 3}


class GridCells:
    """Result of one grid sub-sampling step (``se3_grid_subsample``): the cell of every point (what
    ``torch.unique(keys, return_inverse=True)`` returns in Grid.py:45), the points grouped by cell, the cell sizes
    as inclusive end offsets, and the next level's points / batch ids (PointHierarchy.py:46-49)."""

    def __init__(self, cell_ids, sorted_ids, cell_ends, n_cells, pts, batch_ids):
        self.cell_ids, self.sorted_ids, self.cell_ends, self.n_cells = cell_ids, sorted_ids, cell_ends, n_cells
        self.pts, self.batch_ids = pts, batch_ids


def grid_subsample(pts, batch_ids, cell_size: float, n_batches: Optional[int] = None) -> GridCells:
    lib = _lib.load()
    pts = _as(pts, torch.float32)
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("grid_subsample: only [N,3] point sets are supported")
    if not (cell_size > 0):
        raise ValueError("grid_subsample: cell size must be positive")
    b = _as(batch_ids, torch.int32)
    dev, n = pts.device, pts.shape[0]
    if n_batches is None:
        n_batches = int(b.max().item()) + 1 if n else 1
    i32 = torch.int32
    cell_ids = torch.empty(n, dtype=i32, device=dev)
    sorted_ids = torch.empty(n, dtype=i32, device=dev)
    cell_ends = torch.empty(n, dtype=i32, device=dev)
    n_cells = torch.zeros(1, dtype=i32, device=dev)
    cell_pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
    cell_bid = torch.empty(n, dtype=i32, device=dev)
    ws = _workspace(lib.se3_grid_subsample_workspace_bytes(n, n_batches), dev)
    _lib.check(lib.se3_grid_subsample(
        _ptr(pts, torch.float32, "pts"), _ptr(b, i32, "batch_ids", dev), n, n_batches, float(cell_size),
        C.c_void_p(ws.data_ptr()), ws.numel(), _ptr(cell_ids, i32, "cell_ids"), _ptr(sorted_ids, i32, "sorted_ids"),
        _ptr(cell_ends, i32, "cell_ends"), _ptr(n_cells, i32, "n_cells"), _ptr(cell_pts, torch.float32, "cell_pts"),
        _ptr(cell_bid, i32, "cell_batch_ids"), _stream(dev)), "se3_grid_subsample")
    m = int(n_cells.item())  # the level size has to reach the host: every later allocation depends on it
    return GridCells(cell_ids, sorted_ids, cell_ends[:m], m, cell_pts[:m], cell_bid[:m])


def _rows2d(t):
    return t.reshape(t.shape[0], -1) if t.dim() != 2 else t


def _segment_pool(cells: GridCells, x2, mode: int, want_arg: bool):
    lib = _lib.load()
    c = x2.shape[1]
    out = torch.empty((cells.n_cells, c), dtype=torch.float32, device=x2.device)
    arg = torch.empty((cells.n_cells, c), dtype=torch.int32, device=x2.device) if want_arg else None
    _lib.check(lib.se3_segment_pool(_ptr(x2, torch.float32, "src"), _ptr(cells.sorted_ids, torch.int32, "sorted_ids"),
                                    _ptr(cells.cell_ends, torch.int32, "cell_ends"), cells.n_cells, c, mode,
                                    _ptr(out, torch.float32, "out"), _ptr(arg, torch.int32, "arg"), _stream(x2.device)),
               "se3_segment_pool")
    return out, arg


def _segment_unpool(cells: GridCells, v2, arg, mode: int):
    lib = _lib.load()
    n, c = cells.cell_ids.shape[0], v2.shape[1]
    out = torch.empty((n, c), dtype=torch.float32, device=v2.device)
    _lib.check(lib.se3_segment_unpool(_ptr(v2, torch.float32, "cell_vals"), _ptr(cells.cell_ids, torch.int32, "cell_ids"),
                                      _ptr(cells.cell_ends, torch.int32, "cell_ends"), _ptr(arg, torch.int32, "arg"),
                                      n, c, mode, _ptr(out, torch.float32, "out"), _stream(v2.device)), "se3_segment_unpool")
    return out


def grid_pick(cells: GridCells, u: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """One random point per cell (GridSubSample.py:43-54): ``ids [n_cells]`` = positions in the cell-sorted point list
    (the reference's ``ids_``) and ``picked [n_cells]`` = those points' indices (``sorted_ids_[ids_]``).  ``u``:
    uniform numbers in [0,1), one per cell (default: drawn on the device -- the reference draws them on the host and
    copies them over); nothing here synchronises with the host."""
    lib = _lib.load()
    dev = cells.sorted_ids.device
    if u is None:
        u = torch.rand(cells.n_cells, device=dev)
    u = _as(u, torch.float32).to(dev)
    if u.shape != (cells.n_cells,):
        raise ValueError(f"grid_pick: {tuple(u.shape)} random numbers for {cells.n_cells} cells")
    i32 = torch.int32
    ids = torch.empty(cells.n_cells, dtype=i32, device=dev)
    picked = torch.empty(cells.n_cells, dtype=i32, device=dev)
    _lib.check(lib.se3_grid_pick(_ptr(cells.cell_ends, i32, "cell_ends", dev), _ptr(cells.sorted_ids, i32, "sorted_ids"),
                                 _ptr(u, torch.float32, "u"), cells.n_cells, _ptr(ids, i32, "ids"),
                                 _ptr(picked, i32, "picked"), _stream(dev)), "se3_grid_pick")
    return ids, picked


def _rows_move(lib_fn, name, src, idx, out):
    row_bytes = src[0].numel() * src.element_size() if src.shape[0] else max(1, out[0].numel() * out.element_size())
    _lib.check(lib_fn(C.c_void_p(src.data_ptr()), _ptr(idx, torch.int32, "idx", src.device), idx.shape[0], row_bytes,
                      C.c_void_p(out.data_ptr()), _stream(src.device)), name)


def rows_gather(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """``src[idx]`` along dim 0 for any dtype (``se3_rows_gather``)."""
    if not src.is_cuda:
        raise ValueError("rows_gather: expected a GPU tensor (the HIP path has no CPU fallback)")
    src = src.detach().contiguous()
    out = torch.empty((idx.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if idx.shape[0] and out.numel():
        _rows_move(_lib.load().se3_rows_gather, "se3_rows_gather", src, idx, out)
    return out


def rows_scatter(src: torch.Tensor, idx: torch.Tensor, n_rows: int) -> torch.Tensor:
    """Zeros ``[n_rows, ...]`` with ``out[idx[r]] = src[r]`` (unique ``idx``; ``se3_rows_scatter``)."""
    if not src.is_cuda:
        raise ValueError("rows_scatter: expected a GPU tensor (the HIP path has no CPU fallback)")
    src = src.detach().contiguous()
    out = torch.zeros((int(n_rows),) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if idx.shape[0] and src.numel():
        _rows_move(_lib.load().se3_rows_scatter, "se3_rows_scatter", src, idx, out)
    return out


class RowsGather(torch.autograd.Function):
    """``x[idx]`` for unique row indices (``__subsample_tensor__`` of the random grid sub-sample, GridSubSample.py:67);
    the gradient is the scatter into zeros."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.idx, ctx.n = idx, x.shape[0]
        return rows_gather(x, idx)

    @staticmethod
    def backward(ctx, g):
        return rows_scatter(g, ctx.idx, ctx.n), None


class RowsScatter(torch.autograd.Function):
    """Rows ``idx`` of a zero tensor ``[n_rows, ...]`` set to ``x`` (``__upsample_tensor__``, GridSubSample.py:83-91);
    the gradient is the gather."""

    @staticmethod
    def forward(ctx, x, idx, n_rows):
        ctx.idx = idx
        return rows_scatter(x, idx, n_rows)

    @staticmethod
    def backward(ctx, g):
        return rows_gather(g, ctx.idx), None, None


class GridPool(torch.autograd.Function):
    """``pool_tensor`` of one level step (GridSubSample.py:63-77): rows of level l -> rows of level l+1."""

    @staticmethod
    def forward(ctx, x, cells, method):
        mode = POOL_MODES[method]
        x2 = _rows2d(_as(x, torch.float32))
        out, arg = _segment_pool(cells, x2, mode, mode in (1, 2))
        ctx.cells, ctx.mode, ctx.arg, ctx.shape = cells, mode, arg, x.shape
        return out.reshape((cells.n_cells,) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, g):
        g2 = _rows2d(_as(g, torch.float32))
        return _segment_unpool(ctx.cells, g2, ctx.arg, ctx.mode).reshape(ctx.shape), None, None


class GridUpsample(torch.autograd.Function):
    """``upsample_tensor`` (GridSubSample.py:93: ``p_tensor[cell_ids]``); the gradient is a segment sum in fixed order
    instead of the atomics of an index_add."""

    @staticmethod
    def forward(ctx, x, cells):
        x2 = _rows2d(_as(x, torch.float32))
        ctx.cells, ctx.shape = cells, x.shape
        return _segment_unpool(cells, x2, None, 3).reshape((cells.cell_ids.shape[0],) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, g):
        g2 = _rows2d(_as(g, torch.float32))
        return _segment_pool(ctx.cells, g2, 3, False)[0].reshape(ctx.shape), None


# ------------------------------------------------------------------------------ frame pooling (row f-3)
class FramePool(torch.autograd.Function):
    """``PointcloudRotEquiv.feature_pooling`` (pc/PointcloudRotEquiv.py:224-251): [N*F, C] -> [N, C]."""

    @staticmethod
    def forward(ctx, x, n_frames, method):
        lib = _lib.load()
        mode = POOL_MODES[method]
        x2 = _rows2d(_as(x, torch.float32))
        if x2.shape[0] % n_frames:
            raise ValueError("feature_pooling: rows are not a multiple of the frame count")
        n, c = x2.shape[0] // n_frames, x2.shape[1]
        out = torch.empty((n, c), dtype=torch.float32, device=x2.device)
        arg = torch.empty((n, c), dtype=torch.int32, device=x2.device) if mode in (1, 2) else None
        _lib.check(lib.se3_frame_pool(_ptr(x2, torch.float32, "x"), n, n_frames, c, mode, _ptr(out, torch.float32, "out"),
                                      _ptr(arg, torch.int32, "arg"), _stream(x2.device)), "se3_frame_pool")
        ctx.mode, ctx.arg, ctx.f, ctx.shape = mode, arg, n_frames, x.shape
        return out.reshape((n,) + tuple(x.shape[1:]))

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        g2 = _rows2d(_as(g, torch.float32))
        n, c = g2.shape
        gx = torch.empty((n * ctx.f, c), dtype=torch.float32, device=g2.device)
        _lib.check(lib.se3_frame_unpool(_ptr(g2, torch.float32, "grad_out"), _ptr(ctx.arg, torch.int32, "arg"), n, ctx.f, c,
                                        ctx.mode, _ptr(gx, torch.float32, "grad_x"), _stream(g2.device)), "se3_frame_unpool")
        return gx.reshape(ctx.shape), None, None


# ------------------------------------------------------------------------------- frames (row f-1)
KNN_GRID_MIN_POINTS = 8192  # below this the all-pairs scan is as fast as sorting into cells
KNN_CELL_FACTOR = 1.6       # cell size / estimated k-NN distance (1.26 covers points on a face of the cloud)


def _knn_cell_size(mn, mx, counts, k: int) -> torch.Tensor:
    """Cell size (device scalar): KNN_CELL_FACTOR x the k-NN distance a uniformly filled box of each batch element's extent
    would have -- the larger of the volume, area and length estimates, so flat and thin clouds are covered too.
    Only a speed knob: se3_knn_query_grid is exact for any cell size."""
    ext = (mx - mn).clamp_min(0).sort(dim=1, descending=True)[0]
    cnt = counts.clamp_min(1).to(torch.float32)
    e1, e2, e3 = ext[:, 0], ext[:, 1], ext[:, 2]
    c_vol = (k * e1 * e2 * e3 / (4.19 * cnt)).pow(1.0 / 3.0)
    c_area = (k * e1 * e2 / (3.14 * cnt)).sqrt()
    c_len = k * e1 / (2.0 * cnt)
    c = KNN_CELL_FACTOR * torch.stack([c_vol, c_area, c_len]).max()
    return torch.maximum(c, (e1.max() * 1e-6).clamp_min(1e-30))


def knn_query(pts, batch_ids, k: int, n_batches: Optional[int] = None, method: str = "auto", box=None) -> torch.Tensor:
    """``point_cloud_lib_ops.knn_query``: self-kNN inside each batch element, ``[N,k]`` int32 (self first,
    ascending distance, ties to the lower index, -1 padded).  ``method``: "grid" (cell grid + exact fallback),
    "scan" (all pairs inside the batch element) or "auto" (grid from KNN_GRID_MIN_POINTS points on).  ``box``: the
    cloud's ``batch_aabb`` result when the caller has it (``Pointcloud.aabb()``)."""
    lib = _lib.load()
    pts = _as(pts, torch.float32)
    b = _as(batch_ids, torch.int32)
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise ValueError("knn_query: only [N,3] point sets are supported")
    if method not in ("auto", "grid", "scan"):
        raise ValueError(f"knn_query: unknown method {method!r}")
    n, dev = pts.shape[0], pts.device
    out = torch.empty((n, int(k)), dtype=torch.int32, device=dev)
    f32, i32 = torch.float32, torch.int32
    if not 1 <= int(k) <= 64:
        raise ValueError(f"knn_query: k = {k}; the kernels keep at most 64 neighbours per point (the reference's own "
                         "kernel has the same limit, knn_query.cu:167)")
    if method == "grid" and k > 32:
        raise ValueError("knn_query: the cell-grid search keeps at most 32 neighbours; use method='scan' or 'auto'")
    if method == "scan" or (method == "auto" and (n < KNN_GRID_MIN_POINTS or k > 32)) or n == 0:
        _lib.check(lib.se3_knn_query(_ptr(pts, f32, "pts"), _ptr(b, i32, "batch_ids", dev), n, int(k),
                                     _ptr(out, i32, "out"), _stream(dev)), "se3_knn_query")
        return out
    # boxes, then cell size / shifted minima / cell counts in ONE launch (se3_knn_grid_params: what _knn_cell_size and the
    # lines of BallQuery.py:34-38 compute -- as a dozen torch calls and a bincount they cost more than the search)
    box_mn, mx = box if box is not None else batch_aabb(pts, b, n_batches)
    nb = box_mn.shape[0]
    mn = torch.empty((nb, 3), dtype=f32, device=dev)
    num_cells = torch.empty(3, dtype=i32, device=dev)
    cell3 = torch.empty(3, dtype=f32, device=dev)
    _lib.check(lib.se3_knn_grid_params(_ptr(b, i32, "batch_ids", dev), n, _ptr(box_mn, f32, "box_min", dev),
                                       _ptr(mx, f32, "box_max", dev), nb, int(k), float(KNN_CELL_FACTOR), _ptr(mn, f32, "aabb_min"),
                                       _ptr(num_cells, i32, "num_cells"), _ptr(cell3, f32, "cell_size"), _stream(dev)),
               "se3_knn_grid_params")
    ws = _workspace(lib.se3_knn_query_grid_workspace_bytes(n), dev)
    _lib.check(lib.se3_knn_query_grid(
        _ptr(pts, f32, "pts"), _ptr(b, i32, "batch_ids", dev), _ptr(mn, f32, "aabb_min"), _ptr(num_cells, i32, "num_cells"),
        _ptr(cell3, f32, "cell_size"), n, int(k), _ptr(out, i32, "out"), C.c_void_p(ws.data_ptr()), ws.numel(),
        _stream(dev)), "se3_knn_query_grid")
    return out


def knn_query_pair(pts_src, batch_src, pts_q, batch_q, k: int) -> torch.Tensor:
    """For every query point the ``k`` nearest SOURCE points of the same batch element: ``[N_q, k]`` int32 source
    indices, ascending (distance, index), -1 padded (``se3_knn_query_pair``; the ``torch_cluster.knn`` call of
    pc/KnnNeighborhood.py:77-84).  Both batch-id arrays sorted; k <= 64."""
    lib = _lib.load()
    f32, i32 = torch.float32, torch.int32
    ps, pq = _as(pts_src, f32), _as(pts_q, f32)
    bs, bq = _as(batch_src, i32), _as(batch_q, i32)
    if ps.dim() != 2 or ps.shape[1] != 3 or pq.dim() != 2 or pq.shape[1] != 3:
        raise ValueError("knn_query_pair: only [N,3] point sets are supported")
    if not 1 <= int(k) <= 64:
        raise ValueError(f"knn_query_pair: k = {k}; at most 64 neighbours per point")
    dev = pq.device
    out = torch.empty((pq.shape[0], int(k)), dtype=i32, device=dev)
    _lib.check(lib.se3_knn_query_pair(_ptr(ps, f32, "pts_src", dev), _ptr(bs, i32, "batch_src", dev), ps.shape[0],
                                      _ptr(pq, f32, "pts_q"), _ptr(bq, i32, "batch_q", dev), pq.shape[0], int(k),
                                      _ptr(out, i32, "out"), _stream(dev)), "se3_knn_query_pair")
    return out


class KNNQuery(torch.autograd.Function):
    """Drop-in for ``point_cloud_lib.custom_ops.KNNQuery`` (KNNQuery.py:11-35)."""

    @staticmethod
    def forward(ctx, p_pt_src, p_batch_id_src, p_k):
        return knn_query(p_pt_src, p_batch_id_src, p_k)

    @staticmethod
    def backward(ctx, grad):
        return None, None, None


def pca_frames(pts, knn_ids, axis_fixed=None) -> torch.Tensor:
    """``sample_reference_frames_pca`` on the GPU: ``[N,4,9]`` frames (``[N,2,9]`` with a fixed axis 1 or 2).
    ``axis_fixed`` follows the reference: ``None`` / ``False`` / ``0`` all mean "not fixed"."""
    lib = _lib.load()
    pts = _as(pts, torch.float32)
    ids = _as(knn_ids, torch.int32)
    n, k = ids.shape
    axis = int(axis_fixed) if axis_fixed else -1
    if axis not in (-1, 1, 2):
        raise ValueError(f"axis_fixed = {axis_fixed}")
    nf = 4 if axis < 0 else 2
    frames = torch.empty((n, nf, 9), dtype=torch.float32, device=pts.device)
    _lib.check(lib.se3_pca_frames(_ptr(pts, torch.float32, "pts"), _ptr(ids, torch.int32, "knn", pts.device), n, k, axis,
                                  _ptr(frames, torch.float32, "frames"), _stream(pts.device)), "se3_pca_frames")
    return frames


def shuffle_frames(all_frames, n_frames: int, draws=None) -> torch.Tensor:
    """``n_frames`` of each point's frames in a uniformly random order (``se3_shuffle_frames``; the ``torch.multinomial``
    + gather of PointcloudRotEquiv.py:100-117, 146-167).  ``draws`` ``[N, n_all]`` uniform numbers (``torch.rand`` on
    the device by default)."""
    lib = _lib.load()
    all_frames = _as(all_frames, torch.float32)
    n, n_all = all_frames.shape[0], all_frames.shape[1]
    dev = all_frames.device
    if draws is None:
        draws = torch.rand((n, n_all), device=dev)
    draws = _as(draws, torch.float32)
    if tuple(draws.shape) != (n, n_all):
        raise ValueError(f"shuffle_frames: draws {tuple(draws.shape)}, expected {(n, n_all)}")
    out = torch.empty((n, int(n_frames), all_frames.shape[2]), dtype=torch.float32, device=dev)
    if all_frames.shape[2] != 9:
        raise ValueError("shuffle_frames: frames are [N, n_all, 9]")
    _lib.check(lib.se3_shuffle_frames(_ptr(all_frames, torch.float32, "all_frames"), _ptr(draws, torch.float32, "draws", dev), n,
                                      n_all, int(n_frames), _ptr(out, torch.float32, "out"), _stream(dev)), "se3_shuffle_frames")
    return out


# ------------------------------------------------------------------- the operator's geometry bundle
class PreparedRecords:
    """The packed 64-byte geometry records of one cloud (include/se3conv.h, struct se3conv_prepared): a function of the
    cloud's points and frames alone, so every convolution that touches the cloud -- forward and backward, every layer of a
    level -- shares one image.  The first call that meets an invalid holder fills it inside its own preparation launch.
    Kept on the cloud object (``prepared_records``); re-validated against the tensors' storage and version counters."""

    __slots__ = ("tensor", "valid", "key")

    def __init__(self):
        self.tensor, self.valid, self.key = None, False, None

    def bind(self, pts: torch.Tensor, frames: torch.Tensor) -> "PreparedRecords":
        key = (pts.data_ptr(), pts._version, frames.data_ptr(), frames._version, tuple(frames.shape), str(pts.device))
        if key != self.key:
            rows = frames.shape[0] * frames.shape[1]
            if self.tensor is None or self.tensor.shape[0] != rows or self.tensor.device != pts.device:
                self.tensor = torch.empty((rows, 16), dtype=torch.float32, device=pts.device)
            self.key, self.valid = key, False
        return self


def prepared_records(cloud) -> Optional["PreparedRecords"]:
    """The holder of ``cloud``'s geometry records, created on first use and kept on the object (any object that takes an
    attribute: the reference's own containers and plain namespaces too); None when it cannot be attached."""
    holder = getattr(cloud, "_se3_records_", None)
    if holder is None:
        holder = PreparedRecords()
        try:
            cloud._se3_records_ = holder
        except (AttributeError, TypeError):
            return None
    return holder


def invalidate_prepared(cloud) -> None:
    """Forget the cloud's prepared records (the next convolution that touches the cloud rebuilds them): what a new step
    does implicitly by building new cloud objects; a benchmark that re-uses its clouds calls this once per step."""
    holder = getattr(cloud, "_se3_records_", None)
    if holder is not None:
        holder.valid = False


@dataclass
class ConvGeometry:
    """Everything the operator reads besides features and parameters (all fp32 / int32, GPU)."""

    pts_in: torch.Tensor       # [N_in,3]
    pts_out: torch.Tensor      # [N_out,3]
    frames_in: torch.Tensor    # [N_in,F_in,9]
    frames_out: torch.Tensor   # [N_out,F_out,9]
    neighbors: torch.Tensor    # [E,2] int32
    ends: torch.Tensor         # [N_out] int32
    _transpose: Optional[Tuple[torch.Tensor, torch.Tensor]] = field(default=None, repr=False)
    # the edge relation is symmetric (a radius graph of a cloud with itself: ||s - p|| < r both ways, bit for bit):
    # the source-major edge list is then the sample-major one read the other way round, nothing to build
    symmetric: bool = False
    bounded: bool = False  # `neighbors` is a capacity-sized buffer whose rows past ends[-1] are unset
    sources: Optional[torch.Tensor] = None  # column 1 of `neighbors` as a dense array when the ball query wrote one
    edge_info: Optional[torch.Tensor] = None  # [2] int32 on the device (bounded only): true edge count, overflow flag
    # the neighbourhood's own way to the source-major list (pc.BQNeighborhood.source_major), or None
    source_major_fn: Optional[Callable[[], Optional[Tuple[torch.Tensor, torch.Tensor]]]] = field(default=None, repr=False)
    _edge_ids: Optional[torch.Tensor] = field(default=None, repr=False)  # third result of the library's transposition, if it built the list
    # holders of the clouds' packed geometry records (the same object for a cloud against itself), or None
    records_in: Optional[PreparedRecords] = field(default=None, repr=False)
    records_out: Optional[PreparedRecords] = field(default=None, repr=False)

    @staticmethod
    def build(pts_in, pts_out, frames_in, frames_out, neighbors, ends, symmetric: bool = False) -> "ConvGeometry":
        n_out = pts_out.shape[0]
        ends = _as(ends, torch.int32)
        if ends.shape[0] != n_out:
            raise ValueError(f"start_ids has {ends.shape[0]} entries, expected one per output point ({n_out})")
        fi = _as(frames_in, torch.float32).reshape(pts_in.shape[0], -1, 9)
        fo = _as(frames_out, torch.float32).reshape(n_out, -1, 9)
        nb = _as(neighbors, torch.int32)
        if nb.dim() != 2 or nb.shape[1] != 2:
            raise ValueError("neighbors must be [E,2]")
        if symmetric and pts_in.shape[0] != n_out:
            raise ValueError("a symmetric neighbourhood needs the same cloud on both sides")
        return ConvGeometry(_as(pts_in, torch.float32), _as(pts_out, torch.float32), fi, fo, nb, ends, None, symmetric)

    def transpose(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """Source-major edge list ``(t_samples [E], t_ends [N_in])`` for the feature gradient."""
        if self._transpose is None:
            if self.symmetric:  # samples of source p = sources of sample p
                src = self.sources if self.sources is not None else self.neighbors[:, 1].contiguous()
                self._transpose = (src, self.ends)
            elif self.source_major_fn is not None and (own := self.source_major_fn()) is not None:
                self._transpose = own
            else:
                if self.bounded and self.edge_info is None:
                    raise ValueError("a capacity-bounded edge buffer between two clouds needs its device-side edge count "
                                     "(edge_info) to be transposed: rows past it are unset")
                # bounded: the unset tail of the buffer must not reach the sort (se3_csr_transpose_bounded)
                ts, te, self._edge_ids = csr_transpose(self.neighbors, self.pts_in.shape[0],
                                                       self.edge_info if self.bounded else None, want_edge_ids=True)
                self._transpose = (ts, te)
        return self._transpose

    def shape(self, c_in: int, c_out: int, num_basis: int, precision: Optional[str] = None) -> Se3Shape:
        # n_edges = rows of the edge buffer: an upper bound on the edge count for a bounded neighbourhood (se3conv.h)
        return Se3Shape(self.pts_in.shape[0], self.pts_out.shape[0], self.neighbors.shape[0],
                        self.frames_in.shape[1], self.frames_out.shape[1], c_in, c_out, num_basis,
                        _lib.PRECISIONS[precision or _precision])


def _geom_ptrs(g: ConvGeometry):
    f32, i32 = torch.float32, torch.int32
    dev = g.pts_out.device
    return [_ptr(g.pts_in, f32, "pts_in", dev), _ptr(g.pts_out, f32, "pts_out", dev),
            _ptr(g.frames_in, f32, "frames_in", dev), _ptr(g.frames_out, f32, "frames_out", dev),
            _ptr(g.neighbors, i32, "neighbors", dev), _ptr(g.ends, i32, "ends", dev)]


def _prepared(geom: ConvGeometry, feat_words: Optional[torch.Tensor], feat_words_valid: bool):
    """struct se3conv_prepared of a call (or None) + what to mark valid once the call has been enqueued."""
    r_in, r_out = geom.records_in, geom.records_out
    if r_in is None and r_out is None and feat_words is None:
        return None, ()
    p = _lib.Se3Prepared()
    filled = []
    if r_in is not None:
        r_in.bind(geom.pts_in, geom.frames_in)
        p.geom_in, p.geom_in_valid = r_in.tensor.data_ptr(), int(r_in.valid)
        filled.append(r_in)
    if r_out is not None and r_out is not r_in:
        r_out.bind(geom.pts_out, geom.frames_out)
        p.geom_out, p.geom_out_valid = r_out.tensor.data_ptr(), int(r_out.valid)
        filled.append(r_out)
    elif r_out is r_in and r_in is not None:
        p.geom_out, p.geom_out_valid = p.geom_in, 1  # (ignored for a cloud against itself; valid by the time it is read otherwise)
    if feat_words is not None:
        p.feat_words, p.feat_words_valid = feat_words.data_ptr(), int(feat_words_valid)
    return p, tuple(filled)


def _scalar(t, name, dev) -> torch.Tensor:
    t = torch.as_tensor(t, dtype=torch.float32)
    return t.detach().to(device=dev, dtype=torch.float32).reshape(()).contiguous()


def se3conv_forward(geom: ConvGeometry, feat, proj_axes, proj_biases, conv_weights, rho, nu, save_t: bool = True,
                    precision: Optional[str] = None, feat_words: Optional[torch.Tensor] = None):
    """Raw forward: returns ``(out [N_out*F_out, C_out], T or None)``.  ``T`` is fp32 in "fp32" precision
    and an opaque same-size buffer of packed hi/lo words in "bf16x3" (pass it back with the same precision).
    ``feat_words`` (int32, one word per feature, split-bf16 modes): receives the packed feature words for ``se3conv_backward``."""
    lib = _lib.load()
    f32 = torch.float32
    dev = geom.pts_out.device
    feat = _as(feat, f32)
    a, b, w = _as(proj_axes, f32), _as(proj_biases, f32), _as(conv_weights, f32)
    if a.shape[0] != 9:
        raise ValueError(f"proj_axes_ has {a.shape[0]} rows; only the 9-D ('6D') descriptor is implemented")
    c_in, kb, c_out = w.shape
    if feat.shape != (geom.pts_in.shape[0] * geom.frames_in.shape[1], c_in):
        raise ValueError(f"features are {tuple(feat.shape)}, expected "
                         f"({geom.pts_in.shape[0] * geom.frames_in.shape[1]}, {c_in})")
    shp = geom.shape(c_in, c_out, kb, precision)
    rows = geom.pts_out.shape[0] * geom.frames_out.shape[1]
    out = torch.empty((rows, c_out), dtype=f32, device=dev)
    # T is kept for the weight gradient only on the K = 32 kernels' own layout; other K run as slices of 32 inside the
    # library, which recomputes T in backward (include/se3conv.h)
    t_save = torch.empty((rows, c_in, kb), dtype=f32, device=dev) if (save_t and kb == 32) else None
    ws = _workspace(lib.se3conv_fwd_workspace_bytes(C.byref(shp), 1 if save_t else 0), dev)
    rho_t, nu_t = _scalar(rho, "rho", dev), _scalar(nu, "nu", dev)
    prep, filled = _prepared(geom, feat_words, False)
    _lib.check(lib.se3conv_fwd_prepared(*_geom_ptrs(geom), _ptr(feat, f32, "features", dev), _ptr(a, f32, "proj_axes_", dev),
                                        _ptr(b, f32, "proj_biases_", dev), _ptr(w, f32, "conv_weights_", dev),
                                        _ptr(rho_t, f32, "norm_neigh_dist_"), _ptr(nu_t, f32, "norm_num_neighs_"),
                                        C.byref(shp), _ptr(out, f32, "out"), _ptr(t_save, f32, "t_save"),
                                        C.c_void_p(ws.data_ptr()), ws.numel(), _stream(dev),
                                        C.byref(prep) if prep is not None else None), "se3conv_fwd")
    for h in filled:
        h.valid = True
    return out, t_save


def se3conv_backward(geom: ConvGeometry, feat, proj_axes, proj_biases, conv_weights, rho, nu, t_save, grad_out,
                     want_feat=True, want_params=True, precision: Optional[str] = None,
                     feat_words: Optional[torch.Tensor] = None):
    """Raw backward: returns ``(dX, dA, dbeta, dW)`` (None where not requested).  ``feat_words``: what the forward call of
    the same features wrote (see ``se3conv_forward``), or None."""
    lib = _lib.load()
    f32, i32 = torch.float32, torch.int32
    dev = geom.pts_out.device
    feat = _as(feat, f32)
    a, b, w = _as(proj_axes, f32), _as(proj_biases, f32), _as(conv_weights, f32)
    g = _as(grad_out, f32)
    c_in, kb, c_out = w.shape
    shp = geom.shape(c_in, c_out, kb, precision)
    d_x = torch.empty_like(feat) if want_feat else None
    d_a = torch.empty_like(a) if want_params else None
    d_b = torch.empty_like(b) if want_params else None
    d_w = torch.empty_like(w) if want_params else None
    t_samples, t_ends = geom.transpose() if want_feat else (None, None)
    ws = _workspace(lib.se3conv_bwd_workspace_bytes(C.byref(shp), int(want_feat), int(want_params),
                                                    int(t_save is not None)), dev)
    rho_t, nu_t = _scalar(rho, "rho", dev), _scalar(nu, "nu", dev)
    prep, filled = _prepared(geom, feat_words, feat_words is not None)
    _lib.check(lib.se3conv_bwd_prepared(*_geom_ptrs(geom), _ptr(t_samples, i32, "t_samples"), _ptr(t_ends, i32, "t_ends"),
                                        _ptr(geom._edge_ids if want_feat else None, i32, "t_edge_ids"),
                                        _ptr(feat, f32, "features", dev), _ptr(a, f32, "proj_axes_", dev),
                                        _ptr(b, f32, "proj_biases_", dev), _ptr(w, f32, "conv_weights_", dev),
                                        _ptr(rho_t, f32, "rho"), _ptr(nu_t, f32, "nu"), _ptr(t_save, f32, "t_save"),
                                        _ptr(g, f32, "grad_out", dev), C.byref(shp), _ptr(d_x, f32, "grad_feat"),
                                        _ptr(d_a, f32, "grad_axes"), _ptr(d_b, f32, "grad_biases"),
                                        _ptr(d_w, f32, "grad_weights"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream(dev),
                                        C.byref(prep) if prep is not None else None), "se3conv_bwd")
    for h in filled:
        h.valid = True
    return d_x, d_a, d_b, d_w


class SE3ConvFunction(torch.autograd.Function):
    """The fused operator as one autograd node (replaces the chain matmul -> GELU -> FeatBasisProj
    -> einsum -> scalings of PNEConvLayerRotEquiv.py:199-216).  Differentiable w.r.t. features and
    the three parameters; geometry carries no gradient (reference: built under no_grad, :67)."""

    @staticmethod
    def forward(ctx, feat, proj_axes, proj_biases, conv_weights, geom: ConvGeometry, rho, nu):
        need_params = any(ctx.needs_input_grad[1:4])
        ctx.precision = _precision
        save_t = need_params
        if save_t:
            # T ([rows, C_in, K]: 0.8 GB per layer at the headline shape, the largest activation a layer would keep) is only
            # saved where backward has a use for it: when the feature gradient is wanted too, the library takes the weight
            # gradient from U, the tensor its transposed pass produces anyway (se3conv_bwd_needs_t, include/se3conv.h)
            c_in, kb, c_out = conv_weights.shape
            shp = geom.shape(c_in, c_out, kb, ctx.precision)
            save_t = _lib.load().se3conv_bwd_needs_t(C.byref(shp), int(bool(ctx.needs_input_grad[0]))) != 0
        # the packed feature words of the split-bf16 modes go to backward with the features (the parameter gradients gather
        # them again): one split per step instead of two, for one more word per feature kept
        fw = None
        if need_params and ctx.precision != "fp32" and conv_weights.shape[1] == 32 and feat.is_cuda:
            fw = torch.empty(feat.numel(), dtype=torch.int32, device=feat.device)
        out, t_save = se3conv_forward(geom, feat, proj_axes, proj_biases, conv_weights, rho, nu,
                                      save_t=save_t, precision=ctx.precision, feat_words=fw)
        ctx.geom = geom
        ctx.in_dtype = feat.dtype
        ctx.save_for_backward(feat, proj_axes, proj_biases, conv_weights, _scalar(rho, "rho", out.device),
                              _scalar(nu, "nu", out.device), t_save if t_save is not None else torch.empty(0),
                              fw if fw is not None else torch.empty(0))
        ctx.has_t = t_save is not None
        ctx.has_fw = fw is not None
        return out

    @staticmethod
    def backward(ctx, grad_out):
        feat, a, b, w, rho, nu, t_save, fw = ctx.saved_tensors
        want_feat = ctx.needs_input_grad[0]
        want_params = any(ctx.needs_input_grad[1:4])
        d_x, d_a, d_b, d_w = se3conv_backward(ctx.geom, feat, a, b, w, rho, nu, t_save if ctx.has_t else None,
                                              grad_out, want_feat, want_params, precision=ctx.precision,
                                              feat_words=fw if ctx.has_fw else None)
        if d_x is not None:
            d_x = d_x.to(ctx.in_dtype)
        ng = ctx.needs_input_grad
        return (d_x, d_a if ng[1] else None, d_b if ng[2] else None, d_w if ng[3] else None, None, None, None)


# ------------------------------------------------------------------ API-parity ops (a1, a3, a4, a5)
def rot_tensors(geom: ConvGeometry, rho, rel_rot: str = "6D"):
    """``get_rot_tenors`` materialised on the GPU: ``desc [E',D]``, ``neighbs [E',2] int32`` sorted by
    output row, ``ends [N_out*F_out] int32``.  ``rel_rot`` = the factory's ``p_rel_rot``: "6D" (D = 9), "matrix"
    (D = 12) or "quaternion" (D = 7)."""
    lib = _lib.load()
    if rel_rot not in _lib.REL_ROT:
        raise ValueError(f"rel_rot {rel_rot!r}; expected one of {sorted(_lib.REL_ROT)}")
    mode, dims = _lib.REL_ROT[rel_rot]
    dev = geom.pts_out.device
    shp = geom.shape(1, 1, 32)
    ff = shp.f_in * shp.f_out
    e2 = geom.neighbors.shape[0] * ff
    desc = torch.empty((e2, dims), dtype=torch.float32, device=dev)
    fe_nb = torch.empty((e2, 2), dtype=torch.int32, device=dev)
    fe_ends = torch.zeros(geom.pts_out.shape[0] * shp.f_out, dtype=torch.int32, device=dev)
    rho_t = _scalar(rho, "rho", dev)
    _lib.check(lib.se3_rot_tensors_rel(*_geom_ptrs(geom), _ptr(rho_t, torch.float32, "rho"), C.byref(shp), mode,
                                       _ptr(desc, torch.float32, "desc"), _ptr(fe_nb, torch.int32, "fe_neighbors"),
                                       _ptr(fe_ends, torch.int32, "fe_ends"), _stream(dev)), "se3_rot_tensors_rel")
    return desc, fe_nb, fe_ends


class FeatBasisProj(torch.autograd.Function):
    """Drop-in for ``point_cloud_lib.custom_ops.FeatBasisProj`` (FeatBasisProj.py:4-65)."""

    @staticmethod
    def forward(p_ctx, p_pt_basis, p_pt_features, p_neighbors, p_start_ids):
        lib = _lib.load()
        f32, i32 = torch.float32, torch.int32
        basis, feat = _as(p_pt_basis, f32), _as(p_pt_features, f32)
        nb, ends = _as(p_neighbors, i32), _as(p_start_ids, i32)
        p_ctx.save_for_backward(basis, feat, nb, ends)
        p_ctx.dtypes = (p_pt_basis.dtype, p_pt_features.dtype)
        dev = feat.device
        rows, ch, kb = ends.shape[0], feat.shape[1], basis.shape[1]
        out = torch.empty((rows, ch, kb), dtype=f32, device=dev)
        _lib.check(lib.se3_feat_basis_proj(_ptr(basis, f32, "pt_basis", dev), _ptr(feat, f32, "pt_features"),
                                           _ptr(nb, i32, "neighbors", dev), _ptr(ends, i32, "start_ids", dev),
                                           nb.shape[0], rows, feat.shape[0], ch, kb, _ptr(out, f32, "out"),
                                           _stream(dev)), "se3_feat_basis_proj")
        return out

    @staticmethod
    def backward(p_ctx, p_grads):
        lib = _lib.load()
        f32, i32 = torch.float32, torch.int32
        basis, feat, nb, ends = p_ctx.saved_tensors
        g = _as(p_grads, f32)
        dev = feat.device
        g_feat = torch.empty_like(feat)
        g_basis = torch.empty_like(basis)
        _lib.check(lib.se3_feat_basis_proj_grad(
            _ptr(basis, f32, "pt_basis", dev), _ptr(feat, f32, "pt_features"), _ptr(nb, i32, "neighbors"),
            _ptr(ends, i32, "start_ids"), _ptr(g, f32, "grads", dev), nb.shape[0], ends.shape[0], feat.shape[0],
            feat.shape[1], basis.shape[1], _ptr(g_feat, f32, "g_feat"), _ptr(g_basis, f32, "g_basis"), _stream(dev)),
            "se3_feat_basis_proj_grad")
        return g_basis.to(p_ctx.dtypes[0]), g_feat.to(p_ctx.dtypes[1]), None, None


# ------------------------------------------------------------------ row-wise glue of a block (row f-3)
def _glue_ws(c: int, dev) -> torch.Tensor:
    return _workspace(_lib.load().se3_glue_workspace_bytes(int(c)), dev)


class BatchNormTrain(torch.autograd.Function):
    """Training-mode ``torch.nn.BatchNorm1d`` on ``[rows, C]`` (layers/BatchNormPC.py:22-32): channel sums in fp64 and
    a fixed order, running statistics updated in place, 3 launches forward and 3 backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, num_batches_tracked=None):
        lib = _lib.load()
        f32 = torch.float32
        x2 = _as(x, f32)
        if x2.dim() != 2:
            raise ValueError("BatchNormTrain: expected [rows, C] features")
        rows, c = x2.shape
        dev = x2.device
        y = torch.empty_like(x2)
        mean = torch.empty(c, dtype=f32, device=dev)
        invstd = torch.empty(c, dtype=f32, device=dev)
        w = _as(weight, f32) if weight is not None else None
        b = _as(bias, f32) if bias is not None else None
        ws = _glue_ws(c, dev)
        _lib.check(lib.se3_bn_fwd(_ptr(x2, f32, "x"), _ptr(w, f32, "weight", dev), _ptr(b, f32, "bias", dev), rows, c,
                                  float(eps), float(momentum), _ptr(running_mean, f32, "running_mean", dev),
                                  _ptr(running_var, f32, "running_var", dev),
                                  _ptr(num_batches_tracked, torch.int64, "num_batches_tracked", dev), _ptr(y, f32, "y"),
                                  _ptr(mean, f32, "mean"),
                                  _ptr(invstd, f32, "invstd"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream(dev)),
                   "se3_bn_fwd")
        ctx.save_for_backward(x2, w if w is not None else torch.empty(0, device=dev), mean, invstd)
        ctx.has_w = w is not None
        ctx.mark_non_differentiable(*[t for t in (running_mean, running_var, num_batches_tracked) if t is not None])
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        f32 = torch.float32
        x2, w, mean, invstd = ctx.saved_tensors
        g2 = _as(g, f32)
        rows, c = x2.shape
        dev = x2.device
        dx = torch.empty_like(x2)
        dgamma = torch.empty(c, dtype=f32, device=dev)
        dbeta = torch.empty(c, dtype=f32, device=dev)
        ws = _glue_ws(c, dev)
        _lib.check(lib.se3_bn_bwd(_ptr(g2, f32, "dy", dev), _ptr(x2, f32, "x"), _ptr(mean, f32, "mean"),
                                  _ptr(invstd, f32, "invstd"), _ptr(w if ctx.has_w else None, f32, "weight"), rows, c,
                                  _ptr(dx, f32, "dx"), _ptr(dgamma, f32, "dgamma"), _ptr(dbeta, f32, "dbeta"),
                                  C.c_void_p(ws.data_ptr()), ws.numel(), _stream(dev)), "se3_bn_bwd")
        ng = ctx.needs_input_grad
        return (dx if ng[0] else None, dgamma if ng[1] else None, dbeta if ng[2] else None, None, None, None, None, None)


class SkipDropPath(torch.autograd.Function):
    """``drop_path(x * gamma_) + y`` of layers/SkipConnection.py with the per-batch gate of layers/DropPathPC.py:30-46
    (``gate [B]`` holds keep-mask / keep_prob -- or, with ``gate_keep = keep_prob > 0``, the uniform draws themselves, the
    kernel then evaluates floor(keep + u) / keep; ``row_batch [rows]`` int32 maps rows to batch elements; both None: no drop
    path) -- one launch forward, two backward."""

    @staticmethod
    def forward(ctx, x, y, gamma, gate, row_batch, gate_keep=0.0):
        lib = _lib.load()
        f32, i32 = torch.float32, torch.int32
        x2, y2 = _as(x, f32), _as(y, f32)
        if x2.shape != y2.shape or x2.dim() != 2:
            raise ValueError("SkipDropPath: x and y must be [rows, C] of the same shape")
        rows, c = x2.shape
        dev = x2.device
        ga = _as(gamma, f32).reshape(-1)
        gt = _as(gate, f32) if gate is not None else None
        rb = _as(row_batch, i32) if gate is not None else None
        if gt is not None and rb.shape[0] != rows:
            raise ValueError("SkipDropPath: one batch id per row expected")
        out = torch.empty_like(x2)
        _lib.check(lib.se3_skip_fwd(_ptr(x2, f32, "x"), _ptr(y2, f32, "y", dev), _ptr(ga, f32, "gamma", dev),
                                    _ptr(gt, f32, "gate", dev), float(gate_keep), _ptr(rb, i32, "row_batch", dev), rows, c,
                                    _ptr(out, f32, "out"), _stream(dev)), "se3_skip_fwd")
        ctx.save_for_backward(x2, ga, gt if gt is not None else torch.empty(0, device=dev),
                              rb if rb is not None else torch.empty(0, dtype=i32, device=dev))
        ctx.gated, ctx.gamma_shape, ctx.gate_keep = gt is not None, gamma.shape, float(gate_keep)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        f32, i32 = torch.float32, torch.int32
        x2, ga, gt, rb = ctx.saved_tensors
        g2 = _as(g, f32)
        rows, c = x2.shape
        dev = x2.device
        ng = ctx.needs_input_grad
        dx = torch.empty_like(x2) if ng[0] else None
        dgamma = torch.empty(c, dtype=f32, device=dev)
        ws = _glue_ws(c, dev)
        _lib.check(lib.se3_skip_bwd(_ptr(g2, f32, "g", dev), _ptr(x2, f32, "x"), _ptr(ga, f32, "gamma"),
                                    _ptr(gt if ctx.gated else None, f32, "gate"), ctx.gate_keep,
                                    _ptr(rb if ctx.gated else None, i32, "row_batch"), rows, c, _ptr(dx, f32, "dx"), _ptr(dgamma, f32, "dgamma"), C.c_void_p(ws.data_ptr()),
                                    ws.numel(), _stream(dev)), "se3_skip_bwd")
        return dx, (g2 if ng[1] else None), (dgamma.reshape(ctx.gamma_shape) if ng[2] else None), None, None, None


class BiasGelu(torch.autograd.Function):
    """``GELU(z + bias)`` (exact erf, torch.nn.GELU default) behind a bias-free GEMM: the bias add, the activation and
    the bias gradient's channel sum in one pass each way (layers/ResNetFormer.py:79-81)."""

    @staticmethod
    def forward(ctx, z, bias):
        lib = _lib.load()
        f32 = torch.float32
        z2 = _as(z, f32)
        rows, c = z2.shape
        dev = z2.device
        b = _as(bias, f32) if bias is not None else None
        out = torch.empty_like(z2)
        _lib.check(lib.se3_affine_act(_ptr(z2, f32, "z"), C.c_void_p(0), C.c_void_p(0), _ptr(b, f32, "bias", dev), rows, c, 1,
                                      _ptr(out, f32, "out"), _stream(dev)), "se3_affine_act")
        ctx.save_for_backward(z2, b if b is not None else torch.empty(0, device=dev))
        ctx.has_b = b is not None
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        f32 = torch.float32
        z2, b = ctx.saved_tensors
        g2 = _as(g, f32)
        rows, c = z2.shape
        dev = z2.device
        dz = torch.empty_like(z2)
        db = torch.empty(c, dtype=f32, device=dev)
        ws = _glue_ws(c, dev)
        _lib.check(lib.se3_bias_gelu_bwd(_ptr(g2, f32, "g", dev), _ptr(z2, f32, "z"), _ptr(b if ctx.has_b else None, f32, "bias"),
                                         rows, c, _ptr(dz, f32, "dz"), _ptr(db, f32, "dbias"), C.c_void_p(ws.data_ptr()),
                                         ws.numel(), _stream(dev)), "se3_bias_gelu_bwd")
        return dz, (db if (ctx.has_b and ctx.needs_input_grad[1]) else None)


class Linear(torch.autograd.Function):
    """``x @ weight.T (+ bias)`` with the weight gradient on the library's row-split TN GEMM (``se3_linear_wgrad``): the
    forward and the input gradient are plain BLAS GEMMs (fine as they are, 20-35 us at 131 k rows), the weight gradient --
    a reduction over every row of the cloud into a ``[n_out, n_in]`` matrix -- is what the generic heuristics run on a
    handful of workgroups (0.28-0.36 ms at 131 k rows against ~0.03 here).  The block's ``torch.nn.Linear`` layers
    (layers/ResNetFormer.py:42-49, 80-86)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_b = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        f32 = torch.float32
        gx = gw = gb = None
        g2 = _as(g, f32)
        if ctx.needs_input_grad[0]:
            gx = g2 @ weight
        if ctx.needs_input_grad[1]:
            lib = _lib.load()
            x2 = _as(x, f32)
            rows, n_in = x2.shape
            n_out = weight.shape[0]
            dev = x2.device
            gw = torch.empty(n_out, n_in, dtype=f32, device=dev)
            ws = _workspace(lib.se3_linear_wgrad_workspace_bytes(rows, n_out, n_in), dev)
            _lib.check(lib.se3_linear_wgrad(_ptr(g2, f32, "grad_y", dev), _ptr(x2, f32, "x"), rows, n_out, n_in,
                                            _ptr(gw, f32, "grad_w"), C.c_void_p(ws.data_ptr()), ws.numel(), _stream(dev)),
                       "se3_linear_wgrad")
        if ctx.has_b and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb
