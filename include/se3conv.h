/*
 * se3conv.h -- C ABI of libse3conv_hip.so, the MI355X (gfx950) implementation of the
 * PNEConvLayerRotEquiv hot path of lisaweijler/SE3Conv3D.
 *
 * Each entry point replaces one function of the reference's native boundary
 * (pybind module `point_cloud_lib_ops`, point_cloud_lib/custom_ops/ops_list.cpp:19-25) or one
 * Python-level stage of PNEConvLayerRotEquiv that the reference runs as a chain of torch ops.
 * The citation next to each declaration names the reference interface it stands in for.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - all floating point is fp32, all indices int32 (keys int64), tensors dense row-major;
 *   - caller allocates every output and the workspace (size from the matching *_workspace_bytes);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream) of the CURRENT device -- the library never switches devices -- and makes no hidden
 *     host synchronisation;
 *   - re-entrant: calls on different streams (or from different threads) share no events, streams
 *     or buffers.  Process-wide state, all of it listed here: (1) the profiling switch below;
 *     (2) se3conv_bwd keeps one internal side stream + fork/join event pair per (device, caller
 *     stream) it has been called on, for running its two branches side by side -- OPT-IN since round 5
 *     (SE3_OVERLAP / SE3_OVERLAP_ROWS / se3_set_overlap_rows: with the kernels as they are the fork loses at every size,
 *     and a fork from a stream that is itself a forked branch of a graph capture crashes this HIP runtime's
 *     hipStreamEndCapture, see se3_set_overlap_rows) --
 *     the side stream is always joined back into `stream` before the call returns, on error paths too.
 *     These objects are only ever created by a call whose stream is NOT being captured into a HIP graph:
 *     every eager se3conv_fwd / se3conv_bwd keeps two spare sets per device ready, a capturing stream that
 *     is new to the library takes a spare, and if there is none (no eager call happened before the capture)
 *     the backward pass does not fork -- same results, branches back to back (counted: se3_side_stream_stats).  At most 16
 *     caller streams per device own a set at a time: past that the least recently used set goes back to the spares
 *     and serves the next new caller stream (never rearranged while a capture is involved), so a process that makes
 *     a stream per request does not grow the table; (3) kernel-variant switches read ONCE from the environment at first
 *     use (A/B knobs, none changes results beyond rounding): listed in the appendix at the end of this header.  A
 *     side-stream set in use by a call is pinned: the cap never hands it to another caller.
 *     `t_save` written by se3conv_fwd must be consumed by se3conv_bwd in the same process (same switches);
 *   - graph capture: every entry point that takes a stream can be captured into a HIP graph (no host synchronisation,
 *     nothing allocated) except the two-phase se3_ball_query_count / _store pair.  The library issues NO hipMemsetAsync
 *     (buffers are zeroed by kernels) and its sorts stay on rocPRIM's merge sort at every size: on the HIP runtime
 *     PyTorch 2.10 ships (7.0.51831, RCCL 2.26.6) a captured graph with memset nodes -- rocPRIM's one-sweep radix sort
 *     issues three per pass -- faults on replay once an RCCL collective has run between two replays (round 4,
 *     tools/debug_up_graph.py).
 *   - return value: SE3_OK (0) or a negative SE3_ERR_* code; no exceptions cross the boundary.
 *
 * Layouts: points [N,3]; frames [N,F,9] = row-major 3x3 per (point,frame)
 * whose COLUMNS are the basis vectors; feature rows are point*F + frame; neighbours [E,2] int32
 * (col0 = sample / output point, col1 = source / input point) grouped by col0; `ends[M]` =
 * inclusive end offset of every sample's group (the reference's `start_ids_`).
 */
#ifndef SE3CONV_H_
#define SE3CONV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE3_OK 0
#define SE3_ERR_INVALID_ARGUMENT (-1) /* null pointer, negative size, misaligned buffer        */
#define SE3_ERR_UNSUPPORTED (-2)      /* shape outside what the kernels implement (see below)  */
#define SE3_ERR_WORKSPACE (-3)        /* workspace too small                                   */
#define SE3_ERR_LAUNCH (-4)           /* HIP reported a launch/runtime error                   */

/* Number of descriptor dimensions of the '6D' relative-rotation form (3 local offset + 6):
 * PNEConvLayerRotEquiv.rel_rot_type = '6D', p_dims = 9 (tasks/SemSeg/seg_models.py:72-79). */
#define SE3_DESC_DIMS 9

typedef struct se3conv_shape {
  int64_t n_in;    /* points of the input (source) cloud                    */
  int64_t n_out;   /* points of the output (sample) cloud                   */
  int64_t n_edges; /* rows of the point-level edge list `neighbors`.  An UPPER BOUND on the edge count E: the kernels
                    * walk the offsets `ends` / `t_ends` and never read a row those do not reach, so the capacity-sized
                    * buffer of se3_ball_query_bounded is passed with n_edges = capacity (rows past ends[-1] unset)   */
  int32_t f_in;    /* frames per input point  (PointcloudRotEquiv.n_frames_) */
  int32_t f_out;   /* frames per output point                               */
  int32_t c_in;    /* input feature channels                                */
  int32_t c_out;   /* output feature channels                               */
  int32_t num_basis; /* K = p_num_basis, any K >= 1 (the reference's CUDA op takes 8, 16, 32, 64,
                      * feat_basis_utils.cuh:35-41).  The MFMA kernels work on 32 basis functions: for K != 32
                      * se3conv_fwd / se3conv_bwd run ceil(K / 32) slices of 32 -- the sum over k is separable, a short
                      * slice is padded with basis functions whose axes, bias and conv weights are zero (exact) --
                      * accumulate the outputs and write every slice's columns of the parameter gradients.  For
                      * K != 32 `t_save` is neither written nor read (pass NULL): backward recomputes T per slice */
  int32_t precision; /* SE3_PRECISION_*: arithmetic of the contractions (inputs/outputs are fp32)  */
} se3conv_shape;

/* Arithmetic of the three contractions (kernel MLP, basis (x) feature aggregate, C_in*K -> C_out):
 *   FP32   : v_mfma_f32_32x32x2_f32 -- exact fp32 products, fp32 accumulate (reference numerics up to
 *            summation order; ~1e-6 relative to the oracle).
 *   BF16X3 : every fp32 operand x is split into hi = bf16(x), lo = bf16(x - hi) and every product is
 *            evaluated as hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_bf16 with fp32 accumulate
 *            (~1e-5 relative to the oracle; the north-star tolerance is 1e-4; 16x the MFMA rate at
 *            3 products).  `t_save` is then an opaque buffer of the same size (packed hi/lo words, or
 *            3-byte rows -- the library picks per shape) and must be passed back to se3conv_bwd
 *            with the same precision, shape and process environment. */
/*   BF16X3_T16 (opt-in, round 4): the arithmetic of BF16X3 with the row-sized intermediates T and U (the tensors
 *            [rows, C, K] between an edge kernel and its contraction) kept in a 2.25-byte block floating-point format
 *            instead of 3-byte rows: 4 consecutive channels of one basis function share an 8-bit power-of-two exponent
 *            and keep 16-bit mantissas.  A quarter fewer bytes through HBM for those tensors; the rounding of T / U
 *            alone puts ~2e-5 into every output and gradient (BF16X3: ~6e-6), i.e. ~2.5e-5 in all against the
 *            north-star tolerance of 1e-4.  Shapes the format is not implemented for (rows that are not a multiple of
 *            64 channels, odd frame counts at 64 channels) run exactly as in BF16X3. */
#define SE3_PRECISION_FP32 0
#define SE3_PRECISION_BF16X3 1
#define SE3_PRECISION_BF16X3_T16 2

/* Library / build identification.  SE3_ABI_VERSION changes whenever an entry point's signature does: 2 = round 3
 * (se3_skip_fwd / se3_skip_bwd take gate_keep, se3_bn_fwd takes num_batches_tracked; entry points added since 1:
 * se3_knn_query_pair, se3_grid_pick, se3_rows_gather / _scatter, se3_rot_tensors_rel, se3_csr_transpose_bounded,
 * se3_side_stream_stats, se3_linear_wgrad).  A binding should compare se3_abi_version() with the header it was written
 * against (se3conv3d_amd/_lib.py does). */
/* 3 = round 4: se3_side_stream_stats fills FIVE counters (was three); SE3_PRECISION_BF16X3_T16 and
 * se3conv_intermediate_row_bytes added. */
/* 4 = round 4: se3_csr_transpose / _bounded and se3conv_bwd take `t_edge_ids` (optional).
 * (round 5 added se3_set_overlap_rows and se3conv_bwd_needs_t; no signature changed.) */
/* 5 = round 6: se3conv_fwd_prepared / se3conv_bwd_prepared (operands prepared once per step and shared between calls:
 * struct se3conv_prepared) added; the version also covers round 5's two additions, so that a binding that needs them
 * and meets an older library fails with the "rebuild" message instead of a missing symbol. */
#define SE3_ABI_VERSION 5
int se3_abi_version(void);
const char* se3_error_string(int code);
/* Bytes per element of the row-sized intermediates [rows, C, K] the operator moves through memory for this shape --
 * which = 0: T (forward; read again by the weight gradient), 1: U (feature gradient), 2: grad_T.  4 = fp32 / packed
 * hi|lo words, 3 = 3-byte rows (the library picks per shape: rows of <= 32 or >= 64 channels, even); negative = SE3_ERR_*.
 * For traffic models (bench.py), so that they need not hard-code what the library picked. */
int se3conv_intermediate_bytes_per_element(const se3conv_shape* s, int which);
/* The same as exact bytes per ROW of that tensor (rows x this = what one pass over it moves): 4 / 3 bytes per element, or
 * 72 bytes per channel (2.25 per element) in the T16 block format of SE3_PRECISION_BF16X3_T16, which the call above
 * reports as 2.  Negative = SE3_ERR_*. */
int64_t se3conv_intermediate_row_bytes(const se3conv_shape* s, int which);

/* ---------------------------------------------------------------------------------------------
 * compute_keys  <-  point_cloud_lib_ops.compute_keys
 *   (custom_ops/ball_query/compute_keys.cuh:26-31, kernel compute_keys.cu:32-71,
 *    grid_utils.cuh:57-93).  D = 3 only.
 *   pts [n,3] f32, batch_ids [n] i32, aabb_min [B,3] f32, num_cells [3] i32 (device),
 *   cell_size [3] f32 (device) -> keys [n] i64.
 * ------------------------------------------------------------------------------------------- */
int se3_compute_keys(const float* pts, const int32_t* batch_ids, const float* aabb_min,
                     const int32_t* num_cells, const float* cell_size, int64_t n, int64_t* keys,
                     void* stream);

/* Per-batch axis-aligned bounding boxes: aabb_min / aabb_max [n_batches,3] f32 (batches without points:
 * +inf / -inf).  Stands in for the torch_scatter scatter_min / scatter_max calls of
 * point_cloud_lib/custom_ops/BallQuery.py:35-36 and point_cloud_lib/pc/BoundingBox.py:17-18 (the callers then
 * apply their -1e-6 / +1e-6 shifts themselves). */
int se3_batch_aabb(const float* pts, const int32_t* batch_ids, int64_t n, int32_t n_batches, float* aabb_min,
                   float* aabb_max, void* stream);

/* ---------------------------------------------------------------------------------------------
 * hierarchy build (scope row f-2)  <-  point_cloud_lib/pc/Grid.py:20-51, pc/GridSubSample.py:63-93,
 *                                      pc/PointHierarchy.py:40-57
 * se3_grid_subsample: one level of grid-average sub-sampling.  Bounding boxes (+-1e-6, BoundingBox.py:17-18), cell
 *   counts (Grid.py:28-29), keys (ComputeKeys), then what torch.unique(return_inverse=True) + argsort give:
 *     cell_ids   [n]  cell of every point, cells numbered in ascending key order
 *     sorted_ids [n]  point ids grouped by cell (input order inside a cell)
 *     cell_ends  [n]  inclusive end offsets into sorted_ids, first *n_cells entries valid
 *     n_cells    [1]  (device) number of occupied cells = points of the next level; the caller's one host sync
 *     cell_pts   [n,3] / cell_batch_ids [n]: first *n_cells rows = the next level (cell means / batch ids)
 * se3_segment_pool / se3_segment_unpool: pool_tensor ("avg" = scatter_mean, "max" = scatter_max, also min / sum)
 *   over those cells and the maps back to the rows: mode sum = the gather of upsample_tensor (GridSubSample.py:93)
 *   and, as a pool, its gradient; avg / max / min unpool = the gradients of the pools (max / min route to the row
 *   that supplied the extremum, `arg` [n_cells,C] from the forward, like torch_scatter).
 * Modes: 0 avg, 1 max, 2 min, 3 sum.  Summation order is fixed (input order inside a cell). */
#define SE3_POOL_AVG 0
#define SE3_POOL_MAX 1
#define SE3_POOL_MIN 2
#define SE3_POOL_SUM 3
size_t se3_grid_subsample_workspace_bytes(int64_t n, int32_t n_batches);
int se3_grid_subsample(const float* pts, const int32_t* batch_ids, int64_t n, int32_t n_batches, float cell_size,
                       void* workspace, size_t workspace_bytes, int32_t* cell_ids, int32_t* sorted_ids,
                       int32_t* cell_ends, int32_t* n_cells, float* cell_pts, int32_t* cell_batch_ids, void* stream);
int se3_segment_pool(const float* src, const int32_t* sorted_ids, const int32_t* cell_ends, int64_t n_cells,
                     int32_t channels, int32_t mode, float* out, int32_t* arg, void* stream);
int se3_segment_unpool(const float* cell_vals, const int32_t* cell_ids, const int32_t* cell_ends, const int32_t* arg,
                       int64_t n, int32_t channels, int32_t mode, float* out, void* stream);

/* Random one-point-per-cell sub-sampling  <-  GridSubSample(..., p_rnd_sample=True), pc/GridSubSample.py:43-54, 66-67,
 * 83-91 (the task scripts build their output cloud with it every step: tasks/SemSeg/train_dfaust_rot.py:143-149).
 * se3_grid_pick: u [n_cells] f32 in [0,1) (device; the caller draws it, e.g. torch.rand on the GPU) ->
 *   ids [n_cells] = start(c) + floor(u[c] * count(c)): positions in the cell-sorted point list (the reference's `ids_`)
 *   picked [n_cells] = sorted_ids[ids[c]]: the point that represents cell c.
 *   floor(u * count) is clamped to count - 1 (in fp32 the product can round up to count, which in the reference selects
 *   a point of the next cell).  No host synchronisation.
 * se3_rows_gather: out[r] = src[idx[r]], rows of `row_bytes` bytes of any element type (__subsample_tensor__ :67).
 * se3_rows_scatter: out[idx[r]] = src[r] for unique idx; the caller zero-fills out (__upsample_tensor__ :83-91, and the
 *   gradient of the gather). */
int se3_grid_pick(const int32_t* cell_ends, const int32_t* sorted_ids, const float* u, int64_t n_cells, int32_t* ids,
                  int32_t* picked, void* stream);
int se3_rows_gather(const void* src, const int32_t* idx, int64_t n_out, int64_t row_bytes, void* out, void* stream);
int se3_rows_scatter(const void* src, const int32_t* idx, int64_t n_src, int64_t row_bytes, void* out, void* stream);

/* Frame pooling (scope row f-3)  <-  PointcloudRotEquiv.feature_pooling (pc/PointcloudRotEquiv.py:224-251): the F rows
 * point*F + frame of x [n_points*F, C] -> out [n_points, C]; `arg` [n_points, C] (max / min only) = winning frame.
 * se3_frame_unpool is its gradient: grad_out [n_points, C] -> grad_x [n_points*F, C]. */
int se3_frame_pool(const float* x, int64_t n_points, int32_t frames, int32_t channels, int32_t mode, float* out,
                   int32_t* arg, void* stream);
int se3_frame_unpool(const float* grad_out, const int32_t* arg, int64_t n_points, int32_t frames, int32_t channels,
                     int32_t mode, float* grad_x, void* stream);

/* Grid parameters of a ball query exactly as point_cloud_lib/custom_ops/BallQuery.py:34-38 builds them, in one call
 * and without a host sync: aabb_min [n_batches,3] = per-batch minimum - 1e-6; num_cells [3] = max over batches of
 * int(((max - 1e-6) - aabb_min) / radius) + 1.  aabb_max_scratch: [n_batches,3] floats of scratch. */
int se3_ball_query_grid(const float* pts_src, const int32_t* batch_src, int64_t n_src, int32_t n_batches, float radius,
                        float* aabb_min, float* aabb_max_scratch, int32_t* num_cells, void* stream);
/* The same from bounding boxes that are already known (se3_batch_aabb's box_min / box_max [n_batches,3] of the SOURCE
 * cloud): a cloud is the source of three or four queries per step in the reference's networks (same-level, down, up) and
 * its boxes do not depend on the radius -- one small launch instead of a pass over the points. */
int se3_ball_query_grid_from_box(const float* box_min, const float* box_max, int32_t n_batches, float radius,
                                 float* aabb_min, int32_t* num_cells, void* stream);

/* ---------------------------------------------------------------------------------------------
 * ball query  <-  point_cloud_lib_ops.ball_query
 *   (custom_ops/ball_query/ball_query.cuh:30-38, host ball_query.cu:22-103; max_neighbors = 0,
 *    the only value any model path uses, BQNeighborhood.py:20).
 * Two phases so that the caller (PyTorch) allocates the edge list between them:
 *   count: keys -> radix sort -> 9 pencil windows per sample -> per-sample count -> `ends`
 *          (inclusive scan).  The edge total is ends[n_dst-1]; reading it is the caller's one
 *          host sync (the reference has four, ball_query.cu:46,49,50 + store_neighbors.cu:264).
 *   store: second pass over the same windows; writes neighbors[E,2] = (sample, source) grouped
 *          by sample.  Order inside a sample is deterministic here (pencil order, then sorted
 *          position); the reference's is not (atomics).
 * Predicate: sqrt(sum(((s - p) * (1/r))^2)) < 1 in fp32 and equal batch id
 *   (count_neighbors.cu:84-89).  `aabb_min`/`num_cells` as BallQuery.py:34-38 builds them.
 * The workspace written by `count` must be passed unchanged to `store`.
 * ------------------------------------------------------------------------------------------- */
/* 0 when the source set is small enough for the all-pairs path of the two phases below (aabb_min / num_cells may
 * then be NULL: no boxes, keys or sort are needed), 1 when they search through the cell grid. */
int se3_ball_query_needs_grid(int64_t n_src);
size_t se3_ball_query_workspace_bytes(int64_t n_src, int64_t n_dst);
int se3_ball_query_count(const float* pts_src, const float* pts_dst, const int32_t* batch_src,
                         const int32_t* batch_dst, const float* aabb_min, const int32_t* num_cells,
                         float radius, int64_t n_src, int64_t n_dst, void* workspace,
                         size_t workspace_bytes, int32_t* ends, void* stream);
int se3_ball_query_store(const float* pts_dst, const int32_t* batch_dst, float radius,
                         int64_t n_src, int64_t n_dst, const void* workspace,
                         size_t workspace_bytes, const int32_t* ends, int64_t n_edges,
                         int32_t* neighbors, void* stream);

/* The same query in ONE call without a host round trip (what lets neighbourhood builds be captured into a HIP graph and
 * keeps the host running ahead in eager mode; the reference synchronises four times per query, ball_query.cu:46,49,50 +
 * store_neighbors.cu:264): the caller passes an edge buffer of `capacity` rows, e.g. sized from the previous step.
 *   info [2] int32 (device): info[0] = true number of edges E, info[1] = 1 when E > capacity.
 * On overflow the list is truncated: `ends` are clamped to `capacity`, so consumers never read past the buffer, the
 * first `capacity` edges (sample-major order) are present, and the caller reruns with a larger buffer once it has seen
 * the flag.  Rows [E, capacity) of `neighbors` are left untouched.
 *   n_batches (host): number of batch elements (0 = unknown); for 1 or 2 the search uses 32-bit cell keys of its own (a
 *   fixed stride of 1024 cells per dimension instead of the reference's cell-count products: 4 radix passes instead
 *   of 8 -- only the edge set is defined, and it is the same).
 *   sources [capacity] int32 (optional, may be NULL): column 1 of `neighbors` as a dense array -- for a cloud against
 *   itself the radius graph is symmetric, so this IS the source-major list `t_samples` se3conv_bwd wants (with
 *   `t_ends` = `ends`), written by the same store pass instead of a strided copy afterwards. */
int se3_ball_query_bounded(const float* pts_src, const float* pts_dst, const int32_t* batch_src,
                           const int32_t* batch_dst, const float* aabb_min, const int32_t* num_cells, float radius,
                           int64_t n_src, int64_t n_dst, int32_t n_batches, void* workspace, size_t workspace_bytes,
                           int64_t capacity, int32_t* neighbors, int32_t* sources, int32_t* ends, int32_t* info,
                           void* stream);
/* The same with the SOURCE cloud's cell grid (cell keys, their sorted order, the points in that order) in a buffer of the
 * caller's, so that the queries of a step that search one source cloud with one radius -- a level's same-level, down and
 * up convolutions (PointHierarchy.create_neighborhood, pc/PointHierarchy.py:60-79, builds each from scratch) -- sort it
 * once (round 6: 10 of the 26 queries of a DFaust step are such repeats, each ~50 us of key, sort and gather launches).
 *   grid [se3_ball_query_grid_bytes(n_src)] (device), grid_valid: 0 = build the grid into it (then search), 1 = it holds
 *   the grid an earlier call built from the same pts_src / batch_src / aabb_min / num_cells / radius / n_batches: search
 *   only.  The workspace is sized as for se3_ball_query_bounded.  Same results, bit for bit. */
size_t se3_ball_query_grid_bytes(int64_t n_src);
int se3_ball_query_bounded_shared(const float* pts_src, const float* pts_dst, const int32_t* batch_src,
                                  const int32_t* batch_dst, const float* aabb_min, const int32_t* num_cells, float radius,
                                  int64_t n_src, int64_t n_dst, int32_t n_batches, void* grid, size_t grid_bytes,
                                  int32_t grid_valid, void* workspace, size_t workspace_bytes, int64_t capacity,
                                  int32_t* neighbors, int32_t* sources, int32_t* ends, int32_t* info, void* stream);

/* Source-major (transposed) copy of an edge list, used by the backward pass in place of the
 * reference's global float atomics on the feature gradient (feat_basis_proj_grads.cu:126,140):
 * t_samples[E] = sample id of every edge, grouped by source point, t_ends[n_src] = inclusive end
 * offsets.  se3_csr_transpose* returns every group in ascending sample order; se3conv_bwd does not
 * rely on any order INSIDE a group (a caller may pass a list grouped by source in any order -- the
 * Python drop-in builds long-segment lists as a second ball query with the clouds' roles swapped). */
size_t se3_csr_transpose_workspace_bytes(int64_t n_edges);
/* `t_edge_ids` (optional, may be NULL; ABI 4): [E] the position of every entry in the sample-major list -- entry j of the
 * result is row t_edge_ids[j] of `neighbors`.  se3conv_bwd's edge-major feature gradient (a convolution with many more
 * input than output rows) gathers per-edge rows through it; without it that kernel looks every edge up itself. */
int se3_csr_transpose(const int32_t* neighbors, int64_t n_edges, int64_t n_src, void* workspace,
                      size_t workspace_bytes, int32_t* t_samples, int32_t* t_ends, int32_t* t_edge_ids, void* stream);
/* The same for a capacity-sized buffer of se3_ball_query_bounded between two DIFFERENT clouds: `neighbors` has n_rows
 * rows of which only the first *n_valid (device word, e.g. info[0] of that call; clamped to n_rows) are edges -- the
 * unset tail is ignored (it sorts behind every group and no offset reaches it), t_samples [n_rows], t_ends [n_src].
 * Workspace: se3_csr_transpose_workspace_bytes(n_rows).  No host synchronisation.  If the buffer overflowed
 * (info[1] != 0) the forward list is truncated and so is its transpose: outputs and gradients then belong to the
 * truncated graph, consistently -- rebuild with a larger buffer.  (That consistency holds for THIS call.  A source-major
 * list obtained any other way from an overflowed buffer -- the role-swapped second query of the Python drop-in, which
 * truncates in source-major order -- is a different sub-graph: after an overflow the gradients are not those of the
 * forward pass and the step must be redone, which the overflow flag is there to tell.) */
int se3_csr_transpose_bounded(const int32_t* neighbors, int64_t n_rows, const int32_t* n_valid, int64_t n_src,
                              void* workspace, size_t workspace_bytes, int32_t* t_samples, int32_t* t_ends, int32_t* t_edge_ids,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * rot tensors  <-  PNEConvLayerRotEquiv.get_rot_tenors
 *   (point_cloud_lib/layers/PNEConvLayerRotEquiv.py:62-128).  Materialises what the reference
 *   caches: desc[E',9] (3 local offsets + 6-D relative rotation), frame-level edges
 *   fe_neighbors[E',2] = (s*F_out+a, p*F_in+b) sorted by col0 and fe_ends[N_out*F_out]
 *   (inclusive).  E' = E*F_out*F_in.  Only needed for API parity / FeatBasisProj users; the
 *   fused operator below never materialises these.
 * ------------------------------------------------------------------------------------------- */
int se3_rot_tensors(const float* pts_in, const float* pts_out, const float* frames_in,
                    const float* frames_out, const int32_t* neighbors, const int32_t* ends,
                    const float* rho, const se3conv_shape* shape, float* desc,
                    int32_t* fe_neighbors, int32_t* fe_ends, void* stream);
/* The same with the relative rotation R_out^T R_in in any representation of get_relative_rot
 * (point_cloud_lib/pc/RotationFunctions.py:549-600; the factory's p_rel_rot, PNEConvLayerRotEquiv.py:236-281):
 * "6D" = its first two rows (desc [E',9], what the fused operator implements), "matrix" = all nine entries
 * (desc [E',12]), "quaternion" = real part first (desc [E',7]).  The two others run through the materialised
 * formulation (this call + se3_feat_basis_proj); no shipped configuration uses them. */
#define SE3_REL_ROT_6D 0
#define SE3_REL_ROT_MATRIX 1
#define SE3_REL_ROT_QUATERNION 2
int se3_rot_tensors_rel(const float* pts_in, const float* pts_out, const float* frames_in,
                        const float* frames_out, const int32_t* neighbors, const int32_t* ends,
                        const float* rho, const se3conv_shape* shape, int32_t rel_rot, float* desc,
                        int32_t* fe_neighbors, int32_t* fe_ends, void* stream);

/* ---------------------------------------------------------------------------------------------
 * feat_basis_proj / feat_basis_proj_grad  <-  point_cloud_lib_ops.feat_basis_proj{,_grad}
 *   (custom_ops/feature_aggregation/feat_basis_proj.cuh:27-31, feat_basis_proj_grads.cuh:29-34;
 *    Python wrapper custom_ops/FeatBasisProj.py:10-65).
 *   T[m,c,k] = sum_{e in [ends[m-1], ends[m])} basis[e,k] * feat[neighbors[e,1], c]
 *   basis [E,K] f32, feat [n_feat,C] f32, neighbors [E,2] i32, ends [M] i32 -> T [M,C,K].
 *   grad: gT [M,C,K] -> g_feat [n_feat,C], g_basis [E,K]  (both fully overwritten).
 *   Any C >= 1 and K in {8,16,32,64} (the reference's set, feat_basis_utils.cuh:35-41).
 * ------------------------------------------------------------------------------------------- */
int se3_feat_basis_proj(const float* basis, const float* feat, const int32_t* neighbors,
                        const int32_t* ends, int64_t n_edges, int64_t n_rows, int64_t n_feat,
                        int32_t channels, int32_t num_basis, float* out, void* stream);
int se3_feat_basis_proj_grad(const float* basis, const float* feat, const int32_t* neighbors,
                             const int32_t* ends, const float* grad_out, int64_t n_edges,
                             int64_t n_rows, int64_t n_feat, int32_t channels, int32_t num_basis,
                             float* g_feat, float* g_basis, void* stream);

/* ---------------------------------------------------------------------------------------------
 * the fused operator  <-  PNEConvLayerRotEquiv.__compute_convolution__
 *   (point_cloud_lib/layers/PNEConvLayerRotEquiv.py:160-216: get_rot_tenors :62-128, kernel MLP
 *    :199-203, FeatBasisProj :206-207, einsum :210, scalings :213-216).
 *
 *   out[(s,a),o] = (nu/F_in) * sum_{(s,p) in E} sum_b sum_k sum_i
 *                    GELU_erf([rho*(x_p-y_s)^T R_{s,a}, rows 0,1 of R_{s,a}^T R_{p,b}] . A + beta)_k
 *                    * feat[(p,b),i] * W[i,k,o]
 *
 *   proj_axes [9,K], proj_biases [K], conv_weights [C_in,K,C_out]; rho, nu: device scalars
 *   (the module's norm_neigh_dist_ / norm_num_neighs_ buffers, IConvLayer.py:33-36).
 *   out [N_out*F_out, C_out].  `t_save` (optional, may be NULL): [N_out*F_out, C_in, K] --
 *   the aggregated basis tensor, kept for the weight gradient like autograd keeps the
 *   reference's `result_tensor`.
 *
 * backward: grad_out [N_out*F_out, C_out] -> any of grad_feat [N_in*F_in, C_in],
 *   grad_axes [9,K], grad_biases [K], grad_weights [C_in,K,C_out] (NULL = not wanted).  No
 *   gradient flows to points or frames (the reference builds geometry under no_grad, :67).
 *   `t_samples`/`t_ends` = se3_csr_transpose of `neighbors` (needed only for grad_feat); `t_edge_ids` = its optional
 *   third result (may be NULL);
 *   `t_save` = the tensor written by the forward (needed for grad_weights; if NULL it is
 *   recomputed into the workspace).
 * ------------------------------------------------------------------------------------------- */
size_t se3conv_fwd_workspace_bytes(const se3conv_shape* shape, int save_t);
int se3conv_fwd(const float* pts_in, const float* pts_out, const float* frames_in,
                const float* frames_out, const int32_t* neighbors, const int32_t* ends,
                const float* feat, const float* proj_axes, const float* proj_biases,
                const float* conv_weights, const float* rho, const float* nu,
                const se3conv_shape* shape, float* out, float* t_save, void* workspace,
                size_t workspace_bytes, void* stream);

/* Whether se3conv_bwd (with parameter gradients wanted) makes use of the forward pass's T (`t_save`): 1 = yes -- without
 * it the weight gradient re-runs the forward edge pass; 0 = no -- the weight gradient is taken from U, the tensor the
 * feature gradient's transposed pass produces anyway (dW[i,k,o] = alpha sum_p f[p,i] U[p,o,k]; split-bf16 modes, U form of
 * the feature gradient, C_in a multiple of 4, even C_out), so se3conv_fwd can be called with t_save = NULL and the layer keeps
 * no row-sized activation (0.8 GB per layer at the headline shape).  `want_feat`: whether grad_feat will be asked for.
 * When a `t_save` is passed anyway, se3conv_bwd still prefers U where U has fewer entries than T (an up-convolution). */
int se3conv_bwd_needs_t(const se3conv_shape* shape, int want_feat);
size_t se3conv_bwd_workspace_bytes(const se3conv_shape* shape, int want_feat, int want_params,
                                   int have_t_save);
int se3conv_bwd(const float* pts_in, const float* pts_out, const float* frames_in,
                const float* frames_out, const int32_t* neighbors, const int32_t* ends,
                const int32_t* t_samples, const int32_t* t_ends, const int32_t* t_edge_ids, const float* feat,
                const float* proj_axes, const float* proj_biases, const float* conv_weights,
                const float* rho, const float* nu, const float* t_save, const float* grad_out,
                const se3conv_shape* shape, float* grad_feat, float* grad_axes,
                float* grad_biases, float* grad_weights, void* workspace, size_t workspace_bytes,
                void* stream);

/* Operands the library derives from a call's inputs before its first kernel can run, kept by the CALLER across calls
 * (round 6; the reference has no counterpart: its get_rot_tenors memo, PNEConvLayerRotEquiv.py:53,71-126, is the closest
 * thing -- a per-neighbourhood cache of the descriptors -- and costs a device-to-host copy and a SHA-256 per call):
 *   geom_in / geom_out   packed 64-byte geometry records of the input / output cloud, [rows, 16] fp32 with rows = points x
 *                        frames: a function of the cloud alone, so every convolution that touches the cloud in a step --
 *                        forward and backward, every layer of a level -- can share one image (same cloud on both sides:
 *                        pass the same pointer twice);
 *   feat_words           packed hi | lo words of the input features, [N_in*F_in*C_in] uint32 (split-bf16 modes only):
 *                        written by the forward call, read again by the backward call of the same layer.
 * A pointer may be NULL (the call then builds that operand in its workspace as se3conv_fwd / se3conv_bwd do).  `*_valid`
 * = 0: the buffer is filled by THIS call, inside its one preparation launch (no extra launch), and may be handed to later
 * calls with `*_valid` = 1; the struct itself is never written.  The caller owns validity: records follow their cloud's
 * points and frames, feature words the features.  Inside a captured graph the call that fills a buffer and the calls
 * that read it must be captured together (a replay re-runs the fill). */
typedef struct se3conv_prepared {
  float* geom_in;
  float* geom_out;
  uint32_t* feat_words;
  int32_t geom_in_valid, geom_out_valid, feat_words_valid;
} se3conv_prepared;
/* se3conv_fwd / se3conv_bwd with `prepared` (NULL = exactly those calls).  K != 32 ignores it. */
int se3conv_fwd_prepared(const float* pts_in, const float* pts_out, const float* frames_in,
                         const float* frames_out, const int32_t* neighbors, const int32_t* ends,
                         const float* feat, const float* proj_axes, const float* proj_biases,
                         const float* conv_weights, const float* rho, const float* nu,
                         const se3conv_shape* shape, float* out, float* t_save, void* workspace,
                         size_t workspace_bytes, void* stream, const se3conv_prepared* prepared);
int se3conv_bwd_prepared(const float* pts_in, const float* pts_out, const float* frames_in,
                         const float* frames_out, const int32_t* neighbors, const int32_t* ends,
                         const int32_t* t_samples, const int32_t* t_ends, const int32_t* t_edge_ids, const float* feat,
                         const float* proj_axes, const float* proj_biases, const float* conv_weights,
                         const float* rho, const float* nu, const float* t_save, const float* grad_out,
                         const se3conv_shape* shape, float* grad_feat, float* grad_axes,
                         float* grad_biases, float* grad_weights, void* workspace, size_t workspace_bytes,
                         void* stream, const se3conv_prepared* prepared);

/* ---------------------------------------------------------------------------------------------
 * reference frames (scope row f-1: upstream of the operator, frames are an input of the hot path)
 *   se3_knn_query  <-  point_cloud_lib_ops.knn_query (custom_ops/knn_query/knn_query.cuh:25-28, kernel
 *     knn_query.cu:18-132): exact k nearest neighbours inside the point's batch element (batch ids sorted),
 *     the point itself first, ascending distance, ties to the lower index, -1 padded; k <= 64 (the reference
 *     kernel's limit, knn_query.cu:167).  pts [n,3] f32, batch_ids [n] i32 -> out [n,k] i32.
 *   se3_knn_query_pair <- the torch_cluster.knn call of pc/KnnNeighborhood.py:77-84 (neighbourhoods between two
 *     clouds): for every query point the k nearest SOURCE points of the same batch element, same order and padding;
 *     both batch-id arrays sorted.  src_pts [n_src,3], q_pts [n_q,3] -> out [n_q,k] i32 (source indices).
 *   se3_pca_frames <-  sample_reference_frames_pca (point_cloud_lib/pc/RotationFunctions.py:307-406):
 *     covariance of the k neighbours (missing ones = the point itself), symmetric 3x3 eigen-decomposition,
 *     right-handed orientation fix and the sign-flipped copies.  axis_fixed < 0: frames [n,4,9] (eigenvalues
 *     ascending, column c scaled by the patterns (1,1,1),(1,-1,-1),(-1,1,-1),(-1,-1,1)); axis_fixed = 1 or 2:
 *     frames [n,2,9] (that coordinate zeroed, eigenvalues descending, patterns (1,1,1),(-1,-1,1), the fixed axis
 *     as +e_axis -- its sign is implementation-defined in the reference's LAPACK call).
 * ------------------------------------------------------------------------------------------- */
int se3_knn_query(const float* pts, const int32_t* batch_ids, int64_t n, int32_t k, int32_t* out,
                  void* stream);
int se3_knn_query_pair(const float* src_pts, const int32_t* src_batch, int64_t n_src, const float* q_pts,
                       const int32_t* q_batch, int64_t n_q, int32_t k, int32_t* out, void* stream);
/* Same result through a cell grid (the role of the reference's sorted sweep, knn_query.cu:52-128: prune the
 * candidates, stay exact): cells of size cell_size[0] (device scalar, same value in [0..2]) over the per-batch
 * boxes aabb_min [B,3] / num_cells [3] (device, as for se3_compute_keys); a query whose k-th candidate inside its
 * 27 cells is farther than one cell is recomputed by the all-pairs scan, so any cell size gives the exact
 * answer and a good one (about the expected k-NN distance) gives it ~100x faster on large clouds. */
size_t se3_knn_query_grid_workspace_bytes(int64_t n);
/* Grid parameters for se3_knn_query_grid from the per-batch boxes (se3_batch_aabb), one launch (round 4; as a dozen torch
 * calls and a bincount they cost more than the search itself): cell = cell_factor x the k-NN distance a uniformly
 * filled box of each batch element's extent and point count would have (the larger of the volume, area and length
 * estimates, maximum over the batch elements; batch ids sorted) -> aabb_min [B,3] (box minima - 1e-6), num_cells [3],
 * cell_size [3] (one value three times), all on the device.  Only a speed knob: the grid search is exact for any cell. */
int se3_knn_grid_params(const int32_t* batch_ids, int64_t n, const float* box_min, const float* box_max, int32_t n_batches,
                        int32_t k, float cell_factor, float* aabb_min, int32_t* num_cells, float* cell_size, void* stream);
int se3_knn_query_grid(const float* pts, const int32_t* batch_ids, const float* aabb_min, const int32_t* num_cells,
                       const float* cell_size, int64_t n, int32_t k, int32_t* out, void* workspace,
                       size_t workspace_bytes, void* stream);
int se3_pca_frames(const float* pts, const int32_t* knn, int64_t n, int32_t k, int32_t axis_fixed,
                   float* frames, void* stream);
/* The random choice among a point's PCA frames (point_cloud_lib/pc/PointcloudRotEquiv.py:100-117, 146-167: torch.multinomial
 * without replacement on equal weights, then a gather): out[p, j] = all_frames[p, perm_p[j]] for j < n_frames, perm_p = the
 * order of point p's n_all uniform draws (ascending; ties to the lower index) -- a uniformly random permutation per point,
 * the draws being the caller's (torch.rand on the device: the library owns no generator).  all_frames [n, n_all, 9],
 * draws [n, n_all] f32 -> out [n, n_frames, 9]; n_all <= 8.  One launch instead of the sort + gather (round 6). */
int se3_shuffle_frames(const float* all_frames, const float* draws, int64_t n, int32_t n_all, int32_t n_frames, float* out,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row-wise glue of a block around the convolution (scope row f-3)  <-  the torch element-wise / reduction passes of
 *   layers/ResNetFormer.py:64-88 (norm -> conv -> skip; norm -> linear -> GELU -> linear -> skip),
 *   layers/BatchNormPC.py:22-32 (BatchNorm1d, momentum 0.2), layers/SkipConnection.py (x * gamma_ + y with
 *   gamma_ [1,C]) and layers/DropPathPC.py:30-46 (one keep / drop decision per batch element, indexed through the
 *   per-row batch ids -- batch_ids_considering_frames_ for clouds with frames).
 * All tensors [rows, C] fp32 row-major; per-channel vectors [C]; C <= 1024 (C > 256 needs C % 4 == 0).  Channel
 * sums are accumulated in fp64 in a fixed order (no atomics).  `workspace`: se3_glue_workspace_bytes(C) bytes.
 *   se3_bn_fwd        training-mode torch.nn.BatchNorm1d: save_mean[c], save_invstd[c] = 1 / sqrt(biased var + eps),
 *                     y = (x - mean) * invstd * weight + bias, running_mean / running_var (may be NULL) updated in place
 *                     with `momentum` and the unbiased variance, *num_batches_tracked (int64, may be NULL) += 1;
 *                     weight / bias NULL = 1 / 0
 *   se3_affine_act    y = act((x - center[c]) * scale[c] + shift[c]); center / scale / shift may be NULL (0 / 1 / 0);
 *                     act 0 = none, 1 = exact-erf GELU: eval-mode batch norm (center = running mean, scale = weight /
 *                     sqrt(running var + eps), shift = bias), bias + GELU behind a bias-free GEMM
 *   se3_bn_bwd        dbeta[c] = sum dy, dgamma[c] = sum dy * xhat, dx = gamma * invstd * (dy - dbeta/N - xhat * dgamma/N)
 *                     with xhat = (x - mean) * invstd  (gamma NULL = 1)
 *   se3_skip_fwd      out = x * gamma[c] * factor(row_batch[r]) + y.  gate NULL: factor 1 (no drop path); gate_keep == 0:
 *                     factor = gate[b] (the caller built floor(keep + u) / keep); gate_keep = keep_prob > 0: gate holds
 *                     the uniform draws u themselves (torch.rand, DropPathPC.py:38) and factor = floor(keep + u) *
 *                     float(1 / keep) is evaluated in the kernel -- torch's add / floor / div launches folded in
 *   se3_skip_bwd      dx = g * gamma[c] * gate (dx may be NULL), dgamma[c] = sum_r g * x * gate   (dy = g)
 *   se3_bias_gelu_bwd dz = g * GELU'(z + bias), dbias[c] = sum_r dz   (bias NULL = 0)
 *   se3_linear_wgrad  grad_w[n_out, n_in] = grad_y[rows, n_out]^T x[rows, n_in]: the weight gradient of the block's
 *                     torch.nn.Linear layers (linear_1_, linear_2_, skip_conv_; ResNetFormer.py:42-49, 80-86) -- a reduction
 *                     over every row of the cloud into a small matrix, which generic BLAS heuristics run on the few
 *                     workgroups its output tiles give; here the rows are split over the chip (fp32 MFMA, fixed-order
 *                     reduction of the partials).  `workspace`: se3_linear_wgrad_workspace_bytes(rows, n_out, n_in)
 * ------------------------------------------------------------------------------------------- */
size_t se3_glue_workspace_bytes(int32_t c);
int se3_bn_fwd(const float* x, const float* weight, const float* bias, int64_t rows, int32_t c, float eps, float momentum,
               float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y, float* save_mean,
               float* save_invstd, void* workspace, size_t workspace_bytes, void* stream);
int se3_affine_act(const float* x, const float* center, const float* scale, const float* shift, int64_t rows, int32_t c,
                   int32_t act, float* y, void* stream);
int se3_bn_bwd(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma,
               int64_t rows, int32_t c, float* dx, float* dgamma, float* dbeta, void* workspace,
               size_t workspace_bytes, void* stream);
int se3_skip_fwd(const float* x, const float* y, const float* gamma, const float* gate, float gate_keep,
                 const int32_t* row_batch, int64_t rows, int32_t c, float* out, void* stream);
int se3_skip_bwd(const float* g, const float* x, const float* gamma, const float* gate, float gate_keep,
                 const int32_t* row_batch, int64_t rows, int32_t c, float* dx, float* dgamma, void* workspace,
                 size_t workspace_bytes, void* stream);
int se3_bias_gelu_bwd(const float* g, const float* z, const float* bias, int64_t rows, int32_t c, float* dz,
                      float* dbias, void* workspace, size_t workspace_bytes, void* stream);
size_t se3_linear_wgrad_workspace_bytes(int64_t rows, int32_t n_out, int32_t n_in);
int se3_linear_wgrad(const float* grad_y, const float* x, int64_t rows, int32_t n_out, int32_t n_in, float* grad_w,
                     void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optional per-kernel timing for bench.py's roofline line (no reference counterpart: the reference
 * only prints wall-clock per batch, tasks/SemSeg/train_dfaust_rot.py:239-296).  When enabled, every
 * kernel launch of the fused operator is bracketed by hipEvents on its own launch stream; read
 * accumulates milliseconds and launch counts per stage tag ("edge_t_fwd", "gemm_out", "gemm_gradT",
 * "edge_param_grad", "gemm_gradW", "edge_t_transposed", "gemm_gradX", "edge_t_recompute", "prep"; the grid kNN adds
 * "knn_sort" / "knn_cells" / "knn_fallback").
 * Off by default.
 * ------------------------------------------------------------------------------------------- */
int se3_profile_enable(int on);
/* Introspection of process-wide state (2) above, for tests of the capture contract and as the diagnostic of a graph
 * that was captured without an eager step in front of it.  stats[5]: [0] caller streams that own an internal side
 * stream, [1] spare (stream, events) sets ready on the current device, [2] sets created so far in this process, [3]
 * backward passes that wanted to fork inside a capture and could not (no set prepared: that graph replays its two
 * branches back to back -- the same results; with the fork opt-in since round 5 that is also the default schedule), [4] sets the 16-owner cap handed back to the
 * spares.  Sets are only created by calls whose stream is NOT being captured. */
int se3_side_stream_stats(int32_t* stats);
/* Process-wide switch of state (2): se3conv_bwd runs its feature branch on the internal side stream for layers of at most
 * `rows` output rows (and more than 4096; every size from 2^40 on).  0 = never (the default since round 5), < 0 = back to the
 * environment (SE3_OVERLAP=1 / SE3_OVERLAP_ROWS=n, read once).  Measured on MI355X the fork costs 0.4 - 2.3 % of a step on
 * every workload (profiles/r05_no_fork_ab.txt).  Do NOT turn it on for calls made on a stream that is itself a forked
 * branch of a graph capture: a fork from a forked stream makes hipStreamEndCapture segfault on the HIP runtime PyTorch
 * 2.10+rocm7.0 ships (tools/probes/nested_fork_capture.py reproduces it with torch streams and events alone). */
int se3_set_overlap_rows(int64_t rows);
int se3_profile_reset(void);
int se3_profile_read(const char* tag, double* total_ms, int64_t* launches);
int se3_profile_tags(char* buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* SE3CONV_H_ */

/*
 * Appendix: environment switches (process-wide state (3) above).  Each is read once, at first use; none changes results
 * beyond rounding; each is exercised by tests/test_gpu_variants.py.  What every switch measured: profiles/README.md,
 * "Ledger of lost A/Bs".
 *   SE3_NO_T24            T and U as packed hi/lo words instead of 3-byte rows
 *   SE3_OVERLAP, SE3_OVERLAP_ROWS   two-stream backward pass (see se3_set_overlap_rows; opt-in since round 5)
 *   SE3_BWD_BRANCH_ORDER  backward kernels branch by branch instead of writers first
 *   SE3_NO_PAIR, SE3_FC1  single-wavefront edge kernel instead of the wave pair / one frame per wavefront
 *   SE3_EDGE_STREAM       chunk-stream forms of the edge kernel (round 6: 64-channel rows with two frames per item, 32-channel
 *                         rows with two frames per item): n > 0 = from n items up (default 4096, 1 = every size), 0 = never
 *   SE3_SHARED_GRIDS      (Python host side, se3conv3d_amd/ops.py) =0: every bounded ball query sorts its source cloud itself
 *                         instead of sharing the grid per (cloud, radius) -- se3_ball_query_bounded_shared
 *   SE3_PG_SINGLE, SE3_PG_PAIR (+ _WGS, _C32)   forms of the parameter-gradient kernel
 *   SE3_NN_SPLITS         split-K count of the dense products (default: cost model)
 *   SE3_NN_KG             =2: two k groups per workgroup in the dense products over 3-byte rows of under-filled levels
 *   SE3_T16_GT            SE3_PRECISION_BF16X3_T16 only: grad_T in the block format too
 *   SE3_TR_MERGE_SORT     se3_csr_transpose*: the merge-sort form for every graph, same result
 *   SE3_DX_PATH           feature gradient edge-major: 1 wherever implemented, 0 never; default: the two-term cost model of
 *                         DESIGN.md section 4.11 (microseconds of either form; never above 20 edges per source row)
 * (Removed in round 6 with their code: SE3_SLICE_MB / SE3_SLICE_STREAMS, the row-sliced schedule of round 5 that lost every
 *  A/B -- profiles/r05_slice_ab.txt; SE3_PAIR_PERSIST / SE3_PAIR_OCC.)
 */
