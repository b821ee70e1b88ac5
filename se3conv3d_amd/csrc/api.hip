// extern "C" entry points of the fused operator + the API-parity ops (see include/se3conv.h).
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace se3 {

namespace {

// dst[c][r] = src[r][c]   (src [rows, cols])
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i / rows), r = (int)(i % rows);
    dst[i] = src[(int64_t)r * cols + c];
  }
}

// w2[(o*K + k), i] = w[i, k, o]
__global__ void permute_weights_oki_kernel(const float* __restrict__ w, float* __restrict__ w2, int c_in, int kb,
                                           int c_out) {
  const int64_t total = (int64_t)c_in * kb * c_out;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx % c_in);
    const int ok = (int)(idx / c_in);
    const int k = ok % kb, o = ok / kb;
    w2[idx] = w[((int64_t)i * kb + k) * c_out + o];
  }
}

// one block per output element: 256 threads stride over the per-block partials, then tree-reduce
__global__ __launch_bounds__(256) void reduce_param_partials_kernel(const float* __restrict__ partials, int n_partials,
                                                                    float* __restrict__ grad_axes,
                                                                    float* __restrict__ grad_biases, float scale) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  float s = 0.f;
  for (int p = threadIdx.x; p < n_partials; p += 256) s += partials[(int64_t)p * kDescExt * kBasis + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (i < SE3_DESC_DIMS * kBasis) {
      if (grad_axes) grad_axes[i] = red[0] * scale;
    } else if (grad_biases) {
      grad_biases[i - SE3_DESC_DIMS * kBasis] = red[0] * scale;
    }
  }
}

// ---- API-parity kernels (not on the fused path) ------------------------------------------------

// Rotation matrix (row-major m[9]) -> real-part-first quaternion as RotationFunctions.py:91-151 computes it: the four
// candidates q * 2 q_r, q * 2 q_i, ..., the one with the largest |component| is divided out (denominator floored at 0.1).
__device__ __forceinline__ void matrix_to_quaternion(const float m[9], float q[4]) {
  const float t[4] = {1.0f + m[0] + m[4] + m[8], 1.0f + m[0] - m[4] - m[8], 1.0f - m[0] + m[4] - m[8],
                      1.0f - m[0] - m[4] + m[8]};
  float qa[4];
  int best = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    qa[i] = t[i] > 0.f ? sqrtf(t[i]) : 0.f;
    if (qa[i] > qa[best]) best = i;  // first maximum, as torch.argmax
  }
  const float c[4][4] = {{qa[0] * qa[0], m[7] - m[5], m[2] - m[6], m[3] - m[1]},
                         {m[7] - m[5], qa[1] * qa[1], m[3] + m[1], m[2] + m[6]},
                         {m[2] - m[6], m[3] + m[1], qa[2] * qa[2], m[5] + m[7]},
                         {m[3] - m[1], m[6] + m[2], m[7] + m[5], qa[3] * qa[3]}};
  const float den = 2.0f * fmaxf(qa[best], 0.1f);
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = (best == 0 ? c[0][i] : best == 1 ? c[1][i] : best == 2 ? c[2][i] : c[3][i]) / den;
}

// get_rot_tenors materialised (PNEConvLayerRotEquiv.py:62-128): one thread per (edge, a, b).  REL = representation of
// the relative rotation R_out^T R_in (get_relative_rot, RotationFunctions.py:549-600): 0 "6D" (its first two rows, D = 9),
// 1 "matrix" (all nine entries, D = 12), 2 "quaternion" (D = 7).
template <int REL>
__global__ void rot_tensors_kernel(const float* __restrict__ pts_in, const float* __restrict__ pts_out,
                                   const float* __restrict__ frames_in, const float* __restrict__ frames_out,
                                   const int32_t* __restrict__ neighbors, const int32_t* __restrict__ ends,
                                   const float* __restrict__ rho_p, int64_t n_edges, int f_in, int f_out,
                                   float* __restrict__ desc, int32_t* __restrict__ fe_neighbors) {
  constexpr int D = REL == 0 ? 9 : (REL == 1 ? 12 : 7);
  const float rho = *rho_p;
  const int ff = f_in * f_out;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_edges * ff;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = t / ff;
    const int ab = (int)(t - e * ff);
    const int a = ab / f_in, b = ab % f_in;
    const int s = neighbors[e * 2], p = neighbors[e * 2 + 1];
    const int start = s > 0 ? ends[s - 1] : 0;
    const int deg = ends[s] - start;
    const int64_t idx = (int64_t)ff * start + (int64_t)a * deg * f_in + (e - start) * f_in + b;
    float x[3], y[3], ri[9], ro[9], d[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) x[i] = pts_in[(int64_t)p * 3 + i], y[i] = pts_out[(int64_t)s * 3 + i];
#pragma unroll
    for (int i = 0; i < 9; ++i)
      ri[i] = frames_in[((int64_t)p * f_in + b) * 9 + i], ro[i] = frames_out[((int64_t)s * f_out + a) * 9 + i];
    edge_descriptor(x, ri, y, ro, rho, d);
    if constexpr (REL == 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) desc[idx * 9 + i] = d[i];
    } else {
      float rel[9];  // d[3..8] are rows 0, 1 of R_out^T R_in; row 2 the same way
#pragma unroll
      for (int i = 0; i < 6; ++i) rel[i] = d[3 + i];
#pragma unroll
      for (int c = 0; c < 3; ++c) rel[6 + c] = ro[2] * ri[c] + ro[5] * ri[3 + c] + ro[8] * ri[6 + c];
#pragma unroll
      for (int i = 0; i < 3; ++i) desc[idx * D + i] = d[i];
      if constexpr (REL == 1) {
#pragma unroll
        for (int i = 0; i < 9; ++i) desc[idx * D + 3 + i] = rel[i];
      } else {
        float q[4];
        matrix_to_quaternion(rel, q);
#pragma unroll
        for (int i = 0; i < 4; ++i) desc[idx * D + 3 + i] = q[i];
      }
    }
    fe_neighbors[idx * 2] = s * f_out + a;
    fe_neighbors[idx * 2 + 1] = p * f_in + b;
  }
}

__global__ void rot_tensor_ends_kernel(const int32_t* __restrict__ ends, int64_t n_out, int f_in, int f_out,
                                       int32_t* __restrict__ fe_ends) {
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < n_out * f_out;
       m += (int64_t)gridDim.x * blockDim.x) {
    const int64_t s = m / f_out;
    const int a = (int)(m - s * f_out);
    const int start = s > 0 ? ends[s - 1] : 0;
    const int deg = ends[s] - start;
    fe_ends[m] = f_in * f_out * start + (a + 1) * deg * f_in;
  }
}

// T[m,c,k] = sum_e basis[e,k] feat[src(e),c]  (feat_basis_proj.cu:24-123): one block per row,
// thread = (channel group, k); generic in C and K.
__global__ __launch_bounds__(256) void feat_basis_proj_kernel(const float* __restrict__ basis,
                                                              const float* __restrict__ feat,
                                                              const int32_t* __restrict__ neighbors,
                                                              const int32_t* __restrict__ ends, int channels, int kb,
                                                              float* __restrict__ out) {
  const int64_t m = blockIdx.x;
  const int start = m > 0 ? ends[m - 1] : 0, end = ends[m];
  const int k = threadIdx.x % kb, cg = threadIdx.x / kb, ncg = blockDim.x / kb;
  for (int c = cg; c < channels; c += ncg) {
    float acc = 0.f;
    for (int e = start; e < end; ++e)
      acc += basis[(int64_t)e * kb + k] * feat[(int64_t)neighbors[(int64_t)e * 2 + 1] * channels + c];
    out[(m * channels + c) * kb + k] = acc;
  }
}

__device__ __forceinline__ int row_of_edge(const int32_t* __restrict__ ends, int64_t n_rows, int64_t e) {
  int64_t lo = 0, hi = n_rows;  // first row with ends[row] > e
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (ends[mid] <= e) lo = mid + 1; else hi = mid;
  }
  return (int)lo;
}

// gBasis[e,k] = sum_c gT[m,c,k] feat[p,c]   (feat_basis_proj_grads.cu:113-126)
__global__ void feat_basis_proj_grad_basis_kernel(const float* __restrict__ feat, const int32_t* __restrict__ neighbors,
                                                  const int32_t* __restrict__ ends, const float* __restrict__ grad_t,
                                                  int64_t n_edges, int64_t n_rows, int channels, int kb,
                                                  float* __restrict__ g_basis) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_edges * kb;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = t / kb;
    const int k = (int)(t - e * kb);
    const int64_t m = row_of_edge(ends, n_rows, e);
    const float* f = feat + (int64_t)neighbors[e * 2 + 1] * channels;
    const float* g = grad_t + m * channels * kb + k;
    float acc = 0.f;
    for (int c = 0; c < channels; ++c) acc += g[(int64_t)c * kb] * f[c];
    g_basis[t] = acc;
  }
}

// gFeat[p,c] += sum_k gT[m,c,k] basis[e,k]   (feat_basis_proj_grads.cu:129-140; float atomics as there)
__global__ void feat_basis_proj_grad_feat_kernel(const float* __restrict__ basis, const int32_t* __restrict__ neighbors,
                                                 const int32_t* __restrict__ ends, const float* __restrict__ grad_t,
                                                 int64_t n_edges, int64_t n_rows, int channels, int kb,
                                                 float* __restrict__ g_feat) {
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_edges * channels;
       t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = t / channels;
    const int c = (int)(t - e * channels);
    const int64_t m = row_of_edge(ends, n_rows, e);
    const float* g = grad_t + (m * channels + c) * kb;
    const float* b = basis + e * kb;
    float acc = 0.f;
    for (int k = 0; k < kb; ++k) acc += g[k] * b[k];
    atomicAdd(&g_feat[(int64_t)neighbors[e * 2 + 1] * channels + c], acc);
  }
}

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > 1 << 20) b = 1 << 20;
  return (unsigned)b;
}

bool shape_ok(const se3conv_shape* s) {
  return s && s->n_in >= 0 && s->n_out >= 0 && s->n_edges >= 0 && s->f_in >= 1 && s->f_out >= 1 && s->c_in >= 1 &&
         s->c_out >= 1 && s->num_basis >= 1 &&
         (s->precision == SE3_PRECISION_FP32 || s->precision == SE3_PRECISION_BF16X3 ||
          s->precision == SE3_PRECISION_BF16X3_T16);
}
int shape_supported(const se3conv_shape* s) {
  if (s->num_basis != kBasis) return SE3_ERR_UNSUPPORTED;  // every shipped config uses K = 32
  if (s->n_in * s->f_in >= (1ll << 31) || s->n_out * s->f_out >= (1ll << 31) ||
      s->n_edges * s->f_in * s->f_out >= (1ll << 31))
    return SE3_ERR_UNSUPPORTED;  // row / frame-edge ids are int32 inside the kernels
  if (s->n_in * s->f_in * s->c_in >= (1ll << 29) || s->n_out * s->f_out * s->c_out >= (1ll << 29) ||
      s->n_in * s->f_in >= (1ll << 25) || s->n_out * s->f_out >= (1ll << 25))
    return SE3_ERR_UNSUPPORTED;  // gathered operands (< 2 GB) and the 64-byte geometry records are addressed with 32-bit byte offsets
  return SE3_OK;
}

struct FwdLayout { size_t axes_ext, t, featpk, bt_hi, bt_lo, split, geom_in, geom_out, total; };
FwdLayout fwd_layout(const se3conv_shape* s, int save_t) {
  FwdLayout l{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t kb = s->num_basis;
  l.axes_ext = take(kDescExt * kBasis * 4);
  l.t = save_t ? 0 : take((size_t)s->n_out * s->f_out * s->c_in * kb * 4);
  if (s->precision != SE3_PRECISION_FP32) {
    l.featpk = take((size_t)s->n_in * s->f_in * s->c_in * 4);
    const size_t plane = (size_t)s->c_out * align_up((size_t)s->c_in * kb, 32) * 2;
    l.bt_hi = take(plane);
    l.bt_lo = take(plane);
    l.split = take(gemm_nn_bf16_split_bytes((int64_t)s->n_out * s->f_out, s->c_out, s->c_in * (int)kb));
  } else {
    l.split = take(gemm_nn_split_bytes((int64_t)s->n_out * s->f_out, s->c_out, s->c_in * (int)kb));
  }
  l.geom_in = take((size_t)s->n_in * s->f_in * 64);
  l.geom_out = take((size_t)s->n_out * s->f_out * 64);
  l.total = off;
  return l;
}

struct BwdLayout {
  size_t axes_ext, wt, w2, big, t, param_partials, tn_partials, featpk, gpk, bt_hi, bt_lo, split, geom_in, geom_out, total;
  size_t dx_rows;  // edge-major feature gradient (use_edge_dx): D [edge rows * F_in, C_in] fp32
  size_t big_u, bt2_hi, bt2_lo, split2;  // feature-gradient branch when it runs beside the parameter branch
  int n_param_partials, tn_splits;
};
// The feature gradient of a convolution with many more input rows than edges per row can carry (a down-convolution)
// goes edge-major (edge_dx.hip): D = phi gT^T per frame-edge, summed per source row -- instead of a U row per source
// row and its GEMM.  Decided from the shape alone (bwd_layout and se3conv_bwd must agree): implemented shapes only, split-
// bf16 arithmetic, and the bytes the two forms move through memory -- U written and read (3-byte rows) against D written
// and gathered, plus grad_T when the parameter gradients do not need it anyway -- with a factor of two in favour of the
// default.  SE3_DX_PATH=0 never, =1 whenever implemented (tests force both on the same shapes).
bool use_edge_dx(const se3conv_shape* s, bool want_feat, bool want_params) {
  static const int mode = [] {
    const char* e = getenv("SE3_DX_PATH");
    return e ? atoi(e) : -1;
  }();
  if (!want_feat || mode == 0 || s->precision == SE3_PRECISION_FP32 || s->num_basis != kBasis) return false;
  if (s->n_in == 0 || s->n_out == 0) return false;
  EdgeGeom g{};
  g.f_ctr = s->f_out, g.f_nb = s->f_in;
  if (!edge_dx_bf16_applicable(g, s->c_in)) return false;
  if (s->n_edges * s->f_in * (int64_t)s->c_in >= (1ll << 31)) return false;
  if (mode == 1) return true;
  const double rows_in = (double)s->n_in * s->f_in, rows_out = (double)s->n_out * s->f_out;
  const double u_bytes = rows_in * s->c_out * kBasis * 6.0;
  const double d_bytes = (double)s->n_edges * s->f_in * s->c_in * 8.0 + (want_params ? 0.0 : rows_out * s->c_in * kBasis * 8.0);
  // Cost model in microseconds, fitted in round 5 to the stage times of the reference network's own convolution calls
  // (profiles/r05_faust_network_convs.txt: the default against SE3_DX_PATH=1) and of the bench levels: the U form pays a
  // latency floor for its two launches on levels too small to fill the chip (~45 us: the grad_X GEMM of a 3.7 k-row level
  // takes as long as that of an 18 k-row one) plus its bytes at the rate of a just-written tensor; the edge-major form
  // ~15 us plus its bytes at the rate of its row gathers -- and loses whatever the bytes say where a source owns more than
  // ~20 edges (its per-source sum walks the segment: headline level 2, 27 edges per point, 0.108 -> 0.115 ms; a lateral
  // convolution with 357 edges per source: dx_gather 330 us).  Round 4's rule was 2 d_bytes < u_bytes.
  if ((double)s->n_edges > 20.0 * (double)s->n_in) return false;
  return 15.0 + d_bytes / 2.2e6 < 45.0 + u_bytes / 6.0e6;
}

// The weight gradient from U instead of T (round 5): dW[i,k,o] = alpha sum_p f[p,i] U[p,o,k] -- U is the transposed pass's
// tensor, which backward produces anyway for the feature gradient.  Available in the split-bf16 modes whenever the U form
// of the feature gradient runs; then the forward pass need not keep T (0.8 GB per layer at the headline shape: the largest
// saved activation by a factor of 24), and for an up-convolution the product walks the few rows of the coarse level
// instead of the many of the fine one.  Used when T was not saved, or when U has fewer (row x channel) entries than T.
// From the shape alone: se3conv_bwd_workspace_bytes, se3conv_bwd and se3conv_bwd_needs_t must agree.
bool dw_from_u_available(const se3conv_shape* s, bool want_feat, bool want_params) {
  if (!want_feat || !want_params || s->precision == SE3_PRECISION_FP32 || s->num_basis != kBasis) return false;
  if (s->n_in == 0 || s->n_out == 0 || s->c_in % 4 != 0 || s->c_out % 2 != 0) return false;
  static const bool branch_order = getenv("SE3_BWD_BRANCH_ORDER") != nullptr;  // that order overwrites U before the product would read it
  return !branch_order && !use_edge_dx(s, true, true);
}
bool use_u_for_dw(const se3conv_shape* s, bool want_feat, bool want_params, bool have_t) {
  if (!dw_from_u_available(s, want_feat, want_params)) return false;
  return !have_t || (double)s->n_in * s->f_in * s->c_out < (double)s->n_out * s->f_out * s->c_in;
}

BwdLayout bwd_layout(const se3conv_shape* s, int want_feat, int want_params, int have_t) {
  BwdLayout l{};
  const bool dx = use_edge_dx(s, want_feat != 0, want_params != 0);
  const bool dw_u = use_u_for_dw(s, want_feat != 0, want_params != 0, have_t != 0);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t kb = s->num_basis;
  const size_t rows_out = (size_t)s->n_out * s->f_out, rows_in = (size_t)s->n_in * s->f_in;
  const size_t wsz = (size_t)s->c_in * kb * s->c_out * 4;
  const bool fast = s->precision != SE3_PRECISION_FP32;
  l.axes_ext = take(kDescExt * kBasis * 4);
  if (!fast) {
    l.wt = want_params ? take(wsz) : 0;
    l.w2 = want_feat ? take(wsz) : 0;
    l.split = want_feat ? take(gemm_nn_split_bytes((int64_t)rows_in, s->c_in, s->c_out * (int)kb)) : 0;
  } else {
    // the largest of the three pre-split weight layouts (see prep_weights_kernel)
    size_t plane = (size_t)s->c_in * kb * align_up((size_t)s->c_out, 32) * 2;
    const size_t p2 = (size_t)s->c_in * align_up((size_t)s->c_out * kb, 32) * 2;
    if (p2 > plane) plane = p2;
    const size_t p3 = (size_t)s->c_out * kb * align_up((size_t)s->c_in, 32) * 2;
    if (p3 > plane) plane = p3;
    l.bt_hi = take(plane);
    l.bt_lo = take(plane);
    l.featpk = want_params ? take(rows_in * s->c_in * 4) : 0;
    l.gpk = take(rows_out * s->c_out * 4);
    size_t sp = (want_params || dx) ? gemm_nn_bf16_split_bytes((int64_t)rows_out, s->c_in * (int)kb, s->c_out) : 0;
    const size_t sp3 = want_params ? gemm_nn_bf16_split_bytes((int64_t)rows_in, s->c_out * (int)kb, s->c_in) : 0;
    if (sp3 > sp) sp = sp3;
    const size_t sp2 = (want_feat && !dx) ? gemm_nn_bf16_split_bytes((int64_t)rows_in, s->c_in, s->c_out * (int)kb) : 0;
    if (sp2 > sp) sp = sp2;
    l.split = take(sp);
  }
  l.geom_in = take(rows_in * 64);
  l.geom_out = take(rows_out * 64);
  // (the edge-major feature gradient has no U: none of the U-sized terms is reserved on that path -- at a down-convolution
  // of the headline hierarchy they were 2 GB of workspace nobody touched)
  size_t big = 0;
  if (want_params || dx) big = rows_out * s->c_in * kb * 4;
  if (want_feat && !dx && rows_in * s->c_out * kb * 4 > big) big = rows_in * s->c_out * kb * 4;
  l.big = take(big);
  if (fast && want_feat && !dx) {
    const size_t p2 = (size_t)s->c_in * align_up((size_t)s->c_out * kb, 32) * 2;
    l.bt2_hi = take(p2);
    l.bt2_lo = take(p2);
  }
  if (fast && want_feat && want_params && !dx) {
    l.big_u = take(rows_in * s->c_out * kb * 4);
    l.split2 = take(gemm_nn_bf16_split_bytes((int64_t)rows_in, s->c_in, s->c_out * (int)kb));
  }
  l.dx_rows = dx ? take((size_t)s->n_edges * s->f_in * s->c_in * 4) : 0;
  l.t = (want_params && !have_t && !dw_u) ? take(rows_out * s->c_in * kb * 4) : 0;
  l.n_param_partials = edge_param_grad_blocks((int64_t)rows_out);
  l.param_partials = want_params ? take((size_t)l.n_param_partials * edge_param_grad_bf16_channel_blocks(s->c_in) *
                                         kDescExt * kBasis * 4) : 0;
  l.tn_splits = dw_u ? gemm_tn_splits((int64_t)rows_in, s->c_out * (int)kb, s->c_in)
                     : gemm_tn_splits((int64_t)rows_out, s->c_in * (int)kb, s->c_out);
  l.tn_partials = want_params ? take((size_t)l.tn_splits * wsz) : 0;
  l.total = off;
  return l;
}

// Second stream for the backward pass: the parameter branch (grad_T GEMM -> edge_param_grad, weight-gradient GEMM)
// and the feature branch (transposed edge kernel -> grad_X GEMM) are independent.  OFF by default since round 5: measured
// on MI355X with the kernels as they are now (writers first, non-temporal producer stores), the fork loses at every size --
// headline stack 2.437 -> 2.405 ms without it (its 18 k-row level 0.341 -> 0.330), dfaust_f2 1.918 -> 1.873, dfaust_f4
// 6.97 -> 6.93, scannet150k_f1 2.572 -> 2.561 (profiles/r05_no_fork_ab.txt; at 131 k rows it always lost: 2.15 vs 2.07 ms
// in round 1).  It also keeps the library out of a hazard of this HIP runtime: a fork FROM A FORKED STREAM inside a graph
// capture segfaults in hipStreamEndCapture (tools/probes/nested_fork_capture.py: torch streams and events alone do it), which
// is what a caller who captures the library on a side stream of its own would have triggered.  SE3_OVERLAP=1 (every size) /
// SE3_OVERLAP_ROWS=n / se3_set_overlap_rows(n) turn it on for levels of MORE than kOverlapMinRows and at most n output rows;
// smaller levels fork only when the limit is 2^40 or more (what SE3_OVERLAP=1 sets: "every size") -- below ~4 k rows the fork
// and join cost more than the branches overlap.
constexpr int kOverlapRows = 0, kOverlapMinRows = 4096;
std::atomic<int64_t> g_overlap_rows{-1};  // se3_set_overlap_rows: >= 0 overrides the environment
int64_t overlap_rows_limit() {
  const int64_t v = g_overlap_rows.load(std::memory_order_relaxed);
  if (v >= 0) return v;
  static const int64_t env = [] {
    if (getenv("SE3_OVERLAP") != nullptr) return (int64_t)1 << 62;
    const char* e = getenv("SE3_OVERLAP_ROWS");
    return e ? (int64_t)atoll(e) : (int64_t)kOverlapRows;
  }();
  return env;
}
// One side stream + fork / join event pair per (device, caller stream), kept for the life of the process: two backward
// calls on two caller streams (or threads) never share events or a stream, and a caller stream on device 1 never gets a
// side stream of device 0.  Two calls racing on the SAME caller stream are the caller's race anyway.  The table only
// grows by the number of distinct streams the caller uses.
// Nothing is CREATED while the caller's stream is being captured into a HIP graph (creating runtime objects in the middle
// of a capture is what a capture should not have to survive): every eager call of se3conv_fwd / se3conv_bwd keeps a few
// spare (stream, events) sets per device ready, a capturing caller stream that is new to the table takes one of those,
// and when there is none -- the library was never called outside a capture in this process -- the backward pass simply
// does not fork (same results, the two branches back to back).  INTEGRATION.md: warm up eagerly before capturing.
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  bool ok = false;
  uint64_t last_use = 0;  // SideTable::clock at the last fork (LRU order)
  int pins = 0;           // calls between side_stream_for and release_side_stream on this set: never evicted while > 0
};
constexpr size_t kSpareSideStreams = 2;
// At most this many caller streams own a side stream at a time.  A server that makes a stream per request would grow the
// table without limit otherwise: beyond the cap the least recently used set goes back to the spares (the runtime objects
// live on and are handed to the next new caller stream).  Never inside a capture -- neither when the caller's stream is
// being captured (nothing is rearranged then) nor a set whose own stream is part of a capture in progress.
constexpr size_t kMaxOwnedSideStreams = 16;
struct SideTable {
  std::mutex mu;
  std::map<std::pair<int, hipStream_t>, SideStream> by_stream;
  std::map<int, std::vector<SideStream>> spare;
  uint64_t clock = 0;
  int created = 0;             // (stream, events) sets made so far in this process
  int unforked_in_capture = 0; // captured backward passes that wanted to fork and had no set to fork onto
  int evicted = 0;             // sets taken back from a caller stream by the cap
};
SideTable& side_table() {
  static SideTable t;
  return t;
}
SideStream make_side_stream(SideTable& t) {  // t.mu held
  SideStream v;
  ++t.created;
  v.ok = hipStreamCreateWithFlags(&v.stream, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&v.fork, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&v.join, hipEventDisableTiming) == hipSuccess;
  return v;
}
bool stream_is_capturing(hipStream_t s) {
  if (s == nullptr) return false;  // the legacy default stream cannot be captured (and must not be queried during a capture)
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) != hipSuccess) {
    (void)hipGetLastError();
    return true;  // cannot tell: behave as inside a capture (create nothing)
  }
  return st != hipStreamCaptureStatusNone;
}
// called by every eager se3conv_fwd / se3conv_bwd: the spares a later capture may need
void keep_side_streams_ready(hipStream_t caller) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  SideTable& t = side_table();
  {
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.spare.find(dev);
    if (it != t.spare.end() && it->second.size() >= kSpareSideStreams) return;
  }
  if (stream_is_capturing(caller)) return;
  std::lock_guard<std::mutex> lk(t.mu);
  auto& sp = t.spare[dev];
  while (sp.size() < kSpareSideStreams) {
    SideStream v = make_side_stream(t);
    if (!v.ok) break;
    sp.push_back(v);
  }
}
// t.mu held, the caller's stream is not being captured: hand the least recently used sets of this device back to the spares
// until the device owns at most kMaxOwnedSideStreams (sets that are part of a capture in progress stay where they are)
void evict_side_streams(SideTable& t, int dev) {
  for (;;) {
    size_t owned = 0;
    auto victim = t.by_stream.end();
    for (auto it = t.by_stream.begin(); it != t.by_stream.end(); ++it) {
      if (it->first.first != dev || !it->second.ok) continue;
      ++owned;
      if (it->second.pins > 0) continue;  // between fork and join of another thread's call: not a candidate
      if (victim == t.by_stream.end() || it->second.last_use < victim->second.last_use) victim = it;
    }
    if (owned <= kMaxOwnedSideStreams || victim == t.by_stream.end()) return;
    if (stream_is_capturing(victim->second.stream)) {  // forked into a capture that has not ended: not now
      victim->second.last_use = ++t.clock;
      bool any_idle = false;
      for (auto& kv : t.by_stream)
        if (kv.first.first == dev && kv.second.ok && kv.second.pins == 0 && !stream_is_capturing(kv.second.stream)) any_idle = true;
      if (!any_idle) return;
      continue;
    }
    t.spare[dev].push_back(victim->second);
    t.by_stream.erase(victim);
    ++t.evicted;
  }
}
// Returns a COPY of the set (the table may hand the entry to another caller later; the runtime objects are never
// destroyed), ok = false when there is nothing to fork onto.  A set that is returned ok is PINNED: the cap cannot hand
// it to another caller stream until release_side_stream (ForkJoin's join / destructor) -- with more caller streams in
// flight than the cap, two calls would otherwise record and wait on the same events.
SideStream side_stream_for(hipStream_t caller) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SideStream{};
  SideTable& t = side_table();
  {
    std::lock_guard<std::mutex> lk(t.mu);
    auto it = t.by_stream.find({dev, caller});
    if (it != t.by_stream.end()) {
      it->second.last_use = ++t.clock;
      if (it->second.ok) ++it->second.pins;
      return it->second;
    }
  }
  const bool capturing = stream_is_capturing(caller);
  std::lock_guard<std::mutex> lk(t.mu);
  auto known = t.by_stream.find({dev, caller});  // another thread of this caller stream got here first
  if (known != t.by_stream.end()) {
    if (known->second.ok) ++known->second.pins;
    return known->second;
  }
  SideStream v;
  auto& sp = t.spare[dev];
  if (capturing) {
    if (sp.empty()) {  // nothing prepared outside the capture: no fork (run one eager step first)
      ++t.unforked_in_capture;
      return SideStream{};
    }
    v = sp.back();
    sp.pop_back();
  } else if (sp.size() > kSpareSideStreams) {  // sets the cap took back come first
    v = sp.back();
    sp.pop_back();
  } else {
    v = make_side_stream(t);
  }
  v.last_use = ++t.clock;
  v.pins = v.ok ? 1 : 0;
  t.by_stream.emplace(std::make_pair(dev, caller), v);
  if (!capturing) evict_side_streams(t, dev);
  return v;
}
// the call that pinned the set (side_stream_for returned ok) is done with its events
void release_side_stream(hipStream_t caller) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return;
  SideTable& t = side_table();
  std::lock_guard<std::mutex> lk(t.mu);
  auto it = t.by_stream.find({dev, caller});
  if (it != t.by_stream.end() && it->second.pins > 0) --it->second.pins;
}
// Joins the forked side stream back into the caller's stream on EVERY exit path of se3conv_bwd once the fork has
// happened (an early `return rc` would otherwise leave the side stream writing into buffers the caller is about to
// free, and a stream capture with an unjoined fork).
struct ForkJoin {
  SideStream side;
  hipStream_t main = nullptr;
  bool forked = false, pinned = false;
  // takes over the pin of a set side_stream_for returned ok (released on join / destruction, forked or not)
  void adopt(const SideStream& s, hipStream_t m) { side = s, main = m, pinned = s.ok; }
  int fork(const SideStream& s, hipStream_t m) {
    if (!pinned) adopt(s, m);
    if (hipEventRecord(s.fork, m) != hipSuccess || hipStreamWaitEvent(s.stream, s.fork, 0) != hipSuccess)
      return SE3_ERR_LAUNCH;
    forked = true;
    return SE3_OK;
  }
  int join() {
    int rc = SE3_OK;
    if (forked) {
      forked = false;
      if (hipEventRecord(side.join, side.stream) != hipSuccess || hipStreamWaitEvent(main, side.join, 0) != hipSuccess)
        rc = SE3_ERR_LAUNCH;
    }
    if (pinned) {
      pinned = false;
      release_side_stream(main);
    }
    return rc;
  }
  ~ForkJoin() { (void)join(); }
};

// ---- any number of basis functions on the K = 32 kernels (se3conv_fwd / se3conv_bwd with num_basis != 32) ---------------
// The sum over k is separable: K basis functions are ceil(K / 32) slices of 32, the last one padded with basis functions
// whose projection axes, bias and conv weights are zero (GELU(0) = 0 meets W = 0: exact).  Per slice the padded parameter
// copies are built in the workspace, the K = 32 operator runs on them, outputs are accumulated and the parameter gradients
// copied back into their slice of the caller's tensors.  T is not kept between the calls for K != 32 (`t_save` ignored).
__global__ void slice_params_kernel(const float* __restrict__ axes, const float* __restrict__ biases,
                                    const float* __restrict__ w, int kb, int k0, int kn, int c_in, int c_out,
                                    float* __restrict__ a32, float* __restrict__ b32, float* __restrict__ w32) {
  const int64_t n_w = (int64_t)c_in * kBasis * c_out;
  // the axes / bias tables (288 + 32 entries) ride in the same loop: it runs to whichever is longer (c_in * c_out < 9)
  const int64_t n_all = n_w > SE3_DESC_DIMS * kBasis ? n_w : SE3_DESC_DIMS * kBasis;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (int64_t)gridDim.x * blockDim.x) {
    const int o = (int)(i % c_out);
    const int k = (int)((i / c_out) % kBasis);
    const int ci = (int)(i / ((int64_t)c_out * kBasis));
    if (i < n_w) w32[i] = k < kn ? w[((int64_t)ci * kb + k0 + k) * c_out + o] : 0.f;
    if (i < SE3_DESC_DIMS * kBasis) {
      const int j = (int)(i / kBasis), kk = (int)(i % kBasis);
      a32[i] = kk < kn ? axes[j * kb + k0 + kk] : 0.f;
    }
    if (i < kBasis) b32[i] = i < kn ? biases[k0 + i] : 0.f;
  }
}

__global__ void unslice_grads_kernel(const float* __restrict__ da32, const float* __restrict__ db32,
                                     const float* __restrict__ dw32, int kb, int k0, int kn, int c_in, int c_out,
                                     float* __restrict__ grad_axes, float* __restrict__ grad_biases,
                                     float* __restrict__ grad_weights) {
  const int64_t n_w = (int64_t)c_in * kBasis * c_out;
  const int64_t n_all = n_w > SE3_DESC_DIMS * kBasis ? n_w : SE3_DESC_DIMS * kBasis;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_all; i += (int64_t)gridDim.x * blockDim.x) {
    const int o = (int)(i % c_out);
    const int k = (int)((i / c_out) % kBasis);
    const int ci = (int)(i / ((int64_t)c_out * kBasis));
    if (grad_weights && i < n_w && k < kn) grad_weights[((int64_t)ci * kb + k0 + k) * c_out + o] = dw32[i];
    if (grad_axes && i < SE3_DESC_DIMS * kBasis) {
      const int j = (int)(i / kBasis), kk = (int)(i % kBasis);
      if (kk < kn) grad_axes[j * kb + k0 + kk] = da32[i];
    }
    if (grad_biases && i < kn) grad_biases[k0 + i] = db32[i];
  }
}

__global__ void add_into_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

struct AnyBasisLayout { size_t a32, b32, w32, out_tmp, da32, db32, dw32, inner, total; int slices; };
AnyBasisLayout any_basis_layout(const se3conv_shape* s, size_t out_tmp_bytes, bool grads, size_t inner_bytes) {
  AnyBasisLayout l{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t wsz = (size_t)s->c_in * kBasis * s->c_out * 4;
  l.slices = (s->num_basis + kBasis - 1) / kBasis;
  l.a32 = take(SE3_DESC_DIMS * kBasis * 4);
  l.b32 = take(kBasis * 4);
  l.w32 = take(wsz);
  l.out_tmp = take(l.slices > 1 ? out_tmp_bytes : 0);
  if (grads) {
    l.da32 = take(SE3_DESC_DIMS * kBasis * 4);
    l.db32 = take(kBasis * 4);
    l.dw32 = take(wsz);
  }
  l.inner = take(inner_bytes);
  l.total = off;
  return l;
}
// work items of slice_params_kernel / unslice_grads_kernel: the weights or the 288-entry axes table, whichever is longer
int64_t slice_items(const se3conv_shape* s) {
  const int64_t n_w = (int64_t)s->c_in * kBasis * s->c_out;
  return n_w > SE3_DESC_DIMS * kBasis ? n_w : SE3_DESC_DIMS * kBasis;
}
se3conv_shape with_32_basis(const se3conv_shape* s) {
  se3conv_shape t = *s;
  t.num_basis = kBasis;
  return t;
}

// Where a call's geometry records live and which of them it has to build: the caller's buffers (se3conv_prepared) where it
// keeps them, the workspace otherwise; a cloud against itself has one set of records for both sides.
struct PreparedGeometry {
  float *in, *out;
  bool need_in, need_out;
};
PreparedGeometry prepared_geometry(const se3conv_prepared* prep, bool same_cloud, float* ws_in, float* ws_out) {
  PreparedGeometry g{ws_in, ws_out, true, true};
  if (prep && prep->geom_in) g.in = prep->geom_in, g.need_in = !prep->geom_in_valid;
  if (same_cloud) {
    g.out = g.in, g.need_out = false;
    return g;
  }
  if (prep && prep->geom_out) g.out = prep->geom_out, g.need_out = !prep->geom_out_valid;
  return g;
}

EdgeGeom forward_geom(const float* pts_in, const float* pts_out, const float* frames_in, const float* frames_out,
                      const int32_t* neighbors, const int32_t* ends, const se3conv_shape* s) {
  EdgeGeom g{};
  g.ctr_pts = pts_out, g.ctr_frames = frames_out, g.nb_pts = pts_in, g.nb_frames = frames_in;
  g.nbr = neighbors, g.nbr_stride = 2, g.nbr_offset = 1, g.ends = ends;
  g.n_ctr = s->n_out, g.f_ctr = s->f_out, g.f_nb = s->f_in, g.transposed = 0;
  g.n_nb = s->n_in;
  g.n_edges = s->n_edges;
  return g;
}

}  // namespace
}  // namespace se3

using namespace se3;

extern "C" int se3_abi_version(void) { return SE3_ABI_VERSION; }

extern "C" const char* se3_error_string(int code) {
  switch (code) {
    case SE3_OK: return "ok";
    case SE3_ERR_INVALID_ARGUMENT: return "invalid argument (null pointer, negative size or bad shape)";
    case SE3_ERR_UNSUPPORTED: return "unsupported shape (int32-sized clouds; API-parity ops: K in {8, 16, 32, 64})";
    case SE3_ERR_WORKSPACE: return "workspace too small";
    case SE3_ERR_LAUNCH: return "HIP launch/runtime error";
    default: return "unknown error";
  }
}

extern "C" int se3_rot_tensors_rel(const float* pts_in, const float* pts_out, const float* frames_in,
                                   const float* frames_out, const int32_t* neighbors, const int32_t* ends,
                                   const float* rho, const se3conv_shape* s, int32_t rel_rot, float* desc,
                                   int32_t* fe_neighbors, int32_t* fe_ends, void* stream_) {
  if (!shape_ok(s) || rel_rot < SE3_REL_ROT_6D || rel_rot > SE3_REL_ROT_QUATERNION) return SE3_ERR_INVALID_ARGUMENT;
  if (s->n_edges * s->f_in * s->f_out >= (1ll << 31)) return SE3_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  if (s->n_out > 0) {
    if (!ends || !fe_ends) return SE3_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(rot_tensor_ends_kernel, dim3(grid_for(s->n_out * s->f_out)), dim3(256), 0, stream, ends, s->n_out,
                       s->f_in, s->f_out, fe_ends);
  }
  if (s->n_edges > 0) {
    if (!pts_in || !pts_out || !frames_in || !frames_out || !neighbors || !ends || !rho || !desc || !fe_neighbors)
      return SE3_ERR_INVALID_ARGUMENT;
    const dim3 grid(grid_for(s->n_edges * s->f_in * s->f_out));
#define SE3_ROT(REL)                                                                                                  \
  hipLaunchKernelGGL(rot_tensors_kernel<REL>, grid, dim3(256), 0, stream, pts_in, pts_out, frames_in, frames_out, neighbors, \
                     ends, rho, s->n_edges, s->f_in, s->f_out, desc, fe_neighbors)
    if (rel_rot == SE3_REL_ROT_6D) SE3_ROT(0);
    else if (rel_rot == SE3_REL_ROT_MATRIX) SE3_ROT(1);
    else SE3_ROT(2);
#undef SE3_ROT
  }
  return check_launch();
}

extern "C" int se3_rot_tensors(const float* pts_in, const float* pts_out, const float* frames_in,
                               const float* frames_out, const int32_t* neighbors, const int32_t* ends,
                               const float* rho, const se3conv_shape* s, float* desc, int32_t* fe_neighbors,
                               int32_t* fe_ends, void* stream) {
  return se3_rot_tensors_rel(pts_in, pts_out, frames_in, frames_out, neighbors, ends, rho, s, SE3_REL_ROT_6D, desc,
                             fe_neighbors, fe_ends, stream);
}

static bool basis_count_ok(int kb) { return kb == 8 || kb == 16 || kb == 32 || kb == 64; }

extern "C" int se3_feat_basis_proj(const float* basis, const float* feat, const int32_t* neighbors, const int32_t* ends,
                                   int64_t n_edges, int64_t n_rows, int64_t n_feat, int32_t channels,
                                   int32_t num_basis, float* out, void* stream) {
  if (n_edges < 0 || n_rows < 0 || n_feat < 0 || channels < 1) return SE3_ERR_INVALID_ARGUMENT;
  if (!basis_count_ok(num_basis)) return SE3_ERR_UNSUPPORTED;  // feat_basis_utils.cuh:35-41
  if (n_rows == 0) return SE3_OK;
  if (!ends || !out || (n_edges > 0 && (!basis || !feat || !neighbors))) return SE3_ERR_INVALID_ARGUMENT;
  if (n_rows >= (1ll << 31)) return SE3_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(feat_basis_proj_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, basis, feat,
                     neighbors, ends, channels, num_basis, out);
  return check_launch();
}

extern "C" int se3_feat_basis_proj_grad(const float* basis, const float* feat, const int32_t* neighbors,
                                        const int32_t* ends, const float* grad_out, int64_t n_edges, int64_t n_rows,
                                        int64_t n_feat, int32_t channels, int32_t num_basis, float* g_feat,
                                        float* g_basis, void* stream_) {
  if (n_edges < 0 || n_rows < 0 || n_feat < 0 || channels < 1) return SE3_ERR_INVALID_ARGUMENT;
  if (!basis_count_ok(num_basis)) return SE3_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  if (n_feat > 0) {
    if (!g_feat) return SE3_ERR_INVALID_ARGUMENT;
    if (int rc = launch_fill_words(g_feat, 0u, n_feat * channels, stream)) return rc;
  }
  if (n_edges == 0) return SE3_OK;
  if (!basis || !feat || !neighbors || !ends || !grad_out || !g_basis) return SE3_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(feat_basis_proj_grad_basis_kernel, dim3(grid_for(n_edges * num_basis)), dim3(256), 0, stream, feat,
                     neighbors, ends, grad_out, n_edges, n_rows, channels, num_basis, g_basis);
  hipLaunchKernelGGL(feat_basis_proj_grad_feat_kernel, dim3(grid_for(n_edges * channels)), dim3(256), 0, stream, basis,
                     neighbors, ends, grad_out, n_edges, n_rows, channels, num_basis, g_feat);
  return check_launch();
}

// Format of the row-sized intermediates (T, and U of the feature gradient): 0 packed hi|lo words (4 bytes per element), 1
// the 3-byte rows of common.h when the edge kernel can produce them (edge_t_bf16_t24_rows) and the buffer-load GEMMs consume
// them (SE3_NO_T24=1: packed words everywhere), 2 the 2.25-byte block format T16 -- only in SE3_PRECISION_BF16X3_T16, where
// the wave-pair edge kernel produces it (rows of a multiple of 64 channels); other shapes of that mode fall back to 1 / 0.
// tn_cols = the column count of the TN product that also reads the rows (0: none).
static int row_format(const se3conv_shape* s, const EdgeGeom& g, int channels, int64_t rows, int tn_cols) {
  static const bool on = getenv("SE3_NO_T24") == nullptr;
  (void)rows;  // any row count: the GEMMs that read the rows walk them in blocks their 32-bit offsets reach (gemm_bf16.hip)
  if (s->precision == SE3_PRECISION_BF16X3_T16 && kBasis == 32 && edge_t_bf16_t16_rows(g, channels) && tn_cols % 4 == 0)
    return 2;
  return on && kBasis == 32 && channels % 2 == 0 && edge_t_bf16_t24_rows(g, channels) && tn_cols % 4 == 0 ? 1 : 0;
}

// grad_T in the T16 block format (SE3_PRECISION_BF16X3_T16): when the row-strip GEMM can write it (rows of whole mega tiles,
// c_out <= 64) and the pair form of the parameter-gradient kernel reads it (two frames per point, 64-channel blocks)
// OPT-IN (SE3_T16_GT=1): measured a net loss at the headline shape -- the strip GEMM gains 0.015 ms on 44 % fewer bytes
// (its epilogue, not its stores, now sets its time) and the parameter-gradient kernel loses 0.04 ms to the decode in
// front of every item's first chunk (profiles/r04_t16_ab.txt); T and U alone are the mode's default.
static bool grad_t_t16(const se3conv_shape* s, const EdgeGeom& g) {
  static const bool on = [] {
    const char* e = getenv("SE3_T16_GT");
    return e != nullptr && atoi(e) != 0;
  }();
  const int64_t rows_out = s->n_out * s->f_out;
  return on && s->precision == SE3_PRECISION_BF16X3_T16 && s->num_basis == kBasis &&
         gemm_strip_t16_applicable(rows_out, s->c_in * kBasis, s->c_out) && edge_param_grad_bf16_t16_rows(g, s->c_in);
}

// Bytes per element of the row-sized intermediates this shape would move (what a traffic model has to assume):
// which = 0: T (forward, read again by the weight gradient), 1: U (feature gradient), 2: grad_T.  < 0: bad shape.
// The T16 format's 2.25 bytes are reported as 2 here (an integer interface); se3conv_intermediate_row_bytes is exact.
static int64_t intermediate_row_bytes(const se3conv_shape* s, int which) {
  const int64_t ck = (int64_t)(which == 1 ? s->c_out : s->c_in) * s->num_basis;
  if (s->precision == SE3_PRECISION_FP32 || s->num_basis != kBasis) return ck * 4;
  EdgeGeom g = forward_geom(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, s);
  if (which == 2) return grad_t_t16(s, g) ? t16_row_bytes(s->c_in) : ck * 4;
  EdgeGeom gt{};
  gt.n_ctr = s->n_in, gt.f_ctr = s->f_in, gt.f_nb = s->f_out, gt.n_nb = s->n_out, gt.transposed = 1;
  const int fmt = which == 0 ? row_format(s, g, s->c_in, s->n_out * s->f_out, s->c_out)
                             : row_format(s, gt, s->c_out, s->n_in * s->f_in, 0);
  return fmt == 2 ? t16_row_bytes(which == 0 ? s->c_in : s->c_out) : ck * (fmt == 1 ? 3 : 4);
}
extern "C" int64_t se3conv_intermediate_row_bytes(const se3conv_shape* s, int which) {
  if (!shape_ok(s) || which < 0 || which > 2) return SE3_ERR_INVALID_ARGUMENT;
  return intermediate_row_bytes(s, which);
}
extern "C" int se3conv_intermediate_bytes_per_element(const se3conv_shape* s, int which) {
  if (!shape_ok(s) || which < 0 || which > 2) return SE3_ERR_INVALID_ARGUMENT;
  const int64_t ck = (int64_t)(which == 1 ? s->c_out : s->c_in) * s->num_basis;
  return (int)(intermediate_row_bytes(s, which) / ck);
}

extern "C" size_t se3conv_fwd_workspace_bytes(const se3conv_shape* s, int save_t) {
  if (!shape_ok(s)) return 0;
  if (s->num_basis != kBasis) {
    const se3conv_shape s32 = with_32_basis(s);
    return any_basis_layout(s, (size_t)s->n_out * s->f_out * s->c_out * 4, false, fwd_layout(&s32, 0).total).total;
  }
  return fwd_layout(s, save_t).total;
}

extern "C" int se3conv_fwd(const float* pts_in, const float* pts_out, const float* frames_in, const float* frames_out,
                           const int32_t* neighbors, const int32_t* ends, const float* feat, const float* proj_axes,
                           const float* proj_biases, const float* conv_weights, const float* rho, const float* nu,
                           const se3conv_shape* s, float* out, float* t_save, void* workspace, size_t workspace_bytes,
                           void* stream_) {
  return se3conv_fwd_prepared(pts_in, pts_out, frames_in, frames_out, neighbors, ends, feat, proj_axes, proj_biases, conv_weights,
                              rho, nu, s, out, t_save, workspace, workspace_bytes, stream_, nullptr);
}

extern "C" int se3conv_fwd_prepared(const float* pts_in, const float* pts_out, const float* frames_in, const float* frames_out,
                                    const int32_t* neighbors, const int32_t* ends, const float* feat, const float* proj_axes,
                                    const float* proj_biases, const float* conv_weights, const float* rho, const float* nu,
                                    const se3conv_shape* s, float* out, float* t_save, void* workspace, size_t workspace_bytes,
                                    void* stream_, const se3conv_prepared* prep) {
  if (!shape_ok(s)) return SE3_ERR_INVALID_ARGUMENT;
  if (s->num_basis != kBasis) {
    // slices of 32 basis functions on the K = 32 operator (see slice_params_kernel); t_save is not written
    if (s->n_out == 0) return SE3_OK;
    if (!proj_axes || !proj_biases || !conv_weights || !out || !workspace) return SE3_ERR_INVALID_ARGUMENT;
    const se3conv_shape s32 = with_32_basis(s);
    const int64_t n_out_el = s->n_out * s->f_out * s->c_out;
    const AnyBasisLayout l = any_basis_layout(s, (size_t)n_out_el * 4, false, fwd_layout(&s32, 0).total);
    if (workspace_bytes < l.total) return SE3_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float *a32 = (float*)(ws + l.a32), *b32 = (float*)(ws + l.b32), *w32 = (float*)(ws + l.w32);
    for (int sl = 0; sl < l.slices; ++sl) {
      const int k0 = sl * kBasis, kn = s->num_basis - k0 < kBasis ? s->num_basis - k0 : kBasis;
      hipLaunchKernelGGL(slice_params_kernel, dim3(grid_for(slice_items(s))), dim3(256), 0, stream,
                         proj_axes, proj_biases, conv_weights, s->num_basis, k0, kn, s->c_in, s->c_out, a32, b32, w32);
      float* dst = sl == 0 ? out : (float*)(ws + l.out_tmp);
      if (int rc = se3conv_fwd(pts_in, pts_out, frames_in, frames_out, neighbors, ends, feat, a32, b32, w32, rho, nu, &s32, dst,
                               nullptr, ws + l.inner, l.total - l.inner, stream_))
        return rc;
      if (sl > 0)
        hipLaunchKernelGGL(add_into_kernel, dim3(grid_for(n_out_el)), dim3(256), 0, stream, out, dst, n_out_el);
    }
    return check_launch();
  }
  if (int rc = shape_supported(s)) return rc;
  if (s->n_out == 0) return SE3_OK;
  if (!pts_out || !frames_out || !ends || !proj_axes || !proj_biases || !conv_weights || !rho || !nu || !out ||
      !workspace || (s->n_edges > 0 && (!pts_in || !frames_in || !neighbors || !feat)))
    return SE3_ERR_INVALID_ARGUMENT;
  const FwdLayout l = fwd_layout(s, t_save != nullptr);
  if (workspace_bytes < l.total) return SE3_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  keep_side_streams_ready(stream);  // what a later captured se3conv_bwd may fork onto (never created inside a capture)
  char* ws = (char*)workspace;
  float* axes_ext = (float*)(ws + l.axes_ext);
  float* t = t_save ? t_save : (float*)(ws + l.t);

  EdgeGeom g = forward_geom(pts_in, pts_out, frames_in, frames_out, neighbors, ends, s);
  const int64_t rows_out = s->n_out * s->f_out;
  const int ck = s->c_in * s->num_basis;
  const float inv_fin = 1.0f / (float)s->f_in;
  // einsum('nik,iko->no') :210, /F_in :213, *norm_num_neighs_ :216 are the GEMM + its alpha
  const bool same_cloud = pts_in == pts_out && frames_in == frames_out && s->n_in == s->n_out && s->f_in == s->f_out;
  // the caller's records where it keeps them (se3conv_prepared), the workspace otherwise; `need_*`: built by this call
  const PreparedGeometry pg = prepared_geometry(prep, same_cloud, (float*)(ws + l.geom_in), (float*)(ws + l.geom_out));
  float* geom_in = pg.in;
  float* geom_out = pg.out;
  if (s->precision == SE3_PRECISION_FP32) {
    PrepBatch pb;  // one launch: [A; beta] table and the packed geometry records
    pb.axes(proj_axes, proj_biases, axes_ext);
    if (pg.need_in) pb.geometry(pts_in, frames_in, s->n_in, s->f_in, geom_in);
    if (pg.need_out) pb.geometry(pts_out, frames_out, s->n_out, s->f_out, geom_out);
    if (int rc = pb.launch(stream)) return rc;
    g.ctr_geom = geom_out, g.nb_geom = geom_in;
    if (int rc = launch_edge_t("edge_t_fwd", g, feat, s->c_in, s->n_in * s->f_in, axes_ext, rho, t, stream)) return rc;
    return launch_gemm_nn("gemm_out", t, conv_weights, out, rows_out, s->c_out, ck, nu, inv_fin, stream, (float*)(ws + l.split));
  }
  // (packed feature words: into the caller's buffer where backward will want them again)
  uint32_t* featpk = prep && prep->feat_words ? prep->feat_words : (uint32_t*)(ws + l.featpk);
  const bool need_featpk = !(prep && prep->feat_words && prep->feat_words_valid);
  uint16_t* bt_hi = (uint16_t*)(ws + l.bt_hi);
  uint16_t* bt_lo = (uint16_t*)(ws + l.bt_lo);
  const float inv_phi = inv_fin / kGeluOut;  // the bf16 edge kernels produce kGeluOut * phi (gelu_scaled)
  const int t24 = row_format(s, g, s->c_in, rows_out, s->c_out);  // 0 / 1 / 2 (row_format); se3conv_bwd decides the same way
  {  // one launch: [A; beta] table, packed geometry records, packed feature words, weight planes
    PrepBatch pb;
    pb.axes(proj_axes, proj_biases, axes_ext);
    if (pg.need_in) pb.geometry(pts_in, frames_in, s->n_in, s->f_in, geom_in);
    if (pg.need_out) pb.geometry(pts_out, frames_out, s->n_out, s->f_out, geom_out);
    if (need_featpk) pb.split(feat, featpk, s->n_in * s->f_in * s->c_in);
    pb.weights(conv_weights, s->c_in, s->num_basis, s->c_out, 0, bt_hi, bt_lo, nullptr, 1.0f, false, t24);
    if (int rc = pb.launch(stream)) return rc;
    g.ctr_geom = geom_out, g.nb_geom = geom_in;
  }
  if (int rc = launch_edge_t_bf16("edge_t_fwd", g, featpk, s->c_in, s->n_in * s->f_in, axes_ext, rho, (uint32_t*)t, stream,
                                  -1, -1, t24))
    return rc;
  return launch_gemm_nn_bf16("gemm_out", (const uint32_t*)t, bt_hi, bt_lo, out, false, rows_out, s->c_out, ck,
                             (float*)(ws + l.split), nu, inv_phi, stream, t24);
}

extern "C" int se3conv_bwd_needs_t(const se3conv_shape* s, int want_feat) {
  if (!shape_ok(s)) return SE3_ERR_INVALID_ARGUMENT;
  if (s->num_basis != kBasis) return 0;  // other K: T is recomputed per slice of 32 basis functions, `t_save` is never read
  return dw_from_u_available(s, want_feat != 0, true) ? 0 : 1;
}

extern "C" size_t se3conv_bwd_workspace_bytes(const se3conv_shape* s, int want_feat, int want_params, int have_t) {
  if (!shape_ok(s)) return 0;
  if (s->num_basis != kBasis) {
    const se3conv_shape s32 = with_32_basis(s);
    return any_basis_layout(s, want_feat ? (size_t)s->n_in * s->f_in * s->c_in * 4 : 0, want_params != 0,
                            bwd_layout(&s32, want_feat, want_params, 0).total).total;
  }
  return bwd_layout(s, want_feat, want_params, have_t).total;
}

extern "C" int se3conv_bwd(const float* pts_in, const float* pts_out, const float* frames_in, const float* frames_out,
                           const int32_t* neighbors, const int32_t* ends, const int32_t* t_samples,
                           const int32_t* t_ends, const int32_t* t_edge_ids, const float* feat, const float* proj_axes,
                           const float* proj_biases,
                           const float* conv_weights, const float* rho, const float* nu, const float* t_save,
                           const float* grad_out, const se3conv_shape* s, float* grad_feat, float* grad_axes,
                           float* grad_biases, float* grad_weights, void* workspace, size_t workspace_bytes,
                           void* stream_) {
  return se3conv_bwd_prepared(pts_in, pts_out, frames_in, frames_out, neighbors, ends, t_samples, t_ends, t_edge_ids, feat, proj_axes,
                              proj_biases, conv_weights, rho, nu, t_save, grad_out, s, grad_feat, grad_axes, grad_biases, grad_weights,
                              workspace, workspace_bytes, stream_, nullptr);
}

extern "C" int se3conv_bwd_prepared(const float* pts_in, const float* pts_out, const float* frames_in, const float* frames_out,
                                    const int32_t* neighbors, const int32_t* ends, const int32_t* t_samples,
                                    const int32_t* t_ends, const int32_t* t_edge_ids, const float* feat, const float* proj_axes,
                                    const float* proj_biases, const float* conv_weights, const float* rho, const float* nu,
                                    const float* t_save, const float* grad_out, const se3conv_shape* s, float* grad_feat,
                                    float* grad_axes, float* grad_biases, float* grad_weights, void* workspace,
                                    size_t workspace_bytes, void* stream_, const se3conv_prepared* prep) {
  if (!shape_ok(s)) return SE3_ERR_INVALID_ARGUMENT;
  if (s->num_basis != kBasis) {
    // slices of 32 basis functions (see slice_params_kernel): T is recomputed per slice (t_save ignored), dX is the sum of
    // the slices' feature gradients, every slice writes its own columns of the parameter gradients
    const bool wf = grad_feat != nullptr, wp = grad_axes || grad_biases || grad_weights;
    if (!wf && !wp) return SE3_OK;
    if (!proj_axes || !proj_biases || !conv_weights || !workspace) return SE3_ERR_INVALID_ARGUMENT;
    const se3conv_shape s32 = with_32_basis(s);
    const int64_t n_in_el = s->n_in * s->f_in * s->c_in;
    const AnyBasisLayout l = any_basis_layout(s, wf ? (size_t)n_in_el * 4 : 0, wp, bwd_layout(&s32, wf, wp, 0).total);
    if (workspace_bytes < l.total) return SE3_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    float *a32 = (float*)(ws + l.a32), *b32 = (float*)(ws + l.b32), *w32 = (float*)(ws + l.w32);
    float *da32 = wp ? (float*)(ws + l.da32) : nullptr, *db32 = wp ? (float*)(ws + l.db32) : nullptr;
    float* dw32 = wp ? (float*)(ws + l.dw32) : nullptr;
    const dim3 pgrid(grid_for(slice_items(s)));
    for (int sl = 0; sl < l.slices; ++sl) {
      const int k0 = sl * kBasis, kn = s->num_basis - k0 < kBasis ? s->num_basis - k0 : kBasis;
      hipLaunchKernelGGL(slice_params_kernel, pgrid, dim3(256), 0, stream, proj_axes, proj_biases, conv_weights,
                         s->num_basis, k0, kn, s->c_in, s->c_out, a32, b32, w32);
      float* dx = !wf ? nullptr : (sl == 0 ? grad_feat : (float*)(ws + l.out_tmp));
      if (int rc = se3conv_bwd(pts_in, pts_out, frames_in, frames_out, neighbors, ends, t_samples, t_ends, t_edge_ids, feat, a32, b32, w32, rho,
                               nu, nullptr, grad_out, &s32, dx, da32, db32, dw32, ws + l.inner, l.total - l.inner, stream_))
        return rc;
      if (wf && sl > 0 && n_in_el > 0)
        hipLaunchKernelGGL(add_into_kernel, dim3(grid_for(n_in_el)), dim3(256), 0, stream, grad_feat, dx, n_in_el);
      if (wp)
        hipLaunchKernelGGL(unslice_grads_kernel, pgrid, dim3(256), 0, stream, da32, db32, dw32, s->num_basis, k0, kn, s->c_in,
                           s->c_out, grad_axes, grad_biases, grad_weights);
    }
    return check_launch();
  }
  if (int rc = shape_supported(s)) return rc;
  const bool want_feat = grad_feat != nullptr;
  const bool want_params = grad_axes || grad_biases || grad_weights;
  if (!want_feat && !want_params) return SE3_OK;
  if (!proj_axes || !proj_biases || !conv_weights || !rho || !nu || !workspace) return SE3_ERR_INVALID_ARGUMENT;
  if (s->n_out > 0 && (!pts_out || !frames_out || !ends || !grad_out)) return SE3_ERR_INVALID_ARGUMENT;
  if (s->n_in > 0 && (!pts_in || !frames_in || !feat)) return SE3_ERR_INVALID_ARGUMENT;
  if (s->n_edges > 0 && !neighbors) return SE3_ERR_INVALID_ARGUMENT;
  if (want_feat && s->n_in > 0 && (!t_ends || (s->n_edges > 0 && !t_samples))) return SE3_ERR_INVALID_ARGUMENT;
  const BwdLayout l = bwd_layout(s, want_feat, want_params, t_save != nullptr);
  if (workspace_bytes < l.total) return SE3_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  keep_side_streams_ready(stream);
  char* ws = (char*)workspace;
  float* axes_ext = (float*)(ws + l.axes_ext);
  float* big = (float*)(ws + l.big);
  const int kb = s->num_basis;
  const int ck = s->c_in * kb;
  const int64_t rows_out = s->n_out * s->f_out, rows_in = s->n_in * s->f_in;
  const float inv_fin = 1.0f / (float)s->f_in;

  EdgeGeom g = forward_geom(pts_in, pts_out, frames_in, frames_out, neighbors, ends, s);
  // transposed graph: centre = input point, edges lead to output points (feature gradient)
  EdgeGeom gt{};
  gt.ctr_pts = pts_in, gt.ctr_frames = frames_in, gt.nb_pts = pts_out, gt.nb_frames = frames_out;
  gt.nbr = t_samples, gt.nbr_stride = 1, gt.nbr_offset = 0, gt.ends = t_ends;
  gt.n_ctr = s->n_in, gt.f_ctr = s->f_in, gt.f_nb = s->f_out, gt.transposed = 1;
  gt.n_nb = s->n_out;
  gt.n_edges = s->n_edges;
  float* partials = (float*)(ws + l.param_partials);
  float* tn_partials = (float*)(ws + l.tn_partials);

  if (s->precision == SE3_PRECISION_FP32) {
    {  // one launch: [A; beta] table and the packed geometry records of both sides
      const bool same_cloud = pts_in == pts_out && frames_in == frames_out && s->n_in == s->n_out && s->f_in == s->f_out;
      const PreparedGeometry pg = prepared_geometry(prep, same_cloud, (float*)(ws + l.geom_in), (float*)(ws + l.geom_out));
      float* geom_in = pg.in;
      float* geom_out = pg.out;
      PrepBatch pb;
      pb.axes(proj_axes, proj_biases, axes_ext);
      if (pg.need_in) pb.geometry(pts_in, frames_in, s->n_in, s->f_in, geom_in);
      if (pg.need_out) pb.geometry(pts_out, frames_out, s->n_out, s->f_out, geom_out);
      if (int rc = pb.launch(stream)) return rc;
      g.ctr_geom = geom_out, g.nb_geom = geom_in;
      gt.ctr_geom = geom_in, gt.nb_geom = geom_out;
    }
    if (want_params) {
      // gT[m,(i,k)] = alpha * sum_o g[m,o] W[i,k,o]
      float* wt = (float*)(ws + l.wt);
      hipLaunchKernelGGL(transpose_kernel, dim3(grid_for((int64_t)ck * s->c_out)), dim3(256), 0, stream, conv_weights,
                         wt, ck, s->c_out);
      if (int rc = launch_gemm_nn("gemm_gradT", grad_out, wt, big, rows_out, ck, s->c_out, nu, inv_fin, stream)) return rc;
      if (grad_axes || grad_biases) {
        int n_part = 0;
        if (int rc = launch_edge_param_grad("edge_param_grad", g, feat, s->c_in, rows_in, axes_ext, rho, big, partials,
                                            l.n_param_partials, &n_part, stream))
          return rc;
        hipLaunchKernelGGL(reduce_param_partials_kernel, dim3(kDescExt * kBasis), dim3(256), 0, stream, partials,
                           n_part, grad_axes, grad_biases, 1.0f);
      }
      if (grad_weights) {
        const float* t = t_save;
        if (!t) {
          float* tt = (float*)(ws + l.t);
          if (int rc = launch_edge_t("edge_t_recompute", g, feat, s->c_in, rows_in, axes_ext, rho, tt, stream)) return rc;
          t = tt;
        }
        // dW[(i,k),o] = alpha * sum_m T[m,(i,k)] g[m,o]
        if (int rc = launch_gemm_tn("gemm_gradW", t, grad_out, grad_weights, tn_partials, l.tn_splits, rows_out, ck,
                                    s->c_out, nu, inv_fin, stream))
          return rc;
      }
    }
    if (want_feat && rows_in > 0) {
      // Transposed convolution instead of scatter atomics:
      //   U[(p,b),o,k] = sum_{edges into p} sum_a phi(s,a,p,b)[k] g[(s,a),o];  dX[(p,b),i] = alpha * sum_{o,k} U W[i,k,o]
      if (int rc = launch_edge_t("edge_t_transposed", gt, grad_out, s->c_out, rows_out, axes_ext, rho, big, stream)) return rc;
      float* w2 = (float*)(ws + l.w2);
      hipLaunchKernelGGL(permute_weights_oki_kernel, dim3(grid_for((int64_t)ck * s->c_out)), dim3(256), 0, stream,
                         conv_weights, w2, s->c_in, kb, s->c_out);
      if (int rc = launch_gemm_nn("gemm_gradX", big, w2, grad_feat, rows_in, s->c_in, s->c_out * kb, nu, inv_fin, stream,
                                  (float*)(ws + l.split)))
        return rc;
    }
    return check_launch();
  }

  // ---- split-bf16 path: same stages, operands as packed words -----------------------------------------
  uint16_t* bt_hi = (uint16_t*)(ws + l.bt_hi);    // parameter branch: grad_T weights
  uint16_t* bt_lo = (uint16_t*)(ws + l.bt_lo);
  uint16_t* bx_hi = (uint16_t*)(ws + l.bt2_hi);   // feature branch: grad_X weights
  uint16_t* bx_lo = (uint16_t*)(ws + l.bt2_lo);
  uint32_t* gpk = (uint32_t*)(ws + l.gpk);
  // (the packed feature words the forward call left with the caller, se3conv_prepared, or this call's own)
  const bool have_featpk = prep && prep->feat_words && prep->feat_words_valid;
  uint32_t* featpk = prep && prep->feat_words ? prep->feat_words : (uint32_t*)(ws + l.featpk);
  uint32_t* bigw = (uint32_t*)big;
  const float inv_phi = inv_fin / kGeluOut;  // T and U hold kGeluOut * (the reference's values), see gelu_scaled
  const bool feat_branch = want_feat && rows_in > 0;
  const int t24_t = row_format(s, g, s->c_in, rows_out, s->c_out);  // as se3conv_fwd
  const int t24_u = feat_branch ? row_format(s, gt, s->c_out, rows_in, 0) : 0;
  const bool strip_t = gemm_strip_bf16_applicable(rows_out, ck, s->c_out);            // grad_T = g W^T
  const bool edge_dx = use_edge_dx(s, want_feat, want_params) && feat_branch;         // feature gradient edge-major (edge_dx.hip)
  const bool gt16 = strip_t && !edge_dx && grad_t_t16(s, g);                          // ... written as T16 rows
  {  // one launch: [A; beta] table, packed geometry records, packed words of g and f, weight planes
    const bool same_cloud = pts_in == pts_out && frames_in == frames_out && s->n_in == s->n_out && s->f_in == s->f_out;
    const PreparedGeometry pg = prepared_geometry(prep, same_cloud, (float*)(ws + l.geom_in), (float*)(ws + l.geom_out));
    float* geom_in = pg.in;
    float* geom_out = pg.out;
    PrepBatch pb;
    pb.axes(proj_axes, proj_biases, axes_ext);
    if (pg.need_in) pb.geometry(pts_in, frames_in, s->n_in, s->f_in, geom_in);
    if (pg.need_out) pb.geometry(pts_out, frames_out, s->n_out, s->f_out, geom_out);
    pb.split(grad_out, gpk, rows_out * s->c_out);
    if (feat_branch && !edge_dx) pb.weights(conv_weights, s->c_in, kb, s->c_out, 2, bx_hi, bx_lo, nullptr, 1.0f, false, t24_u);
    if (want_params && !have_featpk) pb.split(feat, featpk, rows_in * s->c_in);
    if (want_params || edge_dx)
      // alpha = nu/F_in is folded into these weights (one multiply per weight instead of one per grad_T element)
      pb.weights(conv_weights, s->c_in, kb, s->c_out, 1, bt_hi, bt_lo, nu, inv_fin, strip_t, 0, gt16);
    if (int rc = pb.launch(stream)) return rc;
    g.ctr_geom = geom_out, g.nb_geom = geom_in;
    gt.ctr_geom = geom_in, gt.nb_geom = geom_out;
  }
  // The sums that end the pass -- split-K partials of the grad_X GEMM, row-range partials of the weight-gradient GEMM,
  // per-workgroup partials of d[A; beta] -- write final outputs nobody in this call reads: one launch folds them all
  ReduceBatch final_sums;
  const bool dw_u = grad_weights != nullptr && feat_branch && !edge_dx && use_u_for_dw(s, want_feat, want_params, t_save != nullptr);
  const uint32_t* u_rows = nullptr;  // where the transposed pass wrote U (set by whoever launches it)
  auto weight_gradient = [&](hipStream_t st) -> int {
    if (!grad_weights) return SE3_OK;
    if (dw_u) {
      if (!u_rows) return SE3_ERR_LAUNCH;  // (a schedule that has not produced U yet: a bug, not an input error)
      // C'[(o,k), i] = sum_p U[p,(o,k)] f[p,i]; the batched reduction stores it as dW[i,k,o]
      return launch_gemm_tn_bf16("gemm_gradW", u_rows, featpk, grad_weights, tn_partials, l.tn_splits, rows_in, s->c_out * kb,
                                 s->c_in, nu, inv_phi, st, t24_u, &final_sums, true);
    }
    const uint32_t* t = (const uint32_t*)t_save;
    if (!t) {
      uint32_t* tt = (uint32_t*)(ws + l.t);
      if (int rc = launch_edge_t_bf16("edge_t_recompute", g, featpk, s->c_in, rows_in, axes_ext, rho, tt, st, -1, -1, t24_t))
        return rc;
      t = tt;
    }
    return launch_gemm_tn_bf16("gemm_gradW", t, gpk, grad_weights, tn_partials, l.tn_splits, rows_out, ck, s->c_out, nu,
                               inv_phi, st, t24_t, &final_sums);
  };
  // d[A; beta]: per-workgroup partial sums; their fixed-order reduction joins the batch (the bf16 kernels accumulate with
  // 2 GELU', gelu_scaled_grad: the 0.5 is applied there)
  auto param_gradients = [&](hipStream_t st) -> int {
    if (!grad_axes && !grad_biases) return SE3_OK;
    int n_part = 0;
    if (int rc = launch_edge_param_grad_bf16("edge_param_grad", g, featpk, s->c_in, rows_in, axes_ext, rho, bigw, partials,
                                             l.n_param_partials, &n_part, st, gt16))
      return rc;
    final_sums.params(partials, n_part, grad_axes, grad_biases, 0.5f);
    return SE3_OK;
  };

  if (edge_dx) {
    // grad_T first (both branches read it), then the edge-major feature gradient and its per-source sums, then the
    // parameter branch -- one stream: at these sizes (a sixth of a level's edges) nothing is worth a fork
    if (strip_t) {
      if (int rc = launch_gemm_strip_bf16("gemm_gradT", gpk, bt_hi, bt_lo, bigw, rows_out, ck, s->c_out, stream, false)) return rc;
    } else if (int rc = launch_gemm_nn_bf16("gemm_gradT", gpk, bt_hi, bt_lo, bigw, true, rows_out, ck, s->c_out,
                                            (float*)(ws + l.split), nullptr, 1.0f, stream)) {
      return rc;
    }
    float* d_rows = (float*)(ws + l.dx_rows);
    if (int rc = launch_edge_dx_bf16("edge_dx", g, axes_ext, rho, bigw, s->c_in, d_rows, stream)) return rc;
    if (int rc = launch_dx_gather_sum("dx_gather", d_rows, neighbors, ends, t_samples, t_ends, t_edge_ids, s->n_in, s->f_in * s->c_in,
                                      1.0f / kGeluOut, grad_feat, stream))
      return rc;
    if (want_params) {
      if (int rc = param_gradients(stream)) return rc;
      if (int rc = weight_gradient(stream)) return rc;
    }
    return final_sums.launch(stream);
  }
  bool branch_forked = false;
  ForkJoin fj;  // joins on every exit path from here on
  if (feat_branch) {
    // feature branch: on the side stream when there is a parameter branch to overlap with (SE3_OVERLAP)
    hipStream_t fs = stream;
    uint32_t* ubuf = bigw;
    float* fsplit = (float*)(ws + l.split);
    // opt-in (see kOverlapRows): the two branches side by side
    const int64_t overlap_rows = overlap_rows_limit();
    SideStream side;
    if (want_params && l.big_u != 0 && rows_out <= overlap_rows && (rows_out > kOverlapMinRows || overlap_rows >= ((int64_t)1 << 40)) &&
        (side = side_stream_for(stream)).ok) {
      if (int rc = fj.fork(side, stream)) return rc;
      fs = side.stream;
      ubuf = (uint32_t*)(ws + l.big_u);
      fsplit = (float*)(ws + l.split2);
      branch_forked = true;
    }
    // One stream: the two kernels that write a row-sized tensor (U, grad_T) go first, their readers after them.
    // Whatever runs right behind a ~1 GB writer is slowed while the caches drain (a memory-bound reader by 15-30 %,
    // whichever tensor it reads): in the order  U-writer, grad_X GEMM, grad_T writer, parameter gradients  two readers
    // sit in that position, here only one does (gemm_gradX 0.221 -> 0.187 ms at the headline shape).
    // SE3_BWD_BRANCH_ORDER=1 restores branch-by-branch order.
    static const bool branch_order = getenv("SE3_BWD_BRANCH_ORDER") != nullptr;
    if (!branch_order && !branch_forked && want_params && l.big_u != 0) {
      ubuf = (uint32_t*)(ws + l.big_u);
      fsplit = (float*)(ws + l.split2);
      u_rows = ubuf;
      if (int rc = launch_edge_t_bf16("edge_t_transposed", gt, gpk, s->c_out, rows_out, axes_ext, rho, ubuf, fs, -1, -1, t24_u))
        return rc;
      if (strip_t) {
        if (int rc = launch_gemm_strip_bf16("gemm_gradT", gpk, bt_hi, bt_lo, bigw, rows_out, ck, s->c_out, stream, gt16)) return rc;
      } else if (int rc = launch_gemm_nn_bf16("gemm_gradT", gpk, bt_hi, bt_lo, bigw, true, rows_out, ck, s->c_out,
                                              (float*)(ws + l.split), nullptr, 1.0f, stream)) {
        return rc;
      }
      if (int rc = param_gradients(stream)) return rc;
      if (int rc = launch_gemm_nn_bf16("gemm_gradX", ubuf, bx_hi, bx_lo, grad_feat, false, rows_in, s->c_in, s->c_out * kb,
                                       fsplit, nu, inv_phi, fs, t24_u, &final_sums))
        return rc;
      if (int rc = weight_gradient(stream)) return rc;
      return final_sums.launch(stream);
    }
    if (int rc = launch_edge_t_bf16("edge_t_transposed", gt, gpk, s->c_out, rows_out, axes_ext, rho, ubuf, fs, -1, -1, t24_u))
      return rc;
    if (int rc = launch_gemm_nn_bf16("gemm_gradX", ubuf, bx_hi, bx_lo, grad_feat, false, rows_in, s->c_in, s->c_out * kb,
                                     fsplit, nu, inv_phi, fs, t24_u, &final_sums))
      return rc;
    if (dw_u) {  // the weight gradient reads U: on the stream that wrote it, before the parameter branch reuses `big`
      u_rows = ubuf;
      if (int rc = weight_gradient(fs)) return rc;
    }
  }
  if (want_params) {
    if (strip_t) {
      if (int rc = launch_gemm_strip_bf16("gemm_gradT", gpk, bt_hi, bt_lo, bigw, rows_out, ck, s->c_out, stream, gt16)) return rc;
    } else if (int rc = launch_gemm_nn_bf16("gemm_gradT", gpk, bt_hi, bt_lo, bigw, true, rows_out, ck, s->c_out,
                                            (float*)(ws + l.split), nullptr, 1.0f, stream)) {
      return rc;
    }
    if (int rc = param_gradients(stream)) return rc;
    if (!(dw_u && feat_branch))
      if (int rc = weight_gradient(stream)) return rc;
  }
  if (int rc = fj.join()) return rc;  // the side stream's grad_X partials are complete before they are folded
  return final_sums.launch(stream);
}

// ---- optional per-kernel timing ------------------------------------------------------------------

namespace se3 {
namespace {
struct ProfRec { std::string tag; hipEvent_t start, stop; };
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::map<std::string, std::pair<double, int64_t>> g_prof_acc;
ProfRec* g_prof_open = nullptr;

void prof_drain_locked() {
  for (auto& r : g_prof_recs) {
    float ms = 0.f;
    if (hipEventSynchronize(r.stop) == hipSuccess && hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
      auto& a = g_prof_acc[r.tag];
      a.first += ms;
      a.second += 1;
    }
    (void)hipEventDestroy(r.start);
    (void)hipEventDestroy(r.stop);
  }
  g_prof_recs.clear();
}
}  // namespace

void prof_begin(const char* tag, hipStream_t stream) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  r.tag = tag;
  if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return;
  (void)hipEventRecord(r.start, stream);
  g_prof_recs.push_back(r);
  g_prof_open = &g_prof_recs.back();
}

void prof_end(hipStream_t stream) {
  if (!g_prof_on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_open) (void)hipEventRecord(g_prof_open->stop, stream);
  g_prof_open = nullptr;
}
}  // namespace se3

extern "C" int se3_side_stream_stats(int32_t* stats) {
  if (!stats) return SE3_ERR_INVALID_ARGUMENT;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SE3_ERR_LAUNCH;
  se3::SideTable& t = se3::side_table();
  std::lock_guard<std::mutex> lk(t.mu);
  stats[0] = (int32_t)t.by_stream.size();
  auto it = t.spare.find(dev);
  stats[1] = it == t.spare.end() ? 0 : (int32_t)it->second.size();
  stats[2] = t.created;
  stats[3] = t.unforked_in_capture;
  stats[4] = t.evicted;
  return SE3_OK;
}

extern "C" int se3_set_overlap_rows(int64_t rows) {
  se3::g_overlap_rows.store(rows < 0 ? -1 : rows, std::memory_order_relaxed);
  return SE3_OK;
}

extern "C" int se3_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(se3::g_prof_mu);
  if (!on) se3::prof_drain_locked();
  se3::g_prof_on = on != 0;
  return SE3_OK;
}

extern "C" int se3_profile_reset(void) {
  std::lock_guard<std::mutex> lk(se3::g_prof_mu);
  se3::prof_drain_locked();
  se3::g_prof_acc.clear();
  return SE3_OK;
}

extern "C" int se3_profile_read(const char* tag, double* total_ms, int64_t* launches) {
  if (!tag || !total_ms || !launches) return SE3_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(se3::g_prof_mu);
  se3::prof_drain_locked();
  auto it = se3::g_prof_acc.find(tag);
  *total_ms = it == se3::g_prof_acc.end() ? 0.0 : it->second.first;
  *launches = it == se3::g_prof_acc.end() ? 0 : it->second.second;
  return SE3_OK;
}

extern "C" int se3_profile_tags(char* buf, size_t len) {
  if (!buf || len == 0) return SE3_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(se3::g_prof_mu);
  se3::prof_drain_locked();
  std::string all;
  for (auto& kv : se3::g_prof_acc) all += (all.empty() ? "" : ",") + kv.first;
  if (all.size() + 1 > len) return SE3_ERR_WORKSPACE;
  std::memcpy(buf, all.c_str(), all.size() + 1);
  return SE3_OK;
}
