// Row-wise glue around the convolution (scope row f-3): what the reference runs as ~10 separate torch elementwise /
// reduction passes per ResNetFormer block (layers/ResNetFormer.py:64-88, layers/BatchNormPC.py:22-32,
// layers/SkipConnection.py, layers/DropPathPC.py:30-46) as one kernel per step:
//   se3_bn_fwd          training-mode BatchNorm1d in 3 launches: channel sums (fp64, fixed order) -> mean / invstd /
//                       running statistics -> y = x * scale + shift
//   se3_affine_act      y = act((x - center[c]) * scale[c] + shift[c])   eval-mode BN apply (act = none), bias + GELU
//   se3_bn_bwd_*        the two reductions and the element-wise pass of the batch-norm gradient
//   se3_skip_fwd/bwd    out = x * gamma[c] * gate[batch(row)] + y   SkipConnection + DropPathPC (frame-aware batch ids)
//   se3_bias_gelu_bwd   dz = g * GELU'(z + b), db = sum dz
// All feature tensors are [rows, C] fp32 row-major.  These are HBM-bound streams: 16-byte accesses along the channels
// when C % 4 == 0, channel sums through per-block partials reduced by a second launch (no atomics, fixed order).
#include "common.h"

namespace se3 {
namespace {

constexpr int kGlueThreads = 256;
constexpr int kGlueMaxBlocks = 1024;

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_exact_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// Thread layout shared by all kernels: `cv` = C / VEC vector columns; thread t handles vector column t % cv of rows
// t / cv, t / cv + rpb, ... (rpb = blockDim / cv rows per sweep); blocks stride over row sweeps.
struct RowWalk {
  int col, row0, rpb;
};
template <int VEC>
__device__ __forceinline__ RowWalk row_walk(int c) {
  const int cv = c / VEC;
  RowWalk w;
  w.rpb = kGlueThreads / cv;
  w.col = (threadIdx.x % cv) * VEC;
  w.row0 = threadIdx.x / cv;
  if (w.row0 >= w.rpb) w.row0 = -1;  // threads beyond rpb * cv idle
  return w;
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float v[VEC]) {
  if (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
  } else {
    v[0] = p[0];
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float v[VEC]) {
  if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else p[0] = v[0];
}

// Block-level channel sums of NQ quantities: per-thread fp64 accumulators -> LDS -> one partial row per block.
// partials layout: [NQ][gridDim.x][C].
template <int VEC, int NQ>
__device__ __forceinline__ void block_channel_sums(const double (&acc)[NQ][VEC], const RowWalk& w, int c,
                                                   double* __restrict__ partials) {
  __shared__ double red[kGlueThreads * 4];
  const int cv = c / VEC;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < VEC; ++v) red[threadIdx.x * VEC + v] = w.row0 >= 0 ? acc[q][v] : 0.0;
    __syncthreads();
    // thread t < C sums channel t over the rpb row slots
    for (int ch = threadIdx.x; ch < c; ch += kGlueThreads) {
      double s = 0.0;
      for (int r = 0; r < w.rpb; ++r) s += red[(r * cv + ch / VEC) * VEC + ch % VEC];
      partials[((int64_t)q * gridDim.x + blockIdx.x) * c + ch] = s;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(kGlueThreads) void bn_stats_kernel(const float* __restrict__ x, int64_t rows, int c,
                                                                 double* __restrict__ partials) {
  const RowWalk w = row_walk<VEC>(c);
  double acc[2][VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[0][v] = acc[1][v] = 0.0;
  if (w.row0 >= 0)
    for (int64_t r = (int64_t)blockIdx.x * w.rpb + w.row0; r < rows; r += (int64_t)gridDim.x * w.rpb) {
      float v[VEC];
      load_vec<VEC>(x + r * c + w.col, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[0][i] += (double)v[i], acc[1][i] += (double)v[i] * (double)v[i];
    }
  block_channel_sums<VEC, 2>(acc, w, c, partials);
}

// Second stage of the channel sums: one 256-thread block per channel, thread t adds the partials of blocks t, t + 256, ...
// and the block folds the 256 values in a fixed tree (a single thread walking all partials is a chain of ~1000
// dependent-latency loads: 0.1 ms per reduction at 1024 partial blocks).
__device__ __forceinline__ double block_sum_256(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// out[q][ch] = sum over blocks of partials[q][block][ch]
__global__ __launch_bounds__(256) void reduce_channel_partials_kernel(const double* __restrict__ partials, int n_blocks,
                                                                       int c, int nq, float* __restrict__ out0,
                                                                       float* __restrict__ out1) {
  __shared__ double red[256];
  const int ch = blockIdx.x;
  for (int q = 0; q < nq; ++q) {
    double s = 0.0;
    for (int b = threadIdx.x; b < n_blocks; b += 256) s += partials[((int64_t)q * n_blocks + b) * c + ch];
    s = block_sum_256(s, red);
    float* out = q == 0 ? out0 : out1;
    if (threadIdx.x == 0 && out) out[ch] = (float)s;
  }
}

// batch-norm statistics from the block partials (fp64 all the way to the variance): mean, invstd, the scale of the
// apply pass, and the running statistics of torch.nn.BatchNorm1d (unbiased variance, momentum m); block = channel
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ partials, int n_blocks, int c,
                                                           int64_t rows, const float* __restrict__ weight, float eps,
                                                           float momentum, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var,
                                                           int64_t* __restrict__ num_batches_tracked,
                                                           float* __restrict__ mean, float* __restrict__ invstd,
                                                           float* __restrict__ scale) {
  __shared__ double red[256];
  const int ch = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int b = threadIdx.x; b < n_blocks; b += 256)
    s += partials[(int64_t)b * c + ch], ss += partials[((int64_t)n_blocks + b) * c + ch];
  s = block_sum_256(s, red);
  ss = block_sum_256(ss, red);
  if (threadIdx.x != 0) return;
  const double n = (double)rows;
  const double mu = rows > 0 ? s / n : 0.0;
  double var = rows > 0 ? ss / n - mu * mu : 0.0;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[ch] = (float)mu, invstd[ch] = is;
  scale[ch] = (weight ? weight[ch] : 1.0f) * is;
  if (running_mean) running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * (float)mu;
  if (running_var) running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * (float)(rows > 1 ? var * n / (n - 1.0) : var);
  if (num_batches_tracked && ch == 0) *num_batches_tracked += 1;  // BatchNorm1d's own counter (one launch less than add_(1))
}

template <int VEC>
__global__ __launch_bounds__(kGlueThreads) void affine_act_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ center,
                                                                   const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, int64_t rows, int c,
                                                                   int act, float* __restrict__ y) {
  // y = act((x - center) * scale + shift): the centred form keeps batch norm exact where the variance is tiny next to
  // the mean (x * scale + (shift - mean * scale) cancels there)
  const RowWalk w = row_walk<VEC>(c);
  if (w.row0 < 0) return;
  float ce[VEC], sc[VEC], sh[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i)
    ce[i] = center ? center[w.col + i] : 0.0f, sc[i] = scale ? scale[w.col + i] : 1.0f, sh[i] = shift ? shift[w.col + i] : 0.0f;
  for (int64_t r = (int64_t)blockIdx.x * w.rpb + w.row0; r < rows; r += (int64_t)gridDim.x * w.rpb) {
    float v[VEC];
    load_vec<VEC>(x + r * c + w.col, v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float t = fmaf(v[i] - ce[i], sc[i], sh[i]);
      v[i] = act == 1 ? gelu_exact(t) : t;
    }
    store_vec<VEC>(y + r * c + w.col, v);
  }
}

// batch-norm gradient, reductions: s1[c] = sum dy, s2[c] = sum dy * xhat,  xhat = (x - mean) * invstd
template <int VEC>
__global__ __launch_bounds__(kGlueThreads) void bn_bwd_reduce_kernel(const float* __restrict__ dy,
                                                                      const float* __restrict__ x,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd, int64_t rows,
                                                                      int c, double* __restrict__ partials) {
  const RowWalk w = row_walk<VEC>(c);
  double acc[2][VEC];
  float mu[VEC], is[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    acc[0][v] = acc[1][v] = 0.0;
    mu[v] = w.row0 >= 0 ? mean[w.col + v] : 0.f, is[v] = w.row0 >= 0 ? invstd[w.col + v] : 0.f;
  }
  if (w.row0 >= 0)
    for (int64_t r = (int64_t)blockIdx.x * w.rpb + w.row0; r < rows; r += (int64_t)gridDim.x * w.rpb) {
      float g[VEC], v[VEC];
      load_vec<VEC>(dy + r * c + w.col, g);
      load_vec<VEC>(x + r * c + w.col, v);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[0][i] += (double)g[i];
        acc[1][i] += (double)g[i] * (double)((v[i] - mu[i]) * is[i]);
      }
    }
  block_channel_sums<VEC, 2>(acc, w, c, partials);
}

// dx = gamma * invstd * (dy - s1 / N - xhat * s2 / N)
template <int VEC>
__global__ __launch_bounds__(kGlueThreads) void bn_bwd_apply_kernel(const float* __restrict__ dy,
                                                                     const float* __restrict__ x,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ s1,
                                                                     const float* __restrict__ s2, int64_t rows, int c,
                                                                     float* __restrict__ dx) {
  const RowWalk w = row_walk<VEC>(c);
  if (w.row0 < 0) return;
  const float inv_n = 1.0f / (float)rows;
  float mu[VEC], is[VEC], k0[VEC], k1[VEC], k2[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    mu[i] = mean[w.col + i], is[i] = invstd[w.col + i];
    k0[i] = (gamma ? gamma[w.col + i] : 1.0f) * is[i];
    k1[i] = s1[w.col + i] * inv_n, k2[i] = s2[w.col + i] * inv_n;
  }
  for (int64_t r = (int64_t)blockIdx.x * w.rpb + w.row0; r < rows; r += (int64_t)gridDim.x * w.rpb) {
    float g[VEC], v[VEC];
    load_vec<VEC>(dy + r * c + w.col, g);
    load_vec<VEC>(x + r * c + w.col, v);
#pragma unroll
    for (int i = 0; i < VEC; ++i) g[i] = k0[i] * (g[i] - k1[i] - (v[i] - mu[i]) * is[i] * k2[i]);
    store_vec<VEC>(dx + r * c + w.col, g);
  }
}

// out = x * gamma[c] * gate[batch(row)] + y;   gate = NULL: no drop path (eval mode or drop probability 0)
// drop-path factor of a row: gate_keep == 0: gate[b] is the factor itself; gate_keep > 0: gate[b] is the uniform draw u of
// DropPathPC.py:38-41 and the factor is floor(keep + u) * (1 / keep) -- torch's add, floor and div launches folded in
__device__ __forceinline__ float gate_factor(const float* __restrict__ gate, float gate_keep, float gate_scale,
                                             const int32_t* __restrict__ row_batch, int64_t r) {
  if (!gate) return 1.0f;
  const float u = gate[row_batch[r]];
  return gate_keep > 0.f ? floorf(gate_keep + u) * gate_scale : u;
}

template <int VEC>
__global__ __launch_bounds__(kGlueThreads) void skip_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ gate, float gate_keep,
                                                                 float gate_scale,
                                                                 const int32_t* __restrict__ row_batch, int64_t rows,
                                                                 int c, float* __restrict__ out) {
  const RowWalk w = row_walk<VEC>(c);
  if (w.row0 < 0) return;
  float ga[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) ga[i] = gamma[w.col + i];
  for (int64_t r = (int64_t)blockIdx.x * w.rpb + w.row0; r < rows; r += (int64_t)gridDim.x * w.rpb) {
    const float gt = gate_factor(gate, gate_keep, gate_scale, row_batch, r);
    float a[VEC], b[VEC];
    load_vec<VEC>(x + r * c + w.col, a);
    load_vec<VEC>(y + r * c + w.col, b);
#pragma unroll
    for (int i = 0; i < VEC; ++i) b[i] = fmaf(a[i] * ga[i], gt, b[i]);
    store_vec<VEC>(out + r * c + w.col, b);
  }
}

// dx = g * gamma[c] * gate;  dgamma[c] = sum_r g * x * gate   (dy = g: no kernel)
template <int VEC>
__global__ __launch_bounds__(kGlueThreads) void skip_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ gate, float gate_keep,
                                                                 float gate_scale,
                                                                 const int32_t* __restrict__ row_batch, int64_t rows,
                                                                 int c, float* __restrict__ dx,
                                                                 double* __restrict__ partials) {
  const RowWalk w = row_walk<VEC>(c);
  double acc[1][VEC];
  float ga[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[0][i] = 0.0, ga[i] = w.row0 >= 0 ? gamma[w.col + i] : 0.f;
  if (w.row0 >= 0)
    for (int64_t r = (int64_t)blockIdx.x * w.rpb + w.row0; r < rows; r += (int64_t)gridDim.x * w.rpb) {
      const float gt = gate_factor(gate, gate_keep, gate_scale, row_batch, r);
      float gv[VEC], xv[VEC];
      load_vec<VEC>(g + r * c + w.col, gv);
      load_vec<VEC>(x + r * c + w.col, xv);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc[0][i] += (double)(gv[i] * xv[i] * gt);
        gv[i] = gv[i] * ga[i] * gt;
      }
      if (dx) store_vec<VEC>(dx + r * c + w.col, gv);
    }
  block_channel_sums<VEC, 1>(acc, w, c, partials);
}

// dz = g * GELU'(z + b);  db[c] = sum_r dz
template <int VEC>
__global__ __launch_bounds__(kGlueThreads) void bias_gelu_bwd_kernel(const float* __restrict__ g,
                                                                      const float* __restrict__ z,
                                                                      const float* __restrict__ bias, int64_t rows,
                                                                      int c, float* __restrict__ dz,
                                                                      double* __restrict__ partials) {
  const RowWalk w = row_walk<VEC>(c);
  double acc[1][VEC];
  float bi[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc[0][i] = 0.0, bi[i] = (w.row0 >= 0 && bias) ? bias[w.col + i] : 0.f;
  if (w.row0 >= 0)
    for (int64_t r = (int64_t)blockIdx.x * w.rpb + w.row0; r < rows; r += (int64_t)gridDim.x * w.rpb) {
      float gv[VEC], zv[VEC];
      load_vec<VEC>(g + r * c + w.col, gv);
      load_vec<VEC>(z + r * c + w.col, zv);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        gv[i] *= gelu_exact_grad(zv[i] + bi[i]);
        acc[0][i] += (double)gv[i];
      }
      store_vec<VEC>(dz + r * c + w.col, gv);
    }
  block_channel_sums<VEC, 1>(acc, w, c, partials);
}

inline int glue_blocks(int64_t rows, int c, int vec) {
  const int rpb = kGlueThreads / (c / vec);
  int64_t b = (rows + rpb - 1) / rpb;
  b = (b + 7) / 8;  // ~8 row sweeps per block
  if (b < 1) b = 1;
  if (b > kGlueMaxBlocks) b = kGlueMaxBlocks;
  return (int)b;
}
inline bool glue_shape_ok(int64_t rows, int c) { return rows >= 0 && c >= 1 && c <= kGlueThreads * 4; }
inline int glue_vec(int c) { return (c % 4 == 0 && c / 4 <= kGlueThreads) ? 4 : 1; }
inline int finish_channel_sums(double* partials, int blocks, int c, int nq, float* out0, float* out1,
                               hipStream_t stream) {
  hipLaunchKernelGGL(reduce_channel_partials_kernel, dim3(c), dim3(256), 0, stream, partials, blocks, c, nq, out0, out1);
  return check_launch();
}

}  // namespace
}  // namespace se3

using namespace se3;

// partial sums of the channel reductions (up to 2 quantities x kGlueMaxBlocks blocks x C doubles) + 2 C floats
extern "C" size_t se3_glue_workspace_bytes(int32_t c) {
  return c >= 1 ? (size_t)2 * kGlueMaxBlocks * c * sizeof(double) + (size_t)2 * c * sizeof(float) : 0;
}

#define SE3_GLUE_DISPATCH(KERNEL, ...)                                                                          \
  do {                                                                                                          \
    if (vec == 4) hipLaunchKernelGGL((KERNEL<4>), dim3(blocks), dim3(kGlueThreads), 0, stream, __VA_ARGS__);    \
    else hipLaunchKernelGGL((KERNEL<1>), dim3(blocks), dim3(kGlueThreads), 0, stream, __VA_ARGS__);             \
  } while (0)

extern "C" int se3_bn_fwd(const float* x, const float* weight, const float* bias, int64_t rows, int32_t c, float eps,
                          float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y,
                          float* save_mean, float* save_invstd, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!glue_shape_ok(rows, c) || !save_mean || !save_invstd || !workspace || (rows > 0 && (!x || !y)))
    return SE3_ERR_INVALID_ARGUMENT;
  if (c > kGlueThreads && c % 4 != 0) return SE3_ERR_UNSUPPORTED;
  if (workspace_bytes < se3_glue_workspace_bytes(c)) return SE3_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int vec = glue_vec(c), blocks = glue_blocks(rows, c, vec);
  double* partials = (double*)workspace;
  // the scale of the apply pass lives behind the partial sums
  float* scale = (float*)((char*)workspace + (size_t)2 * kGlueMaxBlocks * c * sizeof(double));
  SE3_GLUE_DISPATCH(bn_stats_kernel, x, rows, (int)c, partials);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(256), 0, stream, (const double*)partials, blocks,
                     (int)c, rows, weight, eps, momentum, running_mean, running_var, num_batches_tracked, save_mean, save_invstd,
                     scale);
  if (rows > 0)
    SE3_GLUE_DISPATCH(affine_act_kernel, x, (const float*)save_mean, (const float*)scale, bias, rows, (int)c, 0, y);
  return check_launch();
}

extern "C" int se3_affine_act(const float* x, const float* center, const float* scale, const float* shift, int64_t rows,
                              int32_t c, int32_t act, float* y, void* stream_) {
  if (!glue_shape_ok(rows, c) || (act != 0 && act != 1)) return SE3_ERR_INVALID_ARGUMENT;
  if (c > kGlueThreads && c % 4 != 0) return SE3_ERR_UNSUPPORTED;
  if (rows == 0) return SE3_OK;
  if (!x || !y) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const int vec = glue_vec(c), blocks = glue_blocks(rows, c, vec);
  SE3_GLUE_DISPATCH(affine_act_kernel, x, center, scale, shift, rows, (int)c, (int)act, y);
  return check_launch();
}

extern "C" int se3_bn_bwd(const float* dy, const float* x, const float* mean, const float* invstd, const float* gamma,
                          int64_t rows, int32_t c, float* dx, float* dgamma, float* dbeta, void* workspace,
                          size_t workspace_bytes, void* stream_) {
  if (!glue_shape_ok(rows, c) || !mean || !invstd || !dgamma || !dbeta || !workspace || (rows > 0 && (!dy || !x || !dx)))
    return SE3_ERR_INVALID_ARGUMENT;
  if (c > kGlueThreads && c % 4 != 0) return SE3_ERR_UNSUPPORTED;
  if (workspace_bytes < se3_glue_workspace_bytes(c)) return SE3_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int vec = glue_vec(c), blocks = glue_blocks(rows, c, vec);
  double* partials = (double*)workspace;
  SE3_GLUE_DISPATCH(bn_bwd_reduce_kernel, dy, x, mean, invstd, rows, (int)c, partials);
  if (int rc = finish_channel_sums(partials, blocks, c, 2, dbeta, dgamma, stream)) return rc;  // s1 = dbeta, s2 = dgamma
  if (rows == 0) return SE3_OK;
  SE3_GLUE_DISPATCH(bn_bwd_apply_kernel, dy, x, mean, invstd, gamma, (const float*)dbeta, (const float*)dgamma, rows,
                    (int)c, dx);
  return check_launch();
}

extern "C" int se3_skip_fwd(const float* x, const float* y, const float* gamma, const float* gate, float gate_keep,
                            const int32_t* row_batch, int64_t rows, int32_t c, float* out, void* stream_) {
  if (!glue_shape_ok(rows, c)) return SE3_ERR_INVALID_ARGUMENT;
  if (c > kGlueThreads && c % 4 != 0) return SE3_ERR_UNSUPPORTED;
  if (rows == 0) return SE3_OK;
  if (!x || !y || !out || !gamma || (gate && !row_batch)) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const int vec = glue_vec(c), blocks = glue_blocks(rows, c, vec);
  SE3_GLUE_DISPATCH(skip_fwd_kernel, x, y, gamma, gate, gate_keep, gate_keep > 0.f ? 1.0f / gate_keep : 1.0f,
                    row_batch, rows, (int)c, out);
  return check_launch();
}

extern "C" int se3_skip_bwd(const float* g, const float* x, const float* gamma, const float* gate, float gate_keep,
                            const int32_t* row_batch, int64_t rows, int32_t c, float* dx, float* dgamma, void* workspace,
                            size_t workspace_bytes, void* stream_) {
  if (!glue_shape_ok(rows, c) || !gamma || !dgamma || !workspace || (rows > 0 && (!g || !x || (gate && !row_batch))))
    return SE3_ERR_INVALID_ARGUMENT;
  if (c > kGlueThreads && c % 4 != 0) return SE3_ERR_UNSUPPORTED;
  if (workspace_bytes < se3_glue_workspace_bytes(c)) return SE3_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int vec = glue_vec(c), blocks = glue_blocks(rows, c, vec);
  double* partials = (double*)workspace;
  SE3_GLUE_DISPATCH(skip_bwd_kernel, g, x, gamma, gate, gate_keep, gate_keep > 0.f ? 1.0f / gate_keep : 1.0f,
                    row_batch, rows, (int)c, dx, partials);
  return finish_channel_sums(partials, blocks, c, 1, dgamma, nullptr, stream);
}

extern "C" int se3_bias_gelu_bwd(const float* g, const float* z, const float* bias, int64_t rows, int32_t c, float* dz,
                                 float* dbias, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!glue_shape_ok(rows, c) || !dbias || !workspace || (rows > 0 && (!g || !z || !dz))) return SE3_ERR_INVALID_ARGUMENT;
  if (c > kGlueThreads && c % 4 != 0) return SE3_ERR_UNSUPPORTED;
  if (workspace_bytes < se3_glue_workspace_bytes(c)) return SE3_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int vec = glue_vec(c), blocks = glue_blocks(rows, c, vec);
  double* partials = (double*)workspace;
  SE3_GLUE_DISPATCH(bias_gelu_bwd_kernel, g, z, bias, rows, (int)c, dz, partials);
  return finish_channel_sums(partials, blocks, c, 1, dbias, nullptr, stream);
}

// Weight gradient of a point-wise linear layer y = x W^T (+ b): grad_w[n_out, n_in] = grad_y^T x, a reduction over all rows
// of the cloud into a small matrix.  The BLAS heuristics run this shape on the handful of workgroups its output tiles give
// (131 072 rows into 128 x 64 outputs: 0.28-0.36 ms, half of a ResNetFormer block's glue); gemm_tn splits the rows over the
// chip and reduces the partials (gemm.hip: ~512 workgroups, two operand batches in flight each).
extern "C" size_t se3_linear_wgrad_workspace_bytes(int64_t rows, int32_t n_out, int32_t n_in) {
  if (rows < 0 || n_out < 1 || n_in < 1) return 0;
  return (size_t)gemm_tn_splits(rows, n_out, n_in) * n_out * n_in * 4;
}

extern "C" int se3_linear_wgrad(const float* grad_y, const float* x, int64_t rows, int32_t n_out, int32_t n_in, float* grad_w,
                                void* workspace, size_t workspace_bytes, void* stream_) {
  if (rows < 0 || n_out < 1 || n_in < 1 || !grad_w || !workspace || (rows > 0 && (!grad_y || !x)))
    return SE3_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < se3_linear_wgrad_workspace_bytes(rows, n_out, n_in)) return SE3_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  if (rows == 0) return launch_fill_words(grad_w, 0u, (int64_t)n_out * n_in, stream);
  return launch_gemm_tn("linear_wgrad", grad_y, x, grad_w, (float*)workspace, gemm_tn_splits(rows, n_out, n_in), rows, n_out,
                        n_in, nullptr, 1.0f, stream);
}
