// Operand preparation of the split-bf16 path, batched into ONE launch per call: [A; beta] table, packed geometry
// records, packed hi/lo words of the gathered operands, re-laid-out weight planes.  Each of these is a few
// microseconds of work; as separate launches (8 in the backward pass) they cost more in launch latency than in
// execution on the small hierarchy levels.  A block finds its job from blockIdx.x and runs it grid-stride over the
// job's own block count.
#include "common.h"

namespace se3 {

namespace {

enum { kJobAxes = 0, kJobGeometry = 1, kJobSplit = 2, kJobWeights = 3 };

__device__ void job_axes(const PrepJob& j, int block, int) {
  const float* axes = (const float*)j.a;
  const float* biases = (const float*)j.b;
  float* ext = (float*)j.o0;
  const int i = block * blockDim.x + threadIdx.x;
  if (i < SE3_DESC_DIMS * kBasis) ext[i] = axes[i];
  else if (i < kDescExt * kBasis) ext[i] = biases[i - SE3_DESC_DIMS * kBasis];
}

// [N,3] points + [N,F,9] frames -> one 64-byte record per (point, frame) row, see load_geom_record
__device__ void job_geometry(const PrepJob& j, int block, int blocks) {
  const float* pts = (const float*)j.a;
  const float* frames = (const float*)j.b;
  float* records = (float*)j.o0;
  const int f = j.p[0];
  for (int64_t row = (int64_t)block * blockDim.x + threadIdx.x; row < j.n; row += (int64_t)blocks * blockDim.x) {
    const int64_t pt = row / f;
    const float* r = frames + row * 9;
    float4* dst = reinterpret_cast<float4*>(records + row * 16);
    dst[0] = make_float4(pts[pt * 3], pts[pt * 3 + 1], pts[pt * 3 + 2], r[8]);
    dst[1] = make_float4(r[0], r[1], r[2], r[3]);
    dst[2] = make_float4(r[4], r[5], r[6], r[7]);
    dst[3] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

__device__ void job_split(const PrepJob& j, int block, int blocks) {
  const float* src = (const float*)j.a;
  uint32_t* dst = (uint32_t*)j.o0;
  const int64_t n = j.n;
  for (int64_t i4 = ((int64_t)block * blockDim.x + threadIdx.x) * 4; i4 < n; i4 += (int64_t)blocks * blockDim.x * 4) {
    if (i4 + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(src + i4);
      *reinterpret_cast<uint4*>(dst + i4) = make_uint4(split_pack(v.x), split_pack(v.y), split_pack(v.z), split_pack(v.w));
    } else {
      for (int64_t i = i4; i < n; ++i) dst[i] = split_pack(src[i]);
    }
  }
}

// see prep_weights_kernel (gemm_bf16.hip) for the modes and the fragment layout
__device__ void job_weights(const PrepJob& j, int block, int blocks) {
  const float* w = (const float*)j.a;
  const float* scale_num = (const float*)j.b;
  uint16_t* bt_hi = (uint16_t*)j.o0;
  uint16_t* bt_lo = (uint16_t*)j.o1;
  const int kb = j.p[1], c_out = j.p[2], mode = j.p[3], n = j.p[4], k = j.p[5], kp = j.p[6], frag = j.p[7];
  const float sc = (scale_num ? *scale_num : 1.0f) * j.scale;
  const int64_t total = (int64_t)n * kp;
  for (int64_t idx = (int64_t)block * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)blocks * blockDim.x) {
    const int nn = (int)(idx / kp), kk = (int)(idx % kp);
    float v = 0.f;
    if (kk < k) {
      // the A operand is in the 3-byte (2) or the T16 (4) row format: its k order
      const int ks = (frag & 2) ? t24_k_of(kk) : ((frag & 4) ? t16_k_of(kk) : kk);
      if (mode == 0) v = w[(int64_t)ks * c_out + nn];
      else if (mode == 1) v = w[(int64_t)((frag & 8) ? t16_k_of(nn) : nn) * c_out + kk];  // 8: output columns in T16 order
      else if (mode == 2) v = w[((int64_t)nn * kb + (ks % kb)) * c_out + ks / kb];
      else v = w[((int64_t)kk * kb + (nn % kb)) * c_out + nn / kb];
    }
    const uint32_t pk = split_pack(v * sc);
    const int64_t o = (frag & 1) ? ((((int64_t)(nn / 32) * (kp / 16) + kk / 16) * 64 + ((kk % 16) / 8) * 32 + nn % 32) * 8 + kk % 8)
                           : idx;
    bt_hi[o] = (uint16_t)(pk >> 16);
    bt_lo[o] = (uint16_t)(pk & 0xffffu);
  }
}

__global__ __launch_bounds__(256) void prep_batch_kernel(PrepJobs jobs) {
  int block = blockIdx.x, ji = 0;
  while (ji + 1 < jobs.count && block >= jobs.job[ji].blocks) block -= jobs.job[ji++].blocks;
  const PrepJob& j = jobs.job[ji];
  switch (j.type) {
    case kJobAxes: job_axes(j, block, j.blocks); break;
    case kJobGeometry: job_geometry(j, block, j.blocks); break;
    case kJobSplit: job_split(j, block, j.blocks); break;
    default: job_weights(j, block, j.blocks); break;
  }
}

int blocks_for(int64_t work_items, int cap) {
  const int64_t b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

void PrepBatch::axes(const float* axes, const float* biases, float* ext) {
  PrepJob& j = jobs.job[jobs.count++];
  j = PrepJob{};
  j.type = kJobAxes, j.blocks = 2, j.a = axes, j.b = biases, j.o0 = ext;
}

void PrepBatch::geometry(const float* pts, const float* frames, int64_t n, int f, float* records) {
  if (n * f == 0) return;
  if (n * f * 64 >= 0x7fff0000ll) status = SE3_ERR_UNSUPPORTED;  // 32-bit record offsets in the kernels
  PrepJob& j = jobs.job[jobs.count++];
  j = PrepJob{};
  j.type = kJobGeometry, j.blocks = blocks_for(n * f, 2048), j.a = pts, j.b = frames, j.o0 = records, j.n = n * f, j.p[0] = f;
}

void PrepBatch::split(const float* src, uint32_t* dst, int64_t n) {
  if (n == 0) return;
  PrepJob& j = jobs.job[jobs.count++];
  j = PrepJob{};
  j.type = kJobSplit, j.blocks = blocks_for((n + 3) / 4, 4096), j.a = src, j.o0 = dst, j.n = n;
}

void PrepBatch::weights(const float* w, int c_in, int kb, int c_out, int mode, uint16_t* bt_hi, uint16_t* bt_lo,
                        const float* scale_num, float scale, bool frag_layout, int a_rowfmt, bool out_t16) {
  int n, k;
  if (mode == 0) n = c_out, k = c_in * kb;
  else if (mode == 1) n = c_in * kb, k = c_out;
  else if (mode == 2) n = c_in, k = c_out * kb;
  else n = c_out * kb, k = c_in;
  const int kp = (k + 31) / 32 * 32;
  PrepJob& j = jobs.job[jobs.count++];
  j = PrepJob{};
  j.type = kJobWeights, j.blocks = blocks_for((int64_t)n * kp, 2048), j.a = w, j.b = scale_num, j.o0 = bt_hi, j.o1 = bt_lo;
  j.p[0] = c_in, j.p[1] = kb, j.p[2] = c_out, j.p[3] = mode, j.p[4] = n, j.p[5] = k, j.p[6] = kp, j.p[7] = (frag_layout ? 1 : 0) | (a_rowfmt == 1 ? 2 : 0) | (a_rowfmt == 2 ? 4 : 0) | (out_t16 ? 8 : 0);
  j.scale = scale;
}

// ---- batched reductions (ReduceBatch, common.h) ------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void reduce_batch_kernel(ReduceJobs jobs) {
  __shared__ float red[256];
  int block = blockIdx.x, ji = 0;
  while (ji + 1 < jobs.count && block >= jobs.job[ji].blocks) block -= jobs.job[ji++].blocks;
  const ReduceJob& j = jobs.job[ji];
  if (j.type == 0) {
    const float alpha = (j.alpha_num ? *j.alpha_num : 1.0f) * j.alpha_scale;
    for (int64_t i = (int64_t)block * blockDim.x + threadIdx.x; i < j.count; i += (int64_t)j.blocks * blockDim.x) {
      float s = 0.f;
#pragma unroll 8
      for (int z = 0; z < j.splits; ++z) s += j.partials[(int64_t)z * j.count + i];
      if (j.packed) static_cast<uint32_t*>(j.out)[i] = split_pack(alpha * s);
      else if (j.perm_n > 0) {  // C'[(o * kBasis + k), ci] -> dW[(ci * kBasis + k), o]
        const int64_t ok = i / j.perm_n;
        const int ci = (int)(i - ok * j.perm_n), k = (int)(ok % kBasis);
        const int64_t o = ok / kBasis, c_out = j.count / ((int64_t)j.perm_n * kBasis);
        static_cast<float*>(j.out)[((int64_t)ci * kBasis + k) * c_out + o] = alpha * s;
      } else static_cast<float*>(j.out)[i] = alpha * s;
    }
    return;
  }
  // d[A; beta]: one block per output element, 256 threads stride over the per-workgroup partials, then a tree
  const int i = block;
  float s = 0.f;
  for (int p = threadIdx.x; p < j.splits; p += 256) s += j.partials[(int64_t)p * kDescExt * kBasis + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* ga = static_cast<float*>(j.out);
    float* gb = static_cast<float*>(j.out2);
    if (i < SE3_DESC_DIMS * kBasis) {
      if (ga) ga[i] = red[0] * j.alpha_scale;
    } else if (gb) {
      gb[i - SE3_DESC_DIMS * kBasis] = red[0] * j.alpha_scale;
    }
  }
}
}  // namespace

void ReduceBatch::sum(const float* partials, void* out, int64_t count, int splits, const float* alpha_num, float alpha_scale,
                      bool packed, int perm_n) {
  if (count == 0) return;
  ReduceJob& j = jobs.job[jobs.count++];
  j = ReduceJob{};
  j.type = 0, j.blocks = blocks_for(count, 2048), j.partials = partials, j.out = out, j.count = count, j.splits = splits;
  j.packed = packed ? 1 : 0, j.alpha_num = alpha_num, j.alpha_scale = alpha_scale, j.perm_n = perm_n;
}

void ReduceBatch::params(const float* partials, int n_partials, float* grad_axes, float* grad_biases, float scale) {
  ReduceJob& j = jobs.job[jobs.count++];
  j = ReduceJob{};
  j.type = 1, j.blocks = kDescExt * kBasis, j.partials = partials, j.out = grad_axes, j.out2 = grad_biases;
  j.splits = n_partials, j.alpha_scale = scale;
}

int ReduceBatch::launch(hipStream_t stream) {
  if (jobs.count == 0) return check_launch();  // still the caller's last word on the launches before it
  int total = 0;
  for (int i = 0; i < jobs.count; ++i) total += jobs.job[i].blocks;
  ProfScope prof("reductions", stream);
  hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)total), dim3(256), 0, stream, jobs);
  jobs.count = 0;
  return check_launch();
}

int PrepBatch::launch(hipStream_t stream) {
  if (status != SE3_OK) return status;
  if (jobs.count == 0) return SE3_OK;
  int total = 0;
  for (int i = 0; i < jobs.count; ++i) total += jobs.job[i].blocks;
  ProfScope prof("prep", stream);
  hipLaunchKernelGGL(prep_batch_kernel, dim3((unsigned)total), dim3(256), 0, stream, jobs);
  return check_launch();
}


namespace {
__global__ void fill_words_kernel(uint32_t* __restrict__ dst, uint32_t value, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = value;
}
}  // namespace

int launch_fill_words(void* dst, uint32_t value, int64_t n_words, hipStream_t stream) {
  if (n_words <= 0) return SE3_OK;
  int64_t blocks = (n_words + 1023) / 1024;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_words_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (uint32_t*)dst, value, n_words);
  return check_launch();
}

}  // namespace se3
