// fp32 MFMA GEMMs for the dense contractions of the operator (gfx950).
//
//   gemm_nn : C[M,N]  = alpha * A[M,K]  @ B[K,N]       M = rows of the cloud (1e5..1e6), N = channels
//   gemm_tn : C[Ka,N] = alpha * A[M,Ka]^T @ B[M,N]     reduction over the rows (weight gradient)
//
// Both use v_mfma_f32_32x32x2_f32 (exact fp32, k-ordered fma chain), one 32x64 output strip per
// wavefront.  alpha = (*alpha_num) * alpha_scale so that the module's device-resident normaliser
// (norm_num_neighs_) never has to be read back to the host.
#include "common.h"

namespace se3 {

namespace {

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int AS_LD = BK + 1;  // +1: lanes (rows) r and r+1 land on different banks for the A-operand read

__device__ __forceinline__ float4 ld4_guard(const float* p, int64_t avail, bool vec) {
  // up to 4 consecutive floats starting at p, `avail` of them readable (<= 0: none)
  if (avail >= 4 && vec) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (avail > 0) v.x = p[0];
  if (avail > 1) v.y = p[1];
  if (avail > 2) v.z = p[2];
  if (avail > 3) v.w = p[3];
  return v;
}

__global__ __launch_bounds__(256) void gemm_nn_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ c, int64_t m, int n, int k,
                                                      const float* __restrict__ alpha_num, float alpha_scale) {
  __shared__ float as[2][BM][AS_LD];
  __shared__ __attribute__((aligned(16))) float bs[2][BK][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const bool a_vec = (k % 4) == 0, b_vec = (n % 4) == 0;

  float4 ra[4], rb[2];
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = p * 32 + (tid >> 3), kq = (tid & 7) * 4;
      const int64_t gr = m0 + row;
      ra[p] = gr < m ? ld4_guard(a + gr * k + k0 + kq, k - (k0 + kq), a_vec) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int kr = p * 16 + (tid >> 4), nq = (tid & 15) * 4;
      rb[p] = (k0 + kr) < k ? ld4_guard(b + (int64_t)(k0 + kr) * n + n0 + nq, n - (n0 + nq), b_vec)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = p * 32 + (tid >> 3), kq = (tid & 7) * 4;
      as[buf][row][kq + 0] = ra[p].x, as[buf][row][kq + 1] = ra[p].y;
      as[buf][row][kq + 2] = ra[p].z, as[buf][row][kq + 3] = ra[p].w;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int kr = p * 16 + (tid >> 4), nq = (tid & 15) * 4;
      *reinterpret_cast<float4*>(&bs[buf][kr][nq]) = rb[p];
    }
  };

  f32x16 acc0 = zero16(), acc1 = zero16();
  const int nk = (k + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) * BK);
#pragma unroll
    for (int t = 0; t < BK / 2; ++t) {
      const float av = as[buf][wave * 32 + rl][2 * t + h];
      const float b0 = bs[buf][2 * t + h][rl], b1 = bs[buf][2 * t + h][32 + rl];
      acc0 = mfma32(av, b0, acc0);
      acc1 = mfma32(av, b1, acc1);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  const float alpha = (alpha_num ? *alpha_num : 1.0f) * alpha_scale;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t gr = m0 + wave * 32 + acc_row(r, h);
    if (gr < m) {
      const int gc = n0 + rl;
      if (gc < n) c[gr * n + gc] = alpha * acc0[r];
      if (gc + 32 < n) c[gr * n + gc + 32] = alpha * acc1[r];
    }
  }
}

// One block: 128 (ka) x 64 (n) outputs over rows [split*chunk, (split+1)*chunk).  Operands are read
// from global directly in MFMA layout: half-wave h reads 32 consecutive floats of row m+h.
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ partials, int64_t m, int ka, int n,
                                                      int64_t chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rl = lane & 31, h = lane >> 5;
  const int ka0 = blockIdx.x * 128 + wave * 32;
  const int n0 = blockIdx.y * BN;
  if (ka0 >= ka) return;
  const int64_t mb = (int64_t)blockIdx.z * chunk;
  const int64_t me = min(m, mb + chunk);
  const bool a_ok = ka0 + rl < ka, b0_ok = n0 + rl < n, b1_ok = n0 + 32 + rl < n;
  const float* ap = a + (a_ok ? ka0 + rl : 0);
  const float* bp0 = b + (b0_ok ? n0 + rl : 0);
  const float* bp1 = b + (b1_ok ? n0 + 32 + rl : 0);
  f32x16 acc0 = zero16(), acc1 = zero16();
  constexpr int U = 8;
  for (int64_t mm = mb; mm < me; mm += 2 * U) {
    float av[U], bv0[U], bv1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = mm + 2 * u + h;
      const bool ok = row < me;
      const int64_t rr = ok ? row : mb;
      av[u] = ok && a_ok ? ap[rr * ka] : 0.f;
      bv0[u] = ok && b0_ok ? bp0[rr * n] : 0.f;
      bv1[u] = ok && b1_ok ? bp1[rr * n] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc0 = mfma32(av[u], bv0[u], acc0);
      acc1 = mfma32(av[u], bv1[u], acc1);
    }
  }
  float* out = partials + (int64_t)blockIdx.z * ka * n;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ka0 + acc_row(r, h);
    if (row < ka) {
      if (b0_ok) out[(int64_t)row * n + n0 + rl] = acc0[r];
      if (b1_ok) out[(int64_t)row * n + n0 + 32 + rl] = acc1[r];
    }
  }
}

// ---- round-3 forms of the three products for the shapes the operator's full-resolution levels have -------------------
// What the first versions above lose (fp32 mode of bench.py, headline level, v_mfma_f32_32x32x2_f32 peak = 157 TFLOP/s):
//   gemm_nn      59 % of peak: 50 KB of LDS = 3 workgroups per CU, so 1024 row tiles take two rounds, the second a third full;
//   gemm_tn      38 %: every batch of operand loads is waited for before its MFMAs start (no load in flight under them);
//   grad_T (nn)  45 %: 32 768 workgroups of two k-tiles each, prologue + epilogue per 4 096 MFMA cycles.
// All three read through buffer resources (out-of-range rows / columns return 0: no guards, no branches around loads).

constexpr int FK = 16;  // k-tile of gemm_nn_fast_kernel: 25 KB of LDS per workgroup -> up to 6 workgroups per CU
// tiles_per_split > 0: grid.z splits the k loop (levels whose row tiles cannot fill the chip: a block's k loop is serial,
// 128 tiles of 1 024 MFMA cycles whatever M is); block z writes its unscaled partial product to c + z * m * n.
__global__ __launch_bounds__(256) void gemm_nn_fast_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ c, int64_t m, int n, int k,
                                                           int tiles_per_split, const float* __restrict__ alpha_num,
                                                           float alpha_scale) {
  __shared__ float as[2][BM][FK + 1];
  __shared__ __attribute__((aligned(16))) float bs[2][FK][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;
  const __amdgpu_buffer_rsrc_t a_rs = buffer_of(a, m * k * 4);
  const __amdgpu_buffer_rsrc_t b_rs = buffer_of(b, (int64_t)k * n * 4);
  // A tile: 128 rows x 16 k = two passes of 64 rows (4 threads x 16 B per row); B tile: 16 k x 64 columns, one pass
  const int ar = tid >> 2, akq = (tid & 3) * 4, bkr = tid >> 4, bnq = (tid & 15) * 4;
  const uint32_t a_off0 = (uint32_t)(m0 + ar) * (uint32_t)k * 4u + akq * 4u, a_off1 = a_off0 + 64u * (uint32_t)k * 4u;
  const bool b_ok = n0 + bnq < n;  // n % 4 == 0 (host)
  const uint32_t b_off = ((uint32_t)bkr * (uint32_t)n + n0 + bnq) * 4u;
  u32x4 ra0, ra1, rb;
  auto load_tile = [&](int k0) {  // k % 16 == 0 (host): a tile never straddles the row end
    ra0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_off0 + k0 * 4u, 0, 0));
    ra1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_off1 + k0 * 4u, 0, 0));
    rb = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_ok ? b_off + (uint32_t)k0 * (uint32_t)n * 4u : (uint32_t)kOobOffset, 0, 0));
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      as[buf][ar][akq + i] = __uint_as_float(ra0[i]);
      as[buf][64 + ar][akq + i] = __uint_as_float(ra1[i]);
    }
    *reinterpret_cast<u32x4*>(&bs[buf][bkr][bnq]) = rb;
  };
  f32x16 acc0 = zero16(), acc1 = zero16();
  const int kt0 = tiles_per_split > 0 ? blockIdx.z * tiles_per_split : 0;
  const int nk = tiles_per_split > 0 ? min(k / FK, kt0 + tiles_per_split) : k / FK;  // > kt0 by construction
  load_tile(kt0 * FK);
  store_tile(0);
  __syncthreads();
  for (int kt = kt0; kt < nk; ++kt) {
    const int buf = (kt - kt0) & 1;
    load_tile(kt + 1 < nk ? (kt + 1) * FK : 0);  // unconditional (the last one re-reads tile 0 and is dropped)
#pragma unroll
    for (int t = 0; t < FK / 2; ++t) {
      const float av = as[buf][wave * 32 + rl][2 * t + h];
      const float b0 = bs[buf][2 * t + h][rl], b1 = bs[buf][2 * t + h][32 + rl];
      acc0 = mfma32(av, b0, acc0);
      acc1 = mfma32(av, b1, acc1);
    }
    store_tile(buf ^ 1);
    __syncthreads();
  }
  const float alpha = tiles_per_split > 0 ? 1.0f : (alpha_num ? *alpha_num : 1.0f) * alpha_scale;
  if (tiles_per_split > 0) c += (int64_t)blockIdx.z * m * n;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t gr = m0 + wave * 32 + acc_row(r, h);
    if (gr < m) {
      const int gc = n0 + rl;
      if (gc < n) c[gr * n + gc] = alpha * acc0[r];
      if (gc + 32 < n) c[gr * n + gc + 32] = alpha * acc1[r];
    }
  }
}

// C[M, N] = alpha * A[M, 2*KH] @ B[2*KH, N] for a short k (grad_T: k = C_out) and a wide N: a wavefront keeps its 32 rows of
// A in registers as the MFMA A operand of every column tile (KH registers), the workgroup walks its share of the 64-column
// tiles of B through a double-buffered LDS tile, and the stores of one tile run under the MFMAs of the next.
template <int KH>
__global__ __launch_bounds__(256) void gemm_nn_strip_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            float* __restrict__ c, int64_t m, int n, int tiles_per_block,
                                                            const float* __restrict__ alpha_num, float alpha_scale) {
  constexpr int K = 2 * KH;
  __shared__ __attribute__((aligned(16))) float bs[2][K][BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 31, h = lane >> 5;
  const int64_t row = (int64_t)blockIdx.x * BM + wave * 32 + rl;
  const __amdgpu_buffer_rsrc_t a_rs = buffer_of(a, m * K * 4);
  const __amdgpu_buffer_rsrc_t b_rs = buffer_of(b, (int64_t)K * n * 4);
  float av[KH];  // A[row][2 t + h]
#pragma unroll
  for (int j = 0; j < K / 4; ++j) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b128(a_rs, row < m ? (uint32_t)row * (K * 4u) + 16u * j : (uint32_t)kOobOffset, 0, 0);
    av[2 * j] = __uint_as_float(h ? v[1] : v[0]);
    av[2 * j + 1] = __uint_as_float(h ? v[3] : v[2]);
  }
  const int t_begin = blockIdx.y * tiles_per_block;
  const int t_end = min(n / BN, t_begin + tiles_per_block);
  // B tile: K rows x 64 columns = K * 16 pieces of 16 bytes, K / 16 per thread
  constexpr int NP = K / 16;
  const int bkr = tid >> 4, bnq = (tid & 15) * 4;
  u32x4 rb[NP];
  auto load_tile = [&](int t) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
      rb[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            b_rs, ((uint32_t)(16 * p + bkr) * (uint32_t)n + (uint32_t)t * BN + bnq) * 4u, 0, 0));
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(&bs[buf][16 * p + bkr][bnq]) = rb[p];
  };
  const float alpha = (alpha_num ? *alpha_num : 1.0f) * alpha_scale;
  if (t_begin >= t_end) return;
  load_tile(t_begin);
  store_tile(0);
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    load_tile(t + 1 < t_end ? t + 1 : t_begin);  // unconditional
    f32x16 acc0 = zero16(), acc1 = zero16();
#pragma unroll
    for (int kk = 0; kk < KH; ++kk) {
      const float b0 = bs[buf][2 * kk + h][rl], b1 = bs[buf][2 * kk + h][32 + rl];
      acc0 = mfma32(av[kk], b0, acc0);
      acc1 = mfma32(av[kk], b1, acc1);
    }
    store_tile(buf ^ 1);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t gr = (int64_t)blockIdx.x * BM + wave * 32 + acc_row(r, h);
      if (gr < m) {
        float* dst = c + gr * n + (int64_t)t * BN + rl;
        __builtin_nontemporal_store(alpha * acc0[r], dst);  // read once, by the parameter-gradient kernel behind this launch
        __builtin_nontemporal_store(alpha * acc1[r], dst + 32);
      }
    }
  }
}

// gemm_tn with two operand batches in registers: the loads of one batch are in flight under the MFMAs of the other.
// Rows past the end of the range read beyond a buffer that ends there (zeros): no row guards.
__global__ __launch_bounds__(256) void gemm_tn_fast_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ partials, int64_t m, int ka, int n,
                                                           int64_t chunk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rl = lane & 31, h = lane >> 5;
  const int ka0 = blockIdx.x * 128 + wave * 32;
  const int n0 = blockIdx.y * BN;
  if (ka0 >= ka) return;
  const int64_t mb = (int64_t)blockIdx.z * chunk;
  const int64_t me = min(m, mb + chunk);
  const bool a_ok = ka0 + rl < ka, b0_ok = n0 + rl < n, b1_ok = n0 + 32 + rl < n;
  const __amdgpu_buffer_rsrc_t a_rs = buffer_of(a, me * ka * 4);
  const __amdgpu_buffer_rsrc_t b_rs = buffer_of(b, me * n * 4);
  const uint32_t a_col = (uint32_t)(ka0 + rl) * 4u, b_col = (uint32_t)(n0 + rl) * 4u;
  constexpr int U = 8;
  struct Batch { float av[U], bv0[U], bv1[U]; };
  auto load = [&](Batch& t, int64_t mm) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t row = (uint32_t)(mm + 2 * u + h);
      t.av[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(a_rs, a_ok ? row * (uint32_t)ka * 4u + a_col : (uint32_t)kOobOffset, 0, 0));
      t.bv0[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, b0_ok ? row * (uint32_t)n * 4u + b_col : (uint32_t)kOobOffset, 0, 0));
      t.bv1[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(b_rs, b1_ok ? row * (uint32_t)n * 4u + b_col + 128u : (uint32_t)kOobOffset, 0, 0));
    }
  };
  f32x16 acc0 = zero16(), acc1 = zero16();
  auto mfmas = [&](const Batch& t) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc0 = mfma32(t.av[u], t.bv0[u], acc0);
      acc1 = mfma32(t.av[u], t.bv1[u], acc1);
    }
  };
  Batch t0, t1;
  load(t0, mb);
  for (int64_t mm = mb; mm < me; mm += 4 * U) {
    load(t1, mm + 2 * U);
    mfmas(t0);
    load(t0, mm + 4 * U);
    mfmas(t1);
  }
  float* out = partials + (int64_t)blockIdx.z * ka * n;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = ka0 + acc_row(r, h);
    if (row < ka) {
      if (b0_ok) out[(int64_t)row * n + n0 + rl] = acc0[r];
      if (b1_ok) out[(int64_t)row * n + n0 + 32 + rl] = acc1[r];
    }
  }
}

__global__ void reduce_partials_kernel(const float* __restrict__ partials, float* __restrict__ out, int64_t count,
                                       int splits, const float* __restrict__ alpha_num, float alpha_scale) {
  const float alpha = (alpha_num ? *alpha_num : 1.0f) * alpha_scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < splits; ++p) s += partials[(int64_t)p * count + i];
    out[i] = alpha * s;
  }
}

// The same sum for MANY partials of a SMALL result (the weight gradient of a point-wise linear layer: 512 row ranges of a
// 128 x 64 matrix): one thread per output would add 512 values one after the other on 32 workgroups (90 us); here a block
// owns 32 consecutive outputs, its 8 groups of 32 threads take every 8th partial (128-byte reads), and LDS folds the groups.
__global__ __launch_bounds__(256) void reduce_partials_grouped_kernel(const float* __restrict__ partials, float* __restrict__ out,
                                                                      int64_t count, int splits,
                                                                      const float* __restrict__ alpha_num, float alpha_scale) {
  __shared__ float red[8][32];
  const int ol = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + ol;
  float s = 0.f;
  if (i < count)
    for (int p = grp; p < splits; p += 8) s += partials[(int64_t)p * count + i];
  red[grp][ol] = s;
  __syncthreads();
  if (grp == 0 && i < count) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += red[g][ol];  // fixed order
    out[i] = (alpha_num ? *alpha_num : 1.0f) * alpha_scale * t;
  }
}

}  // namespace

// k splits of gemm_nn_fast_kernel: only when the row tiles leave most of the chip idle; at most ~1024 workgroups, at least
// 8 k-tiles per split (the partials cost their traffic and one reduction launch)
int gemm_nn_splits(int64_t m, int n, int k) {
  const int64_t tiles = ((m + BM - 1) / BM) * ((n + BN - 1) / BN);
  const int nk = k / FK;
  if (tiles < 1 || tiles >= 256 || k % FK != 0 || n % 4 != 0 || nk < 16) return 1;  // (tiles = 0: an empty cloud)
  int64_t s = 1024 / tiles;
  if (s > nk / 8) s = nk / 8;
  return (int)(s < 1 ? 1 : s);
}

size_t gemm_nn_split_bytes(int64_t m, int n, int k) {
  const int s = gemm_nn_splits(m, n, k);
  return s > 1 ? (size_t)s * m * n * 4 : 0;
}

int launch_gemm_nn(const char* tag, const float* a, const float* b, float* c, int64_t m, int n, int k,
                   const float* alpha_num, float alpha_scale, hipStream_t stream, float* split_ws) {
  if (m == 0 || n == 0) return SE3_OK;
  ProfScope prof(tag, stream);
  const int64_t row_blocks = (m + BM - 1) / BM;
  // 32-bit byte offsets of the buffer-load forms; kOobOffset must lie beyond both operands
  const bool small = (m + BM) * (int64_t)k * 4 < (int64_t)kOobOffset && (int64_t)k * n * 4 < (int64_t)kOobOffset;
  if (small && (k == 32 || k == 64) && n % BN == 0 && n >= 8 * BN) {
    // short k, wide n (grad_T): row strips; the column tiles are split over grid.y only as far as one resident round needs
    const int n_tiles = n / BN;
    int n_split = row_blocks >= 1024 ? 1 : (int)(1024 / row_blocks);
    if (n_split > n_tiles / 4) n_split = n_tiles / 4;
    const int per = (n_tiles + n_split - 1) / n_split;
    const dim3 grid((unsigned)row_blocks, (unsigned)((n_tiles + per - 1) / per));
    if (k == 64)
      hipLaunchKernelGGL(gemm_nn_strip_kernel<32>, grid, dim3(256), 0, stream, a, b, c, m, n, per, alpha_num, alpha_scale);
    else
      hipLaunchKernelGGL(gemm_nn_strip_kernel<16>, grid, dim3(256), 0, stream, a, b, c, m, n, per, alpha_num, alpha_scale);
    return check_launch();
  }
  const dim3 grid((unsigned)row_blocks, (unsigned)((n + BN - 1) / BN));
  const int splits = small && split_ws ? gemm_nn_splits(m, n, k) : 1;
  if (splits > 1) {
    const int per = (k / FK + splits - 1) / splits;
    const dim3 sgrid(grid.x, grid.y, (unsigned)((k / FK + per - 1) / per));
    hipLaunchKernelGGL(gemm_nn_fast_kernel, sgrid, dim3(256), 0, stream, a, b, split_ws, m, n, k, per, alpha_num, alpha_scale);
    return launch_reduce_partials(split_ws, c, m * n, (int)sgrid.z, alpha_num, alpha_scale, stream);
  }
  if (small && k % FK == 0 && n % 4 == 0)
    hipLaunchKernelGGL(gemm_nn_fast_kernel, grid, dim3(256), 0, stream, a, b, c, m, n, k, 0, alpha_num, alpha_scale);
  else
    hipLaunchKernelGGL(gemm_nn_kernel, grid, dim3(256), 0, stream, a, b, c, m, n, k, alpha_num, alpha_scale);
  return check_launch();
}

// Row ranges of the weight-gradient GEMM: about 512 workgroups in all (two per CU).  More, shorter ranges cost more in
// partial-result traffic and per-block prologue than they gain in parallelism (headline shape: 128 ranges of
// 1024 rows 0.27 ms, 32 ranges of 4096 rows 0.22 ms, 16 ranges 0.24 ms); never less than 256 rows per range.
int gemm_tn_splits(int64_t m, int ka, int n) {
  const int64_t tiles = (int64_t)((ka + 127) / 128) * ((n + BN - 1) / BN);
  int64_t s = 512 / tiles;  // at most 512 workgroups (two per CU): one resident round
  if (s < 1) s = 1;         // wide layers (more than 512 output tiles, e.g. 512 -> 256 channels): one range per tile
  // rows beyond the reach of one launch's 32-bit operand offsets are walked as several groups of ranges (gemm_bf16.hip):
  // a round's worth of ranges for every group
  const int64_t reach_rows = ((1ll << 32) - 64) / ((int64_t)(ka > n ? ka : n) * 4);
  if (m > reach_rows) s *= (4 * m + 3 * reach_rows - 1) / (3 * reach_rows);
  const int64_t max_s = (m + 255) / 256;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;  // (m = 0)
  // a launch of the buffer-load TN kernels must reach at least one range plus the rows it prefetches past it:
  // 2 * chunk * widest < 2^32 (launch_gemm_tn_bf16's `reach`), chunk = ceil(m / s) rounded up to a 32-row stage --
  // applied last, so that neither the 512-workgroup target nor the 256-row floor can leave a range out of reach
  // (c_in * K = 16 384 columns: 32 k rows per range)
  const int64_t max_chunk = reach_rows / 2 - 64;
  if (max_chunk > 0 && (m + s - 1) / s > max_chunk) s = (m + max_chunk - 1) / max_chunk;
  if (s < 1) s = 1;
  return (int)s;
}

int launch_gemm_tn(const char* tag, const float* a, const float* b, float* c, float* partials, int splits, int64_t m,
                   int ka, int n, const float* alpha_num, float alpha_scale, hipStream_t stream) {
  if (ka == 0 || n == 0) return SE3_OK;
  ProfScope prof(tag, stream);
  int64_t chunk = (m + splits - 1) / splits;
  chunk += chunk & 1;  // keep every split's first row even so the (m, m+1) pairing never straddles splits
  if (chunk == 0) chunk = 2;
  const dim3 grid((unsigned)((ka + 127) / 128), (unsigned)((n + BN - 1) / BN), (unsigned)splits);
  // the buffer-load form addresses both operands with 32-bit byte offsets (rows up to one batch past the end)
  if ((m + 64) * (int64_t)ka * 4 < (int64_t)kOobOffset && (m + 64) * (int64_t)n * 4 < (int64_t)kOobOffset)
    hipLaunchKernelGGL(gemm_tn_fast_kernel, grid, dim3(256), 0, stream, a, b, partials, m, ka, n, chunk);
  else
    hipLaunchKernelGGL(gemm_tn_kernel, grid, dim3(256), 0, stream, a, b, partials, m, ka, n, chunk);
  return launch_reduce_partials(partials, c, (int64_t)ka * n, splits, alpha_num, alpha_scale, stream);
}

int launch_reduce_partials(const float* partials, float* out, int64_t count, int splits, const float* alpha_num,
                           float alpha_scale, hipStream_t stream) {
  if (splits >= 64 && count <= (1 << 16)) {
    hipLaunchKernelGGL(reduce_partials_grouped_kernel, dim3((unsigned)((count + 31) / 32)), dim3(256), 0, stream, partials, out,
                       count, splits, alpha_num, alpha_scale);
    return check_launch();
  }
  const int rb = (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(rb), dim3(256), 0, stream, partials, out, count, splits, alpha_num,
                     alpha_scale);
  return check_launch();
}

}  // namespace se3
