// Fused operator stage (split-bf16): edge phase + dense contraction in one launch, the aggregated
// basis tensor T never leaves the CU.
//
//   out[row, o] = alpha * sum_{c,k} T[row, c, k] * W'[(c,k), o],     T[row, c, k] = sum_n feat[q(n), c] * phi(n)[k]
//
// One 512-thread workgroup per CU walks tiles of 8 items (= 16 output rows for 2 frames per item):
//   phase A  wavefront w runs the edge body (edge_bf16_body.h) for item 8*tile + w; its two rows of T
//            (2 x 2048 packed words for 64 channels) go to an LDS tile [16][2048] instead of HBM
//            (optionally also to HBM: the weight gradient of the training path needs T);
//   phase B  the 8 wavefronts contract the tile with the pre-split weight planes Bt[o][(c,k)] on
//            v_mfma_f32_16x16x32_bf16 (3 products per multiply): wavefront w owns output columns
//            16*(w&3).. and half (w>>2) of the 2048-long k range; the two halves meet in LDS.
// Used for the forward (feat = features, W' = W) and for the feature gradient on the transposed graph
// (feat = grad_out, W'[(o,k), i] = W[i,k,o]); replaces edge_t + gemm_nn and their 2 x 1.07 GB round trip.
//
// The weights (0.5 MB as two bf16 planes) are re-read from L2 once per tile: 16 rows is what fits in LDS,
// so that stream (4.3 GB of L2 traffic per launch at the headline shape) is the price of the fusion.
#include <cstdlib>

#include "common.h"
#include "edge_bf16_body.h"

namespace se3 {

namespace {

using f32x4v = __attribute__((__vector_size__(4 * sizeof(float)))) float;

constexpr int kTileRows = 16;               // rows of T per tile
constexpr int kCg = 64;                     // gathered channels this kernel is built for
constexpr int kRowWords = kCg * kBasis;     // 2048
constexpr int kPitch = kRowWords + 4;       // +16 B: the 16 rows of an A-fragment read start 4 banks apart

__device__ __forceinline__ f32x4v mfma16_bf16(u32x4 a, u32x4 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool SAVE_T>
__global__ __launch_bounds__(512, 2) void conv_fused_bf16_kernel(EdgeGeom g, const uint32_t* __restrict__ feat,
                                                                 int64_t feat_rows, const float* __restrict__ axes_ext,
                                                                 const float* __restrict__ rho_p,
                                                                 const uint16_t* __restrict__ bt_hi,
                                                                 const uint16_t* __restrict__ bt_lo, int co,
                                                                 float* __restrict__ out, uint32_t* __restrict__ t_save,
                                                                 const float* __restrict__ alpha_num, float alpha_scale,
                                                                 int64_t n_items, int fnb_shift) {
  __shared__ __attribute__((aligned(16))) uint32_t tile[kTileRows][kPitch];
  __shared__ __attribute__((aligned(16))) uint32_t lds_w[2][2][64][4];
  __shared__ __attribute__((aligned(16))) float red[4][16][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 64) mlp_weights_to_lds<2>(lds_w, axes_ext, threadIdx.x);
  __syncthreads();
  const float rho = *rho_p;
  const float alpha = (alpha_num ? *alpha_num : 1.0f) * alpha_scale;
  const __amdgpu_buffer_rsrc_t feat_rs = buffer_of(feat, feat_rows * kCg * 4);
  const int64_t n_rows = n_items * 2;
  const int64_t n_tiles = (n_items + 7) / 8;
  const int kp = kRowWords;  // weight-plane pitch (k already a multiple of 32)
  const int m_l = lane & 15, g2 = lane >> 4;

  for (int64_t tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x) {
    // ---- phase A: one item per wavefront, rows 2*wave and 2*wave+1 of the tile ----------------------
    const int64_t item = tile_id * 8 + wave;
    if (item < n_items) {
      uint32_t* rows_lds = &tile[2 * wave][0];
      uint32_t* rows_hbm = SAVE_T ? t_save + item * 2 * (int64_t)kRowWords : nullptr;
      edge_item_bf16<2, 2, true>(g, feat_rs, kCg, lds_w, rho, item, fnb_shift, [&](int a, int off, uint32_t w) {
        rows_lds[a * kPitch + off] = w;
        if (SAVE_T) rows_hbm[a * kRowWords + off] = w;
      });
    }
    __syncthreads();

    // ---- phase B: out_tile[16, co] = tile[16, 2048] @ W'[2048, co] -----------------------------------
    const int kh = wave >> 2;
    for (int ct0 = 0; ct0 * 16 < co; ct0 += 4) {
      const int col = (ct0 + (wave & 3)) * 16 + m_l;  // output column of this lane's B fragment / results
      const bool col_ok = col < co;
      const uint16_t* bh = bt_hi + (int64_t)(col_ok ? col : 0) * kp + kh * (kRowWords / 2) + 8 * g2;
      const uint16_t* bl = bt_lo + (int64_t)(col_ok ? col : 0) * kp + kh * (kRowWords / 2) + 8 * g2;
      const uint32_t* arow = &tile[m_l][kh * (kRowWords / 2) + 8 * g2];
      f32x4v acc = {0.f, 0.f, 0.f, 0.f};
      constexpr int U = 8;  // weight fragments fetched U k-steps ahead of their use
#pragma unroll 1
      for (int ks0 = 0; ks0 < (kRowWords / 2) / 32; ks0 += U) {
        u32x4 bhi[U], blo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          bhi[u] = *reinterpret_cast<const u32x4*>(bh + (ks0 + u) * 32);
          blo[u] = *reinterpret_cast<const u32x4*>(bl + (ks0 + u) * 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const u32x4 w0 = *reinterpret_cast<const u32x4*>(arow + (ks0 + u) * 32);
          const u32x4 w1 = *reinterpret_cast<const u32x4*>(arow + (ks0 + u) * 32 + 4);
          const u32x4 a_hi = {pair_hi(w0[0], w0[1]), pair_hi(w0[2], w0[3]), pair_hi(w1[0], w1[1]), pair_hi(w1[2], w1[3])};
          const u32x4 a_lo = {pair_lo(w0[0], w0[1]), pair_lo(w0[2], w0[3]), pair_lo(w1[0], w1[1]), pair_lo(w1[2], w1[3])};
          acc = mfma16_bf16(a_lo, bhi[u], acc);
          acc = mfma16_bf16(a_hi, blo[u], acc);
          acc = mfma16_bf16(a_hi, bhi[u], acc);
        }
      }
      // acc register r, lane (m_l = column, g2) = partial out[row 4*g2 + r][col]
      if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave & 3][4 * g2 + r][m_l] = acc[r];
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = tile_id * kTileRows + 4 * g2 + r;
          if (row < n_rows && col_ok) out[row * co + col] = alpha * (acc[r] + red[wave & 3][4 * g2 + r][m_l]);
        }
      }
      __syncthreads();  // red[] and (after the last column group) the tile are free again
    }
  }
}

}  // namespace

// Measured on MI355X at the headline shape (N=65536, k=32, F=2, C=64): fused 1.02-1.16 ms vs 0.82 ms for
// edge_t + gemm_nn.  With 64 channels only 16 rows of T fit in LDS, so every tile re-streams the whole
// 0.5 MB of weight planes from L2 (4.3 GB per launch, the L1 fill path runs at 64 B/clk/CU) and the eight
// wavefronts idle at the tile barriers behind the slowest item; that costs more than the 2 x 1.07 GB HBM
// round trip it removes.  Kept (and tested: SE3CONV_FUSED=1) as the starting point for a variant with
// >= 64 rows per weight pass; off by default.
// Small levels (where launches are latency, not throughput) do not favour it either: at 2 652 / 254 output rows the
// fused launch takes 32 / 28 us against 15 + 20 / 9 + 16 us for edge kernel + GEMM (its k loop over the 2048-deep weight
// planes is one serial chain per tile; the GEMM splits k over the grid), profiles/r02_levels_fused.txt.
// SE3CONV_FUSED=1 forces the fused kernel at every size, SE3CONV_FUSED_ROWS=n uses it up to n output rows (default 0).
constexpr int64_t kFusedMaxRows = 0;
bool conv_fused_bf16_supported(const EdgeGeom& g, int gathered_channels) {
  static const int64_t max_rows = [] {
    const char* e = getenv("SE3CONV_FUSED");
    if (e && e[0] == '1') return (int64_t)1 << 62;
    if (e && e[0] == '0') return (int64_t)0;
    const char* r = getenv("SE3CONV_FUSED_ROWS");
    return r ? (int64_t)atoll(r) : kFusedMaxRows;
  }();
  return g.n_ctr * g.f_ctr <= max_rows && gathered_channels == kCg && g.f_ctr % 2 == 0;
}

int launch_conv_fused_bf16(const char* tag, const EdgeGeom& g, const uint32_t* feat, int64_t feat_rows,
                           const float* axes_ext, const float* rho, const uint16_t* bt_hi, const uint16_t* bt_lo,
                           int co, float* out, uint32_t* t_save, const float* alpha_num, float alpha_scale,
                           hipStream_t stream) {
  const int64_t rows = g.n_ctr * g.f_ctr;
  if (rows == 0) return SE3_OK;
  ProfScope prof(tag, stream);
  const int64_t items = rows / 2;
  const int64_t tiles = (items + 7) / 8;
  int shift = -1;
  for (int sft = 0; sft < 8; ++sft)
    if ((1 << sft) == g.f_nb) shift = sft;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return SE3_ERR_LAUNCH;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const dim3 grid((unsigned)(tiles < n_cu ? tiles : n_cu)), block(512);
  if (t_save)
    hipLaunchKernelGGL(conv_fused_bf16_kernel<true>, grid, block, 0, stream, g, feat, feat_rows, axes_ext, rho, bt_hi,
                       bt_lo, co, out, t_save, alpha_num, alpha_scale, items, shift);
  else
    hipLaunchKernelGGL(conv_fused_bf16_kernel<false>, grid, block, 0, stream, g, feat, feat_rows, axes_ext, rho, bt_hi,
                       bt_lo, co, out, t_save, alpha_num, alpha_scale, items, shift);
  return check_launch();
}

}  // namespace se3
