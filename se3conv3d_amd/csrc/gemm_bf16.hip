// Split-bf16 GEMMs (v_mfma_f32_32x32x16_bf16, 3 products per multiply, fp32 accumulate).
//
//   gemm_nn_bf16 : C[M,N]  = alpha * A[M,K] @ B[K,N]     A = packed words (hi<<16|lo), row-major;
//                  B = the layer's weights, pre-split by prep_weights_kernel into two bf16 planes
//                  stored transposed, Bt[N][Kp] (Kp = K rounded up to 32, zero padded), so that an
//                  MFMA B fragment (8 consecutive k of one column) is one 16-byte LDS read.
//   gemm_tn_bf16 : C[Ka,N] = alpha * A[M,Ka]^T @ B[M,N]  both packed words; reduction over the rows
//                  (weight gradient).  The k-strided operands go through LDS and are read by column.
//
// These are streaming kernels: A (1 GB at the headline shape) is read once, so they are HBM-bound
// as long as the MFMA side reaches ~1/3 of its peak.
#include <cstdlib>

#include <atomic>

#include "common.h"

namespace se3 {

namespace {

#ifndef SE3_GEMM_ABLATE
#define SE3_GEMM_ABLATE 0  // diagnostic builds: 1 no MFMA stage, 2 no LDS staging, 4 no barriers
#endif
#ifndef SE3_T16_ABLATE
#define SE3_T16_ABLATE 0  // diagnostic builds of gemm_nn_t16_kernel (wrong results): 1 no decode, 2 no weight loads, 4 no MFMA stage
#endif
#ifndef SE3_GEMM_DEPTH
#define SE3_GEMM_DEPTH 4
#endif
constexpr int BM = 128, BN = 64, BK = 32;
constexpr int A_LD = BK + 4;   // words; 144-byte pitch keeps 16-byte alignment and spreads ds_read_b128 over all banks
constexpr int B_LD = BK + 8;   // bf16;  80-byte pitch, same properties

// up to 4 consecutive words starting at p, `avail` of them readable (<= 0: none).  Native vector
// type on purpose: HIP's uint4 struct made the register tiles of the pipeline land in scratch.
__device__ __forceinline__ u32x4 ld4_words(const uint32_t* p, int64_t avail, bool vec) {
  if (avail >= 4 && vec) return *reinterpret_cast<const u32x4*>(p);
  const uint32_t x = avail > 0 ? p[0] : 0u, y = avail > 1 ? p[1] : 0u, z = avail > 2 ? p[2] : 0u,
                 w = avail > 3 ? p[3] : 0u;
  return u32x4{x, y, z, w};
}

// OUT_MODE 0: fp32 C * alpha, 1: packed words of C * alpha, 2: raw fp32 partial of split z (no alpha).
// The A stream is prefetched DEPTH k-tiles ahead in registers: one k-tile of compute (~400 cycles) is
// far shorter than an HBM round trip, so a single tile in flight leaves the kernel latency-bound.
// FAST (k % 32 == 0, operands < 4 GB): every load is an unconditional raw buffer load whose out-of-range
// lanes return 0 -- no branches around the loads, so the compiler keeps all three tiles in flight (with
// guarded loads it drained vmcnt(0) after every tile and the kernel ran at half the HBM rate).
// NB = 64-column blocks per workgroup: 1 for N <= 64; 2 (128 columns, 4 column tiles per wavefront) for wider
// outputs, where the products turn MFMA-bound (24 instead of 12 MFMAs per A fragment set and k-tile).
template <int OUT_MODE, bool FAST, int NB>
__global__ __launch_bounds__(256) void gemm_nn_bf16_kernel(const uint32_t* __restrict__ a,
                                                           const uint16_t* __restrict__ bt_hi,
                                                           const uint16_t* __restrict__ bt_lo, void* __restrict__ c,
                                                           int64_t m, int n, int k, int kp, int kt_per_split,
                                                           const float* __restrict__ alpha_num, float alpha_scale) {
  __shared__ __attribute__((aligned(16))) uint32_t as[2][BM][A_LD];
  constexpr int BNW = BN * NB;  // columns per workgroup
  __shared__ __attribute__((aligned(16))) uint16_t bsh[2][BNW][B_LD];
  __shared__ __attribute__((aligned(16))) uint16_t bsl[2][BNW][B_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 31, h = lane >> 5;
  // (row blocks last-to-first -- A was written front-to-back just before, its tail is still in the memory-side cache --
  // was measured: no difference, 0.245 ms either way)
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BNW;
  const bool a_vec = (k % 4) == 0;
  const int kt_begin = blockIdx.z * kt_per_split;
  const int nk = min(kp / BK - kt_begin, kt_per_split);  // k-tiles of this block (> 0 by construction)

  struct Tile { u32x4 a0, a1, a2, a3, bh[NB], bl[NB]; };
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint32_t*>(a), (short)0, FAST ? (int)(uint32_t)(m * k * 4) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t bh_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_hi), (short)0, FAST ? (int)(uint32_t)((int64_t)n * kp * 2) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t bl_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_lo), (short)0, FAST ? (int)(uint32_t)((int64_t)n * kp * 2) : 0, 0x00020000);
  // Every block walks its k-tiles from a different starting phase (the sum is order-independent up to
  // rounding): with a common phase all resident blocks read the same 128-byte column strip of their
  // 8 KB rows at the same time, which concentrates the traffic on a few HBM channels.
  const int phase = FAST ? (int)((blockIdx.x * 5u) % (unsigned)nk) : 0;
  auto load_tile = [&](Tile& t, int kt_seq) {
    int kt = kt_seq + phase;
    if (kt >= nk) kt -= nk;
    const int k0 = (kt_begin + kt) * BK;
    const int row = tid >> 3, kq = (tid & 7) * 4;
    if constexpr (FAST) {
      const uint32_t aoff = (uint32_t)(((m0 + row) * k + k0 + kq) * 4);  // rows >= m fall outside the buffer -> 0
      const uint32_t rstep = (uint32_t)k * 32u * 4u;
      t.a0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, aoff, 0, 0));
      t.a1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, aoff + rstep, 0, 0));
      t.a2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, aoff + 2 * rstep, 0, 0));
      t.a3 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, aoff + 3 * rstep, 0, 0));
#if SE3_GEMM_ABLATE & 8
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) t.bh[nb] = t.bl[nb] = zero4;
      return;
#endif
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const uint32_t boff = (uint32_t)((((int64_t)(n0 + 64 * nb + (tid >> 2))) * kp + k0 + (tid & 3) * 8) * 2);
        t.bh[nb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bh_rs, boff, 0, 0));
        t.bl[nb] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bl_rs, boff, 0, 0));
      }
      return;
    }
    const uint32_t* ap = a + (m0 + row) * k + k0 + kq;
    const int64_t avail = k - (k0 + kq);
    t.a0 = ld4_words(ap, m0 + row < m ? avail : 0, a_vec);
    t.a1 = ld4_words(ap + (int64_t)32 * k, m0 + row + 32 < m ? avail : 0, a_vec);
    t.a2 = ld4_words(ap + (int64_t)64 * k, m0 + row + 64 < m ? avail : 0, a_vec);
    t.a3 = ld4_words(ap + (int64_t)96 * k, m0 + row + 96 < m ? avail : 0, a_vec);
    const int kq8 = (tid & 3) * 8;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int nl = 64 * nb + (tid >> 2);
      const bool ok = n0 + nl < n;
      const int64_t boff = (int64_t)(ok ? n0 + nl : 0) * kp + k0 + kq8;
      const u32x4 vh = *reinterpret_cast<const u32x4*>(bt_hi + boff), vl = *reinterpret_cast<const u32x4*>(bt_lo + boff);
      t.bh[nb] = ok ? vh : zero4;
      t.bl[nb] = ok ? vl : zero4;
    }
  };
  f32x16 acc[2 * NB];
#pragma unroll
  for (int c = 0; c < 2 * NB; ++c) acc[c] = zero16();
  auto store_tile = [&](const Tile& t, int buf) {
#if SE3_GEMM_ABLATE & 2
    acc[0][0] += __uint_as_float(t.a0[0] ^ t.a1[1] ^ t.a2[2] ^ t.a3[3] ^ t.bh[0][0] ^ t.bl[0][1]);
    return;
#endif
    const int row = tid >> 3, kq = (tid & 7) * 4;
    *reinterpret_cast<u32x4*>(&as[buf][row][kq]) = t.a0;
    *reinterpret_cast<u32x4*>(&as[buf][row + 32][kq]) = t.a1;
    *reinterpret_cast<u32x4*>(&as[buf][row + 64][kq]) = t.a2;
    *reinterpret_cast<u32x4*>(&as[buf][row + 96][kq]) = t.a3;
    const int kq8 = (tid & 3) * 8;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      *reinterpret_cast<u32x4*>(&bsh[buf][64 * nb + (tid >> 2)][kq8]) = t.bh[nb];
      *reinterpret_cast<u32x4*>(&bsl[buf][64 * nb + (tid >> 2)][kq8]) = t.bl[nb];
    }
  };

  auto compute = [&](int buf) {
#if SE3_GEMM_ABLATE & 1
    return;
#endif
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kk = 16 * s + 8 * h;
      const u32x4 w0 = *reinterpret_cast<const u32x4*>(&as[buf][wave * 32 + rl][kk]);
      const u32x4 w1 = *reinterpret_cast<const u32x4*>(&as[buf][wave * 32 + rl][kk + 4]);
      const u32x4 a_hi = {pair_hi(w0[0], w0[1]), pair_hi(w0[2], w0[3]), pair_hi(w1[0], w1[1]), pair_hi(w1[2], w1[3])};
      const u32x4 a_lo = {pair_lo(w0[0], w0[1]), pair_lo(w0[2], w0[3]), pair_lo(w1[0], w1[1]), pair_lo(w1[2], w1[3])};
#pragma unroll
      for (int c = 0; c < 2 * NB; ++c) {
        const u32x4 bh = *reinterpret_cast<const u32x4*>(&bsh[buf][32 * c + rl][kk]);
        const u32x4 bl = *reinterpret_cast<const u32x4*>(&bsl[buf][32 * c + rl][kk]);
        acc[c] = mfma_bf16x3(a_hi, a_lo, bh, bl, acc[c]);
      }
    }
  };
  // k-tile kt lives in register set (kt mod DEPTH) until it is written to LDS buffer (kt & 1).  The kernel's
  // occupancy is set by its LDS tiles (2 blocks per CU), which leaves 256 VGPRs per wavefront: they hold DEPTH
  // tiles in flight.  Measured on MI355X: 3, 4, 6 and 8 tiles give the same time, and so does the kernel with
  // everything but the A loads removed (SE3_GEMM_ABLATE=15): 0.245 ms for 1.07 GB = 4.4 TB/s, while the same walk
  // over a buffer that was not just written by the previous kernel reads at 5.8-6.3 TB/s (tools/probes/): the
  // producer's dirty lines are still draining from L2 / the memory-side cache into HBM while this kernel reads.
  constexpr int DEPTH = SE3_GEMM_DEPTH;
  static_assert(DEPTH % 2 == 0, "the LDS buffer parity of a step must be a compile-time constant");
  Tile t[DEPTH];
#pragma unroll
  for (int u = 0; u < DEPTH; ++u)
    if (u < nk) load_tile(t[u], u);
  store_tile(t[0], 0);
  __syncthreads();
  // steady state without any condition around the loads (conditional loads make the compiler drain
  // vmcnt(0) every step), then a checked tail
  int kt0 = 0;
  for (; kt0 + 2 * DEPTH <= nk; kt0 += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      load_tile(t[u], kt0 + u + DEPTH);
      compute(u & 1);
      store_tile(t[(u + 1) % DEPTH], (u & 1) ^ 1);
#if !(SE3_GEMM_ABLATE & 4)
      __syncthreads();
#endif
    }
  }
  for (; kt0 < nk; kt0 += DEPTH) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) {
      const int kt = kt0 + u;
      if (kt < nk) {
        if (kt + DEPTH < nk) load_tile(t[u], kt + DEPTH);
        compute(u & 1);
        if (kt + 1 < nk) store_tile(t[(u + 1) % DEPTH], (u & 1) ^ 1);
        __syncthreads();
      }
    }
  }

  const float alpha = OUT_MODE == 2 ? 1.0f : (alpha_num ? *alpha_num : 1.0f) * alpha_scale;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t gr = m0 + wave * 32 + acc_row(r, h);
    if (gr < m) {
#pragma unroll
      for (int ct = 0; ct < 2 * NB; ++ct) {
        const int gc = n0 + 32 * ct + rl;
        if (gc >= n) continue;
        if constexpr (OUT_MODE == 1) {
          static_cast<uint32_t*>(c)[gr * n + gc] = split_pack(alpha * acc[ct][r]);
        } else {
          float* out = static_cast<float*>(c) + (OUT_MODE == 2 ? (int64_t)blockIdx.z * m * n : 0);
          out[gr * n + gc] = alpha * acc[ct][r];
        }
      }
    }
  }
}

// NN GEMM over A rows in the 3-byte format (common.h, T24; k a multiple of 64).  What bounds the A stream of these
// products is the number of 64-byte requests, not the bytes (tools/probes/tile_read3.hip: 6144-byte rows read 64 + 32
// bytes at a time take as long as 8192-byte rows read 128 bytes at a time), so the loads work on "super tiles" of 64
// k: 128 B of hi and 64 B of lo per row and instruction.  LDS and the MFMA stage keep the 32-k tiles of the kernel
// above: the two halves of a super tile are the two LDS buffers.  Lane l of load p reads 16-byte column
// (l & 7) ^ (4 * (p & 1)), so every thread holds as many pieces of either half and the stores stay full-width.
// (Measured and not kept: row blocks last-written first -- no difference; the row stream loaded non-temporally -- slower.)
// (256-row workgroups -- 8 wavefronts, the weight planes fetched from L2 half as often -- were measured in round 3:
// nothing at the headline shape, 15 % slower on a 150 k-row scene whose 586 workgroups no longer divide into full rounds:
// profiles/r03_gemm_256row_ab.txt.)
// KG = k groups per workgroup (round 5): 1 -- four wavefronts walk the workgroup's super tiles -- or 2 -- eight wavefronts,
// group g walks half of them with LDS tiles of its own and the two partial tiles are added through LDS at the end.  For
// products whose row blocks cannot fill the chip (levels of 4 k - 40 k rows: fewer workgroups than CUs x 2): a workgroup
// alone on its CU is bound by its own chain of loads, barriers and MFMAs (~0.75 us per 32-k tile whatever is in flight), so
// two chains side by side halve its time without the partial-sum round trip and the reduction launch a split over grid.z costs.
template <int OUT_MODE, int NB, int KG = 1>
__global__ __launch_bounds__(256 * KG) void gemm_nn_t24_kernel(const uint8_t* __restrict__ a,
                                                          const uint16_t* __restrict__ bt_hi,
                                                          const uint16_t* __restrict__ bt_lo, void* __restrict__ c,
                                                          int64_t m, int n, int k, int st_per_split,
                                                          const float* __restrict__ alpha_num, float alpha_scale) {
  constexpr int BNW = BN * NB;
  constexpr int RB = BM;           // rows per workgroup
  constexpr int RP = 32;           // rows one load pass of the hi plane covers (8 threads per row): 4 passes = RB rows
  constexpr int RPL = 64;          // rows one load pass of the lo plane covers (4 threads per row): 2 passes = RB rows
  constexpr int CP = 32;           // weight columns one load pass covers: 2 * NB passes per plane
  constexpr int NBP = 2 * NB;
  // one block of LDS, carved: per k group the four tile images; the partial tile of group 1 reuses group 0's images at the end
  constexpr int kAsh = 2 * RB * (BK + 8) * 2, kAsl = 2 * RB * (BK + 16), kBs = 2 * BNW * B_LD * 2;
  constexpr int kGroup = kAsh + kAsl + 2 * kBs;
  static_assert(KG == 1 || RB * BNW * 4 <= kGroup, "the partial tile must fit the tile images it reuses");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // KG * kGroup bytes (gemm_nn_t24_lds_bytes): 52 / 104 / 72 / 144 KB
  const int grp = KG == 1 ? 0 : (int)(threadIdx.x >> 8);
  auto ash = reinterpret_cast<uint16_t(*)[RB][BK + 8]>(smem + grp * kGroup);                    // [2][RB][BK + 8], 80-byte pitch
  auto asl = reinterpret_cast<uint8_t(*)[RB][BK + 16]>(smem + grp * kGroup + kAsh);             // [2][RB][BK + 16], 48-byte pitch
  auto bsh = reinterpret_cast<uint16_t(*)[BNW][B_LD]>(smem + grp * kGroup + kAsh + kAsl);       // [2][BNW][B_LD]
  auto bsl = reinterpret_cast<uint16_t(*)[BNW][B_LD]>(smem + grp * kGroup + kAsh + kAsl + kBs);
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;  // thread / wavefront inside the k group
  const int rl = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * RB;
  const int n0 = blockIdx.y * BNW;
  const int ns_wg = min(k / 64 - (int)blockIdx.z * st_per_split, st_per_split);  // super tiles of this workgroup (> 0 by construction)
  const int ns = KG == 1 ? ns_wg : ns_wg / 2;                                    // ... of this k group (the host keeps ns_wg even)
  const int st_begin = blockIdx.z * st_per_split + grp * ns;

  struct Super { u32x4 ah[4], al[2], bh[NBP], bl[NBP]; };
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(a), (short)0, (int)(uint32_t)(m * k * 3), 0x00020000);
  const __amdgpu_buffer_rsrc_t bh_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_hi), (short)0, (int)(uint32_t)((int64_t)n * k * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t bl_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_lo), (short)0, (int)(uint32_t)((int64_t)n * k * 2), 0x00020000);
  const int phase = (int)((blockIdx.x * 5u) % (unsigned)ns);  // see gemm_nn_bf16_kernel
  const uint32_t rb = (uint32_t)k * 3u;
  const int hsel = (tid >> 2) & 1, lsel = (tid >> 1) & 1;
  auto load_super = [&](Super& t, int seq) {
    int st = seq + phase;
    if (st >= ns) st -= ns;
    const uint32_t k0 = (uint32_t)(st_begin + st) * 64u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const uint32_t off = (uint32_t)(m0 + p * RP + (tid >> 3)) * rb + k0 * 2u + (uint32_t)((tid & 7) ^ ((p & 1) << 2)) * 16u;
      t.ah[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, off, 0, 0));
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const uint32_t off = (uint32_t)(m0 + p * RPL + (tid >> 2)) * rb + (uint32_t)k * 2u + k0 + (uint32_t)((tid & 3) ^ (p << 1)) * 16u;
      t.al[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, off, 0, 0));
    }
#pragma unroll
    for (int j = 0; j < NBP; ++j) {  // pass j: columns CP * j + (tid >> 3); passes alternate which 32-k half a thread holds
      const uint32_t off = ((uint32_t)(n0 + CP * j + (tid >> 3)) * (uint32_t)k + k0 + (uint32_t)((tid & 7) ^ ((j & 1) << 2)) * 8u) * 2u;
      t.bh[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bh_rs, off, 0, 0));
      t.bl[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bl_rs, off, 0, 0));
    }
  };
  // the 32-k half `hf` of a super tile -> LDS buffer `buf`: pieces whose column bit 2 (hi, B) / bit 1 (lo) equals hf
  auto store_half = [&](const Super& t, int hf, int buf) {
    const bool q = (hf ^ hsel) != 0, ql = (hf ^ lsel) != 0;
    const int r8 = tid >> 3, c4 = (tid & 3) * 8;
    *reinterpret_cast<u32x4*>(&ash[buf][(q ? RP : 0) + r8][c4]) = q ? t.ah[1] : t.ah[0];
    *reinterpret_cast<u32x4*>(&ash[buf][(q ? 3 * RP : 2 * RP) + r8][c4]) = q ? t.ah[3] : t.ah[2];
    *reinterpret_cast<u32x4*>(&asl[buf][(ql ? RPL : 0) + (tid >> 2)][(tid & 1) * 16]) = ql ? t.al[1] : t.al[0];
    // weight passes come in pairs (2 g, 2 g + 1): the even pass holds this thread's half-0 piece iff q == 0
#pragma unroll
    for (int g = 0; g < NBP / 2; ++g) {
      *reinterpret_cast<u32x4*>(&bsh[buf][CP * (2 * g) + (q ? CP : 0) + r8][c4]) = q ? t.bh[2 * g + 1] : t.bh[2 * g];
      *reinterpret_cast<u32x4*>(&bsl[buf][CP * (2 * g) + (q ? CP : 0) + r8][c4]) = q ? t.bl[2 * g + 1] : t.bl[2 * g];
    }
  };
  f32x16 acc[2 * NB];
#pragma unroll
  for (int ct = 0; ct < 2 * NB; ++ct) acc[ct] = zero16();
  auto compute = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kk = 16 * s + 8 * h;
      // the hi fragment is one 16-byte read; the lo fragment is rebuilt from the 8 lo bytes (7 VALU per pair)
      const u32x4 a_hi = *reinterpret_cast<const u32x4*>(&ash[buf][wave * 32 + rl][kk]);
      const uint32_t* lp = reinterpret_cast<const uint32_t*>(&asl[buf][wave * 32 + rl][kk]);
      const uint32_t l0 = lp[0], l1 = lp[1];
      const u32x4 a_lo = {t24_lo_word<0>(a_hi[0], l0), t24_lo_word<2>(a_hi[1], l0), t24_lo_word<0>(a_hi[2], l1),
                          t24_lo_word<2>(a_hi[3], l1)};
#pragma unroll
      for (int ct = 0; ct < 2 * NB; ++ct) {
        const u32x4 bh = *reinterpret_cast<const u32x4*>(&bsh[buf][32 * ct + rl][kk]);
        const u32x4 bl = *reinterpret_cast<const u32x4*>(&bsl[buf][32 * ct + rl][kk]);
        acc[ct] = mfma_bf16x3(a_hi, a_lo, bh, bl, acc[ct]);
      }
    }
  };
  // Two super tiles in registers (= the 4 k-tiles in flight of the kernel above).  Step A computes the first half
  // while the second goes to LDS buffer 1; step B computes the second half, stores the next super tile's first half
  // to buffer 0 and refills the register set that just emptied.  (KG = 2: both k groups run the same number of steps, so
  // the workgroup barriers line up.)
  Super t0, t1;
  load_super(t0, 0);
  if (ns > 1) load_super(t1, 1);
  store_half(t0, 0, 0);
  __syncthreads();
#define SE3_T24_STEP(CUR, NEXT, ST)                       \
  compute(0);                                             \
  store_half(CUR, 1, 1);                                  \
  __syncthreads();                                        \
  if ((ST) + 2 < ns) load_super(CUR, (ST) + 2);           \
  compute(1);                                             \
  if ((ST) + 1 < ns) store_half(NEXT, 0, 0);              \
  __syncthreads();
#define SE3_T24_STEP_FULL(CUR, NEXT, ST) \
  compute(0);                            \
  store_half(CUR, 1, 1);                 \
  __syncthreads();                       \
  load_super(CUR, (ST) + 2);             \
  compute(1);                            \
  store_half(NEXT, 0, 0);                \
  __syncthreads();
  int st = 0;
  const int prio_slot = wave_slot_id();
  for (; st + 4 <= ns; st += 2) {  // no conditions around the loads (see gemm_nn_bf16_kernel)
    rotate_priority(prio_slot + (st >> 1));
    SE3_T24_STEP_FULL(t0, t1, st)
    SE3_T24_STEP_FULL(t1, t0, st + 1)
  }
  for (; st < ns; st += 2) {
    SE3_T24_STEP(t0, t1, st)
    if (st + 1 < ns) {
      SE3_T24_STEP(t1, t0, st + 1)
    }
  }
#undef SE3_T24_STEP
#undef SE3_T24_STEP_FULL

  if constexpr (KG == 2) {
    // group 1's partial tile -> LDS (over the tile images, which nobody reads any more: the loop ended on a barrier),
    // group 0 adds it to its own.  [row][column] floats, column fastest: conflict-free both ways.
    float* red = reinterpret_cast<float*>(smem);
    if (grp == 1) {
#pragma unroll
      for (int ct = 0; ct < 2 * NB; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wave * 32 + acc_row(r, h)) * BNW + 32 * ct + rl] = acc[ct][r];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int ct = 0; ct < 2 * NB; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] += red[(wave * 32 + acc_row(r, h)) * BNW + 32 * ct + rl];
  }
  const float alpha = OUT_MODE == 2 ? 1.0f : (alpha_num ? *alpha_num : 1.0f) * alpha_scale;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t gr = m0 + wave * 32 + acc_row(r, h);
    if (gr < m) {
#pragma unroll
      for (int ct = 0; ct < 2 * NB; ++ct) {
        const int gc = n0 + 32 * ct + rl;
        if (gc >= n) continue;
        if constexpr (OUT_MODE == 1) {
          static_cast<uint32_t*>(c)[gr * n + gc] = split_pack(alpha * acc[ct][r]);
        } else {
          float* out = static_cast<float*>(c) + (OUT_MODE == 2 ? (int64_t)blockIdx.z * m * n : 0);
          out[gr * n + gc] = alpha * acc[ct][r];
        }
      }
    }
  }
}

constexpr int gemm_nn_t24_lds_bytes(int nb, int kg) {
  return kg * (2 * BM * (BK + 8) * 2 + 2 * BM * (BK + 16) + 2 * (2 * BN * nb * B_LD * 2));
}

// NN GEMM over A rows in the 2.25-byte block format (common.h, T16; k a multiple of 256 = one mega tile of the exponent
// plane).  Same skeleton as the 3-byte kernel above -- super tiles of 64 k, the two 32-k halves of a super tile are the
// two LDS buffers -- with these differences: the row stream is the mantissa plane (128 B per row and super tile, as the
// 3-byte format's hi plane) plus ONE 8-byte exponent load per thread and mega tile (4 super tiles; the 8 threads of a
// row read one 64-byte line); a piece (8 mantissas = 2 blocks) is decoded to its hi / lo bf16 planes on the way into
// LDS (t16_unpack2: 5.5 VALU per element, once per element of A -- the MFMA stage then reads both fragments as they
// lie, no rebuild); no per-block phase stagger (4608-byte rows do not alias onto a few channels as 8192-byte rows do).
template <int OUT_MODE, int NB>
__global__ __launch_bounds__(256) void gemm_nn_t16_kernel(const uint8_t* __restrict__ a,
                                                          const uint16_t* __restrict__ bt_hi,
                                                          const uint16_t* __restrict__ bt_lo, void* __restrict__ c,
                                                          int64_t m, int n, int k, int st_per_split,
                                                          const float* __restrict__ alpha_num, float alpha_scale) {
  constexpr int BNW = BN * NB;
  constexpr int RB = BM;           // rows per workgroup
  constexpr int RP = 32;           // rows one load pass covers (8 threads per row): 4 passes = RB rows
  constexpr int CP = 32;           // weight columns one load pass covers: 2 * NB passes per plane
  constexpr int NBP = 2 * NB;
  __shared__ __attribute__((aligned(16))) uint16_t ash[2][RB][BK + 8];   // 80-byte pitch
  __shared__ __attribute__((aligned(16))) uint16_t asl[2][RB][BK + 8];
  __shared__ __attribute__((aligned(16))) uint16_t bsh[2][BNW][B_LD];
  __shared__ __attribute__((aligned(16))) uint16_t bsl[2][BNW][B_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 31, h = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * RB;
  const int n0 = blockIdx.y * BNW;
  const int st_begin = blockIdx.z * st_per_split;          // a multiple of 4 (host)
  const int ns = min(k / 64 - st_begin, st_per_split);     // super tiles of this block (> 0, a multiple of 4)

  struct Super { u32x4 am[4], bh[NBP], bl[NBP]; };
  const uint32_t rb = (uint32_t)k / 32u * 72u;             // row bytes: 2 k of mantissas + k / 4 exponent bytes
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(a), (short)0, (int)(uint32_t)(m * rb), 0x00020000);
  const __amdgpu_buffer_rsrc_t bh_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_hi), (short)0, (int)(uint32_t)((int64_t)n * k * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t bl_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_lo), (short)0, (int)(uint32_t)((int64_t)n * k * 2), 0x00020000);
  const int hsel = (tid >> 2) & 1;
  // lane l of load pass p reads 16-byte column (l & 7) ^ (4 * (p & 1)) of its row (see gemm_nn_t24_kernel)
  auto load_super = [&](Super& t, int st) {
    const uint32_t k0 = (uint32_t)(st_begin + st) * 64u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const uint32_t off = (uint32_t)(m0 + p * RP + (tid >> 3)) * rb + k0 * 2u + (uint32_t)((tid & 7) ^ ((p & 1) << 2)) * 16u;
      t.am[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, off, 0, 0));
    }
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
#if SE3_T16_ABLATE & 2
      t.bh[j] = t.bl[j] = u32x4{k0, k0, k0, k0};
      continue;
#endif
      const uint32_t off = ((uint32_t)(n0 + CP * j + (tid >> 3)) * (uint32_t)k + k0 + (uint32_t)((tid & 7) ^ ((j & 1) << 2)) * 8u) * 2u;
      t.bh[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bh_rs, off, 0, 0));
      t.bl[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bl_rs, off, 0, 0));
    }
  };
  // exponents of this thread's pieces for the 4 super tiles of mega tile `mg` (relative to st_begin): pass p, piece
  // (tid & 7) ^ (4 (p & 1)) -> 8 bytes = [super tile s][block of the piece]
  auto load_exps = [&](u32x2 (&e)[4], int mg) {
    const uint32_t e0 = (uint32_t)k * 2u + (uint32_t)(st_begin / 4 + mg) * 64u;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const uint32_t off = (uint32_t)(m0 + p * RP + (tid >> 3)) * rb + e0 + (uint32_t)((tid & 7) ^ ((p & 1) << 2)) * 8u;
      e[p] = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(a_rs, off, 0, 0));
    }
  };
  // the 32-k half `hf` of super tile number s (0..3) of its mega tile -> LDS buffer `buf`
  auto store_half = [&](const Super& t, const u32x2 (&e)[4], int s, int hf, int buf) {
    const bool q = (hf ^ hsel) != 0;
    const int r8 = tid >> 3, c4 = (tid & 3) * 8;
#pragma unroll
    for (int g = 0; g < 2; ++g) {  // the two passes whose piece lies in this half: p = 2 g + q
      const u32x4 mw = q ? t.am[2 * g + 1] : t.am[2 * g];
      const u32x2 ev = q ? e[2 * g + 1] : e[2 * g];
      const uint32_t e2 = (s < 2 ? ev[0] : ev[1]) >> (16 * (s & 1));  // bytes 2 s, 2 s + 1 of the 8
      u32x4 vh, vl;
#if SE3_T16_ABLATE & 1
      vh = mw, vl = mw ^ u32x4{e2, e2, e2, e2};
#else
      t16_unpack8(mw, e2, vh, vl);
#endif
      const int row = (2 * g + (q ? 1 : 0)) * RP + r8;
      *reinterpret_cast<u32x4*>(&ash[buf][row][c4]) = vh;
      *reinterpret_cast<u32x4*>(&asl[buf][row][c4]) = vl;
    }
#pragma unroll
    for (int g = 0; g < NBP / 2; ++g) {
      *reinterpret_cast<u32x4*>(&bsh[buf][CP * (2 * g) + (q ? CP : 0) + r8][c4]) = q ? t.bh[2 * g + 1] : t.bh[2 * g];
      *reinterpret_cast<u32x4*>(&bsl[buf][CP * (2 * g) + (q ? CP : 0) + r8][c4]) = q ? t.bl[2 * g + 1] : t.bl[2 * g];
    }
  };
  f32x16 acc[2 * NB];
#pragma unroll
  for (int ct = 0; ct < 2 * NB; ++ct) acc[ct] = zero16();
  auto compute = [&](int buf) {
#if SE3_T16_ABLATE & 4
    return;
#endif
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int kk = 16 * s + 8 * h;
      const u32x4 a_hi = *reinterpret_cast<const u32x4*>(&ash[buf][wave * 32 + rl][kk]);
      const u32x4 a_lo = *reinterpret_cast<const u32x4*>(&asl[buf][wave * 32 + rl][kk]);
#pragma unroll
      for (int ct = 0; ct < 2 * NB; ++ct) {
        const u32x4 bh = *reinterpret_cast<const u32x4*>(&bsh[buf][32 * ct + rl][kk]);
        const u32x4 bl = *reinterpret_cast<const u32x4*>(&bsl[buf][32 * ct + rl][kk]);
        acc[ct] = mfma_bf16x3(a_hi, a_lo, bh, bl, acc[ct]);
      }
    }
  };
  // Mega tile = 4 super tiles, fully unrolled (the exponent bytes of super tile s sit at a compile-time position of the
  // thread's 8).  Two super tiles in registers; the next mega tile's exponents are requested during super tile 1 and are
  // first needed when super tile 3 hands over to the next mega tile's super tile 0.
  Super t0, t1;
  u32x2 ex[4], ex_next[4];
  load_exps(ex, 0);
  load_super(t0, 0);
  load_super(t1, 1);
  store_half(t0, ex, 0, 0, 0);
  __syncthreads();
  const int n_mega = ns / 4;
  for (int mg = 0; mg < n_mega; ++mg) {
    const int st = 4 * mg;
    const bool more = mg + 1 < n_mega;
    // super tile 0 (in t0)
    compute(0);
    store_half(t0, ex, 0, 1, 1);
    __syncthreads();
    load_super(t0, st + 2);
    compute(1);
    store_half(t1, ex, 1, 0, 0);
    __syncthreads();
    // super tile 1 (in t1)
    compute(0);
    store_half(t1, ex, 1, 1, 1);
    __syncthreads();
    load_super(t1, st + 3);
    if (more) load_exps(ex_next, mg + 1);
    compute(1);
    store_half(t0, ex, 2, 0, 0);
    __syncthreads();
    // super tile 2 (in t0)
    compute(0);
    store_half(t0, ex, 2, 1, 1);
    __syncthreads();
    if (more) load_super(t0, st + 4);
    compute(1);
    store_half(t1, ex, 3, 0, 0);
    __syncthreads();
    // super tile 3 (in t1)
    compute(0);
    store_half(t1, ex, 3, 1, 1);
    __syncthreads();
    if (more) load_super(t1, st + 5);
    compute(1);
    if (more) {
#pragma unroll
      for (int p = 0; p < 4; ++p) ex[p] = ex_next[p];
      store_half(t0, ex, 0, 0, 0);
    }
    __syncthreads();
  }

  const float alpha = OUT_MODE == 2 ? 1.0f : (alpha_num ? *alpha_num : 1.0f) * alpha_scale;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t gr = m0 + wave * 32 + acc_row(r, h);
    if (gr < m) {
#pragma unroll
      for (int ct = 0; ct < 2 * NB; ++ct) {
        const int gc = n0 + 32 * ct + rl;
        if (gc >= n) continue;
        if constexpr (OUT_MODE == 1) {
          static_cast<uint32_t*>(c)[gr * n + gc] = split_pack(alpha * acc[ct][r]);
        } else {
          float* out = static_cast<float*>(c) + (OUT_MODE == 2 ? (int64_t)blockIdx.z * m * n : 0);
          out[gr * n + gc] = alpha * acc[ct][r];
        }
      }
    }
  }
}

// Row-strip GEMM for the wide, write-dominated products with a short k (grad_T = g W^T, H = f W''):
//   C[m, n] (packed words) = A[m, k] (packed words) * Bt[n, k]^T,   k <= 64, n large (C_in*K or C_out*K = 2048)
// A wavefront keeps its 32 rows of A as MFMA fragments for the whole kernel and walks the n range 32 columns
// at a time: 3*KS MFMAs, then 16 results per lane are split to hi/lo and stored (128 contiguous bytes per
// row and half-wave).  The next tile's weight fragments (L1/L2-resident, 0.5 MB in all) are requested as soon
// as the MFMAs have issued and land during the epilogue.  Everything fits 128 VGPRs, so 4 wavefronts share a
// SIMD and the 4096 strips of the headline shape are resident at once (the previous 64-column version needed
// ~150 VGPRs: 3 per SIMD, i.e. a second, mostly empty round).  alpha is folded into the prepared weights.
#ifndef SE3_STRIP_ROT
#define SE3_STRIP_ROT 17
#endif
template <int KS>  // kp / 16: 2, 4 (k <= 64, 120 VGPRs) or 8 (k <= 128: twice the fragments, 3 waves per SIMD)
__global__ __launch_bounds__(256, KS <= 4 ? 4 : 2) void gemm_strip_bf16_kernel(const uint32_t* __restrict__ a,
                                                                 const uint16_t* __restrict__ bt_hi,
                                                                 const uint16_t* __restrict__ bt_lo,
                                                                 uint32_t* __restrict__ c, int64_t m, int n, int k) {
  constexpr int KP = KS * 16;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // uniform: feeds a buffer descriptor
  const int rl = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
  if (row0 >= m) return;
  const __amdgpu_buffer_rsrc_t a_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a), (short)0, (int)(uint32_t)(m * k * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t bh_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_hi), (short)0, (int)(uint32_t)((int64_t)n * KP * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t bl_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_lo), (short)0, (int)(uint32_t)((int64_t)n * KP * 2), 0x00020000);

  u32x4 a_hi[KS], a_lo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int kk = 16 * ks + 8 * h;
    uint32_t w[8];
    if (k % 8 == 0) {  // block-uniform: whole 8-word groups are inside or outside the row
      const uint32_t off = kk < k ? (uint32_t)(((row0 + rl) * k + kk) * 4) : 0xfffffff0u;
      const auto v0 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, off, 0, 0);
      const auto v1 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, kk < k ? off + 16 : 0xfffffff0u, 0, 0);
      w[0] = v0[0], w[1] = v0[1], w[2] = v0[2], w[3] = v0[3], w[4] = v1[0], w[5] = v1[1], w[6] = v1[2], w[7] = v1[3];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        w[j] = __builtin_amdgcn_raw_buffer_load_b32(
            a_rs, kk + j < k ? (uint32_t)(((row0 + rl) * k + kk + j) * 4) : 0xfffffff0u, 0, 0);
    }
    frags_from_words(w, a_hi[ks], a_lo[ks]);
  }

  // Weight fragments of the next 32 columns.  The loads are issued through inline asm and waited for with an
  // explicit counted s_waitcnt: vmcnt retires loads and stores in order on this chip, so "at most 16 outstanding"
  // after the 16 stores of a tile means the 2*KS older loads have landed while the stores are still in flight.
  // (The compiler's own wait insertion treats mixed loads/stores as unordered and emits vmcnt(0): every tile
  // then waits for its stores to retire, which is what bounded the first version of this kernel.)
  u32x4 bh[KS], bl[KS];
  auto load_b = [&](int n0) {
    const uint32_t off = (uint32_t)(((n0 / 32) * KS * 64 + lane) * 16);  // fragment-ordered planes, 1 KB per k-step
    const uint32_t off4 = off + 4096u;  // the instruction's immediate offset has 12 bits: k-steps 4.. go through here
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3"
                   : "+v"(bh[ks])
                   : "v"(ks < 4 ? off : off4), "s"(bh_rs), "n"(1024 * (ks & 3))
                   : "memory");
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3"
                   : "+v"(bl[ks])
                   : "v"(ks < 4 ? off : off4), "s"(bl_rs), "n"(1024 * (ks & 3))
                   : "memory");
    }
  };
  static_assert(KS == 2 || KS == 4 || KS == 8, "operand lists of SE3_WAIT_B");
#define SE3_WAIT_B(CNT)                                                                                           \
  do {                                                                                                            \
    if constexpr (KS == 2)                                                                                        \
      asm volatile("s_waitcnt vmcnt(" #CNT ")" : "+v"(bh[0]), "+v"(bh[1]), "+v"(bl[0]), "+v"(bl[1])::"memory");   \
    else if constexpr (KS == 4)                                                                                   \
      asm volatile("s_waitcnt vmcnt(" #CNT ")"                                                                    \
                   : "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[KS - 2]), "+v"(bh[KS - 1]), "+v"(bl[0]), "+v"(bl[1]),      \
                     "+v"(bl[KS - 2]), "+v"(bl[KS - 1])::"memory");                                               \
    else                                                                                                          \
      asm volatile("s_waitcnt vmcnt(" #CNT ")"                                                                    \
                   : "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[2]), "+v"(bh[3]), "+v"(bh[KS - 4]), "+v"(bh[KS - 3]),      \
                     "+v"(bh[KS - 2]), "+v"(bh[KS - 1]), "+v"(bl[0]), "+v"(bl[1]), "+v"(bl[2]), "+v"(bl[3]),      \
                     "+v"(bl[KS - 4]), "+v"(bl[KS - 3]), "+v"(bl[KS - 2]), "+v"(bl[KS - 1])::"memory");           \
  } while (0)
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bh[ks] = bl[ks] = u32x4{0u, 0u, 0u, 0u};
  // Output through a buffer descriptor based at the strip: rows >= m fall outside num_records and are dropped by
  // the hardware, the uniform part of every address ((row-of-register * n + n0) * 4) rides in the scalar offset, so
  // a store costs no VALU.  Straight-line loop body: the wait in front of the MFMAs must cover only the weight
  // loads, not the 16 younger stores (vmcnt counts both, in order) -- with per-store branches the compiler fell
  // back to vmcnt(0) and every tile waited for its stores to retire.
  const int64_t c_bytes = (m - row0) * (int64_t)n * 4;
  const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(
      c + row0 * n, (short)0, (int)(uint32_t)(c_bytes > 0xffffffffll ? 0xffffffffll : c_bytes), 0x00020000);
  const int voff = (4 * h * n + rl) * 4;
  const bool cols_full = n % 32 == 0;  // uniform
  // Column order rotated per strip: with a power-of-two row pitch (n * 4 = 8 KB) every wavefront of the chip would
  // otherwise write the same 128-byte column window of its rows at the same time, i.e. all traffic of a moment
  // lands on a handful of L2 / HBM channels.
  // blockIdx.y selects a contiguous range of column tiles (small M: the row strips alone cannot fill the chip)
  const int n_tiles_all = (n + 31) / 32;
  const int per = (n_tiles_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int tile_lo = (int)blockIdx.y * per;
  const int n_tiles = min(per, n_tiles_all - tile_lo);
  if (n_tiles <= 0) return;
  const int n_lo = tile_lo * 32, n_hi = min(n, (tile_lo + n_tiles) * 32);
  int n0 = n_lo + (int)(((blockIdx.x * 4 + wave) * (unsigned)SE3_STRIP_ROT) % (unsigned)n_tiles) * 32;
  load_b(n0);
  SE3_WAIT_B(0);
  // (rotating wave priority, common.h rotate_priority, was measured here too -- every strip is resident for the whole
  // launch -- and loses 2 %: the kernel is bound by its stores, not by issue slots; profiles/r06_rotate_priority_ab.txt)
  for (int it = 0; it < n_tiles; ++it) {
    f32x16 acc = zero16();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) acc = mfma_bf16x3(a_hi[ks], a_lo[ks], bh[ks], bl[ks], acc);
    const int n_next = n0 + 32 < n_hi ? n0 + 32 : n_lo;
    load_b(n_next);  // in flight during the epilogue below (after the last tile: one unused fetch)
    const int lane_off = (cols_full || n0 + rl < n) ? voff : (int)0x7fffff00;  // columns >= n: dropped
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      uint32_t w0, w1;
      split_pack2(acc[r], acc[r + 1], w0, w1);
#ifdef SE3_STRIP_NOSTORE
      if ((w0 ^ w1) != 0x12345678u) continue;
#endif
      __builtin_amdgcn_raw_buffer_store_b32(w0, c_rs, lane_off, (acc_row(r, 0) * n + n0) * 4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(w1, c_rs, lane_off, (acc_row(r + 1, 0) * n + n0) * 4, 0);
    }
    SE3_WAIT_B(16);  // the loads are older than the 16 stores
    n0 = n_next;
  }
#undef SE3_WAIT_B
}

// The row-strip GEMM writing C (= grad_T) in the T16 block format (common.h): 2.25 instead of 4 bytes per element leave
// the kernel, which is bound by its stores.  Column n of C is position n of the T16 row (the weights are prepared in that
// order), so a 32-column tile = one channel quad x 8 basis functions x 4 channels: a block (4 channels of one basis
// function) is four ADJACENT LANES of one accumulator register -- its maximum takes two DPP steps.  Mantissas and exponent
// bytes go through wave-private LDS staging so that every store instruction writes 16 bytes per lane: a tile's 32 rows x
// 64 B of mantissas as two stores, the 64 exponent bytes per row of a mega tile (8 column tiles) as two stores per mega tile
// (2 + 1/4 store instructions per tile; the packed-word kernel issues 16).
template <int KS>
__global__ __launch_bounds__(256, 4) void gemm_strip_t16_kernel(const uint32_t* __restrict__ a,
                                                                const uint16_t* __restrict__ bt_hi,
                                                                const uint16_t* __restrict__ bt_lo,
                                                                uint8_t* __restrict__ c, int64_t m, int n, int k) {
  constexpr int KP = KS * 16;
  __shared__ __attribute__((aligned(16))) uint16_t st_m[4][32][32 + 8];  // [wave][row][column], 80-byte pitch
  __shared__ __attribute__((aligned(16))) uint8_t st_e[4][32][64 + 16];  // [wave][row][byte of the mega tile's exponent line]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int rl = lane & 31, h = lane >> 5;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 32;
  if (row0 >= m) return;
  const __amdgpu_buffer_rsrc_t a_rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a), (short)0, (int)(uint32_t)(m * k * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t bh_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_hi), (short)0, (int)(uint32_t)((int64_t)n * KP * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t bl_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint16_t*>(bt_lo), (short)0, (int)(uint32_t)((int64_t)n * KP * 2), 0x00020000);
  u32x4 a_hi[KS], a_lo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int kk = 16 * ks + 8 * h;
    uint32_t w[8];
    const uint32_t off = kk < k ? (uint32_t)(((row0 + rl) * k + kk) * 4) : 0xfffffff0u;  // k % 8 == 0 (host)
    const auto v0 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, off, 0, 0);
    const auto v1 = __builtin_amdgcn_raw_buffer_load_b128(a_rs, kk < k ? off + 16 : 0xfffffff0u, 0, 0);
    w[0] = v0[0], w[1] = v0[1], w[2] = v0[2], w[3] = v0[3], w[4] = v1[0], w[5] = v1[1], w[6] = v1[2], w[7] = v1[3];
    frags_from_words(w, a_hi[ks], a_lo[ks]);
  }
  // weight fragments of the next tile through inline asm + counted waits, as in gemm_strip_bf16_kernel: vmcnt retires
  // loads and stores in order, so "at most S outstanding" behind the S stores of a tile means the older loads have landed
  static_assert(KS == 2 || KS == 4, "operand lists of SE3_WAIT_B16");
  u32x4 bh[KS], bl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bh[ks] = bl[ks] = u32x4{0u, 0u, 0u, 0u};
  auto load_b = [&](int tile) {
    const uint32_t off = (uint32_t)((tile * KS * 64 + lane) * 16);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "+v"(bh[ks]) : "v"(off), "s"(bh_rs), "n"(1024 * ks) : "memory");
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "+v"(bl[ks]) : "v"(off), "s"(bl_rs), "n"(1024 * ks) : "memory");
    }
  };
#define SE3_WAIT_B16(CNT)                                                                                          \
  do {                                                                                                             \
    if constexpr (KS == 2)                                                                                         \
      asm volatile("s_waitcnt vmcnt(" #CNT ")" : "+v"(bh[0]), "+v"(bh[1]), "+v"(bl[0]), "+v"(bl[1])::"memory");    \
    else                                                                                                           \
      asm volatile("s_waitcnt vmcnt(" #CNT ")"                                                                     \
                   : "+v"(bh[0]), "+v"(bh[1]), "+v"(bh[KS - 2]), "+v"(bh[KS - 1]), "+v"(bl[0]), "+v"(bl[1]),       \
                     "+v"(bl[KS - 2]), "+v"(bl[KS - 1])::"memory");                                                \
  } while (0)
  // rows of C: 72 bytes per channel (n = channels * 32 columns); rows >= m fall outside the buffer and are dropped
  const int64_t rb = (int64_t)n / 32 * 72;
  const int64_t c_bytes = (m - row0) * rb;
  const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(
      c + row0 * rb, (short)0, (int)(uint32_t)(c_bytes > 0xffffffffll ? 0xffffffffll : c_bytes), 0x00020000);
  // blockIdx.y selects a contiguous range of mega tiles; the mega tiles of a strip are walked from a rotated start
  const int n_mega_all = n / 256;
  const int per = (n_mega_all + (int)gridDim.y - 1) / (int)gridDim.y;
  const int mega_lo = (int)blockIdx.y * per;
  const int n_mega = min(per, n_mega_all - mega_lo);
  if (n_mega <= 0) return;
  int mg = mega_lo + (int)(((blockIdx.x * 4 + wave) * 5u) % (unsigned)n_mega);
  // store-side lane roles: lane l writes 16 bytes of row (l >> 2) + 16 i, piece l & 3
  const int s_row = lane >> 2, s_pc = lane & 3;
  load_b(mg * 8);
  SE3_WAIT_B16(0);
  for (int it = 0; it < n_mega; ++it) {
    const int mg_next = mg + 1 < mega_lo + n_mega ? mg + 1 : mega_lo;
#pragma unroll
    for (int tl = 0; tl < 8; ++tl) {
      const int tile = mg * 8 + tl;
      f32x16 acc = zero16();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = mfma_bf16x3(a_hi[ks], a_lo[ks], bh[ks], bl[ks], acc);
      load_b(tl < 7 ? tile + 1 : mg_next * 8);  // in flight during the epilogue (after the last tile: one unused fetch)
      // block = 4 adjacent lanes of one register: maximum by two DPP steps, exponent, mantissa
      const int blk = (tile & 7) * 8 + (rl >> 2);  // block index inside the mega tile: (channel quad & 1) * 32 + basis function
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float mx = fabsf(acc[r]);
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, mx), 0xb1, 0xf, 0xf, true)));  // quad_perm [1,0,3,2]
        mx = fmaxf(mx, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, mx), 0x4e, 0xf, 0xf, true)));  // quad_perm [2,3,0,1]
        int e = __builtin_amdgcn_frexp_expf(mx);
        e = e < -kT16ExpBias ? -kT16ExpBias : (e > 127 ? 127 : e);
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const s16x2 q = __builtin_amdgcn_cvt_pknorm_i16(__builtin_ldexpf(acc[r], -e), 0.f);
        st_m[wave][acc_row(r, h)][rl] = (uint16_t)q[0];
        if ((rl & 3) == 0) st_e[wave][acc_row(r, h)][t16_exp_pos(blk) & 63] = (uint8_t)(e + kT16ExpBias);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(&st_m[wave][s_row + 16 * i][s_pc * 8]);
        __builtin_amdgcn_raw_buffer_store_b128(v, c_rs, (int)((s_row + 16 * i) * rb + s_pc * 16), tile * 64, 0);
      }
      if (tl == 7) {  // the mega tile's exponent line of every row: 64 bytes
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const u32x4 v = *reinterpret_cast<const u32x4*>(&st_e[wave][s_row + 16 * i][s_pc * 16]);
          __builtin_amdgcn_raw_buffer_store_b128(v, c_rs, (int)((s_row + 16 * i) * rb + s_pc * 16), n * 2 + mg * 64, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the staging rows are free for the next tile
      if (tl == 7) SE3_WAIT_B16(4);     // the loads are older than this tile's 2 (+ 2 exponent) stores
      else SE3_WAIT_B16(2);
    }
    mg = mg_next;
  }
#undef SE3_WAIT_B16
}

// out = alpha * sum_z partials[z]  (fp32 or packed words)
template <bool OUT_PACKED>
__global__ void reduce_splits_kernel(const float* __restrict__ partials, void* __restrict__ out, int64_t count,
                                     int splits, const float* __restrict__ alpha_num, float alpha_scale) {
  const float alpha = (alpha_num ? *alpha_num : 1.0f) * alpha_scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
#pragma unroll 8
    for (int z = 0; z < splits; ++z) s += partials[(int64_t)z * count + i];
    if constexpr (OUT_PACKED) static_cast<uint32_t*>(out)[i] = split_pack(alpha * s);
    else static_cast<float*>(out)[i] = alpha * s;
  }
}

// One block: 128 (ka) x 64 (n) outputs over rows [split*chunk, (split+1)*chunk) in stages of 32 rows.
// FAST (n % 4 == 0, operands < 4 GB): unconditional raw buffer loads bounded at the split's last row (rows
// past it and columns past ka / n read as 0), stages prefetched 3 deep in registers like gemm_nn.
// AFMT 1: A rows in the 3-byte format (common.h, T24; ka a multiple of 64); the partial rows are written at t24_k_of().
// AFMT 2: A rows in the 2.25-byte block format (T16; ka a multiple of 128): a stage reads 256 B of mantissas per row
// and the two exponent bytes of every thread's piece, decodes to the same hi / lo 16-bit images (the lo plane is then
// the bf16 lo operand itself: no rebuild in the MFMA stage); partial rows at t16_k_of().
template <bool FAST, int AFMT>
__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const uint32_t* __restrict__ a,
                                                           const uint32_t* __restrict__ b,
                                                           float* __restrict__ partials, int64_t m, int ka, int n,
                                                           int64_t chunk) {
  constexpr bool A16 = AFMT == 2;
  constexpr bool A24 = AFMT != 0;  // "the A operand comes as two 16-bit planes": everything below that is not format-specific
  static_assert(!A24 || FAST, "the 3-byte / T16 A formats are only read through buffer loads");
  // A24: hi / lo planes of A (lo bytes widened to 16 bits) and of B as 16-bit images whose rows are the K index of
  // the MFMAs; the fragments come out of ds_read_b64_tr_b16 (common.h) ready-made: 12 LDS reads and no v_perm per
  // k-step where gathering them with scalar reads took 32 reads + 24 v_perm.  Row pitches of 16 (mod 64) dwords
  // keep the 4-row blocks of a half-wavefront on distinct banks.
  constexpr int APITCH = 128 + 32, BPITCH = BN + 32;  // 320- and 192-byte rows
  __shared__ __attribute__((aligned(16))) uint32_t at[2][A24 ? 1 : BK][128];
  __shared__ __attribute__((aligned(16))) uint16_t ath[2][A24 ? BK : 1][APITCH];
  __shared__ __attribute__((aligned(16))) uint16_t atl[2][A24 ? BK : 1][APITCH];
  __shared__ __attribute__((aligned(16))) uint32_t bt[2][A24 ? 1 : BK][BN];
  __shared__ __attribute__((aligned(16))) uint16_t bth[2][A24 ? BK : 1][BPITCH];
  __shared__ __attribute__((aligned(16))) uint16_t btl[2][A24 ? BK : 1][BPITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 31, h = lane >> 5;
  const int ka0 = blockIdx.x * 128;
  const int n0 = blockIdx.y * BN;
  const int64_t mb = (int64_t)blockIdx.z * chunk;
  const int64_t me = min(m, mb + chunk);
  const bool b_vec = (n % 4) == 0;
  const int64_t nst = me > mb ? (me - mb + BK - 1) / BK : 0;

  struct Stage { u32x4 a0, a1, a2, a3, b0, b1; };
  const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint32_t*>(a), (short)0, FAST ? (int)(uint32_t)(A16 ? me * (ka / 32 * 72) : me * ka * (A24 ? 3 : 4)) : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint32_t*>(b), (short)0, FAST ? (int)(uint32_t)(me * n * 4) : 0, 0x00020000);
  const int arow = tid >> 5, acq = (tid & 31) * 4, brow = tid >> 4, bcq = (tid & 15) * 4;
  const bool a_ok = ka0 + acq < ka, b_ok = n0 + bcq < n;
  auto load_stage = [&](Stage& t, int64_t st) {
    const int64_t mm = mb + st * BK;
    if constexpr (A16) {
      const uint32_t oob = 0xfffffff0u;
      const uint32_t rb = (uint32_t)ka / 32u * 72u;
      constexpr int kNtLoad = 2;
      // mantissas: 32 rows x 256 B = 2 pieces of 16 B per thread (rows tid >> 4 and + 16, piece tid & 15 = 8 mantissas =
      // blocks 2 pc, 2 pc + 1 of the 128 columns); their exponent bytes are adjacent in the plane (t16_exp_pos: bit 0)
      const bool h_ok = ka0 + (tid & 15) * 8 < ka;
      const uint32_t ho = (uint32_t)(mm + (tid >> 4)) * rb + (uint32_t)(ka0 + (tid & 15) * 8) * 2u;
      const uint32_t eo = (uint32_t)(mm + (tid >> 4)) * rb + (uint32_t)ka * 2u + (uint32_t)t16_exp_pos((ka0 + (tid & 15) * 8) >> 2);
      const uint32_t bo = b_ok ? (uint32_t)(((mm + brow) * n + n0 + bcq) * 4) : oob;
      const uint32_t bs16 = (uint32_t)n * 64u;
      t.a0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, h_ok ? ho : oob, 0, kNtLoad));
      t.a1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, h_ok ? ho + 16u * rb : oob, 0, kNtLoad));
      t.a2[0] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(a_rs, h_ok ? eo : oob, 0, kNtLoad);
      t.a2[1] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(a_rs, h_ok ? eo + 16u * rb : oob, 0, kNtLoad);
      t.b0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, bo, 0, 0));
      t.b1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_ok ? bo + bs16 : oob, 0, 0));
    } else if constexpr (A24) {
      const uint32_t oob = 0xfffffff0u;
      const uint32_t rb = (uint32_t)ka * 3u;
      // hi: 32 rows x 256 B = 2 pieces of 16 B per thread; lo: 32 rows x 128 B = 1 piece per thread.  The rows are
      // read once: non-temporal loads keep them from displacing the g tiles the 16 column blocks share in L2
      // (0.178 -> 0.170 ms; the same hint on the NN GEMM's A stream costs it 0.01 ms)
      constexpr int kNtLoad = 2;
      const bool h_ok = ka0 + (tid & 15) * 8 < ka, l_ok = ka0 + (tid & 7) * 16 < ka;
      const uint32_t ho = (uint32_t)(mm + (tid >> 4)) * rb + (uint32_t)(ka0 + (tid & 15) * 8) * 2u;
      const uint32_t lo = (uint32_t)(mm + (tid >> 3)) * rb + (uint32_t)ka * 2u + (uint32_t)(ka0 + (tid & 7) * 16);
      const uint32_t bo = b_ok ? (uint32_t)(((mm + brow) * n + n0 + bcq) * 4) : oob;
      const uint32_t bs16 = (uint32_t)n * 64u;
      t.a0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, h_ok ? ho : oob, 0, kNtLoad));
      t.a1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, h_ok ? ho + 16u * rb : oob, 0, kNtLoad));
      t.a2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, l_ok ? lo : oob, 0, kNtLoad));
      t.b0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, bo, 0, 0));
      t.b1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_ok ? bo + bs16 : oob, 0, 0));
    } else if constexpr (FAST) {
      const uint32_t oob = 0xfffffff0u;  // beyond num_records: the load returns 0
      const uint32_t ao = a_ok ? (uint32_t)(((mm + arow) * ka + ka0 + acq) * 4) : oob;
      const uint32_t bo = b_ok ? (uint32_t)(((mm + brow) * n + n0 + bcq) * 4) : oob;
      const uint32_t as8 = (uint32_t)ka * 32u, bs16 = (uint32_t)n * 64u;
      t.a0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, ao, 0, 0));
      t.a1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_ok ? ao + as8 : oob, 0, 0));
      t.a2 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_ok ? ao + 2 * as8 : oob, 0, 0));
      t.a3 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rs, a_ok ? ao + 3 * as8 : oob, 0, 0));
      t.b0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, bo, 0, 0));
      t.b1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rs, b_ok ? bo + bs16 : oob, 0, 0));
    } else {
      u32x4* ta[4] = {&t.a0, &t.a1, &t.a2, &t.a3};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int64_t gr = mm + p * 8 + arow;
        *ta[p] = ld4_words(a + gr * ka + ka0 + acq, gr < me ? ka - (ka0 + acq) : 0, true);
      }
      u32x4* tb[2] = {&t.b0, &t.b1};
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int64_t gr = mm + p * 16 + brow;
        *tb[p] = ld4_words(b + gr * n + n0 + bcq, gr < me ? n - (n0 + bcq) : 0, b_vec);
      }
    }
  };
  auto store_stage = [&](const Stage& t, int buf) {
    if constexpr (A16) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const u32x4 mw = p ? t.a1 : t.a0;
        const uint32_t e2 = t.a2[p];
        u32x4 vh, vl;
        t16_unpack8(mw, e2, vh, vl);
        *reinterpret_cast<u32x4*>(&ath[buf][16 * p + (tid >> 4)][(tid & 15) * 8]) = vh;
        *reinterpret_cast<u32x4*>(&atl[buf][16 * p + (tid >> 4)][(tid & 15) * 8]) = vl;
      }
      *reinterpret_cast<u32x2*>(&bth[buf][brow][bcq]) = u32x2{pair_hi(t.b0[0], t.b0[1]), pair_hi(t.b0[2], t.b0[3])};
      *reinterpret_cast<u32x2*>(&btl[buf][brow][bcq]) = u32x2{pair_lo(t.b0[0], t.b0[1]), pair_lo(t.b0[2], t.b0[3])};
      *reinterpret_cast<u32x2*>(&bth[buf][brow + 16][bcq]) = u32x2{pair_hi(t.b1[0], t.b1[1]), pair_hi(t.b1[2], t.b1[3])};
      *reinterpret_cast<u32x2*>(&btl[buf][brow + 16][bcq]) = u32x2{pair_lo(t.b1[0], t.b1[1]), pair_lo(t.b1[2], t.b1[3])};
    } else if constexpr (A24) {
      *reinterpret_cast<u32x4*>(&ath[buf][tid >> 4][(tid & 15) * 8]) = t.a0;
      *reinterpret_cast<u32x4*>(&ath[buf][16 + (tid >> 4)][(tid & 15) * 8]) = t.a1;
      u32x4 e0, e1;  // 16 lo bytes -> 16 half-words
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        e0[2 * i] = __builtin_amdgcn_perm(0u, t.a2[i], 0x0c010c00u), e0[2 * i + 1] = __builtin_amdgcn_perm(0u, t.a2[i], 0x0c030c02u);
        e1[2 * i] = __builtin_amdgcn_perm(0u, t.a2[2 + i], 0x0c010c00u), e1[2 * i + 1] = __builtin_amdgcn_perm(0u, t.a2[2 + i], 0x0c030c02u);
      }
      *reinterpret_cast<u32x4*>(&atl[buf][tid >> 3][(tid & 7) * 16]) = e0;
      *reinterpret_cast<u32x4*>(&atl[buf][tid >> 3][(tid & 7) * 16 + 8]) = e1;
      *reinterpret_cast<u32x2*>(&bth[buf][brow][bcq]) = u32x2{pair_hi(t.b0[0], t.b0[1]), pair_hi(t.b0[2], t.b0[3])};
      *reinterpret_cast<u32x2*>(&btl[buf][brow][bcq]) = u32x2{pair_lo(t.b0[0], t.b0[1]), pair_lo(t.b0[2], t.b0[3])};
      *reinterpret_cast<u32x2*>(&bth[buf][brow + 16][bcq]) = u32x2{pair_hi(t.b1[0], t.b1[1]), pair_hi(t.b1[2], t.b1[3])};
      *reinterpret_cast<u32x2*>(&btl[buf][brow + 16][bcq]) = u32x2{pair_lo(t.b1[0], t.b1[1]), pair_lo(t.b1[2], t.b1[3])};
    } else {
      *reinterpret_cast<u32x4*>(&at[buf][arow][acq]) = t.a0;
      *reinterpret_cast<u32x4*>(&at[buf][arow + 8][acq]) = t.a1;
      *reinterpret_cast<u32x4*>(&at[buf][arow + 16][acq]) = t.a2;
      *reinterpret_cast<u32x4*>(&at[buf][arow + 24][acq]) = t.a3;
      *reinterpret_cast<u32x4*>(&bt[buf][brow][bcq]) = t.b0;
      *reinterpret_cast<u32x4*>(&bt[buf][brow + 16][bcq]) = t.b1;
    }
  };

  f32x16 acc0 = zero16(), acc1 = zero16();
  auto compute = [&](int buf) {
#if SE3_GEMM_ABLATE & 1
    return;
#endif
    if constexpr (A24) {
      const int grp = lane >> 4, q = (lane >> 2) & 3, p4 = (lane & 3) * 4;
      const int col = 16 * (grp & 1) + p4;  // this lane's address: row q of the block, 4 columns from here
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int r0 = 16 * s + 8 * (grp >> 1) + q;
        const u32x4 a_hi = lds_frag_tr16(&ath[buf][r0][wave * 32 + col], &ath[buf][r0 + 4][wave * 32 + col]);
        const u32x4 lw = lds_frag_tr16(&atl[buf][r0][wave * 32 + col], &atl[buf][r0 + 4][wave * 32 + col]);
        u32x4 a_lo = lw;  // T16: the lo plane holds the bf16 lo operand itself
        if constexpr (!A16)
          a_lo = u32x4{t24_lo_word<0, 2>(a_hi[0], lw[0]), t24_lo_word<0, 2>(a_hi[1], lw[1]),
                       t24_lo_word<0, 2>(a_hi[2], lw[2]), t24_lo_word<0, 2>(a_hi[3], lw[3])};
        u32x4 b_hi = lds_frag_tr16(&bth[buf][r0][col], &bth[buf][r0 + 4][col]);
        u32x4 b_lo = lds_frag_tr16(&btl[buf][r0][col], &btl[buf][r0 + 4][col]);
        acc0 = mfma_bf16x3(a_hi, a_lo, b_hi, b_lo, acc0);
        b_hi = lds_frag_tr16(&bth[buf][r0][32 + col], &bth[buf][r0 + 4][32 + col]);
        b_lo = lds_frag_tr16(&btl[buf][r0][32 + col], &btl[buf][r0 + 4][32 + col]);
        acc1 = mfma_bf16x3(a_hi, a_lo, b_hi, b_lo, acc1);
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint32_t wa[8], wb0[8], wb1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 16 * s + 8 * h + j;
        wa[j] = at[buf][row][wave * 32 + rl];
        wb0[j] = bt[buf][row][rl];
        wb1[j] = bt[buf][row][32 + rl];
      }
      u32x4 a_hi, a_lo, b_hi, b_lo;
      frags_from_words(wa, a_hi, a_lo);
      frags_from_words(wb0, b_hi, b_lo);
      acc0 = mfma_bf16x3(a_hi, a_lo, b_hi, b_lo, acc0);
      frags_from_words(wb1, b_hi, b_lo);
      acc1 = mfma_bf16x3(a_hi, a_lo, b_hi, b_lo, acc1);
    }
  };
  if (nst > 0) {
    Stage t0, t1, t2;
    load_stage(t0, 0);
    if (1 < nst) load_stage(t1, 1);
    if (2 < nst) load_stage(t2, 2);
    store_stage(t0, 0);
    __syncthreads();
#define SE3_TN_STEP_FULL(ST, CUR, NEXT) \
  load_stage(CUR, (ST) + 3);            \
  compute((int)((ST) & 1));             \
  store_stage(NEXT, (int)((ST) & 1) ^ 1); \
  __syncthreads();
#define SE3_TN_STEP(ST, CUR, NEXT)                                   \
  if ((ST) < nst) {                                                  \
    if ((ST) + 3 < nst) load_stage(CUR, (ST) + 3);                   \
    compute((int)((ST) & 1));                                        \
    if ((ST) + 1 < nst) store_stage(NEXT, (int)((ST) & 1) ^ 1);      \
    __syncthreads();                                                 \
  }
    int64_t st = 0;
    const int prio_slot = wave_slot_id();
    for (; st + 6 <= nst; st += 3) {
      rotate_priority(prio_slot + (int)(st / 3));  // every workgroup of this product is resident for the whole launch (common.h)
      SE3_TN_STEP_FULL(st, t0, t1)
      SE3_TN_STEP_FULL(st + 1, t1, t2)
      SE3_TN_STEP_FULL(st + 2, t2, t0)
    }
    for (; st < nst; st += 3) {
      SE3_TN_STEP(st, t0, t1)
      SE3_TN_STEP(st + 1, t1, t2)
      SE3_TN_STEP(st + 2, t2, t0)
    }
#undef SE3_TN_STEP
#undef SE3_TN_STEP_FULL
  }
  float* out = partials + (int64_t)blockIdx.z * ka * n;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int kq = ka0 + wave * 32 + acc_row(r, h);
    if (kq < ka) {
      const int row = A16 ? t16_k_of(kq) : (A24 ? t24_k_of(kq) : kq);
      if (n0 + rl < n) out[(int64_t)row * n + n0 + rl] = acc0[r];
      if (n0 + 32 + rl < n) out[(int64_t)row * n + n0 + 32 + rl] = acc1[r];
    }
  }
}

// Bt[nn][kk] (two bf16 planes, pitch kp) from the layer weights W[C_in, K, C_out]:
//   mode 0 (out = T W)      : nn = o,          kk = i*K + k     value W[i,k,o]
//   mode 1 (gT = g W^T)     : nn = i*K + k,    kk = o           value W[i,k,o]
//   mode 2 (dX = U W')      : nn = i,          kk = o*K + k     value W[i,k,o]
//   mode 3 (H = f W'')      : nn = o*K + k,    kk = i           value W[i,k,o]
__global__ void prep_weights_kernel(const float* __restrict__ w, int c_in, int kb, int c_out, int mode, int n, int k,
                                    int kp, uint16_t* __restrict__ bt_hi, uint16_t* __restrict__ bt_lo,
                                    const float* __restrict__ scale_num, float scale, int frag) {
  const int64_t total = (int64_t)n * kp;
  const float sc = (scale_num ? *scale_num : 1.0f) * scale;  // the GEMM's alpha, applied to the weights once
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int nn = (int)(idx / kp), kk = (int)(idx % kp);
    float v = 0.f;
    if (kk < k) {
      if (mode == 0) v = w[(int64_t)kk * c_out + nn];
      else if (mode == 1) v = w[(int64_t)nn * c_out + kk];
      else if (mode == 2) v = w[((int64_t)nn * kb + (kk % kb)) * c_out + kk / kb];
      else v = w[((int64_t)kk * kb + (nn % kb)) * c_out + nn / kb];
    }
    const uint32_t pk = split_pack(v * sc);
    // frag layout (gemm_strip_bf16_kernel): the 16 bytes lane (nn % 32, (kk % 16) / 8) needs for k-step kk / 16 of
    // column tile nn / 32 sit at [tile][k-step][lane][8], so one load instruction reads 1 KB of consecutive bytes
    // (row-major planes make it touch 32 cache lines for the same 1 KB)
    const int64_t o = frag ? ((((int64_t)(nn / 32) * (kp / 16) + kk / 16) * 64 + ((kk % 16) / 8) * 32 + nn % 32) * 8 + kk % 8)
                           : idx;
    bt_hi[o] = (uint16_t)(pk >> 16);
    bt_lo[o] = (uint16_t)(pk & 0xffffu);
  }
}

}  // namespace

int launch_prep_weights(const float* w, int c_in, int kb, int c_out, int mode, uint16_t* bt_hi, uint16_t* bt_lo,
                        hipStream_t stream, const float* scale_num, float scale, bool frag_layout) {
  int n, k;
  if (mode == 0) n = c_out, k = c_in * kb;
  else if (mode == 1) n = c_in * kb, k = c_out;
  else if (mode == 2) n = c_in, k = c_out * kb;
  else n = c_out * kb, k = c_in;
  const int kp = (k + 31) / 32 * 32;
  const int64_t total = (int64_t)n * kp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(prep_weights_kernel, dim3(blocks), dim3(256), 0, stream, w, c_in, kb, c_out, mode, n, k, kp, bt_hi,
                     bt_lo, scale_num, scale, frag_layout ? 1 : 0);
  return check_launch();
}

// Split the k loop over grid.z when the row blocks alone cannot fill the chip (small hierarchy levels):
// a block's k loop is serial, so 64 k-tiles on a handful of blocks would cost ~100 us whatever M is.
// 64-column blocks per workgroup (template NB): 128 columns pay when the k loop is long (T W, U W': k = C*K); for the
// short-k, wide-n products (grad_T beyond the strip kernel's k <= 64) the narrow tile keeps more blocks in flight
int gemm_nn_bf16_col_blocks(int n, int k) { return n > BN && k >= 512 ? 2 : 1; }

// k groups per workgroup of the 3-byte-row kernel (gemm_nn_t24_kernel, KG).  OPT-IN (SE3_NN_KG=2: two groups when the
// output tiles alone leave the chip under-filled and the super tiles of a workgroup divide evenly): measured in round 5
// against the split over grid.z it was meant to replace (profiles/r05_nn_kgroups_ab.txt) -- the 18 k-row level of the
// headline stack 0.327 -> 0.336 ms, its 254-row level 0.069 -> 0.074, the stack +0.8 %, scannet150k_f1 +0.6 %, dfaust_f2
// -0.9 %: eight wavefronts sharing one CU's LDS pipe and barriers buy no more than the reduction launch they save.
int gemm_nn_t24_k_groups(int64_t m, int n, int k) {
  static const int forced = [] {
    const char* e = getenv("SE3_NN_KG");
    return e ? atoi(e) : 0;
  }();
  if (forced != 2 || k % 128 != 0) return 1;
  const int bnw = BN * gemm_nn_bf16_col_blocks(n, k);
  const int64_t tiles = ((m + BM - 1) / BM) * ((n + bnw - 1) / bnw);
  return tiles > 0 && tiles <= 384 ? 2 : 1;  // beyond 1.5 workgroups per CU the plain form covers its own latencies
}

int gemm_nn_bf16_splits(int64_t m, int n, int k, int kg) {
  const int bnw = BN * gemm_nn_bf16_col_blocks(n, k);
  const int64_t tiles = ((m + BM - 1) / BM) * ((n + bnw - 1) / bnw);
  const int nkt = (k + BK - 1) / BK;
  const int target = 512 / kg;  // workgroups aimed at (two per CU; one of eight wavefronts); 1024 and 256 measured slower on the 9 k-point level
  if (tiles < 1 || tiles >= target / 2 || nkt < 8) return 1;  // (tiles = 0: an empty cloud)
  // at most `target` workgroups (one more split than fits starts a second, nearly empty round: 144 tiles x 4 splits =
  // 576 on 512 slots took as long as two full rounds), at least 4 k-tiles per split
  int64_t s_max = target / tiles;
  if (s_max > nkt / (4 * kg)) s_max = nkt / (4 * kg);
  if (s_max < 1) s_max = 1;
  static const int forced = [] {  // A/B knob: SE3_NN_SPLITS=n forces the split count (clamped to what the shape allows)
    const char* e = getenv("SE3_NN_SPLITS");
    return e ? atoi(e) : 0;
  }();
  if (forced > 0) return (int)(forced < s_max ? forced : s_max);
  // Cost model in microseconds, fitted to the stage times of the small and mid-sized levels (profiles/r03_nn_split_sweep.txt:
  // forced split counts on the 18 k-row headline level and the 4 k-row / 128-channel DFaust level): a block's k loop is
  // serial at ~0.75 us per 32-k tile (1.1 us with 128-column tiles); the A stream (~3.5 bytes per element) runs at the
  // rate of the level it comes from -- rows of up to ~200 MB were written by the kernel in front and are still in the
  // memory-side cache (9 TB/s), larger ones come from HBM (4.5 TB/s) -- times the share of the 512 workgroup slots that
  // are filled; a split costs its reduction launch (~12 us with the gap in front of it) and the partials written and
  // read back (~0.5 us per MB and split).  64 splits of a 512 x 256 output spent more on 33 MB of partials than on the
  // GEMM; 2 or 4 instead of 3 splits of the 18 k-row level cost 10 us each way.
  const double a_bytes = (double)m * k * 3.5, out_bytes = (double)m * n * 4.0;
  const double t_tile = bnw > BN ? 1.1 : 0.75, rate = a_bytes <= 200e6 ? 9.0e6 : 4.5e6;
  int best = 1;
  double best_cost = 1e30;
  for (int64_t s = 1; s <= s_max; ++s) {
    const int per = (int)((nkt + s - 1) / s);
    const int s_eff = (nkt + per - 1) / per;
    const double fill = (double)(tiles * s_eff) / (double)target;
    const double t_loop = per * t_tile / kg, t_stream = a_bytes / (rate * (fill < 1.0 ? fill : 1.0));
    const double cost = (t_loop > t_stream ? t_loop : t_stream) + (s_eff > 1 ? 12.0 + s_eff * out_bytes * 0.5e-6 : 0.0);
    if (cost < best_cost) best_cost = cost, best = s_eff;
  }
  return best;
}

// Row-strip kernel: packed output, k <= 64, n a multiple of 32, weights prepared with frag_layout and alpha folded in.
bool gemm_strip_bf16_applicable(int64_t m, int n, int k) {
  const int kp = (k + 31) / 32 * 32;
  return kp <= 128 && kp != 96 && n >= 512 && n % 32 == 0 && n <= (1 << 22) && m >= 128 * 16 &&
         (m + 128) * (int64_t)k * 4 < (1ll << 32) - 64 && (int64_t)(n + 64) * kp * 2 < (1ll << 32) - 64;
}

// T16 output: whole mega tiles of 256 columns, k of one or two 32-steps, rows addressed with 32-bit byte offsets
bool gemm_strip_t16_applicable(int64_t m, int n, int k) {
  const int kp = (k + 31) / 32 * 32;
  return gemm_strip_bf16_applicable(m, n, k) && (kp == 32 || kp == 64) && k % 8 == 0 && n % 256 == 0 &&
         (m + 128) * ((int64_t)n / 32 * 72) < (1ll << 32) - 64;
}

int launch_gemm_strip_bf16(const char* tag, const uint32_t* a, const uint16_t* bt_hi, const uint16_t* bt_lo, uint32_t* c,
                           int64_t m, int n, int k, hipStream_t stream, bool out_t16) {
  if (m == 0 || n == 0) return SE3_OK;
  if (!gemm_strip_bf16_applicable(m, n, k)) return SE3_ERR_UNSUPPORTED;
  ProfScope prof(tag, stream);
  if (out_t16) {
    if (!gemm_strip_t16_applicable(m, n, k)) return SE3_ERR_UNSUPPORTED;
    const int64_t rbk = (m + 127) / 128;
    int split = rbk >= 1024 ? 1 : (int)(1024 / rbk);  // <= 1024 workgroups = one resident round
    const int n_mega = n / 256;
    if (split > n_mega) split = n_mega;
    const dim3 g16((unsigned)rbk, (unsigned)split);
    if ((k + 31) / 32 * 32 == 32)
      hipLaunchKernelGGL(gemm_strip_t16_kernel<2>, g16, dim3(256), 0, stream, a, bt_hi, bt_lo, (uint8_t*)c, m, n, k);
    else
      hipLaunchKernelGGL(gemm_strip_t16_kernel<4>, g16, dim3(256), 0, stream, a, bt_hi, bt_lo, (uint8_t*)c, m, n, k);
    return check_launch();
  }
  const int64_t row_blocks = (m + 127) / 128;
  int n_split = row_blocks >= 1024 ? 1 : (int)(1024 / row_blocks);  // <= 1024 workgroups = one resident round (4 per CU)
  const int n_tiles = n / 32;
  if (n_split > n_tiles / 4) n_split = n_tiles / 4 > 0 ? n_tiles / 4 : 1;  // >= 4 column tiles per block
  const dim3 sgrid((unsigned)row_blocks, (unsigned)n_split);
  const int kp = (k + 31) / 32 * 32;
  if (kp == 32)
    hipLaunchKernelGGL(gemm_strip_bf16_kernel<2>, sgrid, dim3(256), 0, stream, a, bt_hi, bt_lo, c, m, n, k);
  else if (kp == 64)
    hipLaunchKernelGGL(gemm_strip_bf16_kernel<4>, sgrid, dim3(256), 0, stream, a, bt_hi, bt_lo, c, m, n, k);
  else
    hipLaunchKernelGGL(gemm_strip_bf16_kernel<8>, sgrid, dim3(256), 0, stream, a, bt_hi, bt_lo, c, m, n, k);
  return check_launch();
}

static int gemm_nn_bf16_rows(const uint32_t* a, const uint16_t* bt_hi, const uint16_t* bt_lo, void* c, bool out_packed,
                             int64_t m, int n, int k, float* split_ws, const float* alpha_num, float alpha_scale,
                             hipStream_t stream, int afmt, ReduceBatch* defer) {
  const bool a24 = afmt == 1, a16 = afmt == 2;  // A rows: 0 packed words, 1 3-byte rows, 2 T16 (common.h)
  const int kp = (k + 31) / 32 * 32;
  const int nkt = kp / BK;
  const int kg = a24 ? gemm_nn_t24_k_groups(m, n, k) : 1;
  int splits = split_ws ? gemm_nn_bf16_splits(m, n, k, kg) : 1;
  const int per = (nkt + splits - 1) / splits;
  splits = (nkt + per - 1) / per;
  const int nbw = gemm_nn_bf16_col_blocks(n, k);
  const dim3 grid((unsigned)((m + BM - 1) / BM), (unsigned)((n + BN * nbw - 1) / (BN * nbw)), (unsigned)splits);
  const bool fast = (k % 32 == 0) && (m * (int64_t)k * 4 < (1ll << 32)) && ((int64_t)n * kp * 2 < (1ll << 32)) &&
                    ((m + BM) * (int64_t)k * 4 < (1ll << 32)) && ((int64_t)(n + 128) * kp * 2 < (1ll << 32));
  if (a24 && (!fast || k % 64 != 0)) return SE3_ERR_UNSUPPORTED;
  // T16: mega tiles of 256 k; its rows are 2.25 bytes per element, addressed with 32-bit byte offsets like the others
  if (a16 && (k % 256 != 0 || (m + BM) * ((int64_t)k / 32 * 72) >= (1ll << 32) || (int64_t)(n + 128) * kp * 2 >= (1ll << 32)))
    return SE3_ERR_UNSUPPORTED;
  int st_per = (per + 1) / 2;  // 3-byte / T16 rows: the kernels walk super tiles of 64 k
  if (a16) st_per = (st_per + 3) / 4 * 4;  // whole mega tiles per split
  if (kg == 2) st_per = (st_per + 1) / 2 * 2;  // two k groups: an even number of super tiles per workgroup (k % 128 == 0)
  if (a24 || a16) splits = (k / 64 + st_per - 1) / st_per;
  const dim3 grid24((unsigned)((m + BM - 1) / BM), grid.y, (unsigned)splits);
#define SE3_NN_LAUNCH(MODE, F, NBV, OUT)                                                                              \
  hipLaunchKernelGGL((gemm_nn_bf16_kernel<MODE, F, NBV>), grid, dim3(256), 0, stream, a, bt_hi, bt_lo, (void*)(OUT), m, n, \
                     k, kp, per, alpha_num, alpha_scale)
#define SE3_NN_LAUNCH24_KG(MODE, NBV, KGV, OUT)                                                                       \
  do {                                                                                                                \
    constexpr int lds_bytes = gemm_nn_t24_lds_bytes(NBV, KGV);                                                         \
    if (lds_bytes > 64 * 1024) {  /* beyond the default dynamic-LDS limit: raised once per (device, instantiation) -- a \
                                     runtime that keeps the attribute per device must see it on every device the      \
                                     process uses; a failure is not cached */                                          \
      static std::atomic<uint64_t> raised{0};                                                                          \
      int dev_ = 0;                                                                                                   \
      if (hipGetDevice(&dev_) != hipSuccess) return SE3_ERR_LAUNCH;                                                   \
      const uint64_t bit_ = 1ull << (dev_ & 63);                                                                      \
      if (!(raised.load(std::memory_order_relaxed) & bit_)) {                                                         \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nn_t24_kernel<MODE, NBV, KGV>),                   \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)                 \
          return SE3_ERR_LAUNCH;                                                                                      \
        raised.fetch_or(bit_, std::memory_order_relaxed);                                                             \
      }                                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL((gemm_nn_t24_kernel<MODE, NBV, KGV>), grid24, dim3(256 * KGV), lds_bytes, stream, (const uint8_t*)a, \
                       bt_hi, bt_lo, (void*)(OUT), m, n, k, st_per, alpha_num, alpha_scale);                          \
  } while (0)
#define SE3_NN_LAUNCH24(MODE, NBV, OUT)                    \
  do {                                                     \
    if (kg == 2) SE3_NN_LAUNCH24_KG(MODE, NBV, 2, OUT);    \
    else SE3_NN_LAUNCH24_KG(MODE, NBV, 1, OUT);            \
  } while (0)
#define SE3_NN_LAUNCH16(MODE, NBV, OUT)                                                                               \
  hipLaunchKernelGGL((gemm_nn_t16_kernel<MODE, NBV>), grid24, dim3(256), 0, stream, (const uint8_t*)a, bt_hi, bt_lo,   \
                     (void*)(OUT), m, n, k, st_per, alpha_num, alpha_scale)
#define SE3_NN(MODE, OUT)                                    \
  do {                                                       \
    if (a16 && nbw == 2) SE3_NN_LAUNCH16(MODE, 2, OUT);      \
    else if (a16) SE3_NN_LAUNCH16(MODE, 1, OUT);             \
    else if (a24 && nbw == 2) SE3_NN_LAUNCH24(MODE, 2, OUT); \
    else if (a24) SE3_NN_LAUNCH24(MODE, 1, OUT);             \
    else if (fast && nbw == 2) SE3_NN_LAUNCH(MODE, true, 2, OUT); \
    else if (fast) SE3_NN_LAUNCH(MODE, true, 1, OUT);        \
    else if (nbw == 2) SE3_NN_LAUNCH(MODE, false, 2, OUT);   \
    else SE3_NN_LAUNCH(MODE, false, 1, OUT);                 \
  } while (0)
  if (splits > 1) {
    SE3_NN(2, split_ws);
    const int64_t count = m * n;
    const int rb = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    if (defer)  // the caller folds the partials together with its other reductions (ReduceBatch)
      defer->sum(split_ws, c, count, splits, alpha_num, alpha_scale, out_packed);
    else if (out_packed)
      hipLaunchKernelGGL(reduce_splits_kernel<true>, dim3(rb), dim3(256), 0, stream, split_ws, c, count, splits,
                         alpha_num, alpha_scale);
    else
      hipLaunchKernelGGL(reduce_splits_kernel<false>, dim3(rb), dim3(256), 0, stream, split_ws, c, count, splits,
                         alpha_num, alpha_scale);
  } else if (out_packed) {
    SE3_NN(1, c);
  } else {
    SE3_NN(0, c);
  }
#undef SE3_NN
#undef SE3_NN_LAUNCH
#undef SE3_NN_LAUNCH24
#undef SE3_NN_LAUNCH24_KG
#undef SE3_NN_LAUNCH16
  return check_launch();
}

// The kernels address A with 32-bit byte offsets (buffer loads): more rows than those reach -- 262 k rows of 2048 values
// and up, i.e. clouds beyond ~130 k points at two frames -- go through the same kernels row block by row block (no split
// of k at that size, so the blocks share nothing but the weights).
int launch_gemm_nn_bf16(const char* tag, const uint32_t* a, const uint16_t* bt_hi, const uint16_t* bt_lo, void* c,
                        bool out_packed, int64_t m, int n, int k, float* split_ws, const float* alpha_num,
                        float alpha_scale, hipStream_t stream, int afmt, ReduceBatch* defer) {
  if (m == 0 || n == 0) return SE3_OK;
  ProfScope prof(tag, stream);
  const int64_t max_rows = (((1ll << 32) - 64) / ((int64_t)k * 4) - 2 * BM) / BM * BM;
  if (m <= max_rows || max_rows < BM || k % 32 != 0)
    return gemm_nn_bf16_rows(a, bt_hi, bt_lo, c, out_packed, m, n, k, split_ws, alpha_num, alpha_scale, stream, afmt, defer);
  const int64_t a_row_bytes = afmt == 2 ? (int64_t)k / 32 * 72 : (int64_t)k * (afmt == 1 ? 3 : 4);
  for (int64_t m0 = 0; m0 < m; m0 += max_rows) {
    const int64_t mb = m - m0 < max_rows ? m - m0 : max_rows;
    if (int rc = gemm_nn_bf16_rows(reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a) + m0 * a_row_bytes), bt_hi,
                                   bt_lo, static_cast<char*>(c) + m0 * (int64_t)n * 4, out_packed, mb, n, k, nullptr, alpha_num,
                                   alpha_scale, stream, afmt, nullptr))
      return rc;
  }
  return SE3_OK;
}

size_t gemm_nn_bf16_split_bytes(int64_t m, int n, int k) {
  // the caller does not know the row format yet: room for whichever form (one or two k groups) splits further
  const int s1 = gemm_nn_bf16_splits(m, n, k, 1), s2 = gemm_nn_bf16_splits(m, n, k, gemm_nn_t24_k_groups(m, n, k));
  const int s = s1 > s2 ? s1 : s2;  // (rounding the super tiles per split up to an even count can only remove a split)
  return s > 1 ? (size_t)s * m * n * 4 : 0;
}

int launch_gemm_tn_bf16(const char* tag, const uint32_t* a, const uint32_t* b, float* c, float* partials, int splits,
                        int64_t m, int ka, int n, const float* alpha_num, float alpha_scale, hipStream_t stream, int afmt,
                        ReduceBatch* defer, bool out_ikn) {
  const bool a24 = afmt == 1, a16 = afmt == 2;
  if (ka == 0 || n == 0) return SE3_OK;
  if (out_ikn && (!defer || ka % kBasis != 0)) return SE3_ERR_INVALID_ARGUMENT;  // the permuted store lives in the batched reduction
  ProfScope prof(tag, stream);
  int64_t chunk = (m + splits - 1) / splits;
  chunk = (chunk + BK - 1) / BK * BK;
  if (chunk == 0) chunk = BK;
  // Both operands are addressed with 32-bit byte offsets from the pointers the kernel gets: row ranges (grid.z) beyond
  // their reach are launched as further groups of ranges, each with its operands' pointers moved to its first row
  // (the partials of group j start at range j * zs of the same buffer).
  const int64_t widest = (int64_t)(ka > n ? ka : n) * 4;
  int64_t zs = (((1ll << 32) - 64) / widest - chunk) / chunk;  // ranges one launch can address
  if (zs < 1) zs = 1;
  if (zs > splits) zs = splits;
  const bool vec = n % 4 == 0, reach = (zs + 1) * chunk * widest < (1ll << 32) - 64;
  if (a24 && (!vec || !reach || ka % 64 != 0)) return SE3_ERR_UNSUPPORTED;
  if (a16 && (!vec || !reach || ka % 256 != 0)) return SE3_ERR_UNSUPPORTED;
  for (int64_t z0 = 0; z0 < splits; z0 += zs) {
    const int64_t zn = splits - z0 < zs ? splits - z0 : zs, r0 = z0 * chunk;
    if (r0 >= m) {  // ranges past the last row (rounding of chunk): their partials are zeros
      if (int rc = launch_fill_words(partials + z0 * ka * n, 0u, (int64_t)(splits - z0) * ka * n, stream)) return rc;
      break;
    }
    const int64_t mb = m - r0 < zn * chunk ? m - r0 : zn * chunk;
    const dim3 grid((unsigned)((ka + 127) / 128), (unsigned)((n + BN - 1) / BN), (unsigned)zn);
    const int64_t a_row_bytes = a16 ? (int64_t)ka / 32 * 72 : (int64_t)ka * (a24 ? 3 : 4);
    const uint32_t* ab = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(a) + r0 * a_row_bytes);
    const uint32_t* bb = b + r0 * n;
    float* pb = partials + z0 * ka * n;
    if (a16)
      hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, 2>), grid, dim3(256), 0, stream, ab, bb, pb, mb, ka, n, chunk);
    else if (a24)
      hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, 1>), grid, dim3(256), 0, stream, ab, bb, pb, mb, ka, n, chunk);
    else if (vec && reach)
      hipLaunchKernelGGL((gemm_tn_bf16_kernel<true, 0>), grid, dim3(256), 0, stream, ab, bb, pb, mb, ka, n, chunk);
    else
      hipLaunchKernelGGL((gemm_tn_bf16_kernel<false, 0>), grid, dim3(256), 0, stream, ab, bb, pb, mb, ka, n, chunk);
  }
  if (defer) {
    defer->sum(partials, c, (int64_t)ka * n, splits, alpha_num, alpha_scale, false, out_ikn ? n : 0);
    return check_launch();
  }
  return launch_reduce_partials(partials, c, (int64_t)ka * n, splits, alpha_num, alpha_scale, stream);
}

}  // namespace se3
