// Probe (round 5): the CONTRACTION phase of an accumulator-resident fused tile, by itself.
//
// Design being priced: a four-wavefront workgroup (one per CU: 512 registers per lane) keeps the T tile of 32 output rows
// x 2048 (channel, basis) values as MFMA accumulators -- 256 registers per lane, exactly as the wave-pair edge kernel
// leaves them (acc tile = 32 channels x 32 basis functions of one row; a wavefront owns 8 rows = 16 tiles) -- and then
// contracts it with the C_in*K x C_out weights WITHOUT a round trip of T through HBM:
//   out[32, 64] = T[32, 2048] . W[2048, 64]      (split-bf16: T and W as hi + lo, three MFMA products)
// T's rows sit in different wavefronts' registers with the contraction index on lanes / registers, so the tile goes through
// LDS once (piece by piece: 8 channels x 32 basis functions = 256 k per piece, 33 KB as hi / lo planes) to become MFMA A
// fragments; W (512 KB of split planes -- more than the 160 KB of LDS) is streamed from L2 once per tile in 32 KB
// half-pieces through a double-buffered LDS image (global -> registers -> LDS, issued one half-piece ahead).
// Wavefront (nh, kh) multiplies k-steps 4 kh .. 4 kh + 3 of every half-piece into columns 32 nh .. 32 nh + 31; the two k
// halves are added through LDS at the end of the tile.
//
// What the probe measures: microseconds per tile with every CU of the chip doing this at once (all streaming the same
// L2-resident weights).  The edge phase is NOT here, and neither are its 256 accumulator registers: hipcc spills as soon
// as a 256-register array is read piecewise under a loop (first version of this file: 636 - 1476 bytes of scratch per
// lane), so a piece's 32 values per lane come from 32 registers that every piece re-uses (T[row][ch][k] depends on ch & 7
// only) -- the same conversions, LDS traffic, weight stream and MFMA work as the real thing; the register budget of the
// real thing = this kernel's count + 256.
// Checked against a host fp64 evaluation of the same tile.  Build: hipcc -O3 --offload-arch=gfx950 fused_tile.hip -o fused_tile
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using bf16x2 = __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16;
using f32x2 = __attribute__((__vector_size__(2 * sizeof(float)))) float;
using u32x4 = __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t;
using u32x2 = __attribute__((__vector_size__(2 * sizeof(uint32_t)))) uint32_t;

constexpr int kRows = 32, kCK = 2048, kN = 64;
constexpr int kPieces = 8, kPieceK = 256, kHalfK = 128;
constexpr int kAPitch = kPieceK * 2 + 16;  // bytes per row of the A image (one plane)
constexpr int kWPitch = kHalfK * 2 + 16;   // bytes per column of the W image (one plane)
constexpr int kAImg = 2 * kRows * kAPitch;           // hi + lo planes: 33 792 B
constexpr int kWImg = 2 * kN * kWPitch;              // hi + lo planes: 34 816 B
constexpr int kLds = 2 * kAImg + 2 * kWImg + 2 * 32 * 32 * 4;  // + the k-half reduction scratch

__host__ __device__ constexpr int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t cvt_pk(float x0, float x1) {
  f32x2 v = {x0, x1};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  hi = cvt_pk(x0, x1);
  lo = cvt_pk(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}
// the stand-in for what the edge phase leaves in the accumulators: T[row][ch][k] of tile `tile`
__host__ __device__ inline float t_value(int row, int ch, int k) {  // (the same for every tile: regenerating it is not what is timed)
  const uint32_t x = (uint32_t)(row * 131 + ch * 17 + k * 3) * 2654435761u;
  return (float)(int)(x >> 20) * (1.0f / 2048.0f) - 1.0f;  // [-1, 1)
}

// wg layout: w_hi / w_lo [piece 8][half 2][n 64][kk 128] bf16, kk = (basis k) * 8 + (channel & 7) inside a piece of 8 channels
#ifndef PROBE_WAVES
#define PROBE_WAVES 1  // 2: the register count under a 256-register cap (what is left beside 256 accumulators)
#endif
__global__ __launch_bounds__(256, PROBE_WAVES) void contract_tiles(const uint16_t* __restrict__ w_hi, const uint16_t* __restrict__ w_lo,
                                                         float* __restrict__ out, int tiles_per_wg, int with_stream) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* a_img = lds;                    // [2 buffers][hi, lo][32 rows][kAPitch]
  char* w_img = lds + 2 * kAImg;        // [2 buffers][hi, lo][64 cols][kWPitch]
  float* red = reinterpret_cast<float*>(lds + 2 * kAImg + 2 * kWImg);  // [2 nh][32 x 32]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kcol = lane & 31, h = lane >> 5;
  const int nh = wave & 1, kh = wave >> 1;

  // ---- one piece of the tile as the edge phase leaves it: rows 8 wave .. + 7, registers 4 pq .. 4 pq + 3 of a tile
  float src[8][4];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr)
#pragma unroll
    for (int i = 0; i < 4; ++i) src[rr][i] = t_value(8 * wave + rr, 4 * h + i, kcol);
  for (int it = 0; it < tiles_per_wg; ++it) {
    const int tile = blockIdx.x * tiles_per_wg + it;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(src[rr][i]));  // opaque per tile: nothing is hoisted out of the tile loop

    f32x16 oacc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // weight half-piece hp (0..15) -> registers: 32 KB = 8 x 16 B per thread; thread -> (plane, col, 16-byte chunk)
    u32x4 wreg[8];
    auto load_w = [&](int hp) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = i * 256 + threadIdx.x;  // 0 .. 2047: plane = idx >> 10, col = (idx >> 4) & 63, chunk = idx & 15
        const uint16_t* src = (idx >> 10) ? w_lo : w_hi;
        const int64_t off = with_stream ? ((int64_t)hp * kN + ((idx >> 4) & 63)) * kHalfK + (idx & 15) * 8 : ((idx >> 4) & 63) * kHalfK + (idx & 15) * 8;
        wreg[i] = *reinterpret_cast<const u32x4*>(src + off);
      }
    };
    auto store_w = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = i * 256 + threadIdx.x;
        char* dst = w_img + buf * kWImg + (idx >> 10) * (kN * kWPitch) + ((idx >> 4) & 63) * kWPitch + (idx & 15) * 16;
        *reinterpret_cast<u32x4*>(dst) = wreg[i];
      }
    };
    // piece p of this wavefront's 8 rows -> A image buffer (p & 1)
    auto store_a = [&](int p) {
      char* img = a_img + (p & 1) * kAImg;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        uint32_t h0, l0, h1, l1;
        split2(src[rr][0], src[rr][1], h0, l0);
        split2(src[rr][2], src[rr][3], h1, l1);
        char* row = img + (8 * wave + rr) * kAPitch + (kcol * 8 + 4 * h) * 2;
        *reinterpret_cast<u32x2*>(row) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(row + kRows * kAPitch) = u32x2{l0, l1};
      }
    };
    load_w(0);
    store_a(0);
    store_w(0);
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
    for (int p = 0; p < kPieces; ++p) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int hp = p * 2 + q;
        if (hp + 1 < 2 * kPieces) load_w(hp + 1);  // next half-piece in flight under this one's MFMAs
        if (q == 1 && p + 1 < kPieces) store_a(p + 1);  // next piece's A image (other buffer) under this half's MFMAs
        const char* ai = a_img + (p & 1) * kAImg + kcol * kAPitch + (q * kHalfK + kh * 64 + 8 * h) * 2;
        const char* wi = w_img + (hp & 1) * kWImg + (32 * nh + kcol) * kWPitch + (kh * 64 + 8 * h) * 2;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4 a_hi = *reinterpret_cast<const u32x4*>(ai + ks * 32);
          const u32x4 a_lo = *reinterpret_cast<const u32x4*>(ai + ks * 32 + kRows * kAPitch);
          const u32x4 b_hi = *reinterpret_cast<const u32x4*>(wi + ks * 32);
          const u32x4 b_lo = *reinterpret_cast<const u32x4*>(wi + ks * 32 + kN * kWPitch);
          oacc = mfma(a_lo, b_hi, oacc);
          oacc = mfma(a_hi, b_lo, oacc);
          oacc = mfma(a_hi, b_hi, oacc);
        }
        if (hp + 1 < 2 * kPieces) store_w((hp + 1) & 1);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);  // nothing of the next half-piece (its weight loads above all) moves up here
      }
    }
    // add the two k halves, store the tile
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[nh * 1024 + acc_row(r, h) * 32 + kcol] = oacc[r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        out[((int64_t)tile * kRows + acc_row(r, h)) * kN + 32 * nh + kcol] = oacc[r] + red[nh * 1024 + acc_row(r, h) * 32 + kcol];
    }
    __syncthreads();
  }
}

static uint16_t bf16_of(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static float f_of(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float x;
  memcpy(&x, &u, 4);
  return x;
}

int main() {
  const int n_cu = 256;
  std::vector<float> w((size_t)kCK * kN);
  srand(1);
  for (auto& v : w) v = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) * 0.02f;
  // weight k index of (piece p, half q, kk): channel = 8 p + (kk' & 7), basis = kk' >> 3 with kk' = q * 128 + kk
  std::vector<uint16_t> whi((size_t)kCK * kN), wlo((size_t)kCK * kN);
  for (int p = 0; p < kPieces; ++p)
    for (int q = 0; q < 2; ++q)
      for (int n = 0; n < kN; ++n)
        for (int kk = 0; kk < kHalfK; ++kk) {
          const int kq = q * kHalfK + kk, ch = 8 * p + (kq & 7), k = kq >> 3;
          const float v = w[(size_t)(ch * 32 + k) * kN + n];
          const uint16_t hi = bf16_of(v);
          const size_t dst = (((size_t)(p * 2 + q) * kN + n) * kHalfK) + kk;
          whi[dst] = hi, wlo[dst] = bf16_of(v - f_of(hi));
        }
  uint16_t *d_hi, *d_lo;
  float* d_out;
  const int max_tiles = 64;
  (void)hipMalloc(&d_hi, whi.size() * 2), (void)hipMalloc(&d_lo, wlo.size() * 2);
  (void)hipMalloc(&d_out, (size_t)n_cu * max_tiles * kRows * kN * 4);
  (void)hipMemcpy(d_hi, whi.data(), whi.size() * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(d_lo, wlo.data(), wlo.size() * 2, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)contract_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, (const void*)contract_tiles);
  printf("contract_tiles: %d VGPRs (arch + acc), %d bytes of scratch, %d B LDS per workgroup\n", fa.numRegs, (int)fa.localSizeBytes, kLds);

  // correctness: tile 5 against fp64 of the same fp32 inputs
  hipLaunchKernelGGL(contract_tiles, dim3(n_cu), dim3(256), kLds, 0, d_hi, d_lo, d_out, 1, 1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<float> got((size_t)kRows * kN);
  const int tile = 5;
  (void)hipMemcpy(got.data(), d_out + (size_t)tile * kRows * kN, got.size() * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (int m = 0; m < kRows; ++m)
    for (int n = 0; n < kN; ++n) {
      double ref = 0;
      for (int ch = 0; ch < 64; ++ch)
        for (int k = 0; k < 32; ++k) ref += (double)t_value(m, ch & 7, k) * (double)w[(size_t)(ch * 32 + k) * kN + n];
      const double d = got[(size_t)m * kN + n] - ref;
      num += d * d, den += ref * ref;
    }
  printf("tile %d vs fp64: rel L2 error %.3g\n", tile, std::sqrt(num / den));

  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  for (int with_stream = 1; with_stream >= 0; --with_stream)
    for (int tiles : {16, 64}) {
      float best = 1e9f;
      for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(contract_tiles, dim3(n_cu), dim3(256), kLds, 0, d_hi, d_lo, d_out, tiles, with_stream);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2 && ms < best) best = ms;
      }
      printf("%s: %2d tiles per CU: %.3f ms per launch = %.2f us per 32-row tile (%.0f GB/s of weight planes per CU)\n",
             with_stream ? "weights streamed (512 KB per tile)" : "one 32 KB weight block re-read (L1/L2-hot; no stream)", tiles, best,
             best * 1e3 / tiles, 512.0 * 1024 / (best * 1e-3 / tiles) / 1e9);
    }
  printf("headline layer: 4096 tiles on 256 CUs = 16 tiles per CU\n");
  return 0;
}
