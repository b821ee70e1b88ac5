// Split-bf16 ("bf16x3") versions of the edge kernels: same decomposition as edge_kernels.hip (one
// wavefront per output row, 32 frame-edges per chunk, kernel-MLP output used in place as the MFMA
// B operand) on v_mfma_f32_32x32x16_bf16 -- 16x the fp32-MFMA rate at 3 products per multiply.
//
// k-dimension bookkeeping (16 per MFMA): lane half h owns k-slots 8h..8h+7.  For the aggregate
// T += feat^T phi, k-step s (0,1) of a chunk covers the frame-edges held in accumulator registers
// 8s..8s+7 of the MLP result, i.e. slot (h, j) <-> frame-edge acc_row(8s + j, h); the A operand
// (gathered features) is loaded in exactly that order, so no lane ever moves data.
//
// Gathered operands (features, grad_out) are read as packed words (hi << 16 | lo) prepared by the batched
// preparation launch (prep.hip), geometry as 64-byte records; intermediates (T, grad_T, U) are written as packed
// words too.  The bf16 kernels evaluate GELU in its scaled form (common.h: gelu_scaled): T and U hold kGeluOut times
// the reference's values, the consuming GEMM divides it out.
//
// Kernels in this file:
//   edge_t_pair_bf16_kernel<CT, FULL, NF>   default for C >= 64: a wave pair per item (two frames, or one row for odd F)
//   edge_t_bf16_kernel<VW, FC, FULL>        single wavefront per item (edge_bf16_body.h), used for C < 64
//   edge_param_grad_bf16_v2_kernel<CH16, NFR>  parameter gradients, 64-channel blocks over blockIdx.y
//   edge_param_grad_bf16_kernel             generic fallback (channel counts that are not multiples of 16)
#include <cstdlib>

// T / U rows leave the wave-pair kernel through non-temporal stores: the kernel is VALU-bound and does not care, and
// the kernel that follows it (the GEMM reading those rows) no longer runs against the write-back of ~300 MB of dirty
// lines left in L2 and the memory-side cache: gemm_out 0.221 -> 0.188 ms.  (The strip GEMM is bound by its stores:
// there the same flag costs 0.03 ms and buys 0.02 in the parameter-gradient kernel behind it -- not used.)
#include "common.h"
#include "edge_bf16_body.h"

// SE3_TIMELINE=1 (diagnostic build only, tools/edge_timeline.py): s_memtime stamps at the phase boundaries of the wave-pair
// edge kernel and of the pair form of the parameter-gradient kernel.  Every stamp is one asm statement with its own
// lgkmcnt(0) between two scheduling barriers (cdna_hip_programming.md section 7, "In-kernel stamps"); a load's wait is made
// a segment of its own by pinning the loaded value between two stamps.  The per-wave records go to a buffer of their own
// (se3_timeline_set) that nothing else reads.  The shipped library contains no stamp and no such symbol.
#ifndef SE3_TIMELINE
#define SE3_TIMELINE 0
#endif
#if SE3_TIMELINE
#define TL(...) __VA_ARGS__
namespace se3 {
constexpr int kTlWords = 64;  // words per record; region r starts at r * cap records (0 edge_t forward, 1 transposed, 2 parameter gradient)
__device__ uint32_t* g_tl_buf = nullptr;
__device__ uint32_t g_tl_cap = 0;
__device__ __forceinline__ uint32_t tl_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return (uint32_t)t;
}
template <class T>
__device__ __forceinline__ void tl_pin(const T& x) { asm volatile("" ::"v"(x)); }
// stamps are parked in a wave-private LDS record (every lane writes the same word: no exec games, two instructions) and
// leave for the buffer once, at the wave's end
__device__ __forceinline__ void tl_mark(uint32_t* rec, int slot) { const uint32_t t = tl_now(); if (slot < kTlWords) rec[slot] = t; }
// persistent kernels: the time since the previous stamp is added to segment `slot`
__device__ __forceinline__ void tl_seg(uint32_t* rec, uint32_t& prev, int slot) { const uint32_t t = tl_now(); rec[slot] += t - prev; prev = t; }
__device__ __forceinline__ void tl_flush(int region, uint32_t idx, const uint32_t* rec, int n_words = kTlWords) {
  if (g_tl_buf == nullptr || idx >= g_tl_cap || (int)(threadIdx.x & 63) >= n_words) return;
  g_tl_buf[((size_t)region * g_tl_cap + idx) * kTlWords + (threadIdx.x & 63)] = rec[threadIdx.x & 63];
}
}  // namespace se3
extern "C" int se3_timeline_set(void* buf, uint32_t cap_records) {
  uint32_t* b = static_cast<uint32_t*>(buf);
  if (hipMemcpyToSymbol(HIP_SYMBOL(se3::g_tl_buf), &b, sizeof(b)) != hipSuccess) return SE3_ERR_LAUNCH;
  if (hipMemcpyToSymbol(HIP_SYMBOL(se3::g_tl_cap), &cap_records, sizeof(cap_records)) != hipSuccess) return SE3_ERR_LAUNCH;
  return SE3_OK;
}
#else
#define TL(...)
#endif

namespace se3 {

namespace {

struct RowInfo {
  int64_t ctr;
  int start, n_total;
};

__device__ __forceinline__ RowInfo row_info(const EdgeGeom& g, int64_t m) {
  RowInfo r;
  r.ctr = m / g.f_ctr;
  r.start = r.ctr > 0 ? g.ends[r.ctr - 1] : 0;
  r.n_total = (g.ends[r.ctr] - r.start) * g.f_nb;
  return r;
}

__device__ __forceinline__ void lane_descriptor(const EdgeGeom& g, const RowInfo& ri, int fe, const float yc[3],
                                                const float rc[9], float rho, float d[kDescExt], int& q) {
  const int e = ri.start + fe / g.f_nb;
  const int fn = fe % g.f_nb;
  const int nb = g.nbr[(int64_t)e * g.nbr_stride + g.nbr_offset];
  q = nb * g.f_nb + fn;
  float xn[3], rn[9];
  load_geom_record(buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64), q, xn, rn);
  if (!g.transposed)
    edge_descriptor(xn, rn, yc, rc, rho, d);
  else
    edge_descriptor(yc, rc, xn, rn, rho, d);
  d[9] = 1.0f;
}

// B operand of the kernel MLP: [A; beta; 0...] as a 16 x 32 matrix, lane (kcol, h) holds rows 8h..8h+7.
__device__ __forceinline__ void load_mlp_weights(const float* __restrict__ axes_ext, int kcol, int h, u32x4& b_hi,
                                                 u32x4& b_lo) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * h + j;
    v[j] = k < kDescExt ? kGeluIn * axes_ext[k * kBasis + kcol] : 0.f;  // kGeluIn * pre, see gelu_scaled
  }
  frags_from_floats(v, b_hi, b_lo);
}

// pre[n, k] for the 32 frame-edges of the chunk (lane n = lane & 31 supplies its descriptor).
__device__ __forceinline__ f32x16 mlp_preactivation(const float d[kDescExt], int h, u32x4 b_hi, u32x4 b_lo) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = h ? (j < 2 ? d[8 + j] : 0.f) : d[j];
  u32x4 a_hi, a_lo;
  frags_from_floats(v, a_hi, a_lo);
  return mfma_bf16x3(a_hi, a_lo, b_hi, b_lo, zero16());
}

// ------------------------------------------------------------------------------------------------
// T[m, c, k] = sum_n feat[q(n), c] * GELU(desc(n) . A + beta)[k];   feat and T are packed words.
//
// One wavefront owns FC (1 or 2) frames of one centre point, i.e. FC output rows that share their
// neighbour list: the gather of the neighbours' geometry, the source-row lookups and the gathered
// feature fragments (the MFMA A operand) are done once and used for both rows.  With FC = 2 lane
// half h computes the descriptor against centre frame a0 + h, so no descriptor is computed twice:
// for row a the half h == a supplies descriptor dims 0..7 of the MLP's k-dimension and the other
// half supplies dims 8, 9 (the bias slot), its dim-8 value arriving through one v_permlane32_swap;
// the MLP weights are held in both arrangements.
// VALU is the bound of this kernel (MFMA ~15 % busy), so everything here is about instruction count:
// 32-bit buffer addressing, ds_bpermute for the per-row source offsets, branch-free GELU.
// ------------------------------------------------------------------------------------------------
template <int VW, int FC, bool FULL, bool T24 = false>
__global__ __launch_bounds__(256, VW == 4 ? 2 : (FC == 1 ? 3 : 2)) void edge_t_bf16_kernel(EdgeGeom g, const uint32_t* __restrict__ feat, int channels,
                                                             int64_t feat_rows, const float* __restrict__ axes_ext,
                                                             const float* __restrict__ rho_p,
                                                             uint32_t* __restrict__ t_out, int64_t n_items,
                                                             int fnb_shift, int64_t item_lo) {  // items item_lo .. n_items - 1
  static_assert(!T24 || VW <= 2, "3-byte rows need the two channels of a pair in one lane (edge_bf16_body.h)");
  __shared__ __attribute__((aligned(16))) uint32_t lds_w[FC][2][64][4];
  if (threadIdx.x < 64) mlp_weights_to_lds<FC>(lds_w, axes_ext, threadIdx.x);
  __syncthreads();
  const int64_t item = __builtin_amdgcn_readfirstlane((int)(item_lo + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (item >= n_items) return;
  const __amdgpu_buffer_rsrc_t feat_rs = buffer_of(feat, feat_rows * channels * 4);
  const int row_words = channels * kBasis;
  const int kcol = threadIdx.x & 31;
  if constexpr (T24) {  // 3-byte rows (common.h): channels ch0, ch0 + 1 of this lane = one hi word + one lo half-word
    char* rows = reinterpret_cast<char*>(t_out) + item * FC * t24_row_bytes(channels);
    edge_item_bf16<VW, FC, FULL>(g, feat_rs, channels, lds_w, *rho_p, item, fnb_shift,
                                 [&](int a, int ch0, int, float x0, float x1, bool ok0, bool ok1) {
                                   if (!ok0) return;  // channels is even and ch0 is: ok1 == ok0
                                   uint32_t hp, lp;
                                   t24_pack2(x0, ok1 ? x1 : 0.f, hp, lp);
                                   char* row = rows + a * t24_row_bytes(channels);
                                   const int idx = (ch0 >> 1) * kBasis + kcol;
                                   __builtin_nontemporal_store(hp, reinterpret_cast<uint32_t*>(row) + idx);
                                   __builtin_nontemporal_store((uint16_t)lp, reinterpret_cast<uint16_t*>(row + (int64_t)row_words * 2) + idx);
                                 });
  } else {
    uint32_t* t_rows = t_out + item * FC * (int64_t)channels * kBasis;  // rows FC*item .. FC*item + FC-1
    edge_item_bf16<VW, FC, FULL>(g, feat_rs, channels, lds_w, *rho_p, item, fnb_shift,
                                 [&](int a, int ch0, int ch1, float x0, float x1, bool ok0, bool ok1) {
                                   uint32_t w0, w1;
                                   split_pack2(x0, x1, w0, w1);
                                   if (ok0) __builtin_nontemporal_store(w0, &t_rows[a * row_words + ch0 * kBasis + kcol]);
                                   if (ok1) __builtin_nontemporal_store(w1, &t_rows[a * row_words + ch1 * kBasis + kcol]);
                                 });
  }
}

// ------------------------------------------------------------------------------------------------
// Wave-pair variant (two frames, C >= 64): the two output rows of an item are produced by a 128-thread
// workgroup.  Wavefront v builds the descriptors, kernel MLP, GELU and hi/lo split for frame a0+v only and
// publishes those B fragments through LDS (double buffered, one barrier per chunk); it then aggregates CT
// tiles of 32 channels for BOTH frames.  No value is computed twice, and each wavefront carries 32*CT
// accumulator registers, which is what lets 3-4 wavefronts share a SIMD (the single-wavefront kernel needs
// ~240 VGPRs: 2 per SIMD, 44 % of their life parked in s_waitcnt).  CT = 1: 64 channels per pass (128 VGPRs,
// 4 waves/SIMD); CT = 2: 128 channels per pass.  Wider rows take several passes, each recomputing phi (the MFMA
// work per pass grows with CT, the GELU work does not).  FULL: C is a multiple of 64*CT.
// ------------------------------------------------------------------------------------------------
#ifndef SE3_PAIR_WAVES
#define SE3_PAIR_WAVES 4
#endif
#ifndef SE3_PAIR2_WAVES
#define SE3_PAIR2_WAVES 2  // wavefronts per SIMD of the two-tile form (rows of 128 channels and up).  At 3 (168 VGPRs) it spilled
                           // 48 - 64 bytes per lane to scratch and was 1 % slower (dfaust_f2 stack 2.02 vs 2.00 ms, a 128-channel
                           // layer 4.67 vs 4.63 ms, profiles/r04_pair2_waves_ab.txt); no kernel the shipped configurations launch uses scratch now
#endif
// (The ablation builds of rounds 2 - 5 -- SE3_PAIR_ABLATE, SE3_PG_ABLATE, SE3_ABLATE: kernels with one ingredient taken out,
// wrong results, for the "what bounds them" tables of profiles/README.md -- were removed in round 6 with the questions they
// answered; the commits that produced a table hold the code that produced it.)
#ifndef SE3_PAIR_PIN
#define SE3_PAIR_PIN 1  // centre record passed through an empty asm at the top of every chunk: nothing derived from it is hoisted out
                        // of the loop (fewer live registers, another schedule).  Measured -2 % (0.366 / 0.360 -> 0.359 / 0.353 ms)
#endif
// POW2: fnb_shift >= 0 is known (no division path, no branch on it).  TR: 0 forward, 1 transposed pass, -1 decided by
// g.transposed at run time (both descriptor paths in the loop)
template <int CT, bool FULL, int NF, bool POW2 = false, int TR = -1>
__global__ __launch_bounds__(128, CT == 1 ? (POW2 ? SE3_PAIR_WAVES : 3) : (FULL ? SE3_PAIR2_WAVES : 2)) void edge_t_pair_bf16_kernel(
    EdgeGeom g, const uint32_t* __restrict__ feat, int C, int64_t feat_rows, const float* __restrict__ axes_ext,
    const float* __restrict__ rho_p, uint32_t* __restrict__ t_out, int64_t item_lo, int64_t n_items, int fnb_shift,
    int t24) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_w[1][2][64][4];
  __shared__ __attribute__((aligned(16))) uint32_t lds_phi[2][2][2][2][64][4];  // [buffer][frame][k-step][hi/lo][lane]
  TL(__shared__ uint32_t tl_lds[2][kTlWords]; uint32_t* tl_rec = tl_lds[threadIdx.x >> 6]; tl_rec[threadIdx.x & 63] = 0u; tl_mark(tl_rec, 0); int tl_ch = 16;)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kcol = lane & 31, h = lane >> 5;
  if (threadIdx.x < 64) mlp_weights_to_lds<1>(lds_w, axes_ext, threadIdx.x);
  __syncthreads();
  TL(tl_mark(tl_rec, 1);)  // [0] entry, [1] arguments, MLP weights into LDS, barrier
  const float rho = *rho_p;
  const int row_bytes = C * 4;
  const __amdgpu_buffer_rsrc_t feat_rs = buffer_of(feat, feat_rows * row_bytes);
  const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
  const __amdgpu_buffer_rsrc_t ctrg_rs = buffer_of(g.ctr_geom, g.n_ctr * g.f_ctr * 64);
  // NF = 2: item = two consecutive frames of a point, wavefront v owns frame a0 + v.  NF = 1: item = one row (any
  // F); both wavefronts hold that frame, wavefront v does the GELU of k-step v only (16 of the 32 frame-edges).
  const int groups = g.f_ctr / NF;
  const int hb = 16 * h;
  int buf = 0;
  // One item per workgroup (the grid-stride loop runs once; resident workgroups are the chunk-stream kernel below).
  // (claiming the next item from a device counter instead of striding was measured in round 4: 65 536 returning atomics on
  // one word take 0.74 ms by themselves, profiles/r04_vmem_diet_ab.txt)
  for (int64_t item = item_lo + blockIdx.x; item < n_items; item += gridDim.x) {  // items item_lo .. n_items-1
  // rows < 2^31 (checked on the host), so 32-bit unsigned division is exact -- the 64-bit one is ~150 scalar instructions
  const int64_t ctr = (uint32_t)item / (uint32_t)groups;
  const int a0 = (int)((uint32_t)item - (uint32_t)ctr * (uint32_t)groups) * NF;
  const int start = ctr > 0 ? g.ends[ctr - 1] : 0;
  const int n_total = (g.ends[ctr] - start) * g.f_nb;
  float yc[3], rc[9];
  load_geom_record(ctrg_rs, (int)(ctr * g.f_ctr + a0 + (NF == 2 ? wv : 0)), yc, rc);  // this wavefront's frame
  TL(tl_mark(tl_rec, 2); asm volatile("" ::"s"(n_total)); tl_mark(tl_rec, 3); tl_rec[15] = (uint32_t)n_total;)  // [2] item set-up issued, [3] row extents in

  // Neighbour ids are fetched two chunks ahead and the geometry records one chunk ahead, so that no load result is
  // needed in the chunk that issues it (the dependent chain ids -> record/feature rows costs one memory latency
  // per link; a wavefront lives for ~2.5 chunks only).  Frame-edge indices past the end clamp to the last one.
  auto nbr_of = [&](int c0) {
    const int fe = min(c0 + kcol, n_total - 1);
    const int e = start + (POW2 || fnb_shift >= 0 ? fe >> fnb_shift : fe / g.f_nb);
    return g.nbr[(int64_t)e * g.nbr_stride + g.nbr_offset];
  };
  auto row_of = [&](int nb, int c0) {
    const int fe = min(c0 + kcol, n_total - 1);
    return (POW2 ? nb << fnb_shift : nb * g.f_nb) + (POW2 || fnb_shift >= 0 ? fe & ((1 << fnb_shift) - 1) : fe % g.f_nb);
  };

  for (int cbase = 0; cbase < C; cbase += 64 * CT) {
    // this wavefront aggregates channels cbase + 32*(CT*wv + t) + kcol, t = 0..CT-1
    int cb4[CT];
    bool ch_ok[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int ch = cbase + 32 * (CT * wv + t) + kcol;
      ch_ok[t] = FULL || ch < C;
      cb4[t] = ch * 4;
    }
    f32x16 acc[NF][CT];  // [frame][tile]
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int t = 0; t < CT; ++t) acc[a][t] = zero16();
    int nb_b = 0, q_a = 0;
    float xn_nx[3], rn_nx[9];
    if (n_total > 0) {
      const int nb_a = nbr_of(0);
      nb_b = nbr_of(32);
      TL(tl_mark(tl_rec, 4); tl_pin(nb_a); tl_mark(tl_rec, 5);)  // [4] id loads issued, [5] ids of chunk 0 in
      q_a = row_of(nb_a, 0);
      load_geom_record(nbg_rs, q_a, xn_nx, rn_nx);
    }
    TL(tl_mark(tl_rec, 6);)  // [6] chunk loop entered
    for (int c0 = 0; c0 < n_total; c0 += 32, buf ^= 1) {
      const int cnt = min(32, n_total - c0);
      // rows past the end of the neighbour list read out of bounds (buffer loads return 0): their phi needs no mask
      const int qoff = c0 + kcol < n_total ? q_a * row_bytes : kOobOffset;
      float xn[3], rn[9], d[9];
      // SE3_PAIR_PIN: the centre's record stays the 12 values it is -- nothing derived from it leaves the loop
#if SE3_PAIR_PIN
#pragma unroll
      for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(yc[i]));
#pragma unroll
      for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(rc[i]));
#endif
#pragma unroll
      for (int i = 0; i < 3; ++i) xn[i] = xn_nx[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) rn[i] = rn_nx[i];
      const int q_b = row_of(nb_b, c0 + 32);
      nb_b = nbr_of(c0 + 64);

      // gathered feature words for this wavefront's channels (shared by both frames): all loads go out now and are
      // only turned into MFMA fragments after the barrier below
      uint32_t fw[CT][2][8];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int src_off = __builtin_amdgcn_ds_bpermute(hb + 4 * acc_row(8 * s + j, 0), qoff);
#pragma unroll
          for (int t = 0; t < CT; ++t)
            fw[t][s][j] = __builtin_amdgcn_raw_buffer_load_b32(feat_rs, ch_ok[t] ? src_off + cb4[t] : kOobOffset, 0, 0);
        }
      load_geom_record(nbg_rs, q_b, xn_nx, rn_nx);
      q_a = q_b;
      // segments of a chunk: gathers issued | this chunk's record in | descriptor + kernel MLP | GELU + split + publish |
      // barrier | feature words of k-step 0 in | aggregation issued
      // (record words 16 + 8 j + i: stamp i of chunk j, chunks 0 .. 5; stamp 7 = the chunk's end)
      TL(tl_mark(tl_rec, tl_ch + 1); tl_pin(xn[0]); tl_pin(rn[7]); tl_pin(rn[8]); tl_mark(tl_rec, tl_ch + 2);)

      if (TR == 0 || (TR < 0 && !g.transposed))
        edge_descriptor(xn, rn, yc, rc, rho, d);
      else
        edge_descriptor(yc, rc, xn, rn, rho, d);

      // kernel MLP + GELU for this wavefront's frame; both lane halves hold the same descriptor: half 0 feeds
      // dims 0..7, half 1 dims 8, 9
      {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = h ? (j == 0 ? d[8] : (j == 1 ? 1.0f : 0.f)) : d[j];
        u32x4 a_hi, a_lo;
        frags_from_floats(v, a_hi, a_lo);
        const u32x4 wb_hi = *reinterpret_cast<const u32x4*>(&lds_w[0][0][lane][0]);
        const u32x4 wb_lo = *reinterpret_cast<const u32x4*>(&lds_w[0][1][lane][0]);
        const f32x16 phi = mfma_bf16x3(a_hi, a_lo, wb_hi, wb_lo, zero16());
        TL(tl_pin(phi[0]); tl_mark(tl_rec, tl_ch + 3);)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s * 16 < cnt && (NF == 2 || s == wv)) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = gelu_scaled(phi[8 * s + j]);
            u32x4 b_hi, b_lo;
            frags_from_floats(pv, b_hi, b_lo);
            *reinterpret_cast<u32x4*>(&lds_phi[buf][NF == 2 ? wv : 0][s][0][lane][0]) = b_hi;
            *reinterpret_cast<u32x4*>(&lds_phi[buf][NF == 2 ? wv : 0][s][1][lane][0]) = b_lo;
          }
        }
      }
      TL(tl_mark(tl_rec, tl_ch + 4);)
      __syncthreads();  // both frames' fragments of this chunk are published (other buffer is used next chunk)
      TL(tl_mark(tl_rec, tl_ch + 5); tl_pin(fw[0][0][7]); tl_mark(tl_rec, tl_ch + 6);)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s * 16 < cnt) {
          u32x4 b_hi[NF], b_lo[NF];
#pragma unroll
          for (int a = 0; a < NF; ++a) {
            b_hi[a] = *reinterpret_cast<const u32x4*>(&lds_phi[buf][a][s][0][lane][0]);
            b_lo[a] = *reinterpret_cast<const u32x4*>(&lds_phi[buf][a][s][1][lane][0]);
          }
#pragma unroll
          for (int t = 0; t < CT; ++t) {
            u32x4 fa_hi, fa_lo;
            frags_from_words(fw[t][s], fa_hi, fa_lo);
#pragma unroll
            for (int a = 0; a < NF; ++a) acc[a][t] = mfma_bf16x3(fa_hi, fa_lo, b_hi[a], b_lo[a], acc[a][t]);
          }
        }
      }
      TL(tl_mark(tl_rec, tl_ch + 7); tl_ch += 8;)
    }
    TL(tl_mark(tl_rec, 7); tl_pin(acc[NF - 1][CT - 1][15]); tl_mark(tl_rec, 8);)  // [7] loop left, [8] accumulators in
    // acc[a][t] register r, lane (kcol, h) = T[row NF*item + a][cbase + 32*(CT*wv + t) + acc_row(r,h)][kcol]
#pragma unroll
    for (int a = 0; a < NF; ++a)
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const int ch0 = cbase + 32 * (CT * wv + t);
        if (t24 == 2) {  // 2.25-byte rows (common.h, T16): registers 4j .. 4j+3 of this lane = 4 consecutive channels = one block
          if (!FULL && ch0 >= C) continue;  // C is a multiple of 64 (host): a 32-channel tile is inside the row or past it
          char* row = reinterpret_cast<char*>(t_out) + (item * NF + a) * t16_row_bytes(C);
          uint8_t* expo = reinterpret_cast<uint8_t*>(row) + (int64_t)C * kBasis * 2;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cq = (ch0 >> 2) + 2 * j + h;  // channel quad of registers 4j .. 4j+3 (acc_row(4j + i, h) = 8j + 4h + i)
            uint32_t m01, m23, eb;
            t16_pack4(acc[a][t][4 * j], acc[a][t][4 * j + 1], acc[a][t][4 * j + 2], acc[a][t][4 * j + 3], m01, m23, eb);
            typedef uint32_t u32x2s __attribute__((ext_vector_type(2)));
            // one instruction = 64 lanes x 8 bytes = 512 consecutive bytes of the row (quads cq, cq + 1 over all 32 k)
            __builtin_nontemporal_store(u32x2s{m01, m23}, reinterpret_cast<u32x2s*>(row) + cq * kBasis + kcol);
            __builtin_nontemporal_store((uint8_t)eb, expo + t16_exp_pos(cq * kBasis + kcol));
          }
          continue;
        }
        if (t24) {  // 3-byte rows (common.h): channels c, c+1 of this lane = one hi word + one lo half-word
          char* row = reinterpret_cast<char*>(t_out) + (item * NF + a) * t24_row_bytes(C);
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int ch = ch0 + acc_row(r, h);  // even
            if (!FULL && ch >= C) continue;
            uint32_t hp, lp;
            t24_pack2(acc[a][t][r], (FULL || ch + 1 < C) ? acc[a][t][r + 1] : 0.f, hp, lp);
            const int idx = (ch >> 1) * kBasis + kcol;
            __builtin_nontemporal_store(hp, reinterpret_cast<uint32_t*>(row) + idx);
            __builtin_nontemporal_store((uint16_t)lp, reinterpret_cast<uint16_t*>(row + (int64_t)C * kBasis * 2) + idx);
          }
          continue;
        }
        uint32_t* t_row = t_out + ((item * NF + a) * (int64_t)C + ch0) * kBasis;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          uint32_t w0, w1;
          split_pack2(acc[a][t][r], acc[a][t][r + 1], w0, w1);
          if (FULL || ch0 + acc_row(r, h) < C) __builtin_nontemporal_store(w0, &t_row[acc_row(r, h) * kBasis + kcol]);
          if (FULL || ch0 + acc_row(r + 1, h) < C) __builtin_nontemporal_store(w1, &t_row[acc_row(r + 1, h) * kBasis + kcol]);
        }
      }
    if (cbase + 64 * CT < C) __syncthreads();  // the next pass reuses the phi buffers from their start
  }
  TL(tl_mark(tl_rec, 9); __builtin_amdgcn_s_waitcnt(0); tl_mark(tl_rec, 10);   // [9] stores issued, [10] stores retired
     tl_rec[11] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); tl_rec[12] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
     tl_rec[13] = (uint32_t)item; tl_rec[14] = (uint32_t)((tl_ch - 16) >> 3); tl_flush(TR == 1 ? 1 : 0, (uint32_t)(item - item_lo) * 2 + wv, tl_rec);)
  }
}

// ------------------------------------------------------------------------------------------------
// Chunk-stream form of the wave-pair kernel (round 6; 64-channel rows, two frames per item, power-of-two neighbour
// frame count, 3-byte rows).  The timeline of the form above (profiles/r06_edge_timeline.txt) shows what a workgroup that
// lives for one item -- two to three chunks -- spends outside its chunks: 37 % of a wavefront's life is the chain of
// exposed round trips in front of the first chunk (kernel arguments and MLP weights, the row's extents, its first ids,
// its first record), and a wavefront slot then idles ~2 300 cycles until the dispatcher has placed the next workgroup.
// Here a workgroup is resident for the whole launch and walks the chunks of its items as ONE stream: the software
// pipeline of the chunk loop (ids two chunks ahead, records one chunk ahead) runs across item boundaries, the row extents
// of 64 items at a time sit in two registers (lane j = the workgroup's j-th item, read with v_readlane: no memory round
// trip per item), and the centre's record is wave-uniform and comes through the scalar cache into SGPRs one chunk ahead
// (12 VGPRs less, no vector-memory instruction).  An item boundary costs the pack + stores of the two finished rows and
// nothing else.  Items go to workgroups round-robin (item_lo + blockIdx.x + j * gridDim.x): neighbouring workgroups read
// neighbouring extents and write neighbouring rows, and a cloud's dense regions are spread over all of them.
// Rows without neighbours are one chunk of zero frame-edges (every lane reads out of bounds: zero rows are stored).
// ------------------------------------------------------------------------------------------------
struct ChunkCursor {
  int j;        // local item index; n_mine = past the end
  int c0;       // first frame-edge of the chunk
  int start;    // first edge of the item's centre point
  int n_total;  // frame-edges of the item
  int crow;     // this wavefront's centre row (record index)
  uint32_t item;
};

template <int TR>
__global__ __launch_bounds__(128, SE3_PAIR_WAVES) void edge_t_stream_bf16_kernel(
    EdgeGeom g, const uint32_t* __restrict__ feat, int64_t feat_rows, const float* __restrict__ axes_ext,
    const float* __restrict__ rho_p, char* __restrict__ t_out, uint32_t item_lo, uint32_t item_hi, int fnb_shift) {
  constexpr int C = 64, row_bytes = C * 4;
  __shared__ __attribute__((aligned(16))) uint32_t lds_w[1][2][64][4];
  __shared__ __attribute__((aligned(16))) uint32_t lds_phi[2][2][2][2][64][4];  // [buffer][frame][k-step][hi/lo][lane]
  // (timeline build: entry / exit stamps and the chunk count of every wavefront -- how evenly the resident workgroups finish)
  TL(__shared__ uint32_t tl_lds[2][kTlWords]; uint32_t* tl_rec = tl_lds[threadIdx.x >> 6]; tl_rec[threadIdx.x & 63] = 0u; tl_mark(tl_rec, 0); uint32_t tl_chunks = 0;)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kcol = lane & 31, h = lane >> 5;
  if (threadIdx.x < 64) mlp_weights_to_lds<1>(lds_w, axes_ext, threadIdx.x);
  __syncthreads();
  const float rho = *rho_p;
  const __amdgpu_buffer_rsrc_t feat_rs = buffer_of(feat, feat_rows * row_bytes);
  const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
  const __amdgpu_buffer_rsrc_t nbr_rs = buffer_of(g.nbr, g.n_edges * g.nbr_stride * 4);  // ids past the list read 0
  const uint32_t groups = (uint32_t)g.f_ctr / 2u;
  const uint32_t item0_wg = item_lo + blockIdx.x;
  const int n_mine = (int)((item_hi - item0_wg + gridDim.x - 1) / gridDim.x);  // >= 1: the grid has at most one workgroup per item
  const int fmask = (1 << fnb_shift) - 1;
  const int hb = 16 * h;

  f32x16 acc[2] = {zero16(), zero16()};
  int buf = 0;
  int prio_step = wave_slot_id();
  // The workgroup's items in windows of 64: the row extents of a window sit in two registers (lane l = the window's l-th
  // item), the chunk pipeline is drained and restarted between two windows (one exposed round trip per 64 items).  A
  // level of up to 64 items per resident workgroup -- 131 072 items, the headline's 65 536 among them -- is one window.
  const int n_all = n_mine;
  for (int win0 = 0; win0 < n_all; win0 += 64) {
  const int n_mine = min(64, n_all - win0);  // (the loop body below sees one window as "its" items)
  const uint32_t item0 = item0_wg + (uint32_t)win0 * gridDim.x;
  int v_lo, v_hi;
  {
    const uint32_t item = item0 + (uint32_t)min(lane, n_mine - 1) * gridDim.x;
    const uint32_t ctr = item / groups;
    v_hi = g.ends[ctr];
    v_lo = g.ends[max((int)ctr - 1, 0)];
    if (ctr == 0) v_lo = 0;
  }
  // (every cursor field is wave-uniform; readfirstlane says so to the compiler: scalar registers, scalar branches)
  auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  auto enter = [&](int j, int crow_keep, uint32_t item_keep) {  // first chunk of local item j, or the end mark
    ChunkCursor c;
    c.j = j, c.c0 = 0, c.start = 0, c.n_total = 0;
    c.crow = crow_keep, c.item = item_keep;  // end mark: the last item's record stays the (valid) prefetch target
    if (j < n_mine) {
      const int lo = __builtin_amdgcn_readlane(v_lo, j), hi = __builtin_amdgcn_readlane(v_hi, j);
      c.start = lo, c.n_total = (hi - lo) << fnb_shift;
      c.item = item0 + (uint32_t)j * gridDim.x;
      const uint32_t ctr = c.item / groups;
      c.crow = uni((int)(ctr * (uint32_t)g.f_ctr + (c.item - ctr * groups) * 2u) + wv);
    }
    return c;
  };
  auto advance = [&](const ChunkCursor& c) {
    ChunkCursor r = c;
    if (c.c0 + 32 < c.n_total) r.c0 = c.c0 + 32;
    else if (c.j < n_mine) r = enter(c.j + 1, c.crow, c.item);
    r.j = uni(r.j), r.c0 = uni(r.c0), r.start = uni(r.start), r.n_total = uni(r.n_total), r.item = (uint32_t)uni((int)r.item);
    return r;
  };
  auto fe_of = [&](const ChunkCursor& c) { return max(min(c.c0 + kcol, c.n_total - 1), 0); };
  auto nbr_of = [&](const ChunkCursor& c) {
    const int e = c.start + (fe_of(c) >> fnb_shift);
    return (int)__builtin_amdgcn_raw_buffer_load_b32(nbr_rs, (e * g.nbr_stride + g.nbr_offset) * 4, 0, 0);
  };
  auto row_of = [&](int nb, const ChunkCursor& c) { return (nb << fnb_shift) + (fe_of(c) & fmask); };
  // the centre's record is wave-uniform: read through the scalar cache (constant address space + a uniform index = s_load)
  typedef const f32x4 __attribute__((address_space(4))) * crec_t;
  const crec_t ctr_rec = (crec_t)(uintptr_t)g.ctr_geom;
  auto centre = [&](const ChunkCursor& c, float yc[3], float rc[9]) {
    const int row = c.crow;
    const f32x4 v0 = ctr_rec[row * 4], v1 = ctr_rec[row * 4 + 1], v2 = ctr_rec[row * 4 + 2];
    yc[0] = v0[0], yc[1] = v0[1], yc[2] = v0[2], rc[8] = v0[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) rc[i] = v1[i], rc[4 + i] = v2[i];
  };

  ChunkCursor cur, n1, n2;
  cur = enter(0, 0, 0u);
  n1 = advance(cur);
  n2 = advance(n1);
  // carried from chunk to chunk: the source rows of this chunk and of the next (ids consumed), this chunk's record
  int q_cur, q_n1;
  float xn_nx[3], rn_nx[9];
  {
    const int nb_cur = nbr_of(cur);
    const int nb_n1 = nbr_of(n1);
    q_cur = row_of(nb_cur, cur);
    load_geom_record(nbg_rs, q_cur, xn_nx, rn_nx);
    q_n1 = row_of(nb_n1, n1);
    // (consumed in front of the loop like every chunk's loads are at its end, see below: the loop is entered with no load in flight)
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(xn_nx[i]));
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(rn_nx[i]));
  }
  float yc[3], rc[9];
  centre(cur, yc, rc);

  while (cur.j < n_mine) {
    rotate_priority(prio_step++);
    const int cnt = min(32, cur.n_total - cur.c0);  // <= 0: a row without neighbours
    // rows past the end of the neighbour list read out of bounds (buffer loads return 0): their phi needs no mask
    const int qoff = cur.c0 + kcol < cur.n_total ? q_cur * row_bytes : kOobOffset;
    float xn[3], rn[9], d[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) xn[i] = xn_nx[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) rn[i] = rn_nx[i];
    const int nb_n2 = nbr_of(n2);  // ids two chunks ahead

    // gathered feature words for this wavefront's channels (shared by both frames): all loads go out now and are only
    // turned into MFMA fragments after the barrier below
    uint32_t fw[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int src_off = __builtin_amdgcn_ds_bpermute(hb + 4 * acc_row(8 * s + j, 0), qoff);
        fw[s][j] = __builtin_amdgcn_raw_buffer_load_b32(feat_rs, src_off + (32 * wv + kcol) * 4, 0, 0);
      }
    load_geom_record(nbg_rs, q_n1, xn_nx, rn_nx);  // record one chunk ahead
    float yn[3], rnc[9];
    centre(n1, yn, rnc);  // the next chunk's centre (the same record until the item changes)

    if (TR == 0)
      edge_descriptor(xn, rn, yc, rc, rho, d);
    else
      edge_descriptor(yc, rc, xn, rn, rho, d);

    // kernel MLP + GELU for this wavefront's frame; both lane halves hold the same descriptor: half 0 feeds dims 0..7,
    // half 1 dims 8, 9
    {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = h ? (j == 0 ? d[8] : (j == 1 ? 1.0f : 0.f)) : d[j];
      u32x4 a_hi, a_lo;
      frags_from_floats(v, a_hi, a_lo);
      const u32x4 wb_hi = *reinterpret_cast<const u32x4*>(&lds_w[0][0][lane][0]);
      const u32x4 wb_lo = *reinterpret_cast<const u32x4*>(&lds_w[0][1][lane][0]);
      const f32x16 phi = mfma_bf16x3(a_hi, a_lo, wb_hi, wb_lo, zero16());
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s * 16 < cnt) {
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pv[j] = gelu_scaled(phi[8 * s + j]);
          u32x4 b_hi, b_lo;
          frags_from_floats(pv, b_hi, b_lo);
          *reinterpret_cast<u32x4*>(&lds_phi[buf][wv][s][0][lane][0]) = b_hi;
          *reinterpret_cast<u32x4*>(&lds_phi[buf][wv][s][1][lane][0]) = b_lo;
        }
      }
    }
    __syncthreads();  // both frames' fragments of this chunk are published (the other buffer is used by the next chunk)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (s * 16 < cnt) {
        u32x4 b_hi[2], b_lo[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          b_hi[a] = *reinterpret_cast<const u32x4*>(&lds_phi[buf][a][s][0][lane][0]);
          b_lo[a] = *reinterpret_cast<const u32x4*>(&lds_phi[buf][a][s][1][lane][0]);
        }
        u32x4 fa_hi, fa_lo;
        frags_from_words(fw[s], fa_hi, fa_lo);
#pragma unroll
        for (int a = 0; a < 2; ++a) acc[a] = mfma_bf16x3(fa_hi, fa_lo, b_hi[a], b_lo[a], acc[a]);
      }
    }
    // Every load this chunk issued is consumed HERE, in front of the stores: gfx9 counts loads and stores in one in-order
    // counter, so a wait behind the stores for a load issued before them would be a wait for the stores (and the
    // compiler's counts at the loop top must hold for the path without stores: they would drain them).  The next wait
    // behind the stores is for the next chunk's feature words, a few thousand cycles away.
    q_cur = q_n1;
    q_n1 = row_of(nb_n2, n2);
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(xn_nx[i]));
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(rn_nx[i]));
    if (n1.j != cur.j) {
      // the item's last chunk: acc[a] register r, lane (kcol, h) = T[row 2 item + a][32 wv + acc_row(r, h)][kcol] leaves
      // as 3-byte rows (channels c, c + 1 of this lane = one hi word + one lo half-word), non-temporal
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        char* row = t_out + ((int64_t)cur.item * 2 + a) * t24_row_bytes(C);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          uint32_t hp, lp;
          t24_pack2(acc[a][r], acc[a][r + 1], hp, lp);
          const int idx = ((32 * wv + acc_row(r, h)) >> 1) * kBasis + kcol;
          __builtin_nontemporal_store(hp, reinterpret_cast<uint32_t*>(row) + idx);
          __builtin_nontemporal_store((uint16_t)lp, reinterpret_cast<uint16_t*>(row + (int64_t)C * kBasis * 2) + idx);
        }
        acc[a] = zero16();
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) yc[i] = yn[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) rc[i] = rnc[i];
    cur = n1, n1 = n2, n2 = advance(n2);
    buf ^= 1;
    TL(++tl_chunks;)
  }
  }  // windows
  TL(tl_mark(tl_rec, 9); __builtin_amdgcn_s_waitcnt(0); tl_mark(tl_rec, 10);
     tl_rec[11] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); tl_rec[12] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
     tl_rec[14] = tl_chunks; tl_rec[15] = (uint32_t)n_all; tl_flush(TR == 1 ? 1 : 0, blockIdx.x * 2 + wv, tl_rec);)
}

// ------------------------------------------------------------------------------------------------
// Chunk-stream form of the single-wavefront kernel (round 6; rows of 32 * VW channels in one pass, power-of-two neighbour
// frame count, 3-byte rows; launched for <VW, FC> = <1, 2> -- 32-channel rows, two frames per item -- the other forms
// measured no better than their one-item kernels, see launch_edge_t_bf16).  Same idea as the
// wave-pair stream above, and simpler: the four wavefronts of a workgroup share nothing but the MLP weights in LDS, so
// each walks the chunks of ITS items (item_lo + wavefront index + j * wavefronts of the grid) with no barrier in the loop.
// FC = 2: lane half h builds descriptors against centre frame a0 + h, whose record it holds in registers -- selected from
// the two records that came through the scalar cache one chunk ahead; FC = 1: the centre record stays in SGPRs.
// ------------------------------------------------------------------------------------------------
template <int VW, int FC, int TR>
__global__ __launch_bounds__(256, (VW == 1 && FC == 1) ? 4 : 3) void edge_t_stream1_bf16_kernel(
    EdgeGeom g, const uint32_t* __restrict__ feat, int64_t feat_rows, const float* __restrict__ axes_ext,
    const float* __restrict__ rho_p, char* __restrict__ t_out, uint32_t item_lo, uint32_t item_hi, int fnb_shift) {
  constexpr int C = 32 * VW, row_bytes = C * 4;
  __shared__ __attribute__((aligned(16))) uint32_t lds_w[FC][2][64][4];
  const int lane = threadIdx.x & 63;
  const int kcol = lane & 31, h = lane >> 5;
  if (threadIdx.x < 64) mlp_weights_to_lds<FC>(lds_w, axes_ext, threadIdx.x);
  __syncthreads();
  const float rho = *rho_p;
  const __amdgpu_buffer_rsrc_t feat_rs = buffer_of(feat, feat_rows * row_bytes);
  const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
  const __amdgpu_buffer_rsrc_t nbr_rs = buffer_of(g.nbr, g.n_edges * g.nbr_stride * 4);  // ids past the list read 0
  const uint32_t groups = (uint32_t)g.f_ctr / (uint32_t)FC;
  const uint32_t n_waves = gridDim.x * 4u;
  const uint32_t item0_wave = item_lo + (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
  // (a wavefront past the end of the range -- the last workgroup's -- has no items and falls through both loops)
  const int n_all = item0_wave < item_hi ? (int)((item_hi - item0_wave + n_waves - 1) / n_waves) : 0;
  const int fmask = (1 << fnb_shift) - 1;
  const int hb = 16 * h;
  const int cb4 = VW * kcol * 4;

  f32x16 acc[FC][VW];
#pragma unroll
  for (int a = 0; a < FC; ++a)
#pragma unroll
    for (int t = 0; t < VW; ++t) acc[a][t] = zero16();
  int prio_step = wave_slot_id();
  for (int win0 = 0; win0 < n_all; win0 += 64) {  // windows of 64 items, as above
  const int n_mine = min(64, n_all - win0);
  const uint32_t item0 = item0_wave + (uint32_t)win0 * n_waves;
  int v_lo, v_hi;
  {
    const uint32_t item = item0 + (uint32_t)min(lane, n_mine - 1) * n_waves;
    const uint32_t ctr = item / groups;
    v_hi = g.ends[ctr];
    v_lo = g.ends[max((int)ctr - 1, 0)];
    if (ctr == 0) v_lo = 0;
  }
  auto uni = [](int x) { return __builtin_amdgcn_readfirstlane(x); };
  auto enter = [&](int j, int crow_keep, uint32_t item_keep) {  // first chunk of local item j, or the end mark
    ChunkCursor c;
    c.j = j, c.c0 = 0, c.start = 0, c.n_total = 0;
    c.crow = crow_keep, c.item = item_keep;
    if (j < n_mine) {
      const int lo = __builtin_amdgcn_readlane(v_lo, j), hi = __builtin_amdgcn_readlane(v_hi, j);
      c.start = lo, c.n_total = (hi - lo) << fnb_shift;
      c.item = item0 + (uint32_t)j * n_waves;
      const uint32_t ctr = c.item / groups;
      c.crow = uni((int)(ctr * (uint32_t)g.f_ctr + (c.item - ctr * groups) * (uint32_t)FC));  // the item's first centre row
    }
    return c;
  };
  auto advance = [&](const ChunkCursor& c) {
    ChunkCursor r = c;
    if (c.c0 + 32 < c.n_total) r.c0 = c.c0 + 32;
    else if (c.j < n_mine) r = enter(c.j + 1, c.crow, c.item);
    r.j = uni(r.j), r.c0 = uni(r.c0), r.start = uni(r.start), r.n_total = uni(r.n_total), r.item = (uint32_t)uni((int)r.item);
    return r;
  };
  auto fe_of = [&](const ChunkCursor& c) { return max(min(c.c0 + kcol, c.n_total - 1), 0); };
  auto nbr_of = [&](const ChunkCursor& c) {
    const int e = c.start + (fe_of(c) >> fnb_shift);
    return (int)__builtin_amdgcn_raw_buffer_load_b32(nbr_rs, (e * g.nbr_stride + g.nbr_offset) * 4, 0, 0);
  };
  auto row_of = [&](int nb, const ChunkCursor& c) { return (nb << fnb_shift) + (fe_of(c) & fmask); };
  typedef const f32x4 __attribute__((address_space(4))) * crec_t;
  const crec_t ctr_rec = (crec_t)(uintptr_t)g.ctr_geom;
  // the centre record(s) of chunk c: through the scalar cache; FC = 2: the record of frame a0 + h per lane half
  auto centre = [&](const ChunkCursor& c, float yc[3], float rc[9]) {
    const int row = c.crow;
    const f32x4 v0 = ctr_rec[row * 4], v1 = ctr_rec[row * 4 + 1], v2 = ctr_rec[row * 4 + 2];
    if constexpr (FC == 2) {
      const f32x4 w0 = ctr_rec[row * 4 + 4], w1 = ctr_rec[row * 4 + 5], w2 = ctr_rec[row * 4 + 6];
      yc[0] = h ? w0[0] : v0[0], yc[1] = h ? w0[1] : v0[1], yc[2] = h ? w0[2] : v0[2], rc[8] = h ? w0[3] : v0[3];
#pragma unroll
      for (int i = 0; i < 4; ++i) rc[i] = h ? w1[i] : v1[i], rc[4 + i] = h ? w2[i] : v2[i];
    } else {
      yc[0] = v0[0], yc[1] = v0[1], yc[2] = v0[2], rc[8] = v0[3];
#pragma unroll
      for (int i = 0; i < 4; ++i) rc[i] = v1[i], rc[4 + i] = v2[i];
    }
  };

  ChunkCursor cur, n1, n2;
  cur = enter(0, 0, 0u);
  n1 = advance(cur);
  n2 = advance(n1);
  int q_cur, q_n1;
  float xn_nx[3], rn_nx[9];
  {
    const int nb_cur = nbr_of(cur);
    const int nb_n1 = nbr_of(n1);
    q_cur = row_of(nb_cur, cur);
    load_geom_record(nbg_rs, q_cur, xn_nx, rn_nx);
    q_n1 = row_of(nb_n1, n1);
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(xn_nx[i]));
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(rn_nx[i]));
  }
  float yc[3], rc[9];
  centre(cur, yc, rc);

  while (cur.j < n_mine) {
    rotate_priority(prio_step++);
    const int cnt = min(32, cur.n_total - cur.c0);  // <= 0: a row without neighbours
    const int qoff = cur.c0 + kcol < cur.n_total ? q_cur * row_bytes : kOobOffset;
    float xn[3], rn[9], d[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) xn[i] = xn_nx[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) rn[i] = rn_nx[i];
    const int nb_n2 = nbr_of(n2);  // ids two chunks ahead

    uint32_t fw[2][VW][8];  // gathered feature words of the chunk's two k-steps (shared by the FC rows)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int src_off = __builtin_amdgcn_ds_bpermute(hb + 4 * acc_row(8 * s + j, 0), qoff);
        if constexpr (VW == 2) {
          const auto v = __builtin_amdgcn_raw_buffer_load_b64(feat_rs, src_off + cb4, 0, 0);
          fw[s][0][j] = v[0], fw[s][1][j] = v[1];
        } else {
          fw[s][0][j] = __builtin_amdgcn_raw_buffer_load_b32(feat_rs, src_off + cb4, 0, 0);
        }
      }
    load_geom_record(nbg_rs, q_n1, xn_nx, rn_nx);  // record one chunk ahead
    float yn[3], rnc[9];
    centre(n1, yn, rnc);  // the next chunk's centre (the same record until the item changes)

    if (TR == 0)
      edge_descriptor(xn, rn, yc, rc, rho, d);
    else
      edge_descriptor(yc, rc, xn, rn, rho, d);

    // MLP A operand pieces of this lane (edge_bf16_body.h): its own dims 0..7, and {dim 8 of the row it serves as "other" half, 1}
    u32x4 own_hi, own_lo, oth_hi, oth_lo;
    frags_from_floats(d, own_hi, own_lo);
    {
      float d8 = d[8];
      if constexpr (FC == 2) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d8), __float_as_uint(d8), false, false);
        d8 = __uint_as_float(h ? sw[0] : sw[1]);
      }
      uint32_t p_hi, p_lo;
      split2(d8, 1.0f, p_hi, p_lo);
      oth_hi = u32x4{p_hi, 0u, 0u, 0u};
      oth_lo = u32x4{p_lo, 0u, 0u, 0u};
    }
    u32x4 fa_hi[2][VW], fa_lo[2][VW];
#pragma unroll
    for (int a = 0; a < FC; ++a) {
      const bool dims07 = FC == 1 ? h == 0 : h == a;
      u32x4 a_hi, a_lo;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        a_hi[i] = dims07 ? own_hi[i] : oth_hi[i];
        a_lo[i] = dims07 ? own_lo[i] : oth_lo[i];
      }
      const u32x4 wb_hi = *reinterpret_cast<const u32x4*>(&lds_w[a][0][lane][0]);
      const u32x4 wb_lo = *reinterpret_cast<const u32x4*>(&lds_w[a][1][lane][0]);
      const f32x16 phi = mfma_bf16x3(a_hi, a_lo, wb_hi, wb_lo, zero16());
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s * 16 < cnt) {
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pv[j] = gelu_scaled(phi[8 * s + j]);
          u32x4 b_hi, b_lo;
          frags_from_floats(pv, b_hi, b_lo);
#pragma unroll
          for (int t = 0; t < VW; ++t) {
            if (a == 0) frags_from_words(fw[s][t], fa_hi[s][t], fa_lo[s][t]);
            acc[a][t] = mfma_bf16x3(fa_hi[s][t], fa_lo[s][t], b_hi, b_lo, acc[a][t]);
          }
        }
      }
    }
    // every load of this chunk is consumed here, in front of the stores (see the wave-pair stream)
    q_cur = q_n1;
    q_n1 = row_of(nb_n2, n2);
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(xn_nx[i]));
#pragma unroll
    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(rn_nx[i]));
    if (n1.j != cur.j) {
      // the item's last chunk: acc[a][t] register r, lane (kcol, h) = T[row FC item + a][VW acc_row(r, h) + t][kcol] leaves
      // as 3-byte rows (two adjacent channels of this lane = one hi word + one lo half-word), non-temporal
#pragma unroll
      for (int a = 0; a < FC; ++a) {
        char* row = t_out + ((int64_t)cur.item * FC + a) * t24_row_bytes(C);
        if constexpr (VW == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            uint32_t hp, lp;
            t24_pack2(acc[a][0][r], acc[a][1][r], hp, lp);
            const int idx = acc_row(r, h) * kBasis + kcol;
            __builtin_nontemporal_store(hp, reinterpret_cast<uint32_t*>(row) + idx);
            __builtin_nontemporal_store((uint16_t)lp, reinterpret_cast<uint16_t*>(row + (int64_t)C * kBasis * 2) + idx);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            uint32_t hp, lp;
            t24_pack2(acc[a][0][r], acc[a][0][r + 1], hp, lp);
            const int idx = (acc_row(r, h) >> 1) * kBasis + kcol;
            __builtin_nontemporal_store(hp, reinterpret_cast<uint32_t*>(row) + idx);
            __builtin_nontemporal_store((uint16_t)lp, reinterpret_cast<uint16_t*>(row + (int64_t)C * kBasis * 2) + idx);
          }
        }
#pragma unroll
        for (int t = 0; t < VW; ++t) acc[a][t] = zero16();
      }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) yc[i] = yn[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) rc[i] = rnc[i];
    cur = n1, n1 = n2, n2 = advance(n2);
  }
  }  // windows
}

// ------------------------------------------------------------------------------------------------
// Gradient of the kernel-MLP parameters (cf. edge_param_grad_kernel in edge_kernels.hip):
//   gphi[n,k] = sum_i feat[q(n), i] * gT[m][i,k]   rows n, cols k, k-dim = channels, 16 per MFMA:
//               A: lane (n,h) reads words feat[q(n)][i0 + 8h .. +7]  (32 contiguous bytes)
//               B: lane (k,h) reads words gT[m][i0 + 8h + j][k], j = 0..7
//   gpre = gphi * GELU'(pre);  d[A;beta][j,k] += desc_ext[n,j] * gpre[n,k]   (fp32 VALU)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_param_grad_bf16_kernel(EdgeGeom g, const uint32_t* __restrict__ feat,
                                                                   int channels, const float* __restrict__ axes_ext,
                                                                   const float* __restrict__ rho_p,
                                                                   const uint32_t* __restrict__ grad_t,
                                                                   float* __restrict__ partials, int64_t rows) {
  __shared__ __attribute__((aligned(16))) float lds_desc[4][32][12];
  __shared__ float lds_red[4][kDescExt][kBasis];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kcol = lane & 31, h = lane >> 5;
  const float rho = *rho_p;
  u32x4 w_hi, w_lo;
  load_mlp_weights(axes_ext, kcol, h, w_hi, w_lo);
  float dacc[kDescExt];
#pragma unroll
  for (int j = 0; j < kDescExt; ++j) dacc[j] = 0.f;
  const bool vec_ok = (channels % 16) == 0;

  for (int64_t m = (int64_t)blockIdx.x * 4 + wave; m < rows; m += (int64_t)gridDim.x * 4) {
    const RowInfo ri = row_info(g, m);
    if (ri.n_total == 0) continue;
    float yc[3], rc[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) yc[i] = g.ctr_pts[ri.ctr * 3 + i];
#pragma unroll
    for (int i = 0; i < 9; ++i) rc[i] = g.ctr_frames[m * 9 + i];
    const uint32_t* gt_row = grad_t + m * (int64_t)channels * kBasis;

    for (int c0 = 0; c0 < ri.n_total; c0 += 32) {
      const int cnt = min(32, ri.n_total - c0);
      const int fe = c0 + min(kcol, cnt - 1);
      float d[kDescExt];
      int q;
      lane_descriptor(g, ri, fe, yc, rc, rho, d, q);
      const f32x16 pre = mlp_preactivation(d, h, w_hi, w_lo);
      if (h == 0) {
        float4* dst = reinterpret_cast<float4*>(&lds_desc[wave][kcol][0]);
        dst[0] = make_float4(d[0], d[1], d[2], d[3]);
        dst[1] = make_float4(d[4], d[5], d[6], d[7]);
        dst[2] = make_float4(d[8], d[9], 0.f, 0.f);
      }

      f32x16 gphi = zero16();
      const uint32_t* f_row = feat + (int64_t)q * channels;
      for (int i0 = 0; i0 < channels; i0 += 16) {
        const int my0 = i0 + 8 * h;
        uint32_t wa[8], wb[8];
        if (vec_ok) {
          const uint4 v0 = *reinterpret_cast<const uint4*>(f_row + my0);
          const uint4 v1 = *reinterpret_cast<const uint4*>(f_row + my0 + 4);
          wa[0] = v0.x, wa[1] = v0.y, wa[2] = v0.z, wa[3] = v0.w, wa[4] = v1.x, wa[5] = v1.y, wa[6] = v1.z, wa[7] = v1.w;
#pragma unroll
          for (int j = 0; j < 8; ++j) wb[j] = gt_row[(int64_t)(my0 + j) * kBasis + kcol];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool ok = my0 + j < channels;
            wa[j] = ok ? f_row[my0 + j] : 0u;
            wb[j] = ok ? gt_row[(int64_t)(my0 + j) * kBasis + kcol] : 0u;
          }
        }
        u32x4 a_hi, a_lo, b_hi, b_lo;
        frags_from_words(wa, a_hi, a_lo);
        frags_from_words(wb, b_hi, b_lo);
        gphi = mfma_bf16x3(a_hi, a_lo, b_hi, b_lo, gphi);
      }

      // wave-private LDS hand-off of the descriptors (LDS ops of one wave complete in order)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = acc_row(r, h);
        const float dy = gelu_scaled_dgrad(pre[r]);  // 2 GELU': the factor 0.5 is applied where the partials are reduced
        const float gp = n < cnt ? gphi[r] * dy : 0.f;
        const float4* src = reinterpret_cast<const float4*>(&lds_desc[wave][n][0]);
        const float4 d0 = src[0], d1 = src[1];
        const float2 d2 = *reinterpret_cast<const float2*>(&lds_desc[wave][n][8]);
        dacc[0] += d0.x * gp, dacc[1] += d0.y * gp, dacc[2] += d0.z * gp, dacc[3] += d0.w * gp;
        dacc[4] += d1.x * gp, dacc[5] += d1.y * gp, dacc[6] += d1.z * gp, dacc[7] += d1.w * gp;
        dacc[8] += d2.x * gp, dacc[9] += d2.y * gp;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }

#pragma unroll
  for (int j = 0; j < kDescExt; ++j) {
    const float v = dacc[j] + __shfl_xor(dacc[j], 32);
    if (h == 0) lds_red[wave][j][kcol] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kDescExt * kBasis; i += blockDim.x) {
    const int j = i / kBasis, k = i % kBasis;
    partials[(int64_t)blockIdx.x * kDescExt * kBasis + i] =
        lds_red[0][j][k] + lds_red[1][j][k] + lds_red[2][j][k] + lds_red[3][j][k];
  }
}

// ------------------------------------------------------------------------------------------------
// edge_param_grad, two frames per wavefront (same sharing scheme as edge_item_bf16) and the d[A;beta]
// accumulation moved from VALU to MFMA:
//   gphi_a[n,k] = sum_i feat[q(n), i] * gT[row a][i,k]      A = gathered rows (shared by both frames),
//                                                            B = gT fragments, fetched once per item
//   gpre_a      = gphi_a * GELU'(pre_a)
//   d[A;beta]^T[k, j] += sum_n gpre_a[n,k] * desc_a[n,j]     A = gpre (the accumulator registers, split in
//                                                            place), B = descriptor columns read back from
//                                                            a wave-private LDS image of packed words
// CH16 = channels / 16 (1..4).  The gT fragments of the item's two rows are built once per item and parked in
// a wave-private LDS image (CH16 * 4 KB per wavefront; in registers they cost CH16 * 16 VGPRs and spilled),
// hence 512-thread blocks: 8 wavefronts share the CU's LDS at the same 2 waves/SIMD as before.
// ------------------------------------------------------------------------------------------------
#ifndef SE3_PG_PAIR_LEAN
#define SE3_PG_PAIR_LEAN 0  // 1: pair form on 19 KB of LDS (8 workgroups = 4 wavefronts per SIMD): MLP weights in registers, ONE
#endif                      // descriptor image per wavefront used by the two frames in turn, centre record re-read per chunk, next
                            // geometry fetched between the frames -- what it takes to fit 128 VGPRs.  Measured (r02_param_grad_lean_ab):
                            // 0.454 ms at 8 per CU against 0.446 for the 26 KB form at 6; at equal occupancy (6) the lean chunk is
                            // 18 % slower (0.528), which the fourth wavefront only just buys back.  Parity-green, off.
#ifndef SE3_PG_PAIR_WAVES
#define SE3_PG_PAIR_WAVES (SE3_PG_PAIR_LEAN ? 4 : 3)  // wavefronts per SIMD the pair form's register budget is set for
#endif
#ifndef SE3_PG_SINGLE_WAVES
#define SE3_PG_SINGLE_WAVES 3  // one frame per wavefront (odd F).  4 (40 KB of LDS per 4-wave workgroup = 4 per CU) was measured:
#endif                         // at 128 VGPRs the 64-channel form spills 11 registers and runs 0.65 instead of 0.47 ms (ScanNet-like)
// PAIR (round 2, two frames only): a 128-thread workgroup = two wavefronts share ONE item -- one grad_T image (16 KB
// instead of 16 KB per wavefront: 26 KB of LDS per workgroup, 6 workgroups = 3 wavefronts per SIMD instead of 2),
// wavefront v builds the image of frame v (32 of the 64 row loads) and takes the chunks v, v + 2, ... of the item for
// both frames; two workgroup barriers per item (image built / image free), partial sums folded per workgroup.
// POW2: the neighbour cloud's frame count is a power of two (fnb_shift >= 0) -- no division path, no branch on it
// GT16: grad_T rows come in the T16 block format (common.h; pair form, 64-channel blocks): a lane's 8 channels of a k-step
// are two blocks of its basis function -- two 8-byte mantissa loads + the exponent line's 8-byte group per k-step instead
// of eight dword loads, decoded to the same fragment image (t16_unpack2).
template <int CH16, int NFR, bool PAIR = false, bool POW2 = false, bool GT16 = false>  // NFR = frames per wavefront: 2 (even F) or 1 (odd F: both lane halves hold the frame)
__global__ __launch_bounds__(PAIR ? 128 : (NFR == 2 ? 512 : 256), PAIR ? SE3_PG_PAIR_WAVES : (NFR == 2 ? 2 : (CH16 == 3 ? 3 : SE3_PG_SINGLE_WAVES))) void edge_param_grad_bf16_v2_kernel(EdgeGeom g, const uint32_t* __restrict__ feat,
                                                                         int row_ch, int64_t feat_rows,
                                                                         const float* __restrict__ axes_ext,
                                                                         const float* __restrict__ rho_p,
                                                                         const uint32_t* __restrict__ grad_t,
                                                                         float* __restrict__ partials, int64_t n_items,
                                                                         int fnb_shift, int64_t item_lo = 0, int pipe = 0) {
  // Rows wider than 64 channels are covered by blockIdx.y: block row y handles channels 64y .. 64y+63 (gphi, hence
  // d[A;beta], is linear in the channel sum, so every channel block contributes an independent partial; each block
  // row recomputes the descriptors and GELU').  row_ch = channels per row (a multiple of 16).
  // wavefronts per block: the gT images (8 KB per frame and wavefront) bound the occupancy
  static_assert(!PAIR || NFR == 2, "the pair form shares the two frames of a point");
  static_assert(!GT16 || (PAIR && CH16 == 4), "T16 grad_T rows: pair form on whole 64-channel blocks");
  constexpr bool LEAN = PAIR && SE3_PG_PAIR_LEAN;
  constexpr int NW = PAIR ? 2 : (NFR == 2 ? 8 : 4);
  constexpr int NIMG = PAIR ? 1 : NW;        // grad_T images per workgroup
  constexpr int CSTEP = PAIR ? 64 : 32;      // frame-edges between two chunks of one wavefront
  const int c_off = 64 * (int)blockIdx.y;
  const int row_bytes = row_ch * 4;
  __shared__ __attribute__((aligned(16))) uint32_t lds_w[LEAN ? 1 : NFR][LEAN ? 1 : 2][LEAN ? 1 : 64][4];  // LEAN: in registers
  // descriptor image for the d[A;beta] product: hi / lo bf16 planes, row = frame-edge, 12 columns ([desc(9), 1, 0, 0]).
  // Its MFMA fragments have the K index over the rows: read with ds_read_b64_tr_b16 (common.h), 4 per (frame, k-step).
  __shared__ __attribute__((aligned(16))) uint16_t lds_desc[NW][LEAN ? 1 : NFR][2][32][12];
  __shared__ __attribute__((aligned(16))) uint32_t lds_gt[NIMG][NFR][CH16][2][64][4];  // [wave][row][step][hi/lo][lane]
  // the final block reduction reuses the gT image (NW * 10 * 32 floats <= NIMG * NFR * CH16 * 512 words)
  float(*lds_red)[kDescExt][kBasis] = reinterpret_cast<float(*)[kDescExt][kBasis]>(&lds_gt[0][0][0][0][0][0]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (parameter-gradient timeline: segment sums, words 0 .. 15 as named at the stamps; [16] items, [17] chunks, [18] first stamp, [19] last)
  // (32 words per wavefront: with 64 the pair form's workgroup crosses an LDS allocation granule and only five fit a CU)
  TL(__shared__ uint32_t tl_lds[NW][32]; uint32_t* tl_rec = tl_lds[wave]; if (lane < 32) tl_rec[lane] = 0u; uint32_t tl_prev = tl_now(); tl_rec[18] = tl_prev;)
  const int img = PAIR ? 0 : wave;  // which grad_T image this wavefront reads
  const int kcol = lane & 31, h = lane >> 5;
  // MLP weights [A; beta] as the MFMA B operand.  LEAN: arrangement 0 (lane half 0 holds descriptor dims 0..7, half 1
  // dims 8, 9) lives in 8 registers; frame 1 wants the halves exchanged, one v_permlane32_swap per register and chunk
  u32x4 w_hi = {0u, 0u, 0u, 0u}, w_lo = {0u, 0u, 0u, 0u};
  if (LEAN) {
    load_mlp_weights(axes_ext, kcol, h, w_hi, w_lo);
  } else {
    if (threadIdx.x < 64) mlp_weights_to_lds<NFR>(reinterpret_cast<uint32_t(*)[2][64][4]>(&lds_w[0][0][0][0]), axes_ext, threadIdx.x);
    __syncthreads();
  }
  const float rho = *rho_p;
  const __amdgpu_buffer_rsrc_t feat_rs = buffer_of(feat, feat_rows * row_bytes);
  const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
  const __amdgpu_buffer_rsrc_t ctrg_rs = buffer_of(g.ctr_geom, g.n_ctr * g.f_ctr * 64);
  f32x16 dacc = zero16();  // lane (j = kcol, h), register r: d[A;beta][j][k = acc_row(r,h)]
  TL(tl_seg(tl_rec, tl_prev, 0);)  // 0: arguments, MLP weights into LDS, barrier

  // Items are taken last-to-first: grad_T was written front-to-back by the GEMM just before this kernel, so its tail
  // (what fits the memory-side cache) is still on chip -- reading it first turns those rows into cache hits instead of
  // letting the front-to-back walk evict them unread (0.492 -> 0.477 ms).
#ifndef SE3_PG_REVERSE
#define SE3_PG_REVERSE 1
#endif
  // Software pipeline over the items (round 6): the centre's record and the ids of an item's first two chunks are issued
  // when this wavefront has finished its last chunk of the PREVIOUS item, i.e. in front of the barrier that frees the
  // image, into registers that item no longer uses: the barrier wait (the partner wavefront still on its chunk: 12 % of a
  // wavefront's life in profiles/r06_edge_timeline.txt) and those round trips overlap.  pipe: the row extents of the workgroup's items (at most
  // 64, one per lane) are read once at the start and taken from there with v_readlane -- no extent -> ids -> rows chain
  // per item.
  const int64_t item_stride = PAIR ? (int64_t)gridDim.x : (int64_t)gridDim.x * NW;
  const int groups = g.f_ctr / NFR;
  int v_lo = 0, v_hi = 0;  // pipe: extents of local item `lane`
  if (pipe) {
    const int64_t item_f = min((int64_t)blockIdx.x + (int64_t)lane * item_stride, n_items - 1);
    const uint32_t ctr = (uint32_t)(item_lo + (SE3_PG_REVERSE ? n_items - 1 - item_f : item_f)) / (uint32_t)groups;
    v_hi = g.ends[ctr];
    v_lo = g.ends[max((int)ctr - 1, 0)];
    if (ctr == 0) v_lo = 0;
  }
  int64_t item_f = PAIR ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * NW + wave;
  int64_t item = 0;
  int start = 0, n_total = 0, ctr_row = 0, local_j = 0;
  float yc[3], rc[9];
  constexpr int NBUILD = PAIR ? 1 : NFR;
  uint32_t gw[NBUILD][CH16][8];
  int nb_a = 0, nb_b = 0;
  const int c_first = PAIR ? 32 * wave : 0;  // this wavefront's first chunk
  // ids two chunks ahead, geometry one chunk ahead (see edge_t_pair_bf16_kernel); indices past the end clamp
  auto nbr_of = [&](int c0) {
    const int fe = min(c0 + kcol, n_total - 1);
    const int e = start + (POW2 || fnb_shift >= 0 ? fe >> fnb_shift : fe / g.f_nb);
    return g.nbr[(int64_t)e * g.nbr_stride + g.nbr_offset];
  };
  auto row_of = [&](int nb, int c0) {
    const int fe = min(c0 + kcol, n_total - 1);
    return (POW2 ? nb << fnb_shift : nb * g.f_nb) + (POW2 || fnb_shift >= 0 ? fe & ((1 << fnb_shift) - 1) : fe % g.f_nb);
  };
  // the next item with neighbours at or behind item_f (items without any are skipped: uniform over the workgroup in the
  // pair form, both wavefronts skip their barriers); false: none left
  auto find_item = [&]() {
    for (; item_f < n_items; item_f += item_stride, ++local_j) {
      item = item_lo + (SE3_PG_REVERSE ? n_items - 1 - item_f : item_f);  // items item_lo .. item_lo + n_items - 1
      // rows < 2^31 (checked on the host), so 32-bit unsigned division is exact -- the 64-bit one is ~150 scalar instructions
      const int64_t ctr = (uint32_t)item / (uint32_t)groups;
      const int a0 = (int)((uint32_t)item - (uint32_t)ctr * (uint32_t)groups) * NFR;
      if (pipe) {
        start = __builtin_amdgcn_readlane(v_lo, local_j);
        n_total = (__builtin_amdgcn_readlane(v_hi, local_j) - start) * g.f_nb;
      } else {
        start = ctr > 0 ? g.ends[ctr - 1] : 0;
        n_total = (g.ends[ctr] - start) * g.f_nb;
      }
      ctr_row = (int)(ctr * g.f_ctr + a0 + (NFR == 2 ? h : 0));
      TL(tl_rec[16] += 1;)
      if (n_total != 0) return true;
    }
    return false;
  };
  auto issue_item_loads = [&]() {
    if (!LEAN) load_geom_record(ctrg_rs, ctr_row, yc, rc);  // LEAN: fetched again per chunk (a cache hit; 12 registers)
    nb_a = nbr_of(c_first);
    nb_b = nbr_of(c_first + CSTEP);
  };
  // (the grad_T rows stay behind the barrier: their 32 registers in flight across it made the compiler spill 16 values
  // around the chunk loop and reload them through the vector-memory queue in the middle of this very issue sequence --
  // 0.41 -> 0.475 ms)
  auto issue_image_loads = [&]() {
    // gT fragments (MFMA B operand) of the item's two rows: lane (k = kcol, h) holds channels 16*st + 8h + j
      // (pair form: this wavefront fetches and builds the image of frame `wave` only)
      if constexpr (GT16) {
        // row `wave` of the item: mantissas of channel quad cq = c_off / 4 + 4 st + 2 h + jj at (cq * 32 + kcol) * 8; the
        // exponents of both jj sit in one 8-byte group of the row's exponent plane (t16_exp_pos: mega tile = cq >> 1, piece
        // = (kcol >> 1) & 7; byte (2 jj + (kcol >> 4)) * 2 + (kcol & 1))
        const int64_t rbytes = t16_row_bytes(row_ch);
        const uint64_t gt_addr = reinterpret_cast<uint64_t>(reinterpret_cast<const char*>(grad_t) + (item * NFR + wave) * rbytes);
        const uint64_t gt_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(gt_addr >> 32)) << 32) |
                                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)gt_addr);
        const __amdgpu_buffer_rsrc_t gt_rs = buffer_of(reinterpret_cast<const void*>(gt_base), rbytes);
        const int cq0 = (c_off >> 2) + 2 * h;
        const int e_off = row_ch * 64 + ((kcol >> 1) & 7) * 8, e_sh = 8 * ((kcol >> 4) * 2 + (kcol & 1));
#pragma unroll
        for (int st = 0; st < CH16; ++st) {
          const auto m0 = __builtin_amdgcn_raw_buffer_load_b64(gt_rs, ((cq0 + 4 * st) * kBasis + kcol) * 8, 0, 0);
          const auto m1 = __builtin_amdgcn_raw_buffer_load_b64(gt_rs, ((cq0 + 4 * st + 1) * kBasis + kcol) * 8, 0, 0);
          const auto eg = __builtin_amdgcn_raw_buffer_load_b64(gt_rs, e_off + ((cq0 + 4 * st) >> 1) * 64, 0, 0);
          gw[0][st][0] = m0[0], gw[0][st][1] = m0[1], gw[0][st][2] = m1[0], gw[0][st][3] = m1[1];
          gw[0][st][4] = (eg[0] >> e_sh) & 0xffu, gw[0][st][5] = (eg[1] >> e_sh) & 0xffu;
        }
      } else
#pragma unroll
      for (int ab = 0; ab < NBUILD; ++ab) {
        const int a = PAIR ? wave : ab;
        const uint32_t* gt_row = grad_t + ((item * NFR + a) * (int64_t)row_ch + c_off) * kBasis;
        // one buffer per row (its base is wave-uniform): channels past the row read as zeros through the bounds check of
        // the buffer load -- a guarded global load per element compiled into a branch per load
        const int row_left = min(row_ch - c_off, 16 * CH16);
        // (`wave` is uniform but the compiler cannot know: without readfirstlane every load becomes a waterfall loop)
        const uint64_t gt_addr = reinterpret_cast<uint64_t>(gt_row);
        const uint64_t gt_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(gt_addr >> 32)) << 32) |
                                 (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)gt_addr);  // (the builtin returns int)
        const __amdgpu_buffer_rsrc_t gt_rs = buffer_of(reinterpret_cast<const void*>(gt_base), (int64_t)row_left * kBasis * 4);
#pragma unroll
        for (int st = 0; st < CH16; ++st)
#pragma unroll
          for (int j = 0; j < 8; ++j)
            gw[ab][st][j] = __builtin_amdgcn_raw_buffer_load_b32(gt_rs, (8 * h * kBasis + kcol) * 4, (16 * st + j) * kBasis * 4, 0);
      }
  };
  int prio_step = wave_slot_id();
  bool have = find_item();
  TL(tl_seg(tl_rec, tl_prev, 1);)  // 1: item set-up (extents, rows)
  if (have) issue_item_loads();
  TL(tl_seg(tl_rec, tl_prev, 3);)  // 3: centre record, ids, grad_T row loads issued
  while (have) {
    issue_image_loads();
    TL(tl_pin(nb_a); tl_seg(tl_rec, tl_prev, 4);)  // 4: ids in
    int q_a = row_of(nb_a, c_first);
    float xn_nx[3], rn_nx[9];
    load_geom_record(nbg_rs, q_a, xn_nx, rn_nx);
    TL(tl_pin(gw[NBUILD - 1][CH16 - 1][7]); tl_seg(tl_rec, tl_prev, 5);)  // 5: grad_T words in
#pragma unroll
    for (int ab = 0; ab < NBUILD; ++ab)
#pragma unroll
      for (int st = 0; st < CH16; ++st) {
        const int a = PAIR ? wave : ab;
        u32x4 f_hi, f_lo;
        if constexpr (GT16) {
          const float sc0 = t16_scale(gw[ab][st][4]), sc1 = t16_scale(gw[ab][st][5]);
          uint32_t hw, lw;
          t16_unpack2(gw[ab][st][0], sc0, hw, lw), f_hi[0] = hw, f_lo[0] = lw;
          t16_unpack2(gw[ab][st][1], sc0, hw, lw), f_hi[1] = hw, f_lo[1] = lw;
          t16_unpack2(gw[ab][st][2], sc1, hw, lw), f_hi[2] = hw, f_lo[2] = lw;
          t16_unpack2(gw[ab][st][3], sc1, hw, lw), f_hi[3] = hw, f_lo[3] = lw;
        } else
        frags_from_words(gw[ab][st], f_hi, f_lo);
        *reinterpret_cast<u32x4*>(&lds_gt[img][a][st][0][lane][0]) = f_hi;
        *reinterpret_cast<u32x4*>(&lds_gt[img][a][st][1][lane][0]) = f_lo;
      }
    TL(tl_seg(tl_rec, tl_prev, 6);)  // 6: grad_T fragments built and parked
    if (PAIR) __syncthreads();  // both frames' images are in place
    TL(tl_seg(tl_rec, tl_prev, 7);)  // 7: barrier (image complete)

    for (int c0 = c_first; c0 < n_total; c0 += CSTEP) {
      rotate_priority(prio_step++);  // resident workgroups (every form of this kernel walks its items grid-stride): common.h
      const int cnt = min(32, n_total - c0);
      // rows past the end of the edge list read zeros (out-of-bounds buffer loads): gphi = 0 there, no mask needed
      const int qoff = c0 + kcol < n_total ? q_a * row_bytes + c_off * 4 : kOobOffset;
      float d[9];
      const int q_b = row_of(nb_b, c0 + CSTEP);
      nb_b = nbr_of(c0 + 2 * CSTEP);

      // gathered feature rows (MFMA A operand of gphi): lane (n = kcol, h) reads its own source row; the words are
      // turned into fragments only after the GELU' work below
      uint32_t fw[CH16][8];
#pragma unroll
      for (int st = 0; st < CH16; ++st) {
        const int voff = c_off + 16 * st < row_ch ? qoff + (16 * st + 8 * h) * 4 : kOobOffset;  // past the row: zeros
        const auto v0 = __builtin_amdgcn_raw_buffer_load_b128(feat_rs, voff, 0, 0);
        const auto v1 = __builtin_amdgcn_raw_buffer_load_b128(feat_rs, voff + 16, 0, 0);
        fw[st][0] = v0[0], fw[st][1] = v0[1], fw[st][2] = v0[2], fw[st][3] = v0[3];
        fw[st][4] = v1[0], fw[st][5] = v1[1], fw[st][6] = v1[2], fw[st][7] = v1[3];
      }
      if (LEAN) load_geom_record(ctrg_rs, ctr_row, yc, rc);
      q_a = q_b;
      TL(tl_seg(tl_rec, tl_prev, 8); tl_pin(xn_nx[0]); tl_pin(rn_nx[7]); tl_pin(rn_nx[8]); tl_seg(tl_rec, tl_prev, 9); tl_rec[17] += 1;)  // 8: gathers issued, 9: this chunk's record in

      if (!g.transposed)
        edge_descriptor(xn_nx, rn_nx, yc, rc, rho, d);
      else
        edge_descriptor(yc, rc, xn_nx, rn_nx, rho, d);
      // the next chunk's record goes out once this chunk's has been consumed (issuing it only at the
      // end of the chunk body -- measured slower, 0.434 vs 0.419 ms)
      if (!LEAN) load_geom_record(nbg_rs, q_b, xn_nx, rn_nx);

      u32x4 own_hi, own_lo, oth_hi = {0u, 0u, 0u, 0u}, oth_lo = {0u, 0u, 0u, 0u};
      frags_from_floats(d, own_hi, own_lo);
      // descriptor image: half h writes the rows of frame a0+h -- the split pairs above are the rows of the two planes
      uint32_t own8_hi, own8_lo;
      split2(d[8], 1.0f, own8_hi, own8_lo);
      auto write_desc_rows = [&](int slot) {
        typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
        uint16_t* dh = &lds_desc[wave][slot][0][kcol][0];
        uint16_t* dl = &lds_desc[wave][slot][1][kcol][0];
        *reinterpret_cast<u32x2v*>(dh) = u32x2v{own_hi[0], own_hi[1]};
        *reinterpret_cast<u32x2v*>(dh + 4) = u32x2v{own_hi[2], own_hi[3]};
        *reinterpret_cast<u32x2v*>(dh + 8) = u32x2v{own8_hi, 0u};
        *reinterpret_cast<u32x2v*>(dl) = u32x2v{own_lo[0], own_lo[1]};
        *reinterpret_cast<u32x2v*>(dl + 4) = u32x2v{own_lo[2], own_lo[3]};
        *reinterpret_cast<u32x2v*>(dl + 8) = u32x2v{own8_lo, 0u};
      };
      if (!LEAN && (NFR == 2 || h == 0)) write_desc_rows(NFR == 2 ? h : 0);
      if (!LEAN) {
        float d8 = d[8];
        if (NFR == 2) {  // dims 8, 9 of frame a come from the half that did not build frame a's descriptor
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d8), __float_as_uint(d8), false, false);
          d8 = __uint_as_float(h ? sw[0] : sw[1]);
        }
        uint32_t p_hi, p_lo;
        split2(d8, 1.0f, p_hi, p_lo);
        oth_hi = u32x4{p_hi, 0u, 0u, 0u};
        oth_lo = u32x4{p_lo, 0u, 0u, 0u};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // this lane's address for the transposed reads: row (lane >> 2) & 3 of a 4-row block, columns 4 (lane & 3)
      // (the image has 12 columns: the last quad points at columns 8..11 again, what it returns lands in unused columns)
      const int tr_row = (lane >> 2) & 3, tr_col = min((lane & 3) * 4, 8);
      auto gelu_grad_of_frame = [&](int a, float (&dy)[16]) {
        const bool dims07 = NFR == 2 ? h == a : h == 0;
        u32x4 a_hi, a_lo;
        u32x4 wb_hi, wb_lo;
        if (LEAN) {
          // one weight arrangement (lane half 0: dims 0..7, half 1: dims 8, 9) for both frames; the descriptor of
          // frame a was built by lane half a, so one of the two operand parts crosses the halves (v_permlane32_swap)
          wb_hi = w_hi, wb_lo = w_lo;
          auto other_half = [&](uint32_t v) {
            const auto sw = __builtin_amdgcn_permlane32_swap(v, v, false, false);
            return h ? sw[0] : sw[1];
          };
          uint32_t d8_hi = own8_hi, d8_lo = own8_lo;
          u32x4 r_hi = own_hi, r_lo = own_lo;
          if (a == 0) {
            d8_hi = other_half(own8_hi), d8_lo = other_half(own8_lo);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) r_hi[i] = other_half(own_hi[i]), r_lo[i] = other_half(own_lo[i]);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            a_hi[i] = h == 0 ? r_hi[i] : (i == 0 ? d8_hi : 0u);
            a_lo[i] = h == 0 ? r_lo[i] : (i == 0 ? d8_lo : 0u);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            a_hi[i] = dims07 ? own_hi[i] : oth_hi[i];
            a_lo[i] = dims07 ? own_lo[i] : oth_lo[i];
          }
          const auto* wl = reinterpret_cast<const uint32_t(*)[2][64][4]>(&lds_w[0][0][0][0]);
          wb_hi = *reinterpret_cast<const u32x4*>(&wl[a][0][lane][0]);
          wb_lo = *reinterpret_cast<const u32x4*>(&wl[a][1][lane][0]);
        }
        const f32x16 pre = mfma_bf16x3(a_hi, a_lo, wb_hi, wb_lo, zero16());
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          dy[r] = gelu_scaled_dgrad(pre[r]);  // 2 GELU': the 0.5 is applied where the partials are reduced
        }
      };
      // gphi = F gT on the gathered rows, gpre = gphi * GELU', and the d[A;beta] product of frame a
      auto accumulate_frame = [&](int a, const u32x4 (&fa_hi)[CH16], const u32x4 (&fa_lo)[CH16], const float (&dy)[16]) {
        f32x16 gphi = zero16();
#pragma unroll
        for (int st = 0; st < CH16; ++st) {
          const u32x4 bg_hi = *reinterpret_cast<const u32x4*>(&lds_gt[img][a][st][0][lane][0]);
          const u32x4 bg_lo = *reinterpret_cast<const u32x4*>(&lds_gt[img][a][st][1][lane][0]);
          gphi = mfma_bf16x3(fa_hi[st], fa_lo[st], bg_hi, bg_lo, gphi);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s * 16 < cnt) {
            float gp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) gp[j] = gphi[8 * s + j] * dy[8 * s + j];
            u32x4 ga_hi, ga_lo;
            frags_from_floats(gp, ga_hi, ga_lo);
            // K slot (h, j) of this k-step is frame-edge acc_row(8 s + j, h) = 16 s + 4 h + (j & 3) + 8 (j >> 2)
            const int r0 = 16 * s + 4 * h + tr_row;
            const int slot = LEAN ? 0 : a;
            const u32x4 db_hi = lds_frag_tr16(&lds_desc[wave][slot][0][r0][tr_col], &lds_desc[wave][slot][0][r0 + 8][tr_col]);
            const u32x4 db_lo = lds_frag_tr16(&lds_desc[wave][slot][1][r0][tr_col], &lds_desc[wave][slot][1][r0 + 8][tr_col]);
            dacc = mfma_bf16x3(ga_hi, ga_lo, db_hi, db_lo, dacc);
          }
        }
      };
      u32x4 fa_hi[CH16], fa_lo[CH16];
      if (LEAN) {
        // one descriptor image per wavefront: frame a's lane half writes its rows, the frame is accumulated, then the
        // other half takes the image over (two wavefront barriers per frame instead of one per chunk)
#pragma unroll
        for (int a = 0; a < NFR; ++a) {
          float dy[16];
          gelu_grad_of_frame(a, dy);  // pure VALU in front of the first use of the gathered words
          if (h == a) write_desc_rows(0);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          if (a == 0) {
#pragma unroll
            for (int st = 0; st < CH16; ++st) frags_from_words(fw[st], fa_hi[st], fa_lo[st]);
          }
          accumulate_frame(a, fa_hi, fa_lo, dy);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          // the next chunk's geometry goes out between the frames: 12 registers that would not fit next to frame 0's
          // fragments, and frame 1's work still covers the latency
          if (a == 0) load_geom_record(nbg_rs, q_b, xn_nx, rn_nx);
        }
        continue;
      }
      // GELU' of both frames first (pure VALU, covers the gather latency) ...
      TL(tl_seg(tl_rec, tl_prev, 10);)  // 10: descriptor, splits, descriptor image
      float dyv[NFR][16];
#pragma unroll
      for (int a = 0; a < NFR; ++a) gelu_grad_of_frame(a, dyv[a]);
      TL(tl_pin(dyv[NFR - 1][15]); tl_seg(tl_rec, tl_prev, 11); tl_pin(fw[CH16 - 1][7]); tl_seg(tl_rec, tl_prev, 12);)  // 11: kernel MLP + GELU', 12: feature words in
      // ... then the gathered words become fragments and both frames are accumulated
#pragma unroll
      for (int st = 0; st < CH16; ++st) frags_from_words(fw[st], fa_hi[st], fa_lo[st]);
#pragma unroll
      for (int a = 0; a < NFR; ++a) accumulate_frame(a, fa_hi, fa_lo, dyv[a]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      TL(tl_seg(tl_rec, tl_prev, 13);)  // 13: gphi, gpre, d[A;beta] products issued
    }
    // this wavefront's chunks of the item are done: the next item's loads go out in front of the barrier
    item_f += item_stride, ++local_j;
    have = find_item();
    TL(tl_seg(tl_rec, tl_prev, 1);)
    if (have) issue_item_loads();
    TL(tl_seg(tl_rec, tl_prev, 3);)
    if (PAIR) __syncthreads();  // both wavefronts are done with the images before the next item overwrites them
    TL(tl_seg(tl_rec, tl_prev, 14);)  // 14: barrier (item done)
  }

  // dacc: rows = k (acc_row(r,h)), columns = descriptor dim j = kcol (only j < 10 are meaningful)
  __syncthreads();  // every wavefront is done with its gT image
  if (kcol < kDescExt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_red[wave][kcol][acc_row(r, h)] = dacc[r];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kDescExt * kBasis; i += blockDim.x) {
    const int j = i / kBasis, k = i % kBasis;
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) sum += lds_red[w][j][k];
    partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * kDescExt * kBasis + i] = sum;
  }
  TL(tl_seg(tl_rec, tl_prev, 15); tl_rec[19] = tl_prev;  // 15: accumulator drain + workgroup reduction
     tl_rec[20] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); tl_rec[21] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
     tl_flush(2, (blockIdx.y * gridDim.x + blockIdx.x) * NW + wave, tl_rec, 32);)
}

// [N,3] points + [N,F,9] frames -> one 64-byte record per (point, frame) row, see load_geom_record
__global__ void pack_geometry_kernel(const float* __restrict__ pts, const float* __restrict__ frames, int64_t rows,
                                     int f, float* __restrict__ records) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const int64_t pt = row / f;
  const float* r = frames + row * 9;
  float4* dst = reinterpret_cast<float4*>(records + row * 16);
  dst[0] = make_float4(pts[pt * 3], pts[pt * 3 + 1], pts[pt * 3 + 2], r[8]);
  dst[1] = make_float4(r[0], r[1], r[2], r[3]);
  dst[2] = make_float4(r[4], r[5], r[6], r[7]);
  dst[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void split_pack_kernel(const float* __restrict__ src, uint32_t* __restrict__ dst, int64_t n) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 + 3 < n) {
    const float4 v = *reinterpret_cast<const float4*>(src + i4);
    *reinterpret_cast<uint4*>(dst + i4) = make_uint4(split_pack(v.x), split_pack(v.y), split_pack(v.z), split_pack(v.w));
  } else {
    for (int64_t i = i4; i < n; ++i) dst[i] = split_pack(src[i]);
  }
}

}  // namespace

int launch_pack_geometry(const float* pts, const float* frames, int64_t n, int f, float* records, hipStream_t stream) {
  const int64_t rows = n * f;
  if (rows == 0) return SE3_OK;
  if (rows * 64 >= (int64_t)kOobOffset) return SE3_ERR_UNSUPPORTED;  // 32-bit record offsets in the kernels
  hipLaunchKernelGGL(pack_geometry_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, pts, frames, rows, f,
                     records);
  return check_launch();
}

int launch_split_pack(const float* src, uint32_t* dst, int64_t n, hipStream_t stream) {
  if (n == 0) return SE3_OK;
  ProfScope prof("split_pack", stream);
  const int64_t blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, src, dst, n);
  return check_launch();
}

// Which kernel a shape takes: the wave pair for rows of >= 64 channels -- except 64-channel rows with an odd frame count
// (ScanNet's F = 1), where both wavefronts of a pair would evaluate every descriptor for half the aggregation work each:
// the single-wavefront kernel at two channels per lane is 5 % faster there since it is pipelined like the pair
// (profiles/r03_f1_forms_ab.txt).  SE3_NO_PAIR=1: the single-wavefront kernel everywhere.
static bool edge_t_bf16_uses_pair(const EdgeGeom& g, int channels) {
  return channels >= 64 && getenv("SE3_NO_PAIR") == nullptr && !(channels == 64 && g.f_ctr % 2 == 1);
}

// row ranges (producer / consumer interleaving over slices of the rows): every form (multiples of the rows per item)
bool edge_t_bf16_row_ranges(const EdgeGeom& g, int channels) { return true; }

// Which launches can write their rows in the 3-byte format: the wave-pair kernel, and the single-wavefront kernel at one
// or two channels per lane (rows of up to 32 channels, and the 64-channel rows it takes), where the two channels of a
// pair are values of one lane
bool edge_t_bf16_t24_rows(const EdgeGeom& g, int channels) {
  if (channels % 2 != 0) return false;
  if (edge_t_bf16_uses_pair(g, channels)) return true;
  return channels <= 32 || channels == 64;
}

// 2.25-byte rows (T16): the wave-pair kernel on full 64-channel passes (a lane's four consecutive registers are a block)
bool edge_t_bf16_t16_rows(const EdgeGeom& g, int channels) {
  return channels % 64 == 0 && edge_t_bf16_uses_pair(g, channels);
}

#ifndef SE3_STREAM1
#define SE3_STREAM1 1  // 0: diagnostic build without the single-wavefront chunk-stream kernel (A/B against the one-item form)
#endif
// items below which the chunk-stream kernels are not used (SE3_EDGE_STREAM=n: n items; 0: never)
static int edge_stream_min_items() {
  static const int v = [] {
    const char* e = getenv("SE3_EDGE_STREAM");
    return e ? (atoi(e) > 0 ? atoi(e) : INT32_MAX) : 4096;
  }();
  return v;
}

static int device_cu_count() {  // of the current device (the resident grids are sized by it); <= 0: the query failed
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return n_cu;
}

// row_lo / row_hi (multiples of 2 for even F; < 0: everything): only the rows in that range are produced -- the
// wave-pair kernel supports it (edge_t_bf16_row_ranges), which lets the caller interleave producer and consumer
// launches over slices of the rows
int launch_edge_t_bf16(const char* tag, const EdgeGeom& g, const uint32_t* feat, int channels, int64_t feat_rows,
                       const float* axes_ext, const float* rho, uint32_t* t_out, hipStream_t stream, int64_t row_lo,
                       int64_t row_hi, int rowfmt) {
  const bool t24 = rowfmt == 1;
  const int64_t rows = g.n_ctr * g.f_ctr;
  if (rows == 0) return SE3_OK;
  if (row_lo >= 0 && !edge_t_bf16_row_ranges(g, channels)) return SE3_ERR_UNSUPPORTED;
  if (t24 && !edge_t_bf16_t24_rows(g, channels)) return SE3_ERR_UNSUPPORTED;
  if (rowfmt == 2 && !edge_t_bf16_t16_rows(g, channels)) return SE3_ERR_UNSUPPORTED;
  // 32-bit byte offsets into the gathered operand; kOobOffset must lie beyond it
  if (feat_rows * (int64_t)channels * 4 >= (int64_t)kOobOffset) return SE3_ERR_UNSUPPORTED;
  ProfScope prof(tag, stream);
  // two frames per wavefront share the gather; with 4 channel tiles per frame that would spill, so VW = 4 stays at 1
  const int fc = (getenv("SE3_FC1") == nullptr && g.f_ctr % 2 == 0 && channels % 128 != 0) ? 2 : 1;
  const int64_t items = rows / fc;
  int shift = -1;
  for (int sft = 0; sft < 8; ++sft)
    if ((1 << sft) == g.f_nb) shift = sft;
  const dim3 block(256);
  if (edge_t_bf16_uses_pair(g, channels)) {
    // a 128-thread workgroup per two frames of a point (even F) or per single row (odd F)
    const bool two = g.f_ctr % 2 == 0;
    const int64_t pair_items = two ? rows / 2 : rows;
    const int per = two ? 2 : 1;
    const int64_t item_lo = row_lo >= 0 ? row_lo / per : 0;
    const int64_t item_hi = row_lo >= 0 ? row_hi / per : pair_items;
    const int64_t n_range = item_hi - item_lo;
    if (n_range <= 0) return SE3_OK;
    // chunk-stream form (resident workgroups, the chunk pipeline running across item boundaries): 64-channel rows, two
    // frames per item, power-of-two neighbour frame count, 3-byte rows; SE3_EDGE_STREAM=0 keeps the one-item workgroups
    if (channels == 64 && two && shift >= 0 && rowfmt == 1 && n_range >= edge_stream_min_items() && g.n_edges > 0 &&
        g.n_edges * g.nbr_stride * 4 < (int64_t)kOobOffset && item_hi < (1ll << 31)) {
      const int n_cu = device_cu_count();
      if (n_cu <= 0) return SE3_ERR_LAUNCH;
      constexpr int per_cu = 2 * SE3_PAIR_WAVES;  // 18 KB of LDS and <= 128 VGPRs: eight two-wavefront workgroups per CU
      int64_t wgs = (int64_t)n_cu * per_cu;  // resident workgroups only, whatever the level's size (windows of 64 items inside)
      if (wgs > n_range) wgs = n_range;
      const dim3 sgrid((unsigned)wgs), sblock(128);
      if (g.transposed)
        hipLaunchKernelGGL((edge_t_stream_bf16_kernel<1>), sgrid, sblock, 0, stream, g, feat, feat_rows, axes_ext, rho,
                           reinterpret_cast<char*>(t_out), (uint32_t)item_lo, (uint32_t)item_hi, shift);
      else
        hipLaunchKernelGGL((edge_t_stream_bf16_kernel<0>), sgrid, sblock, 0, stream, g, feat, feat_rows, axes_ext, rho,
                           reinterpret_cast<char*>(t_out), (uint32_t)item_lo, (uint32_t)item_hi, shift);
      return check_launch();
    }
    const dim3 pgrid((unsigned)n_range), pblock(128);
#define SE3_PAIR_T(CT, FULL, NF, P2, TR)                                                                                \
  hipLaunchKernelGGL((edge_t_pair_bf16_kernel<CT, FULL, NF, P2, TR>), pgrid, pblock, 0, stream, g, feat, channels, feat_rows, \
                     axes_ext, rho, t_out, item_lo, item_hi, shift, rowfmt)
#define SE3_PAIR_L(CT, FULL, NF, P2)                 \
  do {                                               \
    if (!(P2)) SE3_PAIR_T(CT, FULL, NF, P2, -1);     \
    else if (g.transposed) SE3_PAIR_T(CT, FULL, NF, P2, 1); \
    else SE3_PAIR_T(CT, FULL, NF, P2, 0);            \
  } while (0)
#define SE3_PAIR(CT, FULL)                                   \
  do {                                                       \
    if (two && shift >= 0) SE3_PAIR_L(CT, FULL, 2, true);    \
    else if (two) SE3_PAIR_L(CT, FULL, 2, false);            \
    else if (shift >= 0) SE3_PAIR_L(CT, FULL, 1, true);      \
    else SE3_PAIR_L(CT, FULL, 1, false);                     \
  } while (0)
    if (channels == 64) SE3_PAIR(1, true);
    else if (channels % 128 == 0) SE3_PAIR(2, true);
    else SE3_PAIR(2, false);
#undef SE3_PAIR
#undef SE3_PAIR_L
#undef SE3_PAIR_T
    return check_launch();
  }
  // row range: items of fc rows each
  const int64_t s_item_lo = row_lo >= 0 ? row_lo / fc : 0, s_item_hi = row_lo >= 0 ? row_hi / fc : items;
  if (s_item_hi <= s_item_lo) return SE3_OK;
  // chunk-stream form of the single-wavefront kernel: rows of 32 channels, two frames per item.  Measured per form
  // (profiles/r06_edge_stream1_ab.txt): <1, 2> -8 % at two frames (items of ~2 chunks: dfaust_f2 0.191 -> 0.177 ms), -1 % at
  // four; <2, 1> (64-channel rows at one frame, the ScanNet scene: items of ONE chunk, so every chunk ends in its stores
  // and there is no pipeline to carry across) +1 %, and +6 % when squeezed to four wavefronts per SIMD (7 spilled
  // registers) -- those rows keep the one-item form.
  if (SE3_STREAM1 && t24 && shift >= 0 && channels == 32 && fc == 2 && s_item_hi - s_item_lo >= edge_stream_min_items() &&
      g.n_edges > 0 && g.n_edges * g.nbr_stride * 4 < (int64_t)kOobOffset && s_item_hi < (1ll << 31)) {
    const int n_cu = device_cu_count();
    if (n_cu <= 0) return SE3_ERR_LAUNCH;
    constexpr int per_cu = 3;  // = the kernel's launch bounds: resident workgroups only
    int64_t wgs = (int64_t)n_cu * per_cu;
    if (wgs > (s_item_hi - s_item_lo + 3) / 4) wgs = (s_item_hi - s_item_lo + 3) / 4;
    const dim3 sgrid((unsigned)wgs);
    if (g.transposed)
      hipLaunchKernelGGL((edge_t_stream1_bf16_kernel<1, 2, 1>), sgrid, block, 0, stream, g, feat, feat_rows, axes_ext, rho,
                         reinterpret_cast<char*>(t_out), (uint32_t)s_item_lo, (uint32_t)s_item_hi, shift);
    else
      hipLaunchKernelGGL((edge_t_stream1_bf16_kernel<1, 2, 0>), sgrid, block, 0, stream, g, feat, feat_rows, axes_ext, rho,
                         reinterpret_cast<char*>(t_out), (uint32_t)s_item_lo, (uint32_t)s_item_hi, shift);
    return check_launch();
  }
  const dim3 grid((unsigned)((s_item_hi - s_item_lo + 3) / 4));
#define SE3_LAUNCH(VW, FC, FULL, T24)                                                                                 \
  hipLaunchKernelGGL((edge_t_bf16_kernel<VW, FC, FULL, T24>), grid, block, 0, stream, g, feat, channels, feat_rows,     \
                     axes_ext, rho, t_out, s_item_hi, shift, s_item_lo)
#define SE3_LAUNCH_T(VW, FC, FULL)                   \
  do {                                               \
    if (t24) SE3_LAUNCH(VW, FC, FULL, true);         \
    else SE3_LAUNCH(VW, FC, FULL, false);            \
  } while (0)
  if (channels % 128 == 0) {
    SE3_LAUNCH(4, 1, true, false);
  } else if (channels % 64 == 0) {
    if (fc == 2) SE3_LAUNCH_T(2, 2, true); else SE3_LAUNCH_T(2, 1, true);
  } else if (channels % 32 == 0) {
    if (fc == 2) SE3_LAUNCH_T(1, 2, true); else SE3_LAUNCH_T(1, 1, true);
  } else {
    if (fc == 2) SE3_LAUNCH_T(1, 2, false); else SE3_LAUNCH_T(1, 1, false);
  }
#undef SE3_LAUNCH_T
#undef SE3_LAUNCH
  return check_launch();
}

// grad_T rows in the T16 block format: the pair form (two frames per point) on whole 64-channel blocks
bool edge_param_grad_bf16_t16_rows(const EdgeGeom& g, int channels) {
  static const bool pair_on = [] {
    const char* e = getenv("SE3_PG_PAIR");
    return e == nullptr || atoi(e) != 0;
  }();
  return pair_on && getenv("SE3_PG_SINGLE") == nullptr && g.f_ctr % 2 == 0 && channels % 64 == 0 && channels >= 64;
}

// row ranges (slices of the rows): every form for rows of a multiple of 16 channels
bool edge_param_grad_bf16_row_ranges(const EdgeGeom& g, int channels) {
  static const bool pair_on = [] {
    const char* e = getenv("SE3_PG_PAIR");
    return e == nullptr || atoi(e) != 0;
  }();
  (void)pair_on;
  return channels % 16 == 0 && channels > 0;  // every MFMA form of the kernel walks an item range
}

// partials: room for n_partials x edge_param_grad_bf16_channel_blocks(channels) slots of 320 floats; *n_used = slots written
int edge_param_grad_bf16_channel_blocks(int channels) { return channels > 64 && channels % 16 == 0 ? (channels + 63) / 64 : 1; }

int launch_edge_param_grad_bf16(const char* tag, const EdgeGeom& g, const uint32_t* feat, int channels,
                                int64_t feat_rows, const float* axes_ext, const float* rho, const uint32_t* grad_t,
                                float* partials, int n_partials, int* n_used, hipStream_t stream, bool gt16, int64_t row_lo,
                                int64_t row_hi) {
  const int64_t rows = g.n_ctr * g.f_ctr;
  if (row_lo >= 0 && !edge_param_grad_bf16_row_ranges(g, channels)) return SE3_ERR_UNSUPPORTED;
  if (feat_rows * (int64_t)channels * 4 >= (int64_t)kOobOffset) return SE3_ERR_UNSUPPORTED;
  if (gt16 && !edge_param_grad_bf16_t16_rows(g, channels)) return SE3_ERR_UNSUPPORTED;
  ProfScope prof(tag, stream);
  *n_used = n_partials;
  // Rows of fewer than 16 channels (the networks' first layers: C_in = 1 for DFaust, 3 for ScanNet colours) take the MFMA
  // form with one k-step too (round 6): a lane's eight feature words then run past its own row into the next rows', and meet
  // grad_T fragments that are zero there (the row's buffer ends behind its channels: out-of-range loads return 0), so
  // gphi = f . gT is exact -- 0.153 -> see profiles/r06_faust_network_convs.txt for call 00 -- instead of the generic kernel's
  // scalar loads and VALU outer products.
  if (channels > 0 && (channels % 16 == 0 || channels < 16)) {
    int shift = -1;
    for (int sft = 0; sft < 8; ++sft)
      if ((1 << sft) == g.f_nb) shift = sft;
    // two frames per wavefront share the gather (2 waves/SIMD: 16 KB of gT fragments per wavefront); one row per
    // wavefront (odd F, or SE3_PG_SINGLE for any F) gathers each neighbour row once per centre frame but runs at 3
    static const bool force_single = getenv("SE3_PG_SINGLE") != nullptr;
    const bool two = g.f_ctr % 2 == 0 && !force_single;
    const int64_t items = two ? rows / 2 : rows;
    // row range (slices of the rows, api.hip): items item_lo .. item_lo + n_range - 1 only
    const int64_t item_lo = row_lo >= 0 ? row_lo / (two ? 2 : 1) : 0;
    const int64_t n_range = row_lo >= 0 ? row_hi / (two ? 2 : 1) - item_lo : items;
    if (n_range <= 0) {
      *n_used = 0;
      return SE3_OK;
    }
    const int blocks_y = edge_param_grad_bf16_channel_blocks(channels);
    // pair form (two wavefronts share an item and its grad_T image, 3 wavefronts per SIMD): SE3_PG_PAIR=0 turns it off
    static const bool pair_on = [] {
      const char* e = getenv("SE3_PG_PAIR");
      return e == nullptr || atoi(e) != 0;
    }();
    // 32-channel rows (DFaust's first level) take the pair form too, with half the grad_T image (CH16 = 2): 0.258 against
    // 0.284 ms on the DFaust F = 2 batch (profiles/r03_c32_pair_forms_ab.txt); SE3_PG_PAIR_C32=0 keeps the 512-thread form.
    // (The same idea for the edge_t kernel -- a split-K wave pair for 32-channel rows -- measured 7-10 % SLOWER than the
    // single-wavefront kernel and is not kept, same file.)
    static const bool pair32_on = [] {
      const char* e = getenv("SE3_PG_PAIR_C32");
      return e == nullptr || atoi(e) != 0;
    }();
    if (pair_on && two && (channels >= 64 || (channels == 32 && pair32_on))) {
      const int n_cu = device_cu_count();
      if (n_cu <= 0) return SE3_ERR_LAUNCH;
      static const int per_cu = [] {
        const char* e = getenv("SE3_PG_PAIR_WGS");
        return e ? atoi(e) : (SE3_PG_PAIR_LEAN ? 8 : 6);  // 19 KB / 26 KB of LDS per workgroup
      }();
      int64_t wgs = (int64_t)n_cu * per_cu;
      // extents in lane registers (pipe) where a resident workgroup walks at most 64 items; the per-item extent loads
      // otherwise -- never more workgroups than the chip holds at once for the registers' sake: 3 450 workgroups of 64 items
      // on dfaust_f4's level 0 ran as 2.25 rounds of 1 536 and took 1.52 instead of 1.23 ms, and the extent loads were never
      // what the kernel waited on (profiles/r06_param_grad_pipeline_ab.txt)
      if (wgs > n_partials) wgs = n_partials;
      if (wgs > n_range) wgs = n_range;
      if (wgs < 1) wgs = 1;
      const int pipe = (n_range + wgs - 1) / wgs <= 64 && item_lo + n_range < (1ll << 31) ? 1 : 0;
      *n_used = (int)wgs * blocks_y;
      const dim3 pgrid((unsigned)wgs, (unsigned)blocks_y);
      if (channels == 32 && shift >= 0)
        hipLaunchKernelGGL((edge_param_grad_bf16_v2_kernel<2, 2, true, true>), pgrid, dim3(128), 0, stream, g, feat, channels,
                           feat_rows, axes_ext, rho, grad_t, partials, n_range, shift, item_lo, pipe);
      else if (channels == 32)
        hipLaunchKernelGGL((edge_param_grad_bf16_v2_kernel<2, 2, true, false>), pgrid, dim3(128), 0, stream, g, feat, channels,
                           feat_rows, axes_ext, rho, grad_t, partials, n_range, shift, item_lo, pipe);
      else if (gt16 && shift >= 0)
        hipLaunchKernelGGL((edge_param_grad_bf16_v2_kernel<4, 2, true, true, true>), pgrid, dim3(128), 0, stream, g, feat, channels,
                           feat_rows, axes_ext, rho, grad_t, partials, n_range, shift, item_lo, pipe);
      else if (gt16)
        hipLaunchKernelGGL((edge_param_grad_bf16_v2_kernel<4, 2, true, false, true>), pgrid, dim3(128), 0, stream, g, feat, channels,
                           feat_rows, axes_ext, rho, grad_t, partials, n_range, shift, item_lo, pipe);
      else if (shift >= 0)
        hipLaunchKernelGGL((edge_param_grad_bf16_v2_kernel<4, 2, true, true>), pgrid, dim3(128), 0, stream, g, feat, channels,
                           feat_rows, axes_ext, rho, grad_t, partials, n_range, shift, item_lo, pipe);
      else
        hipLaunchKernelGGL((edge_param_grad_bf16_v2_kernel<4, 2, true, false>), pgrid, dim3(128), 0, stream, g, feat, channels,
                           feat_rows, axes_ext, rho, grad_t, partials, n_range, shift, item_lo, pipe);
      return check_launch();
    }
    const int n_blocks = n_partials < 512 ? n_partials : 512;  // the 512-thread form: one workgroup per CU and round
    *n_used = n_blocks * blocks_y;
    const dim3 grid((unsigned)n_blocks, (unsigned)blocks_y);
#define SE3_PG_L(CH16, NFR, P2, THREADS)                                                                                  \
  hipLaunchKernelGGL((edge_param_grad_bf16_v2_kernel<CH16, NFR, false, P2>), grid, dim3(THREADS), 0, stream, g, feat, channels, \
                     feat_rows, axes_ext, rho, grad_t, partials, n_range, shift, item_lo)
#define SE3_PG(CH16)                                       \
  do {                                                     \
    if (two && shift >= 0) SE3_PG_L(CH16, 2, true, 512);   \
    else if (two) SE3_PG_L(CH16, 2, false, 512);           \
    else if (shift >= 0) SE3_PG_L(CH16, 1, true, 256);     \
    else SE3_PG_L(CH16, 1, false, 256);                    \
  } while (0)
    switch (channels >= 64 ? 4 : (channels + 15) / 16) {
      case 1: SE3_PG(1); break;
      case 2: SE3_PG(2); break;
      case 3: SE3_PG(3); break;
      default: SE3_PG(4); break;
    }
#undef SE3_PG
#undef SE3_PG_L
    return check_launch();
  }
  *n_used = n_partials < 512 ? n_partials : 512;  // generic fallback: two workgroups per CU, as the fp32 kernel
  hipLaunchKernelGGL(edge_param_grad_bf16_kernel, dim3(*n_used), dim3(256), 0, stream, g, feat, channels, axes_ext,
                     rho, grad_t, partials, rows);
  return check_launch();
}

}  // namespace se3
