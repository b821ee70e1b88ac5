// Feature gradient of a convolution whose INPUT cloud is much larger than its output cloud (a down-convolution, level
// l -> l + 1 of an encoder; reference: models/Encoder.py:137-147,167-171), split-bf16 arithmetic.
//
// The default feature gradient is a transposed convolution: U[(p,b), o, k] = sum over the edges into source row (p,b) of
// phi[e,k] g[m(e),o], then dX = U W' -- one row of U (C_out * 32 values) per SOURCE row, whatever the number of edges.  A
// down-convolution has 8 frame-edges per source row: its U (131 k rows x 6 KB at the headline hierarchy) is written and
// read back for a sixth of a same-level layer's edges, 0.54 of the convolution's 0.72 ms.  Here the
// gradient goes the way the reference's kernel does (feat_basis_proj_grads.cu:91-143: gFeat += sum_k gT[m,i,k] basis[e,k])
// but without its global float atomics:
//
//   D[e', i]   = sum_{a < F_out} sum_k phi_a[e', k] gT[(s,a), i, k]     one row per POINT-edge x input frame (e' = e F_in + b):
//                                                                       edge_dx_bf16_kernel, sample-major, grad_T rows
//                                                                       (there anyway for the parameter gradients) in LDS
//   dX[(p,b)]  = (1 / kGeluOut) sum over the edges e into p of D[e F_in + b]        dx_gather_sum_kernel, source-major list
//
// Work and bytes are proportional to the edges (D: E F_in C_in floats, written once and gathered once), not to the source
// rows; every sum has a fixed order (deterministic).
//
// edge_dx_bf16_kernel, one 128-thread workgroup per sample point, its centre frames two at a time (a0, a0 + 1):
//   * wavefront v parks the grad_T row of frame a0 + v in LDS as MFMA B fragments of D = phi gT^T (K index = basis
//     function, N = channel): lane (i, h') holds gT[i][k] for the 8 k of its half of a k-step -- two 16-byte loads;
//   * chunks of 32 frame-edges, wavefront v takes chunks v, v + 2, ...: lane n (both halves) gathers the neighbour's
//     geometry record, half h builds the descriptor against centre frame a0 + h (as edge_item_bf16<.., FC = 2>);
//   * the kernel MLP is evaluated TRANSPOSED, pre^T[k, n] = [A; beta]^T[k, d] desc^T[d, n] (the two operands of the edge
//     kernels' MLP product exchanged): the accumulator then holds, per lane n, the 16 basis functions k = acc_row(r, h) --
//     after GELU and the hi / lo split exactly the A fragment of D (M = frame-edge n, K-slot (h, j) of k-step s <-> basis
//     function acc_row(8 s + j, h)), no lane moves data;
//   * D[n, i] accumulates over both centre frames and both k-steps (24 MFMAs per chunk at 64 channels) and is stored as
//     fp32 rows, 128 contiguous bytes per half-wavefront and register.
#include <cstdlib>

#include "common.h"
#include "edge_bf16_body.h"

#ifndef SE3_DX_WAVES
#define SE3_DX_WAVES 3
#endif

namespace se3 {

namespace {

// CT = 32-channel tiles per workgroup: 1 (rows of 32 channels) or 2 (64-channel blocks over blockIdx.y)
template <int CT, bool POW2>
__global__ __launch_bounds__(128, SE3_DX_WAVES) void edge_dx_bf16_kernel(EdgeGeom g, const float* __restrict__ axes_ext,
                                                              const float* __restrict__ rho_p,
                                                              const uint32_t* __restrict__ grad_t, int row_ch,
                                                              float* __restrict__ d_out, int64_t n_items, int fnb_shift) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_w[2][2][64][4];
  __shared__ __attribute__((aligned(16))) uint32_t lds_gt[2][CT][2][2][64][4];  // [frame][tile][k-step][hi/lo][lane]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kcol = lane & 31, h = lane >> 5;
  const int c_off = 32 * CT * (int)blockIdx.y;
  if (threadIdx.x < 64) mlp_weights_to_lds<2>(lds_w, axes_ext, threadIdx.x);
  const float rho = *rho_p;
  const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
  const __amdgpu_buffer_rsrc_t ctrg_rs = buffer_of(g.ctr_geom, g.n_ctr * g.f_ctr * 64);

  // item = sample point; its centre frames are taken two at a time, every pair adding into the same D rows (a D row
  // belongs to a point-edge and an INPUT frame: all F_out centre frames contribute).  The wavefront that stored a chunk's
  // rows for the first pair is the one that reads them back for the next (same chunk assignment): program order suffices.
  for (int64_t ctr = blockIdx.x; ctr < n_items; ctr += gridDim.x) {
    const int start = ctr > 0 ? g.ends[ctr - 1] : 0;
    const int n_total = (g.ends[ctr] - start) * g.f_nb;
    if (n_total == 0) continue;  // uniform over the workgroup
  for (int a0 = 0; a0 < g.f_ctr; a0 += 2) {
    __syncthreads();             // the previous pair's fragments are no longer read (first one: the MLP weights are in place)

    auto row_of_fe = [&](int fe) {
      const int e = start + (POW2 ? fe >> fnb_shift : fe / g.f_nb);
      const int nb = g.nbr[(int64_t)e * g.nbr_stride + g.nbr_offset];
      return (POW2 ? nb << fnb_shift : nb * g.f_nb) + (POW2 ? fe & ((1 << fnb_shift) - 1) : fe % g.f_nb);
    };
    const int c_first = 32 * wv;
    float xn_nx[3], rn_nx[9], yc[3], rc[9];
    // Every load of the item's start goes out before any result is needed: the centre's record (half h: centre frame a0 + h),
    // the first chunk's neighbour id, this wavefront's grad_T row (its frame of the pair), and -- once the id is there --
    // the neighbour's record; the fragments are built and parked in LDS behind all of that.
    load_geom_record(ctrg_rs, (int)(ctr * g.f_ctr + a0 + h), yc, rc);
    const int fe0 = min(c_first + kcol, n_total - 1);
    int nb0 = 0;
    if (c_first < n_total) nb0 = g.nbr[(int64_t)(start + (POW2 ? fe0 >> fnb_shift : fe0 / g.f_nb)) * g.nbr_stride + g.nbr_offset];
    uint4 gw[CT][2][2];
    {
      const uint32_t* gt_row = grad_t + (((int64_t)ctr * g.f_ctr + a0 + wv) * row_ch + c_off) * kBasis;
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const uint32_t* src = gt_row + (32 * t + kcol) * kBasis + 16 * s + 4 * h;
          gw[t][s][0] = *reinterpret_cast<const uint4*>(src);       // k = 16 s + 4 h + 0..3  = acc_row(8 s + j, h), j = 0..3
          gw[t][s][1] = *reinterpret_cast<const uint4*>(src + 8);   // k = 16 s + 8 + 4 h + .. = acc_row(8 s + j, h), j = 4..7
        }
    }
    if (c_first < n_total)
      load_geom_record(nbg_rs, (POW2 ? nb0 << fnb_shift : nb0 * g.f_nb) + (POW2 ? fe0 & ((1 << fnb_shift) - 1) : fe0 % g.f_nb), xn_nx, rn_nx);
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int s = 0; s < 2; ++s) {  // grad_T words -> B fragments of D = phi gT^T
        const uint32_t w[8] = {gw[t][s][0].x, gw[t][s][0].y, gw[t][s][0].z, gw[t][s][0].w,
                               gw[t][s][1].x, gw[t][s][1].y, gw[t][s][1].z, gw[t][s][1].w};
        u32x4 f_hi, f_lo;
        frags_from_words(w, f_hi, f_lo);
        *reinterpret_cast<u32x4*>(&lds_gt[wv][t][s][0][lane][0]) = f_hi;
        *reinterpret_cast<u32x4*>(&lds_gt[wv][t][s][1][lane][0]) = f_lo;
      }
    __syncthreads();  // both frames' fragments are in place

    for (int c0 = c_first; c0 < n_total; c0 += 64) {
      const int cnt = min(32, n_total - c0);
      float xn[3], rn[9], d[9];
#pragma unroll
      for (int i = 0; i < 3; ++i) xn[i] = xn_nx[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) rn[i] = rn_nx[i];
      if (c0 + 64 < n_total) load_geom_record(nbg_rs, row_of_fe(min(c0 + 64 + kcol, n_total - 1)), xn_nx, rn_nx);  // next chunk of this wavefront
      edge_descriptor(xn, rn, yc, rc, rho, d);

      // descriptor pieces as in edge_item_bf16<.., FC = 2>: own dims 0..7, and {dim 8 of the other half's frame, 1}
      u32x4 own_hi, own_lo, oth_hi, oth_lo;
      frags_from_floats(d, own_hi, own_lo);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d[8]), __float_as_uint(d[8]), false, false);
        const float d8 = __uint_as_float(h ? sw[0] : sw[1]);
        uint32_t p_hi, p_lo;
        split2(d8, 1.0f, p_hi, p_lo);
        oth_hi = u32x4{p_hi, 0u, 0u, 0u};
        oth_lo = u32x4{p_lo, 0u, 0u, 0u};
      }
      f32x16 dacc[CT];
#pragma unroll
      for (int t = 0; t < CT; ++t) dacc[t] = zero16();
#pragma unroll 1
      for (int a = 0; a < 2; ++a) {  // one centre frame at a time (unrolled, the two frames' fragments all stay live: 256 VGPRs)
        const bool dims07 = h == a;
        u32x4 b_hi, b_lo;  // desc^T as the B operand: K = descriptor dim, N = frame-edge
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          b_hi[i] = dims07 ? own_hi[i] : oth_hi[i];
          b_lo[i] = dims07 ? own_lo[i] : oth_lo[i];
        }
        const u32x4 wa_hi = *reinterpret_cast<const u32x4*>(&lds_w[a][0][lane][0]);  // [A; beta]^T as the A operand: M = basis function
        const u32x4 wa_lo = *reinterpret_cast<const u32x4*>(&lds_w[a][1][lane][0]);
        const f32x16 pre_t = mfma_bf16x3(wa_hi, wa_lo, b_hi, b_lo, zero16());  // register r, lane (n, h): k = acc_row(r, h)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pv[j] = gelu_scaled(pre_t[8 * s + j]);
          u32x4 a_hi, a_lo;
          frags_from_floats(pv, a_hi, a_lo);
#pragma unroll
          for (int t = 0; t < CT; ++t) {
            const u32x4 g_hi = *reinterpret_cast<const u32x4*>(&lds_gt[a][t][s][0][lane][0]);
            const u32x4 g_lo = *reinterpret_cast<const u32x4*>(&lds_gt[a][t][s][1][lane][0]);
            dacc[t] = mfma_bf16x3(a_hi, a_lo, g_hi, g_lo, dacc[t]);
          }
        }
      }
      // dacc[t] register r, lane (i = kcol, h) = D[frame-edge c0 + acc_row(r, h)][c_off + 32 t + i]
      float* rows = d_out + ((int64_t)start * g.f_nb + c0) * row_ch + c_off + kcol;
#pragma unroll
      for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = acc_row(r, h);
          if (n < cnt) {
            float* dst = rows + (int64_t)n * row_ch + 32 * t;
            *dst = a0 == 0 ? dacc[t][r] : *dst + dacc[t][r];
          }
        }
    }
  }
  }
}

// dX block of source point p (its F_in rows are contiguous: `width` = F_in * channels floats, a multiple of 4) = scale * sum
// over the edges into p of D's block of that edge.  A group of G lanes per source point (G = 16, 32 or 64: 16 bytes per
// lane and step), 64 / G points per wavefront; the source-major list names the sample of every edge, the edge's position in
// the sample-major list comes from the transposition (t_edge_ids) or, without it, from scanning that sample's (short)
// neighbour list, one list entry per lane of the group.
template <int G>
__global__ __launch_bounds__(256) void dx_gather_sum_kernel(const float* __restrict__ d_rows, const int32_t* __restrict__ neighbors,
                                                            const int32_t* __restrict__ ends, const int32_t* __restrict__ t_samples,
                                                            const int32_t* __restrict__ t_ends, const int32_t* __restrict__ t_edge_ids,
                                                            int64_t n_src, int width, float scale, float* __restrict__ grad_feat) {
  constexpr int PER_WAVE = 64 / G;
  const int lane = threadIdx.x & 63, gl = lane & (G - 1), grp = lane / G;
  const int w4 = width >> 2;  // 16-byte pieces per block
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  for (int64_t p0 = wave * PER_WAVE; p0 < n_src; p0 += n_waves * PER_WAVE) {
    const int64_t p = p0 + grp;
    const bool live = p < n_src;
    float4 acc[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    const int t0 = live ? (p > 0 ? t_ends[p - 1] : 0) : 0, t1 = live ? t_ends[p] : 0;
    // the groups of a wavefront walk their lists in lockstep: G entries at a time, as many rounds as the longest list needs
    int rounds = (t1 - t0 + G - 1) / G;
#pragma unroll
    for (int m = G; m < 64; m <<= 1) rounds = max(rounds, __shfl_xor(rounds, m));
    for (int r = 0; r < rounds; ++r) {
      const int j = t0 + r * G + gl;
      int e = -1;
      if (j < t1) {
        if (t_edge_ids) {  // the transposition's own record of where every entry came from (se3_csr_transpose, ABI 4)
          e = t_edge_ids[j];
        } else {
          const int s = t_samples[j];
          const int st = s > 0 ? ends[s - 1] : 0, en = ends[s];
#pragma unroll 8
          for (int i = st; i < en; ++i) e = neighbors[(int64_t)i * 2 + 1] == (int)p ? i : e;  // (no match: not an edge, nothing to add)
        }
      }
      const int cnt = min(G, t1 - (t0 + r * G));  // <= 0 for a group whose list has ended
      // four blocks in flight per lane: the loads of a step are issued before any of them is added (fixed order of the sum)
      for (int j4 = 0; j4 < G; j4 += 4) {
        int ee[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ee[u] = __shfl(e, grp * G + j4 + u);
        if (__ballot(j4 < cnt) == 0) break;  // every group of the wavefront is past the end of its list
        float4 v0[4], v1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = j4 + u < cnt && ee[u] >= 0;
          const float4* blk = reinterpret_cast<const float4*>(d_rows + (int64_t)(ok ? ee[u] : 0) * width);
          v0[u] = ok && gl < w4 ? blk[gl] : make_float4(0.f, 0.f, 0.f, 0.f);
          if (G == 64) v1[u] = ok && gl + 64 < w4 ? blk[gl + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[0].x += v0[u].x, acc[0].y += v0[u].y, acc[0].z += v0[u].z, acc[0].w += v0[u].w;
          if (G == 64) acc[1].x += v1[u].x, acc[1].y += v1[u].y, acc[1].z += v1[u].z, acc[1].w += v1[u].w;
        }
      }
    }
    if (!live) continue;
    float4* out = reinterpret_cast<float4*>(grad_feat + p * width);
    if (gl < w4) out[gl] = make_float4(acc[0].x * scale, acc[0].y * scale, acc[0].z * scale, acc[0].w * scale);
    if (G == 64 && gl + 64 < w4) out[gl + 64] = make_float4(acc[1].x * scale, acc[1].y * scale, acc[1].z * scale, acc[1].w * scale);
  }
}

}  // namespace

// Shapes the edge-major feature gradient is implemented for: pairs of centre frames, rows of 32 channels or of whole
// 64-channel blocks, a dX block of at most 512 floats per source point.
bool edge_dx_bf16_applicable(const EdgeGeom& g, int channels) {
  return g.f_ctr % 2 == 0 && (channels == 32 || (channels % 64 == 0 && channels > 0)) && g.f_nb * channels <= 512;
}

// grad_t: packed words [rows_out, channels, 32] (the grad_T GEMM's output, alpha folded in); d_rows: [edge rows * f_nb, channels]
int launch_edge_dx_bf16(const char* tag, const EdgeGeom& g, const float* axes_ext, const float* rho, const uint32_t* grad_t,
                        int channels, float* d_rows, hipStream_t stream) {
  if (!edge_dx_bf16_applicable(g, channels)) return SE3_ERR_UNSUPPORTED;
  const int64_t items = g.n_ctr;  // one workgroup per sample point (all its centre-frame pairs)
  if (items == 0) return SE3_OK;
  ProfScope prof(tag, stream);
  int shift = -1;
  for (int sft = 0; sft < 8; ++sft)
    if ((1 << sft) == g.f_nb) shift = sft;
  const int ct = channels == 32 ? 1 : 2;
  const dim3 grid((unsigned)(items < (1 << 20) ? items : (1 << 20)), (unsigned)(channels / (32 * ct)));
#define SE3_DX(CT, P2) \
  hipLaunchKernelGGL((edge_dx_bf16_kernel<CT, P2>), grid, dim3(128), 0, stream, g, axes_ext, rho, grad_t, channels, d_rows, items, shift)
  if (ct == 1) {
    if (shift >= 0) SE3_DX(1, true); else SE3_DX(1, false);
  } else {
    if (shift >= 0) SE3_DX(2, true); else SE3_DX(2, false);
  }
#undef SE3_DX
  return check_launch();
}

int launch_dx_gather_sum(const char* tag, const float* d_rows, const int32_t* neighbors, const int32_t* ends,
                         const int32_t* t_samples, const int32_t* t_ends, const int32_t* t_edge_ids, int64_t n_src, int width,
                         float scale, float* grad_feat, hipStream_t stream) {
  if (n_src == 0) return SE3_OK;
  if (width < 1 || width > 512) return SE3_ERR_UNSUPPORTED;
  if (width % 4 != 0) return SE3_ERR_UNSUPPORTED;
  ProfScope prof(tag, stream);
  const int g = width <= 64 ? 16 : (width <= 128 ? 32 : 64);  // lanes per source point: 16 bytes per lane and step
  int64_t blocks = (n_src + 4 * (64 / g) - 1) / (4 * (64 / g));
  if (blocks > (1 << 20)) blocks = 1 << 20;
#define SE3_GATHER(G)                                                                                                      \
  hipLaunchKernelGGL(dx_gather_sum_kernel<G>, dim3((unsigned)blocks), dim3(256), 0, stream, d_rows, neighbors, ends, t_samples, \
                     t_ends, t_edge_ids, n_src, width, scale, grad_feat)
  if (g == 16) SE3_GATHER(16);
  else if (g == 32) SE3_GATHER(32);
  else SE3_GATHER(64);
#undef SE3_GATHER
  return check_launch();
}

}  // namespace se3
