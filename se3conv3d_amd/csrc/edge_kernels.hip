// Edge-phase kernels of the SE(3) point convolution for gfx950 (wave64, fp32 MFMA).
//
// One wavefront owns one output row m = (centre point, centre frame).  For that row it walks
// the centre's edge group in chunks of 32 frame-edges n = (neighbour point, neighbour frame):
//
//   1. lane n builds the 9-D descriptor of frame-edge n                      (VALU, 1 edge/lane)
//   2. pre[n,k] = [desc, 1] . [A; beta]  as 5 x v_mfma_f32_32x32x2_f32       (rows n, cols k)
//      -> the accumulator holds pre[n = acc_row(r,h)][k = lane&31] in register r
//   3. phi = GELU(pre) elementwise on the accumulator registers              (VALU)
//   4. T[m][i,k] += sum_n feat[q(n), i] * phi[n,k] as 32x32x2 MFMAs whose B operand IS the
//      register file of step 3 (k-step r pairs the frame-edges acc_row(r,0) and acc_row(r,1)),
//      and whose A operand is gathered straight from HBM/L2: half-wave h reads 32*VW
//      consecutive channels of source row q(acc_row(r,h)) -- 128*VW contiguous bytes.
//
// Nothing E'-sized (descriptors, basis values, frame-level edge lists) is ever written: the
// reference materialises all three (PNEConvLayerRotEquiv.py:62-128,199-203).
//
// The same kernel serves the transposed graph (centre = input point, edges lead to output
// points) for the feature gradient, see api.hip.
#include "common.h"

namespace se3 {

namespace {

struct RowInfo {
  int64_t ctr;
  int fc;
  int start, n_total;
};

__device__ __forceinline__ RowInfo row_info(const EdgeGeom& g, int64_t m) {
  RowInfo r;
  r.ctr = m / g.f_ctr;
  r.fc = (int)(m - r.ctr * g.f_ctr);
  r.start = r.ctr > 0 ? g.ends[r.ctr - 1] : 0;
  r.n_total = (g.ends[r.ctr] - r.start) * g.f_nb;
  return r;
}

// Descriptor of the frame-edge handled by this lane (+ the source feature row it reads).
// `fe` must be a valid frame-edge index of the row (callers clamp).
__device__ __forceinline__ void lane_descriptor(const EdgeGeom& g, const RowInfo& ri, int fe, const float yc[3],
                                                const float rc[9], float rho, float d[kDescExt], int& q) {
  const int e = ri.start + fe / g.f_nb;
  const int fn = fe % g.f_nb;
  const int nb = g.nbr[(int64_t)e * g.nbr_stride + g.nbr_offset];
  q = nb * g.f_nb + fn;
  float xn[3], rn[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) xn[i] = g.nb_pts[(int64_t)nb * 3 + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) rn[i] = g.nb_frames[(int64_t)q * 9 + i];
  if (!g.transposed)
    edge_descriptor(xn, rn, yc, rc, rho, d);  // neighbour = input side, centre = output side
  else
    edge_descriptor(yc, rc, xn, rn, rho, d);  // centre = input side, neighbour = output side
  d[9] = 1.0f;
}

__device__ __forceinline__ f32x16 mlp_preactivation(const float d[kDescExt], const float bmlp[5], int h) {
  f32x16 pre = zero16();
#pragma unroll
  for (int t = 0; t < 5; ++t) pre = mfma32(h ? d[2 * t + 1] : d[2 * t], bmlp[t], pre);
  return pre;
}

// ------------------------------------------------------------------------------------------------
// T[m, c, k] = sum over the row's frame-edges of feat[q, c] * GELU(desc . A + beta)[k]
//
// Software pipeline as in the split-bf16 kernels (edge_bf16_body.h): packed geometry records, neighbour ids two chunks
// ahead, records one chunk ahead, the chunk's 16 x VW gathered feature values loaded at its top -- unconditionally:
// frame-edges past the end of the list read out of bounds (zeros), so phi needs no mask and no counted s_waitcnt has to
// assume a skipped load -- and consumed after descriptor, kernel MLP and GELU.  (Round-1 version: ids -> points and
// frames from the reference's separate arrays -> descriptor -> MLP -> GELU -> gathers in four guarded groups, each
// link waiting for the one before: 0.78 ms at the headline level against 0.28 ms of MFMA issue.)
// ------------------------------------------------------------------------------------------------
#ifndef SE3_E32_ABLATE
#define SE3_E32_ABLATE 0  // diagnostic builds of edge_t_kernel (wrong results): 1 no GELU, 2 no feature gather, 4 no stores, 8 no aggregation MFMAs
#endif
template <int VW>
__global__ __launch_bounds__(256, VW == 4 ? 2 : 3) void edge_t_kernel(EdgeGeom g, const float* __restrict__ feat,
                                                                      int channels, int64_t feat_rows,
                                                                      const float* __restrict__ axes_ext,
                                                                      const float* __restrict__ rho_p,
                                                                      float* __restrict__ t_out, int64_t rows,
                                                                      int fnb_shift) {
  const int lane = threadIdx.x & 63;
  const int64_t m = __builtin_amdgcn_readfirstlane((int)((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)));
  if (m >= rows) return;
  const int kcol = lane & 31, h = lane >> 5;
  const float rho = *rho_p;
  float bmlp[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) bmlp[t] = axes_ext[(2 * t + h) * kBasis + kcol];

  // rows < 2^31 (checked on the host): 32-bit division
  const int64_t ctr = (uint32_t)m / (uint32_t)g.f_ctr;
  const int start = ctr > 0 ? g.ends[ctr - 1] : 0;
  const int n_total = (g.ends[ctr] - start) * g.f_nb;
  const __amdgpu_buffer_rsrc_t feat_rs = buffer_of(feat, feat_rows * channels * 4);
  const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
  float yc[3], rc[9];
  load_geom_record(buffer_of(g.ctr_geom, g.n_ctr * g.f_ctr * 64), (int)m, yc, rc);
  const int row_bytes = channels * 4;
  const int hb = 16 * h;  // ds_bpermute byte address of lane 4h

  auto nbr_of = [&](int c0) {
    const int fe = min(c0 + kcol, n_total - 1);
    const int e = start + (fnb_shift >= 0 ? fe >> fnb_shift : fe / g.f_nb);
    return g.nbr[(int64_t)e * g.nbr_stride + g.nbr_offset];
  };
  auto row_of = [&](int nb, int c0) {
    const int fe = min(c0 + kcol, n_total - 1);
    return nb * g.f_nb + (fnb_shift >= 0 ? fe & ((1 << fnb_shift) - 1) : fe % g.f_nb);
  };

  float* t_row = t_out + m * (int64_t)channels * kBasis;
  for (int cbase = 0; cbase < channels; cbase += 32 * VW) {
    const int cb = cbase + VW * kcol;     // first channel this lane feeds as MFMA row `kcol`
    const bool ch_ok = cb < channels;     // channels % VW == 0 (host guarantees) => whole vector valid
    const int cb4 = (ch_ok ? cb : 0) * 4;
    f32x16 acc[VW];
#pragma unroll
    for (int t = 0; t < VW; ++t) acc[t] = zero16();

    int nb_b = 0, q_a = 0;
    float xn_nx[3], rn_nx[9];
    if (n_total > 0) {
      const int nb_a = nbr_of(0);
      nb_b = nbr_of(32);
      q_a = row_of(nb_a, 0);
      load_geom_record(nbg_rs, q_a, xn_nx, rn_nx);
    }
    for (int c0 = 0; c0 < n_total; c0 += 32) {
      const int cnt = min(32, n_total - c0);
      const int qoff = c0 + kcol < n_total ? q_a * row_bytes : kOobOffset;
      float xn[3], rn[9], d[kDescExt];
#pragma unroll
      for (int i = 0; i < 3; ++i) xn[i] = xn_nx[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) rn[i] = rn_nx[i];
      const int q_b = row_of(nb_b, c0 + 32);
      nb_b = nbr_of(c0 + 64);

      // k-step r of the aggregation pairs the frame-edges acc_row(r, 0) and acc_row(r, 1): lane half h reads VW
      // consecutive channels of the source row of frame-edge acc_row(r, h), whose byte offset it fetches from that lane
      float a[16][VW];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int src_off = __builtin_amdgcn_ds_bpermute(hb + 4 * acc_row(r, 0), qoff);
        const int voff = ch_ok ? src_off + cb4 : kOobOffset;
        if constexpr ((SE3_E32_ABLATE & 2) != 0) {
#pragma unroll
          for (int t = 0; t < VW; ++t) a[r][t] = __uint_as_float((uint32_t)(voff + t) * 2654435761u) * 1e-30f;
        } else if constexpr (VW == 4) {
          const auto v = __builtin_amdgcn_raw_buffer_load_b128(feat_rs, voff, 0, 0);
#pragma unroll
          for (int t = 0; t < 4; ++t) a[r][t] = __uint_as_float(v[t]);
        } else if constexpr (VW == 2) {
          const auto v = __builtin_amdgcn_raw_buffer_load_b64(feat_rs, voff, 0, 0);
          a[r][0] = __uint_as_float(v[0]), a[r][1] = __uint_as_float(v[1]);
        } else {
          a[r][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(feat_rs, voff, 0, 0));
        }
      }
      load_geom_record(nbg_rs, q_b, xn_nx, rn_nx);
      q_a = q_b;

      if (!g.transposed)
        edge_descriptor(xn, rn, yc, rc, rho, d);  // neighbour = input side, centre = output side
      else
        edge_descriptor(yc, rc, xn, rn, rho, d);  // centre = input side, neighbour = output side
      d[9] = 1.0f;
      f32x16 phi = mlp_preactivation(d, bmlp, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) phi[r] = (SE3_E32_ABLATE & 1) ? phi[r] : gelu_erf(phi[r]);

#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (g4 * 8 < cnt) {  // wave-uniform: frame-edges 8*g4 .. 8*g4+7 of the chunk
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int t = 0; t < VW; ++t) {
              if (SE3_E32_ABLATE & 8) acc[t][rr] += a[g4 * 4 + rr][t] * phi[g4 * 4 + rr];
              else acc[t] = mfma32(a[g4 * 4 + rr][t], phi[g4 * 4 + rr], acc[t]);
            }
        }
      }
    }
    // acc[t] register r, lane (kcol,h) = T[m][cbase + VW*acc_row(r,h) + t][kcol]
#pragma unroll
    for (int t = 0; t < VW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ch = cbase + VW * acc_row(r, h) + t;
        // non-temporal: the GEMM that follows does not run against this kernel's write-back (see edge_bf16.hip)
        if ((SE3_E32_ABLATE & 4) && __float_as_uint(acc[t][r]) != 0x12345678u) continue;
        if (ch < channels) __builtin_nontemporal_store(acc[t][r], &t_row[(int64_t)ch * kBasis + kcol]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// Gradient of the kernel-MLP parameters.  Per row m and chunk of 32 frame-edges:
//   gphi[n,k] = sum_i feat[q(n), i] * gT[m][i,k]          (MFMA: rows n, cols k, k-dim = channels)
//   gpre      = gphi * GELU'(pre)
//   d[A;beta][j,k] += desc_ext[n,j] * gpre[n,k]           (VALU, descriptor broadcast through LDS)
// Every block writes one [10,32] partial; a second kernel sums them (no atomics: all blocks would
// hit the same 320 addresses).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void edge_param_grad_kernel(EdgeGeom g, const float* __restrict__ feat,
                                                              int channels, const float* __restrict__ axes_ext,
                                                              const float* __restrict__ rho_p,
                                                              const float* __restrict__ grad_t,
                                                              float* __restrict__ partials, int64_t rows) {
  __shared__ __attribute__((aligned(16))) float lds_desc[4][32][12];
  __shared__ float lds_red[4][kDescExt][kBasis];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kcol = lane & 31, h = lane >> 5;
  const float rho = *rho_p;
  float bmlp[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) bmlp[t] = axes_ext[(2 * t + h) * kBasis + kcol];
  float dacc[kDescExt];
#pragma unroll
  for (int j = 0; j < kDescExt; ++j) dacc[j] = 0.f;

  const bool vec_ok = (channels % 4) == 0;
  for (int64_t m = (int64_t)blockIdx.x * 4 + wave; m < rows; m += (int64_t)gridDim.x * 4) {
    const RowInfo ri = row_info(g, m);
    if (ri.n_total == 0) continue;
    float yc[3], rc[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) yc[i] = g.ctr_pts[ri.ctr * 3 + i];
#pragma unroll
    for (int i = 0; i < 9; ++i) rc[i] = g.ctr_frames[m * 9 + i];
    const float* gt_row = grad_t + m * (int64_t)channels * kBasis;

    for (int c0 = 0; c0 < ri.n_total; c0 += 32) {
      const int cnt = min(32, ri.n_total - c0);
      const int fe = c0 + min(kcol, cnt - 1);
      float d[kDescExt];
      int q;
      lane_descriptor(g, ri, fe, yc, rc, rho, d, q);
      const f32x16 pre = mlp_preactivation(d, bmlp, h);
      if (h == 0) {
        float4* dst = reinterpret_cast<float4*>(&lds_desc[wave][kcol][0]);
        dst[0] = make_float4(d[0], d[1], d[2], d[3]);
        dst[1] = make_float4(d[4], d[5], d[6], d[7]);
        dst[2] = make_float4(d[8], d[9], 0.f, 0.f);
      }

      // gphi: k-dimension = channels, split as [cb0 + h*hs, cb0 + h*hs + hs) per lane half.
      f32x16 gphi = zero16();
      const float* f_row = feat + (int64_t)q * channels;
      for (int cb0 = 0; cb0 < channels; cb0 += 64) {
        const int crem = min(64, channels - cb0);
        const int hs = (crem + 1) >> 1;  // channels per half in this block (<= 32)
        const int my0 = cb0 + h * hs;    // first channel of this lane's half
        if (vec_ok && (hs % 4) == 0) {
#pragma unroll
          for (int t4 = 0; t4 < 8; ++t4) {
            if (t4 * 4 < hs) {
              const float4 av = *reinterpret_cast<const float4*>(f_row + my0 + t4 * 4);
              const float a[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
              for (int u = 0; u < 4; ++u)
                gphi = mfma32(a[u], gt_row[(int64_t)(my0 + t4 * 4 + u) * kBasis + kcol], gphi);
            }
          }
        } else {
          // (rolled: channel counts that are not multiples of 4 are the rare case)
          for (int t = 0; t < hs; ++t) {
            const int ch = my0 + t;
            const bool ok = ch < cb0 + crem;
            const float a = ok ? f_row[ch] : 0.f;
            const float b = ok ? gt_row[(int64_t)ch * kBasis + kcol] : 0.f;
            gphi = mfma32(a, b, gphi);
          }
        }
      }

      // wave-private LDS hand-off of the descriptors (LDS ops of one wave complete in order).
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = acc_row(r, h);
        float y, dy;
        gelu_erf_grad(pre[r], y, dy);
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

  // block reduction: halves of a wave, then the four waves
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

// The same product for rows of exactly 2 * HS channels (32 or 64: one k block), pipelined like edge_t_kernel above:
// the row's grad_T values (the MFMA B operand, HS registers) are loaded once per row and serve all its chunks, neighbour
// ids run two chunks ahead, geometry records one chunk ahead, and the chunk's own feature values (the lane's source
// row, HS registers) go out at its top, unconditionally -- frame-edges past the end read out of bounds (zeros).
template <int HS>
__global__ __launch_bounds__(256, 3) void edge_param_grad_fast_kernel(EdgeGeom g, const float* __restrict__ feat,
                                                                      int64_t feat_rows,
                                                                      const float* __restrict__ axes_ext,
                                                                      const float* __restrict__ rho_p,
                                                                      const float* __restrict__ grad_t,
                                                                      float* __restrict__ partials, int64_t rows,
                                                                      int fnb_shift) {
  constexpr int C = 2 * HS;
  __shared__ __attribute__((aligned(16))) float lds_desc[4][32][12];
  __shared__ float lds_red[4][kDescExt][kBasis];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kcol = lane & 31, h = lane >> 5;
  const float rho = *rho_p;
  float bmlp[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) bmlp[t] = axes_ext[(2 * t + h) * kBasis + kcol];
  float dacc[kDescExt];
#pragma unroll
  for (int j = 0; j < kDescExt; ++j) dacc[j] = 0.f;
  const __amdgpu_buffer_rsrc_t feat_rs = buffer_of(feat, feat_rows * C * 4);
  const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
  const __amdgpu_buffer_rsrc_t ctrg_rs = buffer_of(g.ctr_geom, g.n_ctr * g.f_ctr * 64);
  const __amdgpu_buffer_rsrc_t gt_rs = buffer_of(grad_t, rows * C * kBasis * 4);

  for (int64_t m = (int64_t)blockIdx.x * 4 + wave; m < rows; m += (int64_t)gridDim.x * 4) {
    const int64_t ctr = (uint32_t)m / (uint32_t)g.f_ctr;  // rows < 2^31 (host)
    const int start = ctr > 0 ? g.ends[ctr - 1] : 0;
    const int n_total = (g.ends[ctr] - start) * g.f_nb;
    if (n_total == 0) continue;
    float yc[3], rc[9];
    load_geom_record(ctrg_rs, (int)m, yc, rc);
    // B operand of every chunk: lane (k = kcol, half h) holds grad_T[m][h * HS + t][k]
    float b[HS];
#pragma unroll
    for (int t = 0; t < HS; ++t)
      b[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gt_rs, (int)(((m * C + h * HS + t) * kBasis + kcol) * 4), 0, 0));

    auto nbr_of = [&](int c0) {
      const int fe = min(c0 + kcol, n_total - 1);
      const int e = start + (fnb_shift >= 0 ? fe >> fnb_shift : fe / g.f_nb);
      return g.nbr[(int64_t)e * g.nbr_stride + g.nbr_offset];
    };
    auto row_of = [&](int nb, int c0) {
      const int fe = min(c0 + kcol, n_total - 1);
      return nb * g.f_nb + (fnb_shift >= 0 ? fe & ((1 << fnb_shift) - 1) : fe % g.f_nb);
    };
    const int nb_a = nbr_of(0);
    int nb_b = nbr_of(32);
    int q_a = row_of(nb_a, 0);
    float xn_nx[3], rn_nx[9];
    load_geom_record(nbg_rs, q_a, xn_nx, rn_nx);

    for (int c0 = 0; c0 < n_total; c0 += 32) {
      const int cnt = min(32, n_total - c0);
      const int voff = c0 + kcol < n_total ? q_a * (C * 4) + h * (HS * 4) : kOobOffset;
      float xn[3], rn[9], d[kDescExt];
#pragma unroll
      for (int i = 0; i < 3; ++i) xn[i] = xn_nx[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) rn[i] = rn_nx[i];
      const int q_b = row_of(nb_b, c0 + 32);
      nb_b = nbr_of(c0 + 64);
      // A operand: lane (n = kcol, half h) reads channels h * HS .. + HS - 1 of its own frame-edge's source row
      float a[HS];
#pragma unroll
      for (int t4 = 0; t4 < HS / 4; ++t4) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(feat_rs, voff + 16 * t4, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) a[4 * t4 + u] = __uint_as_float(v[u]);
      }
      load_geom_record(nbg_rs, q_b, xn_nx, rn_nx);
      q_a = q_b;

      if (!g.transposed)
        edge_descriptor(xn, rn, yc, rc, rho, d);
      else
        edge_descriptor(yc, rc, xn, rn, rho, d);
      d[9] = 1.0f;
      const f32x16 pre = mlp_preactivation(d, bmlp, h);
      if (h == 0) {
        float4* dst = reinterpret_cast<float4*>(&lds_desc[wave][kcol][0]);
        dst[0] = make_float4(d[0], d[1], d[2], d[3]);
        dst[1] = make_float4(d[4], d[5], d[6], d[7]);
        dst[2] = make_float4(d[8], d[9], 0.f, 0.f);
      }
      f32x16 gphi = zero16();
#pragma unroll
      for (int t = 0; t < HS; ++t) gphi = mfma32(a[t], b[t], gphi);

      // wave-private LDS hand-off of the descriptors (LDS ops of one wave complete in order).
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = acc_row(r, h);
        float y, dy;
        gelu_erf_grad(pre[r], y, dy);
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

  // block reduction: halves of a wave, then the four waves
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

}  // namespace

int launch_edge_t(const char* tag, const EdgeGeom& g, const float* feat, int channels, int64_t feat_rows,
                  const float* axes_ext, const float* rho, float* t_out, hipStream_t stream) {
  const int64_t rows = g.n_ctr * g.f_ctr;
  if (rows == 0) return SE3_OK;
  // 32-bit byte offsets into the gathered operand and the packed records
  if (feat_rows * (int64_t)channels * 4 >= (int64_t)kOobOffset || rows >= (1ll << 31) || !g.ctr_geom || !g.nb_geom)
    return SE3_ERR_UNSUPPORTED;
  ProfScope prof(tag, stream);
  int shift = -1;
  for (int sft = 0; sft < 8; ++sft)
    if ((1 << sft) == g.f_nb) shift = sft;
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
#define SE3_LAUNCH(VW) \
  hipLaunchKernelGGL(edge_t_kernel<VW>, grid, block, 0, stream, g, feat, channels, feat_rows, axes_ext, rho, t_out, rows, shift)
  if (channels % 128 == 0) SE3_LAUNCH(4);
  else if (channels % 64 == 0) SE3_LAUNCH(2);
  else SE3_LAUNCH(1);
#undef SE3_LAUNCH
  return check_launch();
}

// slots of partial sums the parameter-gradient kernels may use: up to 2048 workgroups (the pair form of the split-bf16
// kernel keeps 6 x 256 resident; the other forms use at most 512 of them)
int edge_param_grad_blocks(int64_t rows) {
  const int64_t want = (rows + 3) / 4;
  return (int)(want < 2048 ? (want > 0 ? want : 1) : 2048);
}

int launch_edge_param_grad(const char* tag, const EdgeGeom& g, const float* feat, int channels, int64_t feat_rows,
                           const float* axes_ext, const float* rho, const float* grad_t, float* partials,
                           int n_partials, int* n_used, hipStream_t stream) {
  const int64_t rows = g.n_ctr * g.f_ctr;
  ProfScope prof(tag, stream);
  // two workgroups per CU (2048 measured 3 % slower, 4096 11 %): the slot capacity is sized for the pair form of the
  // split-bf16 kernel, this kernel uses at most 512 of the slots
  *n_used = n_partials < 512 ? n_partials : 512;
  // 32-bit byte offsets in the pipelined form: the gathered operand, the packed records, the grad_T rows
  const bool fast = (channels == 32 || channels == 64) && g.ctr_geom && g.nb_geom && rows < (1ll << 31) &&
                    feat_rows * (int64_t)channels * 4 < (int64_t)kOobOffset &&
                    rows * (int64_t)channels * kBasis * 4 < (1ll << 31);
  if (fast) {
    // the pipelined form holds 168 registers: three workgroups per CU (768: 0.78 ms against 0.86 at 512 and 0.82 at 1024)
    *n_used = n_partials < 768 ? n_partials : 768;
    int shift = -1;
    for (int sft = 0; sft < 8; ++sft)
      if ((1 << sft) == g.f_nb) shift = sft;
    if (channels == 64)
      hipLaunchKernelGGL(edge_param_grad_fast_kernel<32>, dim3(*n_used), dim3(256), 0, stream, g, feat, feat_rows, axes_ext,
                         rho, grad_t, partials, rows, shift);
    else
      hipLaunchKernelGGL(edge_param_grad_fast_kernel<16>, dim3(*n_used), dim3(256), 0, stream, g, feat, feat_rows, axes_ext,
                         rho, grad_t, partials, rows, shift);
    return check_launch();
  }
  hipLaunchKernelGGL(edge_param_grad_kernel, dim3(*n_used), dim3(256), 0, stream, g, feat, channels, axes_ext,
                     rho, grad_t, partials, rows);
  return check_launch();
}

}  // namespace se3
