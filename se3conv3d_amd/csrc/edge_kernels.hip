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
// ------------------------------------------------------------------------------------------------
template <int VW>
__global__ __launch_bounds__(256) void edge_t_kernel(EdgeGeom g, const float* __restrict__ feat, int channels,
                                                     const float* __restrict__ axes_ext,
                                                     const float* __restrict__ rho_p, float* __restrict__ t_out,
                                                     int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= rows) return;
  const int kcol = lane & 31, h = lane >> 5;
  const float rho = *rho_p;
  float bmlp[5];
#pragma unroll
  for (int t = 0; t < 5; ++t) bmlp[t] = axes_ext[(2 * t + h) * kBasis + kcol];

  const RowInfo ri = row_info(g, m);
  float yc[3], rc[9];
#pragma unroll
  for (int i = 0; i < 3; ++i) yc[i] = g.ctr_pts[ri.ctr * 3 + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) rc[i] = g.ctr_frames[m * 9 + i];

  float* t_row = t_out + m * (int64_t)channels * kBasis;
  for (int cbase = 0; cbase < channels; cbase += 32 * VW) {
    const int cb = cbase + VW * kcol;     // first channel this lane feeds as MFMA row `kcol`
    const bool ch_ok = cb < channels;     // channels % VW == 0 (host guarantees) => whole vector valid
    const int cb_ld = ch_ok ? cb : 0;
    f32x16 acc[VW];
#pragma unroll
    for (int t = 0; t < VW; ++t) acc[t] = zero16();

    for (int c0 = 0; c0 < ri.n_total; c0 += 32) {
      const int cnt = min(32, ri.n_total - c0);
      const int fe = c0 + min(kcol, cnt - 1);
      float d[kDescExt];
      int q;
      lane_descriptor(g, ri, fe, yc, rc, rho, d, q);
      f32x16 phi = mlp_preactivation(d, bmlp, h);
#pragma unroll
      for (int r = 0; r < 16; ++r) phi[r] = acc_row(r, h) < cnt ? gelu_erf(phi[r]) : 0.f;

#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        if (g4 * 8 < cnt) {  // wave-uniform: frame-edges 8*g4 .. 8*g4+7 of the chunk
          float a[4][VW];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int r = g4 * 4 + rr;
            const int q_lo = __builtin_amdgcn_readlane(q, acc_row(r, 0));
            const int q_hi = __builtin_amdgcn_readlane(q, acc_row(r, 1));
            const float* src = feat + (int64_t)(h ? q_hi : q_lo) * channels + cb_ld;
            if constexpr (VW == 4) {
              const float4 v = *reinterpret_cast<const float4*>(src);
              a[rr][0] = v.x, a[rr][1] = v.y, a[rr][2] = v.z, a[rr][3] = v.w;
            } else if constexpr (VW == 2) {
              const float2 v = *reinterpret_cast<const float2*>(src);
              a[rr][0] = v.x, a[rr][1] = v.y;
            } else {
              a[rr][0] = *src;
            }
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int t = 0; t < VW; ++t)
              acc[t] = mfma32(ch_ok ? a[rr][t] : 0.f, phi[g4 * 4 + rr], acc[t]);
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
#pragma unroll 4
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

}  // namespace

int launch_edge_t(const char* tag, const EdgeGeom& g, const float* feat, int channels, const float* axes_ext,
                  const float* rho, float* t_out, hipStream_t stream) {
  const int64_t rows = g.n_ctr * g.f_ctr;
  if (rows == 0) return SE3_OK;
  ProfScope prof(tag, stream);
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (channels % 128 == 0)
    hipLaunchKernelGGL(edge_t_kernel<4>, grid, block, 0, stream, g, feat, channels, axes_ext, rho, t_out, rows);
  else if (channels % 64 == 0)
    hipLaunchKernelGGL(edge_t_kernel<2>, grid, block, 0, stream, g, feat, channels, axes_ext, rho, t_out, rows);
  else
    hipLaunchKernelGGL(edge_t_kernel<1>, grid, block, 0, stream, g, feat, channels, axes_ext, rho, t_out, rows);
  return check_launch();
}

// slots of partial sums the parameter-gradient kernels may use: up to 2048 workgroups (the pair form of the split-bf16
// kernel keeps 6 x 256 resident; the other forms use at most 512 of them)
int edge_param_grad_blocks(int64_t rows) {
  const int64_t want = (rows + 3) / 4;
  return (int)(want < 2048 ? (want > 0 ? want : 1) : 2048);
}

int launch_edge_param_grad(const char* tag, const EdgeGeom& g, const float* feat, int channels,
                           const float* axes_ext, const float* rho, const float* grad_t, float* partials,
                           int n_partials, int* n_used, hipStream_t stream) {
  const int64_t rows = g.n_ctr * g.f_ctr;
  ProfScope prof(tag, stream);
  // two workgroups per CU (2048 measured 3 % slower, 4096 11 %): the slot capacity is sized for the pair form of the
  // split-bf16 kernel, this kernel uses at most 512 of the slots
  *n_used = n_partials < 512 ? n_partials : 512;
  hipLaunchKernelGGL(edge_param_grad_kernel, dim3(*n_used), dim3(256), 0, stream, g, feat, channels, axes_ext,
                     rho, grad_t, partials, rows);
  return check_launch();
}

}  // namespace se3
