// Backward edge pass of the split-bf16 path: feature gradient AND kernel-MLP parameter gradients from one
// walk over the source-major (transposed) graph.
//
// Centre = input row (p,b); its edges lead to the output rows (s,a) whose grad_out rows g[(s,a), :] are
// gathered.  With H[(p,b)][o,k] = alpha * sum_i f[(p,b),i] W[i,k,o] (one dense GEMM in front of this kernel)
//
//   U[(p,b)][o,k]   = sum_n g[q(n), o] * phi(n)[k]                  transposed convolution, then dX = U W'
//   gphi[n,k]       = sum_o g[q(n), o] * H[(p,b)][o,k]              == sum_i f[(p,b),i] * grad_T[q(n)][i,k]
//   d[A;beta][j,k] += desc(n)[j] * gphi[n,k] * GELU'(pre(n)[k])
//
// i.e. the parameter gradient needs exactly the operands the transposed convolution already has in flight
// (the gathered g rows, the descriptors, the pre-activations); a separate output-major pass
// (edge_param_grad*: its own gather of the feature rows, descriptors, MLP, GELU') and the grad_T tensor
// disappear.  The reference reaches the same numbers through autograd over E'-sized tensors
// (PNEConvLayerRotEquiv.py:199-216 backward, feat_basis_proj_grads.cu:100-141).
//
// Workgroup = 2 wavefronts = the two frames of one centre point (persistent over items).  Wavefront v
//   * builds descriptors / MLP / GELU / GELU' for centre frame a0+v, publishes phi through LDS (as in
//     edge_t_pair_bf16_kernel) and aggregates channels 32v..32v+31 of U for both frames;
//   * computes gphi for its own frame: A = the gathered g rows in row layout (lane n reads 8 consecutive
//     channels of its own row per k-step), B = H fragments of row (item, v) parked in LDS once per item;
//   * accumulates d[A;beta]^T on MFMA: A = gpre (accumulator registers split in place), B = descriptor
//     columns from a wave-private LDS image.
#include <cstdlib>

#include "common.h"
#include "edge_bf16_body.h"

namespace se3 {

namespace {

constexpr int kBwdMaxBlocks = 1024;

__global__ __launch_bounds__(128, 2) void edge_bwd_pair_bf16_kernel(EdgeGeom g, const uint32_t* __restrict__ gpk,
                                                                    int64_t g_rows, const float* __restrict__ axes_ext,
                                                                    const float* __restrict__ rho_p,
                                                                    const uint32_t* __restrict__ h_rows,
                                                                    uint32_t* __restrict__ u_out,
                                                                    float* __restrict__ partials, int64_t n_items,
                                                                    int fnb_shift) {
  constexpr int C = 64;
  __shared__ __attribute__((aligned(16))) uint32_t lds_w[1][2][64][4];
  __shared__ __attribute__((aligned(16))) uint32_t lds_phi[2][2][2][2][64][4];  // [buffer][frame][k-step][hi/lo][lane]
  __shared__ __attribute__((aligned(16))) uint32_t lds_h[2][4][2][64][4];       // [wave][k-step][hi/lo][lane]
  __shared__ __attribute__((aligned(16))) uint32_t lds_desc[2][32][12];         // [wave][frame-edge][dim]
  float(*lds_red)[kDescExt][kBasis] = reinterpret_cast<float(*)[kDescExt][kBasis]>(&lds_h[0][0][0][0][0]);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int kcol = lane & 31, h = lane >> 5;
  if (threadIdx.x < 64) mlp_weights_to_lds<1>(lds_w, axes_ext, threadIdx.x);
  __syncthreads();
  const float rho = *rho_p;
  const __amdgpu_buffer_rsrc_t g_rs = buffer_of(gpk, g_rows * C * 4);
  const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
  const __amdgpu_buffer_rsrc_t ctrg_rs = buffer_of(g.ctr_geom, g.n_ctr * g.f_ctr * 64);
  const int groups = g.f_ctr / 2;
  const int hb = 16 * h;
  const int cb4 = (32 * wv + kcol) * 4;  // this wavefront aggregates channels 32*wv .. 32*wv+31 of U
  const int jcol = min(kcol, 11);
  f32x16 dacc = zero16();  // lane (j = kcol, h), register r: d[A;beta][j][k = acc_row(r,h)]
  int buf = 0;

  for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t ctr = item / groups;
    const int a0 = (int)(item - ctr * groups) * 2;
    const int start = ctr > 0 ? g.ends[ctr - 1] : 0;
    const int n_total = (g.ends[ctr] - start) * g.f_nb;
    float yc[3], rc[9];
    load_geom_record(ctrg_rs, (int)(ctr * g.f_ctr + a0 + wv), yc, rc);

    // ids two chunks ahead, geometry one chunk ahead (see edge_t_pair_bf16_kernel); indices past the end clamp
    auto nbr_of = [&](int c0) {
      const int fe = min(c0 + kcol, n_total - 1);
      const int e = start + (fnb_shift >= 0 ? fe >> fnb_shift : fe / g.f_nb);
      return g.nbr[(int64_t)e * g.nbr_stride + g.nbr_offset];
    };
    auto row_of = [&](int nb, int c0) {
      const int fe = min(c0 + kcol, n_total - 1);
      return nb * g.f_nb + (fnb_shift >= 0 ? fe & ((1 << fnb_shift) - 1) : fe % g.f_nb);
    };

    f32x16 acc[2] = {zero16(), zero16()};  // [frame] of U
    int nb_b = 0, q_a = 0;
    float xn_nx[3], rn_nx[9];
    if (n_total > 0) {
      const int nb_a = nbr_of(0);
      nb_b = nbr_of(32);
      // H fragments (MFMA B operand of gphi) of row (item, wv): lane (k = kcol, h) holds channels 16*st + 8h + j
      const uint32_t* hrow = h_rows + (item * 2 + wv) * (int64_t)C * kBasis;
      uint32_t hw[4][8];
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) hw[st][j] = hrow[(16 * st + 8 * h + j) * kBasis + kcol];
      q_a = row_of(nb_a, 0);
      load_geom_record(nbg_rs, q_a, xn_nx, rn_nx);
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        u32x4 f_hi, f_lo;
        frags_from_words(hw[st], f_hi, f_lo);
        *reinterpret_cast<u32x4*>(&lds_h[wv][st][0][lane][0]) = f_hi;
        *reinterpret_cast<u32x4*>(&lds_h[wv][st][1][lane][0]) = f_lo;
      }
    }

    for (int c0 = 0; c0 < n_total; c0 += 32, buf ^= 1) {
      const int cnt = min(32, n_total - c0);
      // rows past the end of the edge list read out of bounds (buffer loads return 0): no masks needed below
      const int qoff = c0 + kcol < n_total ? q_a * (C * 4) : kOobOffset;
      float xn[3], rn[9], d[9];
#pragma unroll
      for (int i = 0; i < 3; ++i) xn[i] = xn_nx[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) rn[i] = rn_nx[i];
      const int q_b = row_of(nb_b, c0 + 32);
      nb_b = nbr_of(c0 + 64);

      // all gathers of the chunk go out now; fragments are built where they are consumed
      // grad_out rows, row layout (A operand of gphi): lane (n = kcol, h), 8 channels per k-step
      uint32_t rw[4][8];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int voff = qoff + (16 * st + 8 * h) * 4;
        const auto v0 = __builtin_amdgcn_raw_buffer_load_b128(g_rs, voff, 0, 0);
        const auto v1 = __builtin_amdgcn_raw_buffer_load_b128(g_rs, voff + 16, 0, 0);
        rw[st][0] = v0[0], rw[st][1] = v0[1], rw[st][2] = v0[2], rw[st][3] = v0[3];
        rw[st][4] = v1[0], rw[st][5] = v1[1], rw[st][6] = v1[2], rw[st][7] = v1[3];
      }
      // the same rows, channel layout (A operand of U): channels 32*wv + kcol of rows acc_row(8s+j, h)
      uint32_t fw[2][8];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int src_off = __builtin_amdgcn_ds_bpermute(hb + 4 * acc_row(8 * s + j, 0), qoff);
          fw[s][j] = __builtin_amdgcn_raw_buffer_load_b32(g_rs, src_off + cb4, 0, 0);
        }
      load_geom_record(nbg_rs, q_b, xn_nx, rn_nx);
      q_a = q_b;

      edge_descriptor(yc, rc, xn, rn, rho, d);  // centre is the source side of the edge

      // descriptor image for the d[A;beta] product (both lane halves hold the same descriptor)
      if (h == 0) {
        uint32_t* dst = &lds_desc[wv][kcol][0];
        uint32_t pw[12];
#pragma unroll
        for (int i = 0; i < 8; i += 2) split_pack2(d[i], d[i + 1], pw[i], pw[i + 1]);
        split_pack2(d[8], 1.0f, pw[8], pw[9]);
        pw[10] = pw[11] = 0u;
        *reinterpret_cast<u32x4*>(dst) = u32x4{pw[0], pw[1], pw[2], pw[3]};
        *reinterpret_cast<u32x4*>(dst + 4) = u32x4{pw[4], pw[5], pw[6], pw[7]};
        *reinterpret_cast<u32x4*>(dst + 8) = u32x4{pw[8], pw[9], pw[10], pw[11]};
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

      // kernel MLP, GELU and GELU' for this wavefront's frame (half 0 feeds descriptor dims 0..7, half 1 dims 8, 9);
      // phi is published for the partner wavefront, GELU' stays in registers
      float dyv[16];
      {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = h ? (j == 0 ? d[8] : (j == 1 ? 1.0f : 0.f)) : d[j];
        u32x4 a_hi, a_lo;
        frags_from_floats(v, a_hi, a_lo);
        const u32x4 wb_hi = *reinterpret_cast<const u32x4*>(&lds_w[0][0][lane][0]);
        const u32x4 wb_lo = *reinterpret_cast<const u32x4*>(&lds_w[0][1][lane][0]);
        const f32x16 pre = mfma_bf16x3(a_hi, a_lo, wb_hi, wb_lo, zero16());
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s * 16 < cnt) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) gelu_scaled_grad(pre[8 * s + j], pv[j], dyv[8 * s + j]);
            u32x4 b_hi, b_lo;
            frags_from_floats(pv, b_hi, b_lo);
            *reinterpret_cast<u32x4*>(&lds_phi[buf][wv][s][0][lane][0]) = b_hi;
            *reinterpret_cast<u32x4*>(&lds_phi[buf][wv][s][1][lane][0]) = b_lo;
          }
        }
      }

      // gphi = G H on the gathered rows, gpre = gphi * GELU', d[A;beta]^T += gpre^T desc
      {
        f32x16 gphi = zero16();
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          u32x4 ra_hi, ra_lo;
          frags_from_words(rw[st], ra_hi, ra_lo);
          const u32x4 bh_hi = *reinterpret_cast<const u32x4*>(&lds_h[wv][st][0][lane][0]);
          const u32x4 bh_lo = *reinterpret_cast<const u32x4*>(&lds_h[wv][st][1][lane][0]);
          gphi = mfma_bf16x3(ra_hi, ra_lo, bh_hi, bh_lo, gphi);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s * 16 < cnt) {
            float gp[8];
            uint32_t wd[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              gp[j] = gphi[8 * s + j] * dyv[8 * s + j];
              wd[j] = lds_desc[wv][acc_row(8 * s + j, h)][jcol];
            }
            u32x4 ga_hi, ga_lo, db_hi, db_lo;
            frags_from_floats(gp, ga_hi, ga_lo);
            frags_from_words(wd, db_hi, db_lo);
            dacc = mfma_bf16x3(ga_hi, ga_lo, db_hi, db_lo, dacc);
          }
        }
      }
      __syncthreads();  // both frames' phi fragments of this chunk are published (other buffer is used next chunk)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s * 16 < cnt) {
          u32x4 fa_hi, fa_lo;
          frags_from_words(fw[s], fa_hi, fa_lo);
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const u32x4 b_hi = *reinterpret_cast<const u32x4*>(&lds_phi[buf][a][s][0][lane][0]);
            const u32x4 b_lo = *reinterpret_cast<const u32x4*>(&lds_phi[buf][a][s][1][lane][0]);
            acc[a] = mfma_bf16x3(fa_hi, fa_lo, b_hi, b_lo, acc[a]);
          }
        }
      }
    }
    // acc[a] register r, lane (kcol, h) = U[row 2*item + a][32*wv + acc_row(r,h)][kcol]
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      uint32_t* u_row = u_out + ((item * 2 + a) * (int64_t)C + 32 * wv) * kBasis;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        uint32_t w0, w1;
        split_pack2(acc[a][r], acc[a][r + 1], w0, w1);
        u_row[acc_row(r, h) * kBasis + kcol] = w0;
        u_row[acc_row(r + 1, h) * kBasis + kcol] = w1;
      }
    }
  }

  // dacc: rows = k (acc_row(r,h)), columns = descriptor dim j = kcol (only j < 10 are meaningful)
  __syncthreads();  // both wavefronts are done with their H images
  if (kcol < kDescExt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_red[wv][kcol][acc_row(r, h)] = dacc[r];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kDescExt * kBasis; i += blockDim.x) {
    const int j = i / kBasis, k = i % kBasis;
    partials[(int64_t)blockIdx.x * kDescExt * kBasis + i] = lds_red[0][j][k] + lds_red[1][j][k];
  }
}

}  // namespace

bool edge_bwd_pair_bf16_supported(int f_ctr, int gathered_channels) {
  static const bool enabled = getenv("SE3_BWD_MERGE") != nullptr;
  return enabled && gathered_channels == 64 && f_ctr > 0 && f_ctr % 2 == 0;
}

int edge_bwd_pair_bf16_blocks(int64_t items) {
  return (int)(items < kBwdMaxBlocks ? (items > 0 ? items : 1) : kBwdMaxBlocks);
}

// gt: transposed graph (centres = input points).  h_rows: [n_ctr*f_ctr, 64, 32] packed words (H above),
// u_out: same shape.  partials: edge_bwd_pair_bf16_blocks(items) x 320 floats.
int launch_edge_bwd_pair_bf16(const char* tag, const EdgeGeom& gt, const uint32_t* gpk, int64_t g_rows,
                              const float* axes_ext, const float* rho, const uint32_t* h_rows, uint32_t* u_out,
                              float* partials, int* n_partials, hipStream_t stream) {
  const int64_t items = gt.n_ctr * gt.f_ctr / 2;
  if (g_rows * 64 * 4 >= (int64_t)kOobOffset) return SE3_ERR_UNSUPPORTED;
  const int blocks = edge_bwd_pair_bf16_blocks(items);
  *n_partials = blocks;
  ProfScope prof(tag, stream);
  int shift = -1;
  for (int sft = 0; sft < 8; ++sft)
    if ((1 << sft) == gt.f_nb) shift = sft;
  hipLaunchKernelGGL(edge_bwd_pair_bf16_kernel, dim3(blocks), dim3(128), 0, stream, gt, gpk, g_rows, axes_ext, rho,
                     h_rows, u_out, partials, items, shift);
  return check_launch();
}

}  // namespace se3
