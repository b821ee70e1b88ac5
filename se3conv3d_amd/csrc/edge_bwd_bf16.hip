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
// See edge_bwd_quad_bf16_kernel for the work split.
#include <cstdlib>

#include "common.h"
#include "edge_bf16_body.h"

namespace se3 {

namespace {

constexpr int kBwdMaxBlocks = 1024;

// ------------------------------------------------------------------------------------------------
// Role-specialised wavefronts (256 threads = the two frames of a centre point):
//   phi-waves (one per frame)  : geometry records -> descriptor -> kernel MLP -> GELU and GELU' -> hi/lo split;
//                                publish phi fragments, GELU' and the packed descriptors through LDS.  Pure VALU,
//                                they never touch grad_out.
//   acc-waves (one per frame)  : gather the grad_out rows (row layout for gphi, channel layout for U), gphi = G H on
//                                MFMA, gpre = gphi * GELU', d[A;beta]^T on MFMA (own frame); U for their half of the
//                                channels and both frames.  Memory + MFMA work, ~200 VALU per chunk.
// Per chunk two block barriers (A: last chunk's LDS data consumed, B: this chunk's data published) keep the
// exchange single-buffered (36 KB LDS per block, 164 VGPRs: 3 blocks = 12 wavefronts per CU).  The role of a
// wavefront flips with the block parity so that every SIMD sees both kinds.  VALU work per (item, chunk): ~1300
// instructions against ~1900 for edge_t_transposed + edge_param_grad.
//
// Measured on MI355X at the headline shape: 1.01 ms (+ 0.20 ms for H) against 0.36 + 0.50 ms (+ 0.20 ms for grad_T)
// for the two separate kernels -- correct, not faster, hence opt-in (SE3_BWD_MERGE=1).  Only the two phi-waves of
// a block issue VALU work in bulk, i.e. 1.5 VALU-heavy wavefronts per SIMD where edge_t_pair has 4; an earlier
// version with two symmetric wavefronts per item (every wavefront does phi, gphi and U for its frame) needed
// 222-248 VGPRs + 37 KB LDS (2 per SIMD) and ran at the same 1.0 ms.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 3) void edge_bwd_quad_bf16_kernel(EdgeGeom g, const uint32_t* __restrict__ gpk,
                                                                    int64_t g_rows, const float* __restrict__ axes_ext,
                                                                    const float* __restrict__ rho_p,
                                                                    const uint32_t* __restrict__ h_rows,
                                                                    uint32_t* __restrict__ u_out,
                                                                    float* __restrict__ partials, int64_t n_items,
                                                                    int fnb_shift) {
  constexpr int C = 64;
  __shared__ __attribute__((aligned(16))) uint32_t lds_phi[2][2][2][64][4];  // [frame][k-step][hi/lo][lane]
  __shared__ __attribute__((aligned(16))) float lds_dy[2][4][64][4];         // [frame][register quad][lane]
  __shared__ __attribute__((aligned(16))) uint32_t lds_desc[2][32][12];      // [frame][frame-edge][dim]
  __shared__ __attribute__((aligned(16))) uint32_t lds_h[2][4][2][64][4];    // [frame][k-step][hi/lo][lane]
  float(*lds_red)[kDescExt][kBasis] = reinterpret_cast<float(*)[kDescExt][kBasis]>(&lds_h[0][0][0][0][0]);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int role = wave ^ (((int)blockIdx.x & 1) << 1);
  const int v = role & 1;         // frame of this wavefront
  const bool phi_wave = role < 2;
  const int kcol = lane & 31, h = lane >> 5;
  const float rho = *rho_p;
  const int groups = g.f_ctr / 2;

  if (phi_wave) {
    // ------------------------------------------------------------------------------------------ phi-wave
    const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
    const __amdgpu_buffer_rsrc_t ctrg_rs = buffer_of(g.ctr_geom, g.n_ctr * g.f_ctr * 64);
    u32x4 wb_hi, wb_lo;  // [A; beta] as the MLP's B operand: half 0 holds descriptor dims 0..7, half 1 dims 8, 9
    {
      float wv8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        wv8[j] = k < kDescExt ? kGeluIn * axes_ext[k * kBasis + kcol] : 0.f;
      }
      frags_from_floats(wv8, wb_hi, wb_lo);
    }
    for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
      const int64_t ctr = item / groups;
      const int a0 = (int)(item - ctr * groups) * 2;
      const int start = ctr > 0 ? g.ends[ctr - 1] : 0;
      const int n_total = (g.ends[ctr] - start) * g.f_nb;
      if (n_total == 0) continue;
      float yc[3], rc[9];
      load_geom_record(ctrg_rs, (int)(ctr * g.f_ctr + a0 + v), yc, rc);
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
      float xn_nx[3], rn_nx[9];
      load_geom_record(nbg_rs, row_of(nb_a, 0), xn_nx, rn_nx);
      for (int c0 = 0; c0 < n_total; c0 += 32) {
        const int cnt = min(32, n_total - c0);
        float xn[3], rn[9], d[9];
#pragma unroll
        for (int i = 0; i < 3; ++i) xn[i] = xn_nx[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) rn[i] = rn_nx[i];
        const int q_b = row_of(nb_b, c0 + 32);
        nb_b = nbr_of(c0 + 64);
        load_geom_record(nbg_rs, q_b, xn_nx, rn_nx);
        edge_descriptor(yc, rc, xn, rn, rho, d);  // centre is the source side of the edge
        uint32_t pw[12];
#pragma unroll
        for (int i = 0; i < 8; i += 2) split_pack2(d[i], d[i + 1], pw[i], pw[i + 1]);
        split_pack2(d[8], 1.0f, pw[8], pw[9]);
        pw[10] = pw[11] = 0u;
        float mv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mv[j] = h ? (j == 0 ? d[8] : (j == 1 ? 1.0f : 0.f)) : d[j];
        u32x4 a_hi, a_lo;
        frags_from_floats(mv, a_hi, a_lo);
        const f32x16 pre = mfma_bf16x3(a_hi, a_lo, wb_hi, wb_lo, zero16());
        u32x4 b_hi[2], b_lo[2];
        float dyv[16];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (s * 16 < cnt) gelu_scaled_grad(pre[8 * s + j], pv[j], dyv[8 * s + j]);
            else pv[j] = 0.f, dyv[8 * s + j] = 0.f;
          }
          frags_from_floats(pv, b_hi[s], b_lo[s]);
        }
        __syncthreads();  // A: the acc-waves are done with the previous chunk's exchange data
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          *reinterpret_cast<u32x4*>(&lds_phi[v][s][0][lane][0]) = b_hi[s];
          *reinterpret_cast<u32x4*>(&lds_phi[v][s][1][lane][0]) = b_lo[s];
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
          *reinterpret_cast<f32x4*>(&lds_dy[v][qd][lane][0]) =
              f32x4{dyv[4 * qd], dyv[4 * qd + 1], dyv[4 * qd + 2], dyv[4 * qd + 3]};
        if (h == 0) {
          uint32_t* dst = &lds_desc[v][kcol][0];
          *reinterpret_cast<u32x4*>(dst) = u32x4{pw[0], pw[1], pw[2], pw[3]};
          *reinterpret_cast<u32x4*>(dst + 4) = u32x4{pw[4], pw[5], pw[6], pw[7]};
          *reinterpret_cast<u32x4*>(dst + 8) = u32x4{pw[8], pw[9], pw[10], pw[11]};
        }
        __syncthreads();  // B: published
      }
    }
    // the acc-waves end with two more barriers (reduction of d[A;beta]); take part in them
    __syncthreads();
    __syncthreads();
    return;
  }

  // -------------------------------------------------------------------------------------------- acc-wave
  const __amdgpu_buffer_rsrc_t g_rs = buffer_of(gpk, g_rows * C * 4);
  const int hb = 16 * h;
  const int cb4 = (32 * v + kcol) * 4;  // this wavefront aggregates channels 32*v .. 32*v+31 of U (both frames)
  const int jcol = min(kcol, 11);
  f32x16 dacc = zero16();  // lane (j = kcol, h), register r: d[A;beta][j][k = acc_row(r,h)] of frame v
  for (int64_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int64_t ctr = item / groups;
    const int a0 = (int)(item - ctr * groups) * 2;
    (void)a0;
    const int start = ctr > 0 ? g.ends[ctr - 1] : 0;
    const int n_total = (g.ends[ctr] - start) * g.f_nb;
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
    if (n_total > 0) {
      int nb_a = nbr_of(0);
      int nb_b = nbr_of(32);
      // H fragments (MFMA B operand of gphi) of row (item, v): lane (k = kcol, h) holds channels 16*st + 8h + j
      const uint32_t* hrow = h_rows + (item * 2 + v) * (int64_t)C * kBasis;
      uint32_t hw[4][8];
#pragma unroll
      for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int j = 0; j < 8; ++j) hw[st][j] = hrow[(16 * st + 8 * h + j) * kBasis + kcol];
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        u32x4 f_hi, f_lo;
        frags_from_words(hw[st], f_hi, f_lo);
        *reinterpret_cast<u32x4*>(&lds_h[v][st][0][lane][0]) = f_hi;
        *reinterpret_cast<u32x4*>(&lds_h[v][st][1][lane][0]) = f_lo;
      }
      for (int c0 = 0; c0 < n_total; c0 += 32) {
        const int cnt = min(32, n_total - c0);
        // rows past the end of the edge list read out of bounds (buffer loads return 0): no masks needed below
        const int qoff = c0 + kcol < n_total ? row_of(nb_a, c0) * (C * 4) : kOobOffset;
        nb_a = nb_b;
        nb_b = nbr_of(c0 + 64);
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
        // the same rows, channel layout (A operand of U): channels 32*v + kcol of rows acc_row(8s+j, h)
        uint32_t fw[2][8];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int src_off = __builtin_amdgcn_ds_bpermute(hb + 4 * acc_row(8 * s + j, 0), qoff);
            fw[s][j] = __builtin_amdgcn_raw_buffer_load_b32(g_rs, src_off + cb4, 0, 0);
          }
        __syncthreads();  // A: this wavefront is done with the previous chunk's exchange data
        __syncthreads();  // B: the phi-waves have published this chunk

        // gphi = G H on the gathered rows, gpre = gphi * GELU', d[A;beta]^T += gpre^T desc   (frame v)
        f32x16 gphi = zero16();
#pragma unroll
        for (int st = 0; st < 4; ++st) {
          u32x4 ra_hi, ra_lo;
          frags_from_words(rw[st], ra_hi, ra_lo);
          const u32x4 bh_hi = *reinterpret_cast<const u32x4*>(&lds_h[v][st][0][lane][0]);
          const u32x4 bh_lo = *reinterpret_cast<const u32x4*>(&lds_h[v][st][1][lane][0]);
          gphi = mfma_bf16x3(ra_hi, ra_lo, bh_hi, bh_lo, gphi);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s * 16 < cnt) {
            const f32x4 dy0 = *reinterpret_cast<const f32x4*>(&lds_dy[v][2 * s][lane][0]);
            const f32x4 dy1 = *reinterpret_cast<const f32x4*>(&lds_dy[v][2 * s + 1][lane][0]);
            float gp[8];
            uint32_t wd[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              gp[j] = gphi[8 * s + j] * (j < 4 ? dy0[j] : dy1[j - 4]);
              wd[j] = lds_desc[v][acc_row(8 * s + j, h)][jcol];
            }
            u32x4 ga_hi, ga_lo, db_hi, db_lo;
            frags_from_floats(gp, ga_hi, ga_lo);
            frags_from_words(wd, db_hi, db_lo);
            dacc = mfma_bf16x3(ga_hi, ga_lo, db_hi, db_lo, dacc);
          }
        }
        // U for this wavefront's channels, both frames
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s * 16 < cnt) {
            u32x4 fa_hi, fa_lo;
            frags_from_words(fw[s], fa_hi, fa_lo);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              const u32x4 b_hi = *reinterpret_cast<const u32x4*>(&lds_phi[a][s][0][lane][0]);
              const u32x4 b_lo = *reinterpret_cast<const u32x4*>(&lds_phi[a][s][1][lane][0]);
              acc[a] = mfma_bf16x3(fa_hi, fa_lo, b_hi, b_lo, acc[a]);
            }
          }
        }
      }
    }
    // acc[a] register r, lane (kcol, h) = U[row 2*item + a][32*v + acc_row(r,h)][kcol]
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      uint32_t* u_row = u_out + ((item * 2 + a) * (int64_t)C + 32 * v) * kBasis;
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
  __syncthreads();  // all H images are free (the phi-waves wait here too)
  if (kcol < kDescExt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lds_red[v][kcol][acc_row(r, h)] = dacc[r];
  }
  __syncthreads();
  if (v == 0)
    for (int i = lane; i < kDescExt * kBasis; i += 64) {
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
  hipLaunchKernelGGL(edge_bwd_quad_bf16_kernel, dim3(blocks), dim3(256), 0, stream, gt, gpk, g_rows, axes_ext, rho,
                     h_rows, u_out, partials, items, shift);
  return check_launch();
}

}  // namespace se3
