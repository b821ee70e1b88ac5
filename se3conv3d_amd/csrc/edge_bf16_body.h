// Per-item body of the single-wavefront split-bf16 edge kernel (edge_bf16.hip, rows narrower than 64 channels).
// See edge_bf16.hip for the scheme.
#pragma once

#include "common.h"

namespace se3 {

// MLP weights [A; beta] as MFMA B fragments in LDS: arrangement a has descriptor dims 0..7 in the lane
// half that builds row a's descriptors.  Called by the first wavefront of a block.
template <int FC>
__device__ __forceinline__ void mlp_weights_to_lds(uint32_t (*lds_w)[2][64][4], const float* __restrict__ axes_ext,
                                                   int lane) {
  const int kcol = lane & 31, h = lane >> 5;
  float v07[8], v89[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v07[j] = kGeluIn * axes_ext[j * kBasis + kcol];  // the MLP delivers kGeluIn * pre (gelu_scaled)
    v89[j] = j < 2 ? kGeluIn * axes_ext[(8 + j) * kBasis + kcol] : 0.f;
  }
#pragma unroll
  for (int a = 0; a < FC; ++a) {
    const bool dims07 = FC == 1 ? h == 0 : h == a;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = dims07 ? v07[j] : v89[j];
    u32x4 w_hi, w_lo;
    frags_from_floats(v, w_hi, w_lo);
    *reinterpret_cast<u32x4*>(&lds_w[a][0][lane][0]) = w_hi;
    *reinterpret_cast<u32x4*>(&lds_w[a][1][lane][0]) = w_lo;
  }
}

// One item = FC frames of one centre point.  sink(a, ch0, ch1, x0, x1, ok0, ok1) receives the values of row
// (ctr*f_ctr + a0 + a), channels ch0 / ch1 (ch1 = ch0 + VW), basis function k = lane & 31, and packs / stores them.
template <int VW, int FC, bool FULL, class Sink>
__device__ __forceinline__ void edge_item_bf16(const EdgeGeom& g, const __amdgpu_buffer_rsrc_t feat_rs, int channels,
                                               const uint32_t (*lds_w)[2][64][4], float rho, int64_t item,
                                               int fnb_shift, Sink&& sink) {
  const int lane = threadIdx.x & 63;
  const int kcol = lane & 31, h = lane >> 5;
  const int groups = g.f_ctr / FC;
  // rows < 2^31 (checked on the host), so 32-bit unsigned division is exact -- the 64-bit one is ~150 scalar instructions
  const int64_t ctr = (uint32_t)item / (uint32_t)groups;
  const int a0 = (int)((uint32_t)item - (uint32_t)ctr * (uint32_t)groups) * FC;

  const int start = ctr > 0 ? g.ends[ctr - 1] : 0;
  const int n_total = (g.ends[ctr] - start) * g.f_nb;
  const __amdgpu_buffer_rsrc_t nbg_rs = buffer_of(g.nb_geom, g.n_nb * g.f_nb * 64);
  float yc[3], rc[9];
  // the centre frame this lane builds descriptors for
  load_geom_record(buffer_of(g.ctr_geom, g.n_ctr * g.f_ctr * 64), (int)(ctr * g.f_ctr + a0 + (FC == 2 ? h : 0)), yc, rc);
  const int row_bytes = channels * 4;
  const int hb = 16 * h;  // ds_bpermute byte address of lane 4h

  for (int cbase = 0; cbase < channels; cbase += 32 * VW) {
    const int cb = cbase + VW * kcol;
    const bool ch_ok = FULL || cb < channels;
    const int cb4 = (ch_ok ? cb : 0) * 4;
    f32x16 acc[FC][VW];
#pragma unroll
    for (int a = 0; a < FC; ++a)
#pragma unroll
      for (int t = 0; t < VW; ++t) acc[a][t] = zero16();

    // frame-edge -> neighbour id / source row: lane n of both halves handles frame-edge c0 + n (indices past the end
    // clamp to the last one)
    auto nbr_of = [&](int c0) {
      const int fe = min(c0 + kcol, n_total - 1);
      const int e = start + (fnb_shift >= 0 ? fe >> fnb_shift : fe / g.f_nb);
      return g.nbr[(int64_t)e * g.nbr_stride + g.nbr_offset];
    };
    auto row_of = [&](int nb, int c0) {
      const int fe = min(c0 + kcol, n_total - 1);
      return nb * g.f_nb + (fnb_shift >= 0 ? fe & ((1 << fnb_shift) - 1) : fe % g.f_nb);
    };
    auto geom_of = [&](int nb, int q, float xn[3], float rn[9]) {
      load_geom_record(nbg_rs, q, xn, rn);
    };
    // Software pipeline (as in the wave-pair kernel): neighbour ids are fetched two chunks ahead and geometry records
    // one chunk ahead, and the chunk's own feature words go out at its top and are only turned into MFMA fragments
    // where the first product needs them -- no load result is needed by the instructions right behind the load (the
    // first version converted the words where they were loaded: one full memory latency per k-step with nothing of
    // this wavefront to overlap it).
    int nb_b = 0, q_a = 0;
    float xn_nx[3], rn_nx[9];
    if (n_total > 0) {
      const int nb_a = nbr_of(0);
      nb_b = nbr_of(32);
      q_a = row_of(nb_a, 0);
      geom_of(nb_a, q_a, xn_nx, rn_nx);
    }
    for (int c0 = 0; c0 < n_total; c0 += 32) {
      const int cnt = min(32, n_total - c0);
      // rows past the end of the edge list are read out of bounds (raw buffer loads return 0), so phi needs no mask
      const int qoff = c0 + kcol < n_total ? q_a * row_bytes : kOobOffset;
      float xn[3], rn[9], d[9];
#pragma unroll
      for (int i = 0; i < 3; ++i) xn[i] = xn_nx[i];
#pragma unroll
      for (int i = 0; i < 9; ++i) rn[i] = rn_nx[i];
      const int q_b = row_of(nb_b, c0 + 32);
      const int nb_q = nb_b;
      nb_b = nbr_of(c0 + 64);

      // gathered feature words of the chunk's two k-steps (shared by the FC rows)
      uint32_t fw[2][VW][8];
      // (no branch around the second k-step's loads: a conditional load makes every counted s_waitcnt behind it assume
      // the loads were not issued, i.e. wait for all of them; past the end of the list they read out of bounds)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          // byte offset of the source row of frame-edge acc_row(8s+j, h), fetched from the lane that owns it
          const int src_off = __builtin_amdgcn_ds_bpermute(hb + 4 * acc_row(8 * s + j, 0), qoff);
          const int voff = ch_ok ? src_off + cb4 : kOobOffset;
          if constexpr (VW == 4) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(feat_rs, voff, 0, 0);
            fw[s][0][j] = v[0], fw[s][1][j] = v[1], fw[s][2][j] = v[2], fw[s][3][j] = v[3];
          } else if constexpr (VW == 2) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(feat_rs, voff, 0, 0);
            fw[s][0][j] = v[0], fw[s][1][j] = v[1];
          } else {
            fw[s][0][j] = __builtin_amdgcn_raw_buffer_load_b32(feat_rs, voff, 0, 0);
          }
        }
      }
      geom_of(nb_q, q_b, xn_nx, rn_nx);
      q_a = q_b;

      if (!g.transposed)
        edge_descriptor(xn, rn, yc, rc, rho, d);
      else
        edge_descriptor(yc, rc, xn, rn, rho, d);

      // MLP A operand pieces of this lane: its own dims 0..7, and {dim 8 of the row it serves as "other" half, 1}
      u32x4 own_hi, own_lo, oth_hi, oth_lo;
      frags_from_floats(d, own_hi, own_lo);
      {
        float d8 = d[8];
        if constexpr (FC == 2) {
          // lanes of half h hold the descriptor against frame a0+h; row a's dims 8,9 live in half 1-a
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
        f32x16 phi = mfma_bf16x3(a_hi, a_lo, wb_hi, wb_lo, zero16());
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (s * 16 < cnt) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              pv[j] = gelu_scaled(phi[8 * s + j]);
            }
            u32x4 b_hi, b_lo;
            frags_from_floats(pv, b_hi, b_lo);
#pragma unroll
            for (int t = 0; t < VW; ++t) {
              if (a == 0) frags_from_words(fw[s][t], fa_hi[s][t], fa_lo[s][t]);  // first use: the words have had a chunk's work to arrive
              acc[a][t] = mfma_bf16x3(fa_hi[s][t], fa_lo[s][t], b_hi, b_lo, acc[a][t]);
            }
          }
        }
      }
    }
    // acc[a][t] register r, lane (kcol, h) = T[row a][cbase + VW*acc_row(r,h) + t][kcol].  The values go to the sink in
    // pairs of ADJACENT channels where the layout has them in one lane (what the 3-byte row format stores together):
    // registers r, r + 1 of one tile at one channel per lane (VW = 1), tiles 0 and 1 of one register at two (VW = 2).
#pragma unroll
    for (int a = 0; a < FC; ++a) {
      if constexpr (VW == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ch0 = cbase + 2 * acc_row(r, h);
          sink(a, ch0, ch0 + 1, acc[a][0][r], acc[a][1][r], FULL || ch0 < channels, FULL || ch0 + 1 < channels);
        }
      } else {
#pragma unroll
        for (int t = 0; t < VW; ++t)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int ch0 = cbase + VW * acc_row(r, h) + t, ch1 = cbase + VW * acc_row(r + 1, h) + t;
            sink(a, ch0, ch1, acc[a][t][r], acc[a][t][r + 1], FULL || ch0 < channels, FULL || ch1 < channels);
          }
      }
    }
  }
}


}  // namespace se3
