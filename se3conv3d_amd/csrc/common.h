// Shared device/host helpers for libse3conv_hip (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/se3conv.h"

namespace se3 {

using f32x16 = __attribute__((__vector_size__(16 * sizeof(float)))) float;
using f32x4 = __attribute__((__vector_size__(4 * sizeof(float)))) float;

constexpr int kWave = 64;         // CDNA wavefront
constexpr int kBasis = 32;        // K of the MFMA kernels (every shipped config)
constexpr int kDescExt = 10;      // 9 descriptor dims + 1 constant (carries the bias)

// Row of a 32x32 MFMA accumulator held in register `reg` of a lane in half `h` (= lane >> 5):
// cdna_hip_programming.md section 3, "row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)".
__host__ __device__ constexpr int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// ---- split-bf16 ("bf16x3") arithmetic ------------------------------------------------------------
// An fp32 value x is carried as hi = bf16(x), lo = bf16(x - hi) (16 significant bits together); a
// product a*b is evaluated as a_hi*b_hi + a_lo*b_hi + a_hi*b_lo on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation (relative error ~2^-17 per product, vs 2^-24 for the fp32 MFMA at 1/16 the rate).
// In memory such operands are "packed words": (hi << 16) | lo, the same 4 bytes as the fp32 value.
using bf16x8 = __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16;
using bf16x2 = __attribute__((__vector_size__(2 * sizeof(__bf16)))) __bf16;
using f32x2 = __attribute__((__vector_size__(2 * sizeof(float)))) float;
using u32x4 = __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t;

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// c += a*b with both operands split
__device__ __forceinline__ f32x16 mfma_bf16x3(u32x4 a_hi, u32x4 a_lo, u32x4 b_hi, u32x4 b_lo, f32x16 c) {
  c = mfma_bf16(a_lo, b_hi, c);
  c = mfma_bf16(a_hi, b_lo, c);
  return mfma_bf16(a_hi, b_hi, c);
}
__device__ __forceinline__ uint32_t cvt_pk_bf16(float x0, float x1) {  // low half = bf16(x0), high half = bf16(x1)
  f32x2 v = {x0, x1};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
// two values -> packed pair of their hi parts and packed pair of their lo parts
__device__ __forceinline__ void split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  hi = cvt_pk_bf16(x0, x1);
  lo = cvt_pk_bf16(x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xffff0000u));
}
// one value -> packed word (hi << 16) | lo
__device__ __forceinline__ uint32_t split_pack(float x) {
  const uint32_t h = cvt_pk_bf16(x, 0.f) << 16;
  return h | (cvt_pk_bf16(x - __uint_as_float(h), 0.f) & 0xffffu);
}
// packed words w0, w1 -> packed pair of hi parts / of lo parts (element 0 = w0)
__device__ __forceinline__ uint32_t pair_hi(uint32_t w0, uint32_t w1) { return __builtin_amdgcn_perm(w1, w0, 0x07060302u); }
__device__ __forceinline__ uint32_t pair_lo(uint32_t w0, uint32_t w1) { return __builtin_amdgcn_perm(w1, w0, 0x05040100u); }
// two values -> their two packed words, sharing the conversions (4 VALU per value instead of 7)
__device__ __forceinline__ void split_pack2(float x0, float x1, uint32_t& w0, uint32_t& w1) {
  uint32_t hi, lo;
  split2(x0, x1, hi, lo);
  w0 = pair_lo(lo, hi);
  w1 = pair_hi(lo, hi);
}
// ---- 3-byte row format ("T24", opt-in) of the row-sized intermediates T and U ---------------------------------
// An element keeps the upper 24 bits of its fp32 value (rounded): hi = bits 31..16 (a truncated bf16), lo = bits
// 15..8.  hi + lo reproduce 16 significant bits; the MFMA operands are a_hi = hi (as bf16) and
// a_lo = bf16(value24 - hi), exact because that difference has at most 8 significant bits.
// Row layout for C channels x 32 basis functions (k' order = channel pair, basis, channel parity, so that the two
// values a lane of the edge kernels owns -- channels c and c+1 of one basis function -- are one word + one half-word):
//   bytes [0, 2*C*32)        uint16 hi[(c/2)*64 + k*2 + (c&1)]
//   bytes [2*C*32, 3*C*32)   uint8  lo[same index]
__host__ __device__ inline int64_t t24_row_bytes(int channels) { return (int64_t)channels * kBasis * 3; }
// position kq in a T24 row -> the index c*32 + k the fp32 / packed-word rows use
__host__ __device__ inline int t24_k_of(int kq) { return ((((kq >> 6) << 1) | (kq & 1)) << 5) + ((kq >> 1) & 31); }
__device__ __forceinline__ void t24_pack2(float x0, float x1, uint32_t& hi_pair, uint32_t& lo_pair) {
  const uint32_t u0 = __float_as_uint(x0) + 0x80u, u1 = __float_as_uint(x1) + 0x80u;  // round to 24 bits
  hi_pair = __builtin_amdgcn_perm(u1, u0, 0x07060302u);  // u0[31:16] | u1[31:16] << 16
  lo_pair = __builtin_amdgcn_perm(u1, u0, 0x0c0c0501u);  // u0[15:8] | u1[15:8] << 8
}
// hi pair word H (two bf16) + the word L holding their lo bytes at byte positions B0, B0+1 -> the lo fragment word
template <int B0, int B1 = B0 + 1>
__device__ __forceinline__ uint32_t t24_lo_word(uint32_t H, uint32_t L) {
  const uint32_t w0 = __builtin_amdgcn_perm(H, L, 0x0504000cu | ((uint32_t)B0 << 8));
  const uint32_t w1 = __builtin_amdgcn_perm(H, L, 0x0706000cu | ((uint32_t)B1 << 8));
  const float l0 = __uint_as_float(w0) - __uint_as_float(H << 16);
  const float l1 = __uint_as_float(w1) - __uint_as_float(H & 0xffff0000u);
  return cvt_pk_bf16(l0, l1);
}

// ---- 2.25-byte row format ("T16", SE3_PRECISION_BF16X3_T16) of the row-sized intermediates ---------------------------
// Block floating point: 4 consecutive channels of one basis function share a power-of-two exponent and keep 16-bit signed
// mantissas -- the block a lane of the edge kernels owns as four consecutive accumulator registers, so the producer
// needs no cross-lane reduction.  (What the block shape costs in accuracy was simulated on rows with the operator's
// statistics, profiles/r04_t16_format_simulation.txt: 2.0e-5 on every output and gradient for this shape, 3.1e-5 for
// a block per (row, k, 32 channels), 5.7e-5 for one scale per row -- against 6.5e-6 for the 3-byte rows.)
// Row layout for C channels x 32 basis functions, C a multiple of 8:
//   bytes [0, 64 C)        int16 mant[(c >> 2) * 128 + k * 4 + (c & 3)]      (k' order = channel quad, basis, channel)
//   bytes [64 C, 72 C)     uint8 expo[t16_exp_pos(b)], b = (c >> 2) * 32 + k  (one per block = per 4 mantissas)
//   value = mant * 2^(expo - kT16ExpBias) / 32767,   2^(expo - bias) > max |value| of the block
// The exponent plane is stored piece-major inside "mega tiles" of 64 blocks (256 mantissas): the NN GEMM walks a row in
// 16-byte pieces (8 mantissas = 2 blocks), thread c of the 8 threads of a row takes piece c of every 64-mantissa super
// tile -- its exponents of 4 consecutive super tiles are 8 consecutive bytes, and the 8 threads of a row read one
// 64-byte line per mega tile (the row stream is bound by the number of 64-byte requests, gemm_bf16.hip).
constexpr int kT16ExpBias = 64;
__host__ __device__ inline int64_t t16_row_bytes(int channels) { return (int64_t)channels * 72; }
// position kq in a T16 row -> the index c*32 + k the fp32 / packed-word rows use
__host__ __device__ inline int t16_k_of(int kq) { return ((((kq >> 7) << 2) | (kq & 3)) << 5) + ((kq >> 2) & 31); }
// block b -> byte position in the exponent plane: [mega tile b >> 6][piece (b >> 1) & 7][super tile (b >> 4) & 3][b & 1]
__host__ __device__ inline int t16_exp_pos(int b) { return (b & ~63) | (((b >> 1) & 7) << 3) | (((b >> 4) & 3) << 1) | (b & 1); }
// four values -> two words of packed mantissas + the exponent byte
__device__ __forceinline__ void t16_pack4(float x0, float x1, float x2, float x3, uint32_t& m01, uint32_t& m23, uint32_t& eb) {
  const float m = fmaxf(fmaxf(fabsf(x0), fabsf(x1)), fmaxf(fabsf(x2), fabsf(x3)));
  int e = __builtin_amdgcn_frexp_expf(m);  // m = f * 2^e, f in [0.5, 1): |x| * 2^-e < 1 (0 for m = 0)
  e = e < -kT16ExpBias ? -kT16ExpBias : (e > 127 ? 127 : e);  // below 2^-64 of anything this operator produces: flushed
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 a = __builtin_amdgcn_cvt_pknorm_i16(__builtin_ldexpf(x0, -e), __builtin_ldexpf(x1, -e));  // rint(x * 32767)
  const s16x2 b = __builtin_amdgcn_cvt_pknorm_i16(__builtin_ldexpf(x2, -e), __builtin_ldexpf(x3, -e));
  m01 = __builtin_bit_cast(uint32_t, a);
  m23 = __builtin_bit_cast(uint32_t, b);
  eb = (uint32_t)(e + kT16ExpBias);
}
// scale of a block from its exponent byte
__device__ __forceinline__ float t16_scale(uint32_t eb) { return __builtin_ldexpf(1.0f / 32767.0f, (int)eb - kT16ExpBias); }
// one word of two mantissas -> the packed pair of hi parts / of lo parts of the two values (the MFMA operand halves)
__device__ __forceinline__ void t16_unpack2(uint32_t w, float sc, uint32_t& hi_pair, uint32_t& lo_pair) {
  const float x0 = (float)(int)(int16_t)(w & 0xffffu) * sc, x1 = (float)((int)w >> 16) * sc;
  hi_pair = cvt_pk_bf16(x0, x1);
  lo_pair = cvt_pk_bf16(x0 - __uint_as_float(hi_pair << 16), x1 - __uint_as_float(hi_pair & 0xffff0000u));
}

// a 16-byte piece (8 mantissas = two blocks with the exponent bytes e2 & 0xff, e2 >> 8) -> its hi / lo fragment words
__device__ __forceinline__ void t16_unpack8(u32x4 mw, uint32_t e2, u32x4& vh, u32x4& vl) {
  const float sc0 = t16_scale(e2 & 0xffu), sc1 = t16_scale((e2 >> 8) & 0xffu);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t hw, lw;
    t16_unpack2(mw[i], i < 2 ? sc0 : sc1, hw, lw);
    vh[i] = hw, vl[i] = lw;
  }
}

// ds_read_b64_tr_b16: every group of 16 lanes reads a block of 4 rows x 16 columns of 16-bit values and gets it back
// column-major -- lane i of the group receives column i, rows 0..3 packed as two words (row0 | row1 << 16, row2 |
// row3 << 16).  Lane 4q + p of the group supplies the address of row q, columns 4p .. 4p+3 (8-byte aligned).  That is
// the bf16 MFMA fragment of an operand whose K index runs over the ROWS of the LDS image (the TN GEMM's operands);
// mapping verified by tools/probes/tr_read.hip.  EXEC must be all ones.
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 lds_read_tr16(const uint16_t* p) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)p));
}
// fragment = rows r .. r+3 and r+4 .. r+7 of the image for this lane's column
__device__ __forceinline__ u32x4 lds_frag_tr16(const uint16_t* p_rows03, const uint16_t* p_rows47) {
  const u32x2 a = lds_read_tr16(p_rows03), b = lds_read_tr16(p_rows47);
  return u32x4{a[0], a[1], b[0], b[1]};
}

// 8 packed words -> the two 8 x bf16 MFMA fragments
__device__ __forceinline__ void frags_from_words(const uint32_t w[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hi[i] = pair_hi(w[2 * i], w[2 * i + 1]);
    lo[i] = pair_lo(w[2 * i], w[2 * i + 1]);
  }
}
// 8 fp32 values -> the two fragments
__device__ __forceinline__ void frags_from_floats(const float v[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t a, b;
    split2(v[2 * i], v[2 * i + 1], a, b);
    hi[i] = a;
    lo[i] = b;
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// Exact-erf GELU (torch.nn.GELU() default; reference PNEConvLayer.py:94-95) and its derivative.
// erfc through Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 on erf, i.e. <= 7.5e-8 on the normal
// CDF -- fp32 rounding level), evaluated on |x| so the negative tail keeps its relative accuracy:
//   Phi(-|x|) = 0.5 * erfc(|x|/sqrt2) = 0.5 * t*(a1 + t*(a2 + ...)) * exp(-x^2/2),  t = 1/(1 + p|x|/sqrt2)
// The same exponential gives the density for GELU'.
// Select-free arrangement (no compare / cndmask, 0.5 folded into the coefficients):
//   q = Phi(-|x|) = (0.5 a(t)) t e,  cdf = 0.5 + s (0.5 - q),  s = sign(x),  y = x cdf,
//   dy = 0.5 + s ((0.5 - q) + |x| e / sqrt(2 pi))
// 12 (y) / 16 (y and dy) full-rate ops + v_rcp_f32 + v_exp_f32 per value.  The negative tail loses its relative
// accuracy (absolute error <= 1 ulp of 0.5 on cdf), which is below the 7.5e-8 of the approximation itself.
__device__ __forceinline__ void gelu_erf_core(float x, float& s, float& a, float& e, float& hq) {
  a = fabsf(x);
  s = __builtin_copysignf(1.0f, x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f, a, 1.0f));  // 0.3275911 / sqrt2
  float p = fmaf(0.5307027145f, t, -0.7265760135f);                     // the 7.1.26 coefficients, halved
  p = fmaf(p, t, 0.7107068705f);
  p = fmaf(p, t, -0.142248368f);
  p = fmaf(p, t, 0.127414796f);
  e = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);  // exp(-x^2/2)
  hq = fmaf(-(p * t), e, 0.5f);
}
__device__ __forceinline__ void gelu_erf_grad(float x, float& y, float& dy) {
  float s, a, e, hq;
  gelu_erf_core(x, s, a, e, hq);
  y = fmaf(a, hq, 0.5f * x);
  dy = fmaf(s, fmaf(a * e, 0.39894228040143267794f, hq), 0.5f);
}
__device__ __forceinline__ float gelu_erf(float x) {
  float s, a, e, hq;
  gelu_erf_core(x, s, a, e, hq);
  return fmaf(a, hq, 0.5f * x);  // == x * (0.5 + s hq), without needing s
}

// Scaled form used by the split-bf16 edge kernels.  The kernel MLP hands over x' = kGeluIn * x (the factor is folded
// into [A; beta] when the weights are staged), so exp(-x^2/2) = exp2(-x'^2) needs no constant multiply; the result
// is y'' = kGeluOut * GELU(x) = |x'| (1 - 2q) + x', which needs no 0.5 x either (kGeluOut is divided out in the
// alpha of the GEMM that consumes T / U).  dy2 = 2 GELU'(x).  14 / 18 issue slots per value instead of 16 / 20
// (value only: gelu_scaled below drops the reciprocal).
constexpr float kGeluIn = 0.84932180028801904272f;   // sqrt(0.5 * log2(e))
constexpr float kGeluOut = 2.0f * kGeluIn;
__device__ __forceinline__ void gelu_scaled_core(float xp, float& a, float& e, float& hq2) {
  a = fabsf(xp);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.23164189f / kGeluIn, a, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  e = __builtin_amdgcn_exp2f(-(xp * xp));  // exp(-x^2/2)
  hq2 = fmaf(-(p * t), e, 1.0f);           // 1 - 2 Phi(-|x|)
}
// Value only (the forward and transposed edge passes, the edge-major feature gradient): 2 Phi(-|x|) = exp2(P(|x'|)) with P a
// polynomial (round 5: degree 7; round 6: degree 5).  Weighted minimax fit of log2(2 Phi(-a)) on |x| <= 6.4 for the
// absolute error of |x| * 2 Phi(-|x|) (Lawson iteration on the linearised problem, tools/fit_gelu_poly.py).  The degree-5
// fit's leading coefficient comes out negative by itself and P' < 0 on the whole half line (checked to |x'| = 2000 and
// symbolically beyond: every term of P' is negative past |x'| = 10.2), so exp2(P) keeps falling beyond the fitted range:
// no clamp, no overflow into the result for any finite input.  |error| on GELU in fp32: max 7.2e-7 / rms 2.7e-7 on
// [-7, 7] (degree 7: 5.7e-7 / 0.9e-7, the 7.1.26 form 5.6e-7 / 1.05e-7) -- a fifth of the 2^-17 relative rounding the
// split-bf16 operands put on a phi of typical size (3e-6), and the maximum is the fp32 rounding of the final fused step
// in every form.  7 full-rate ops + v_exp_f32 = 36 issue cycles per value instead of 44 (degree 7) / 52 (7.1.26).  Once the
// chunk-stream kernel had taken the wave's idle time out, the edge passes follow their VALU count at ~2/3 of the
// proportional rate (degree 6 instead of 7: -3 % on edge_t, profiles/r06_gelu_degree_ab.txt).
// SE3_GELU_POLY=7 builds the degree-7 fit, =0 the 7.1.26 form.
#ifndef SE3_GELU_POLY
#define SE3_GELU_POLY 5
#endif
__device__ __forceinline__ float gelu_scaled(float xp) {
#if SE3_GELU_POLY == 0
  float a, e, hq2;
  gelu_scaled_core(xp, a, e, hq2);
  return fmaf(a, hq2, xp);
#else
  const float a = fabsf(xp);
#if SE3_GELU_POLY == 7
  float p = fmaf(-3.000000106e-06f, a, 1.157411680e-04f);
  p = fmaf(p, a, -1.839424018e-03f);
  p = fmaf(p, a, 1.570610516e-02f);
  p = fmaf(p, a, -8.734710515e-02f);
  p = fmaf(p, a, -6.359119415e-01f);
  p = fmaf(p, a, -1.355453730e+00f);
  p = fmaf(p, a, 8.137106306e-06f);
#else
  float p = fmaf(-1.070951186e-03f, a, 1.361500331e-02f);
  p = fmaf(p, a, -8.459421670e-02f);
  p = fmaf(p, a, -6.376852093e-01f);
  p = fmaf(p, a, -1.354949055e+00f);
  p = fmaf(p, a, -3.763227806e-05f);
#endif
  const float q2 = __builtin_amdgcn_exp2f(p);  // 2 Phi(-|x|)
  return fmaf(-a, q2, a + xp);                 // |x'| (1 - q2) + x'
#endif
}
__device__ __forceinline__ void gelu_scaled_grad(float xp, float& y2, float& dy2) {
  float a, e, hq2;
  gelu_scaled_core(xp, a, e, hq2);
  y2 = fmaf(a, hq2, xp);
  dy2 = fmaf(__builtin_copysignf(1.0f, xp), fmaf(a * e, 0.79788456080286535588f / kGeluIn, hq2), 1.0f);
}
// Derivative only (the parameter-gradient kernels): 2 GELU'(x) = 1 + sign(x) (1 - H(|x|)), H(a) = 2 Phi(-a) - 2 a pdf(a).
// H changes sign once, at a0 = 0.751791524693... (the fixed point of the Mills ratio), and H(a) / (a0 - a) is positive and
// smooth with a Gaussian tail, so H(a) = (a0 - a) exp2(P(a)) with P a degree-7 polynomial (round 5; weighted minimax fit of
// the absolute error of H on |x| <= 6.4, leading coefficient negative: P keeps falling beyond the range, no clamp).
// |error| <= 1.85e-7 on 2 GELU' before rounding; in fp32 max 4.2e-7 / rms 1.5e-7 on [-3, 3] (7.1.26 form: 5.6e-7 / 1.0e-7).
// 11 full-rate ops + v_exp_f32 = 52 issue cycles per wavefront instead of 12 + v_rcp_f32 + v_exp_f32 = 64.
// (degree 7 here: the derivative's degree-6 fit is not monotone beyond the fitted range and its degree-5 fit is 1e-5 off,
// tools/fit_gelu_poly.py)
#ifndef SE3_GELU_DPOLY
#define SE3_GELU_DPOLY (SE3_GELU_POLY ? 7 : 0)  // 0: the 7.1.26 form
#endif
__device__ __forceinline__ float gelu_scaled_dgrad(float xp) {
#if SE3_GELU_DPOLY == 0
  float y2, dy2;
  gelu_scaled_grad(xp, y2, dy2);
  return dy2;
#else
  const float a = fabsf(xp);
  float p = fmaf(-6.029059296e-05f, a, 9.206497925e-04f);
  p = fmaf(p, a, -6.434103474e-03f);
  p = fmaf(p, a, 2.816033363e-02f);
  p = fmaf(p, a, -8.932273835e-02f);
  p = fmaf(p, a, -7.773189545e-01f);
  p = fmaf(p, a, -4.511650503e-01f);
  p = fmaf(p, a, 6.472119689e-01f);
  const float g = fmaf(a - 6.385129094e-01f, __builtin_amdgcn_exp2f(p), 1.0f);  // 1 - H(|x|)
  return fmaf(__builtin_copysignf(1.0f, xp), g, 1.0f);
#endif
}

// 9-D edge descriptor (reference PNEConvLayerRotEquiv.py:68-90):
//   d[0..2] = (rho * (x_in - y_out))^T R_out          (RotationFunctions.py:637-665)
//   d[3..8] = rows 0,1 of R_out^T R_in                 (RotationFunctions.py:549-600, 236-252)
// R* are row-major 3x3 whose columns are the frame's basis vectors.
__device__ __forceinline__ void edge_descriptor(const float x_in[3], const float r_in[9],
                                                const float y_out[3], const float r_out[9],
                                                float rho, float d[9]) {
  const float v0 = (x_in[0] - y_out[0]) * rho;
  const float v1 = (x_in[1] - y_out[1]) * rho;
  const float v2 = (x_in[2] - y_out[2]) * rho;
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = v0 * r_out[c] + v1 * r_out[3 + c] + v2 * r_out[6 + c];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      d[3 + 3 * r + c] = r_out[r] * r_in[c] + r_out[3 + r] * r_in[3 + c] + r_out[6 + r] * r_in[6 + c];
}

// ---- gather helpers shared by the edge kernels of both arithmetic modes ------------------------------------------
// byte offset no gathered buffer reaches: a raw buffer load from it returns 0
constexpr int kOobOffset = 0x7fff0000;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void* p, int64_t bytes) {
  const uint32_t n = bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)n, 0x00020000);
}

// Packed geometry record of one (point, frame) row: 16 floats = [x, y, z, R8 | R0..R3 | R4..R7 | pad] (64 bytes, built
// by pack_geometry_kernel).  A lane fetches its row with three 16-byte loads out of ONE cache line; from the
// reference's separate [N,3] / [N,F,9] arrays it took 12 dword loads, and a gather instruction costs the L1 one
// cycle per distinct line it touches -- that alone (12 x 32..64 lines per 32 frame-edges) bounded the edge kernels.
__device__ __forceinline__ void load_geom_record(const __amdgpu_buffer_rsrc_t rs, int row, float x[3], float r[9]) {
  const int off = row * 64;
  const auto v0 = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
  const auto v1 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
  const auto v2 = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 32, 0, 0);
  x[0] = __uint_as_float(v0[0]), x[1] = __uint_as_float(v0[1]), x[2] = __uint_as_float(v0[2]);
  r[8] = __uint_as_float(v0[3]);
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = __uint_as_float(v1[i]), r[4 + i] = __uint_as_float(v2[i]);
}

// Rotating wave priority (round 6).  The SIMD's arbiter prefers the OLDEST wavefront among equals, so the resident
// wavefronts of a SIMD do not advance at one rate: stamped, the four of a SIMD finish at 0.59 / 0.70 / 0.89 / 1.00 of its
// span although they hold the same work to 5 % (profiles/r06_edge_timeline.txt) -- a fifth of the slot time is idle and the
// last wavefront runs alone, where nothing covers its latencies.  Every wavefront therefore takes priority
// (step + its slot) mod 4 and moves on by one every chunk: each is the preferred one a quarter of the time.
#ifndef SE3_ROTATE_PRIO
#define SE3_ROTATE_PRIO 1
#endif
__device__ __forceinline__ void rotate_priority(int step_plus_slot) {
#if SE3_ROTATE_PRIO
  switch (step_plus_slot & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
  }
#endif
}
__device__ __forceinline__ int wave_slot_id() { return (int)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (3 << 11)) ; }  // HW_ID[3:0]

inline int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SE3_OK : SE3_ERR_LAUNCH;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Fill `n_words` 32-bit words with `value` by a kernel (prep.hip).  Used wherever hipMemsetAsync would do: a memset NODE of
// a captured graph faults on replay once an RCCL collective has run between two replays on the HIP runtime PyTorch 2.10
// ships (7.0.51831) -- round 4, tools/debug_up_graph.py; kernel nodes and device-to-device copy nodes are fine.
int launch_fill_words(void* dst, uint32_t value, int64_t n_words, hipStream_t stream);

// Optional per-kernel timing (se3_profile_* in se3conv.h): when enabled every launcher brackets its
// kernel with hipEvents on the launch stream.  Off by default; the only process-wide state.
void prof_begin(const char* tag, hipStream_t stream);
void prof_end(hipStream_t stream);
struct ProfScope {
  hipStream_t s;
  ProfScope(const char* tag, hipStream_t stream) : s(stream) { prof_begin(tag, s); }
  ~ProfScope() { prof_end(s); }
};

// ---- kernels implemented in the other translation units (host launchers) ----------------------
struct EdgeGeom {            // one side-agnostic view of the geometry for the edge kernels
  const float* ctr_pts;      // [Nc,3]   points the rows are centred on
  const float* ctr_frames;   // [Nc,Fc,9]
  const float* nb_pts;       // [Nn,3]   points the edges lead to
  const float* nb_frames;    // [Nn,Fn,9]
  const float* ctr_geom;     // packed 64-byte records per (point, frame) row (pack_geometry_kernel); bf16 kernels
  const float* nb_geom;
  int64_t n_nb;              // number of points on the neighbour side
  int64_t n_edges;           // rows of the id list (the edge buffer's capacity); 0: unknown (kernels that bounds-check ids are not used)
  const int32_t* nbr;        // neighbour id of edge e at nbr[e*nbr_stride + nbr_offset]
  int nbr_stride, nbr_offset;
  const int32_t* ends;       // [Nc] inclusive end offsets of every centre's edge group
  int64_t n_ctr;
  int f_ctr, f_nb;
  int transposed;            // 0: centre = output point (descriptor "out" side); 1: centre = input point
};

int launch_edge_t(const char* tag, const EdgeGeom& g, const float* feat, int channels, int64_t feat_rows,
                  const float* axes_ext, const float* rho, float* t_out, hipStream_t stream);
int launch_edge_param_grad(const char* tag, const EdgeGeom& g, const float* feat, int channels, int64_t feat_rows,
                           const float* axes_ext, const float* rho, const float* grad_t, float* partials,
                           int n_partials, int* n_used, hipStream_t stream);
int edge_param_grad_blocks(int64_t rows);

int launch_gemm_nn(const char* tag, const float* a, const float* b, float* c, int64_t m, int n, int k,
                   const float* alpha_num, float alpha_scale, hipStream_t stream, float* split_ws = nullptr);
size_t gemm_nn_split_bytes(int64_t m, int n, int k);  // room for split_ws (0: the shape is not split)
int launch_gemm_tn(const char* tag, const float* a, const float* b, float* c, float* partials, int splits, int64_t m,
                   int ka, int n, const float* alpha_num, float alpha_scale, hipStream_t stream);
int gemm_tn_splits(int64_t m, int ka, int n);
int launch_reduce_partials(const float* partials, float* out, int64_t count, int splits, const float* alpha_num,
                           float alpha_scale, hipStream_t stream);

// Batched reductions behind the backward pass's last kernels (prep.hip): the split-K partials of the grad_X GEMM, the
// row-range partials of the weight-gradient GEMM and the per-workgroup partial sums of d[A; beta] are independent
// fixed-order sums -- collected here and folded by ONE launch at the end of se3conv_bwd instead of three (a launch is
// 5-7 us on the small hierarchy levels whatever it does).
struct ReduceJob {
  int type, blocks;         // 0: out[i] = alpha * sum_z partials[z * count + i];  1: d[A; beta] from n_partials slots of 320
  const float* partials;
  void *out, *out2;         // type 1: grad_axes, grad_biases (either may be NULL)
  int64_t count;
  int splits, packed;       // packed: out holds packed hi/lo words
  const float* alpha_num;
  float alpha_scale;
  int perm_n;               // type 0, > 0: the partials are C'[(o, k), i] with i < perm_n, the output is dW[i, k, o] (kBasis k's)
};
struct ReduceJobs {
  int count;
  ReduceJob job[4];
};
struct ReduceBatch {
  ReduceJobs jobs{};
  void sum(const float* partials, void* out, int64_t count, int splits, const float* alpha_num, float alpha_scale, bool packed,
           int perm_n = 0);
  void params(const float* partials, int n_partials, float* grad_axes, float* grad_biases, float scale);
  int launch(hipStream_t stream);
};

// split-bf16 path (edge_bf16.hip, gemm_bf16.hip)
int launch_split_pack(const float* src, uint32_t* dst, int64_t n, hipStream_t stream);
bool edge_t_bf16_row_ranges(const EdgeGeom& g, int channels);
bool edge_t_bf16_t24_rows(const EdgeGeom& g, int channels);
bool edge_t_bf16_t16_rows(const EdgeGeom& g, int channels);
int launch_edge_t_bf16(const char* tag, const EdgeGeom& g, const uint32_t* feat, int channels, int64_t feat_rows,
                       const float* axes_ext, const float* rho, uint32_t* t_out, hipStream_t stream,
                       int64_t row_lo = -1, int64_t row_hi = -1, int rowfmt = 0);  // rowfmt: 0 packed words, 1 3-byte rows, 2 T16
int edge_param_grad_bf16_channel_blocks(int channels);
int launch_edge_param_grad_bf16(const char* tag, const EdgeGeom& g, const uint32_t* feat, int channels,
                                int64_t feat_rows, const float* axes_ext, const float* rho, const uint32_t* grad_t,
                                float* partials, int n_partials, int* n_used, hipStream_t stream, bool gt16 = false,  // gt16: grad_t rows in the T16 block format
                                int64_t row_lo = -1, int64_t row_hi = -1);  // rows row_lo .. row_hi - 1 only (edge_param_grad_bf16_row_ranges)
bool edge_param_grad_bf16_row_ranges(const EdgeGeom& g, int channels);
bool edge_param_grad_bf16_t16_rows(const EdgeGeom& g, int channels);
// edge-major feature gradient of a convolution with many more input than output rows (edge_dx.hip)
bool edge_dx_bf16_applicable(const EdgeGeom& g, int channels);
int launch_edge_dx_bf16(const char* tag, const EdgeGeom& g, const float* axes_ext, const float* rho, const uint32_t* grad_t,
                        int channels, float* d_rows, hipStream_t stream);
int launch_dx_gather_sum(const char* tag, const float* d_rows, const int32_t* neighbors, const int32_t* ends,
                         const int32_t* t_samples, const int32_t* t_ends, const int32_t* t_edge_ids, int64_t n_src, int width,
                         float scale, float* grad_feat, hipStream_t stream);
int launch_prep_weights(const float* w, int c_in, int kb, int c_out, int mode, uint16_t* bt_hi, uint16_t* bt_lo,
                        hipStream_t stream, const float* scale_num = nullptr, float scale = 1.0f,
                        bool frag_layout = false);
// squared distance of the kNN kernels: one expression for the all-pairs scan and the grid search, so that both
// order equal-looking candidates identically
__device__ __forceinline__ float knn_dist2(float dx, float dy, float dz) { return fmaf(dz, dz, fmaf(dy, dy, dx * dx)); }
int launch_knn_bruteforce(const float* pts, const int32_t* batch_ids, int64_t n, const float* qpts, const int32_t* qbatch,
                          int64_t m, int k, int32_t* out, hipStream_t stream);
int launch_knn_listed(const float* pts, const int32_t* batch_ids, int64_t n, int k, int32_t* out, const int32_t* list,
                      const int32_t* list_count, hipStream_t stream);

// The K best (distance, index) pairs of a stream of candidates, ascending in (distance, index) -- the order of the
// reference's sweep (knn_query.cu:68, strict '<' while scanning ascending indices).  Lists live in registers (K is a
// power of two, all indices compile-time); lanes that scanned disjoint candidate sets combine their lists with a
// bitonic merge through shuffles: c[e] = min(a[e], b[K-1-e]) holds the K smallest of both lists as a bitonic
// sequence, log2(K) compare-exchange stages sort it.
template <int K>
struct TopK {
  // Entry = (bits of the squared distance << 32) | index: distances are >= +0, so their bit patterns order as the values do
  // and ONE unsigned 64-bit comparison is the lexicographic (distance, index) order -- a compare-exchange is a v_cmp_lt_u64
  // and four v_cndmask instead of three compares, two logic ops and four v_cndmask on separate (float, int) arrays
  // (round 6: the ordered insertion is what the cell-grid k-NN kernel spends its time on).
  uint64_t key[K];
  static __device__ __forceinline__ uint64_t pack(float d, int i) {
    return ((uint64_t)__float_as_uint(d) << 32) | (uint64_t)(uint32_t)i;
  }
  __device__ __forceinline__ float dist(int e) const { return __uint_as_float((uint32_t)(key[e] >> 32)); }
  __device__ __forceinline__ int idx(int e) const { return (int)(uint32_t)key[e]; }
  __device__ __forceinline__ void set(int e, float d, int i) { key[e] = pack(d, i); }
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int e = 0; e < K; ++e) key[e] = pack(3.0e38f, 0x7fffffff);
  }
  __device__ __forceinline__ void cas(int a, int b) {  // afterwards entry a <= entry b
    const uint64_t x = key[a], y = key[b];
    const bool sw = y < x;
    key[a] = sw ? y : x, key[b] = sw ? x : y;
  }
  __device__ __forceinline__ void insert(float dd, int jj) {
    const uint64_t k = pack(dd, jj);
    if (!(k < key[K - 1])) return;
    key[K - 1] = k;
#pragma unroll
    for (int e = K - 1; e > 0; --e) cas(e - 1, e);
  }
  __device__ __forceinline__ void merge_xor(int mask) {  // both partner lanes end up with the merged list
    uint64_t tmp[K];
#pragma unroll
    for (int e = 0; e < K; ++e) {
      const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key[K - 1 - e], mask);
      const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(key[K - 1 - e] >> 32), mask);
      tmp[e] = ((uint64_t)hi << 32) | lo;
    }
#pragma unroll
    for (int e = 0; e < K; ++e)
      if (tmp[e] < key[e]) key[e] = tmp[e];
#pragma unroll
    for (int s = K / 2; s > 0; s >>= 1)
#pragma unroll
      for (int e = 0; e < K; ++e)
        if ((e & s) == 0) cas(e, e + s);
  }
};

// Batched operand preparation (prep.hip): collect the jobs of one call, launch them as one kernel.
struct PrepJob {
  int type, blocks;
  const void *a, *b;
  void *o0, *o1;
  int64_t n;
  int p[8];
  float scale;
};
struct PrepJobs {
  int count;
  PrepJob job[8];
};
struct PrepBatch {
  PrepJobs jobs{};
  int status = SE3_OK;
  void axes(const float* axes, const float* biases, float* ext);
  void geometry(const float* pts, const float* frames, int64_t n, int f, float* records);
  void split(const float* src, uint32_t* dst, int64_t n);
  void weights(const float* w, int c_in, int kb, int c_out, int mode, uint16_t* bt_hi, uint16_t* bt_lo,
               const float* scale_num = nullptr, float scale = 1.0f, bool frag_layout = false, int a_rowfmt = 0,
               bool out_t16 = false);  // a_rowfmt: k order of the A rows (0 natural, 1 3-byte rows, 2 T16); out_t16: mode 1's output columns in T16 order
  int launch(hipStream_t stream);
};

int launch_pack_geometry(const float* pts, const float* frames, int64_t n, int f, float* records, hipStream_t stream);
bool gemm_strip_bf16_applicable(int64_t m, int n, int k);
int launch_gemm_strip_bf16(const char* tag, const uint32_t* a, const uint16_t* bt_hi, const uint16_t* bt_lo, uint32_t* c,
                           int64_t m, int n, int k, hipStream_t stream, bool out_t16 = false);  // out_t16: C rows in the T16 block format
bool gemm_strip_t16_applicable(int64_t m, int n, int k);
int launch_gemm_nn_bf16(const char* tag, const uint32_t* a, const uint16_t* bt_hi, const uint16_t* bt_lo, void* c,
                        bool out_packed, int64_t m, int n, int k, float* split_ws, const float* alpha_num,
                        float alpha_scale, hipStream_t stream, int afmt = 0, ReduceBatch* defer = nullptr);  // afmt: A rows 0 packed, 1 3-byte, 2 T16
size_t gemm_nn_bf16_split_bytes(int64_t m, int n, int k);
int launch_gemm_tn_bf16(const char* tag, const uint32_t* a, const uint32_t* b, float* c, float* partials, int splits,
                        int64_t m, int ka, int n, const float* alpha_num, float alpha_scale, hipStream_t stream, int afmt = 0,
                        ReduceBatch* defer = nullptr,  // defer (both GEMMs): the reduction joins the caller's ReduceBatch
                        bool out_ikn = false);  // out_ikn (needs defer): ka = (o, k), n = i -> c is dW[i, k, o] (the weight gradient from U)

}  // namespace se3
