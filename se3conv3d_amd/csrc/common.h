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

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// Exact-erf GELU (torch.nn.GELU() default; reference PNEConvLayer.py:94-95) and its derivative.
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ void gelu_erf_grad(float x, float& y, float& dy) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  y = x * cdf;
  dy = cdf + x * pdf;
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

inline int check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? SE3_OK : SE3_ERR_LAUNCH;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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
  const int32_t* nbr;        // neighbour id of edge e at nbr[e*nbr_stride + nbr_offset]
  int nbr_stride, nbr_offset;
  const int32_t* ends;       // [Nc] inclusive end offsets of every centre's edge group
  int64_t n_ctr;
  int f_ctr, f_nb;
  int transposed;            // 0: centre = output point (descriptor "out" side); 1: centre = input point
};

int launch_edge_t(const char* tag, const EdgeGeom& g, const float* feat, int channels, const float* axes_ext,
                  const float* rho, float* t_out, hipStream_t stream);
int launch_edge_param_grad(const char* tag, const EdgeGeom& g, const float* feat, int channels,
                           const float* axes_ext, const float* rho, const float* grad_t, float* partials,
                           int n_partials, hipStream_t stream);
int edge_param_grad_blocks(int64_t rows);

int launch_gemm_nn(const char* tag, const float* a, const float* b, float* c, int64_t m, int n, int k,
                   const float* alpha_num, float alpha_scale, hipStream_t stream);
int launch_gemm_tn(const char* tag, const float* a, const float* b, float* c, float* partials, int splits, int64_t m,
                   int ka, int n, const float* alpha_num, float alpha_scale, hipStream_t stream);
int gemm_tn_splits(int64_t m, int ka, int n);

}  // namespace se3
