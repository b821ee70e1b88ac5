// Probe: how many cycles does one wave64 VALU instruction occupy a SIMD's issue port on gfx950 when W waves share the
// SIMD?  (MI355X_MICROARCH.md says 2 per v_fma_f32 at >= 2 waves / SIMD and 4 for a lone wave; the edge kernels of this
// library were accounted at 4.)  Each wave runs a long loop of INDEPENDENT v_fma_f32 (8 accumulators) written in
// inline asm; the kernel stamps s_memtime around the loop.  Also: the same with one v_exp_f32 per 7 fma (the GELU mix)
// and with one MFMA 32x32x16 bf16 per 16 fma (the edge kernels' mix).
// Output: cycles per VALU instruction per SIMD = loop cycles * 1 / (instructions per wave * waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float m = 0.999f, c = 1e-4f;
  f32x16 acc = {0};
  bf16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {8, 7, 6, 5, 4, 3, 2, 1};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                   : "v"(m), "v"(c));
      if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a7));
    }
    if (MODE == 2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  for (int r = 0; r < 16; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves_per_simd, int valu_per_iter) {
  const int cus = 256, iters = 20000;
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD; the dispatcher fills CUs evenly
  float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
  (void)hipMalloc(&cyc, (size_t)blocks * 4 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 4);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2];
  const double n_valu = (double)iters * valu_per_iter;
  // s_memtime ticks at 100 MHz on gfx9 (constant clock): convert through the wall time of the launch instead
  printf("%-22s %d waves/SIMD: launch %.3f ms; per wave %.0f VALU instr; wall-time cycles @2.4GHz per (VALU instr x SIMD): %.2f"
         "  (memtime ticks median %.0f)\n",
         name, waves_per_simd, ms, n_valu, ms * 1e-3 * 2.4e9 / (n_valu * waves_per_simd), med);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4, 8}) run<0>("fma only", w, 16);
  for (int w : {1, 2, 4}) run<1>("fma + 2 exp per 16", w, 18);
  for (int w : {1, 2, 4}) run<2>("16 fma + 1 mfma", w, 16);
  return 0;
}
