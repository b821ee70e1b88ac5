// Probe: operand maps of ds_read_b64_tr_b16 as the TN GEMM would use them.  A [32 rows (m)][128 columns (ka)] image of
// 16-bit values; wave w, k-step s wants for lane (rl, h) the 8 values of column w*32 + rl, rows 16 s + 8 h + 0..7,
// packed in pairs -- i.e. the bf16 MFMA A fragment.  Prints the number of mismatches.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
constexpr int PITCH = 128 + 8;
__global__ void k(unsigned* out) {
  __shared__ __attribute__((aligned(16))) unsigned short img[32][PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32 * 128; i += 256) img[i / 128][i % 128] = (unsigned short)((i / 128) * 256 + (i % 128));  // row*256 + col
  __syncthreads();
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  for (int s = 0; s < 2; ++s)
    for (int half = 0; half < 2; ++half) {
      const int r0 = 16 * s + 8 * (g >> 1) + 4 * half, c0 = wave * 32 + 16 * (g & 1);
      v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)&img[r0 + q][c0 + 4 * p]);
      unsigned* o = out + ((wave * 2 + s) * 64 + lane) * 4 + half * 2;
      o[0] = (unsigned short)v[0] | ((unsigned)(unsigned short)v[1] << 16);
      o[1] = (unsigned short)v[2] | ((unsigned)(unsigned short)v[3] << 16);
    }
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 4 * 2 * 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d);
  unsigned h[4 * 2 * 64 * 4]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w) for (int s = 0; s < 2; ++s) for (int lane = 0; lane < 64; ++lane) for (int j = 0; j < 8; ++j) {
    const int rl = lane & 31, hh = lane >> 5;
    const unsigned want = (16 * s + 8 * hh + j) * 256 + w * 32 + rl;
    const unsigned word = h[((w * 2 + s) * 64 + lane) * 4 + j / 2];
    const unsigned got = (j & 1) ? word >> 16 : word & 0xffffu;
    if (got != want) { if (bad < 5) printf("w%d s%d lane%d j%d: got %u want %u\n", w, s, lane, j, got, want); ++bad; }
  }
  printf("mismatches: %d\n", bad);
  return bad != 0;
}
