// Probe 2: same 128-row x 128-byte tile walk, now with the NN GEMM's resource shape: LDS-limited occupancy
// (WGS blocks per CU) and DEPTH tiles of 4 x 16-byte loads per thread in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int LDS_KB, bool BUF>
__global__ __launch_bounds__(256) void walk(const char* a, long m, int row_bytes, unsigned* sink) {
  __shared__ char pad[LDS_KB * 1024];
  const int t = threadIdx.x;
  const long row0 = (long)blockIdx.x * 128;
  const int ntile = row_bytes / 128;
  const int phase = (blockIdx.x * 5) % ntile;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[DEPTH][4];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a), (short)0, (int)(unsigned)(m * row_bytes), 0x00020000);
  auto load = [&](u32x4* dst, int it) {
    const int kt = (it + phase) % ntile;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const long r = row0 + p * 32 + t / 8;
      if constexpr (BUF) {
        const unsigned off = (unsigned)(r * row_bytes + (long)kt * 128 + (t % 8) * 16);
        dst[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
      } else {
        dst[p] = *(const u32x4*)(a + r * row_bytes + (long)kt * 128 + (t % 8) * 16);
      }
    }
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(buf[d], d);
  for (int it0 = 0; it0 < ntile; it0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int p = 0; p < 4; ++p) acc ^= buf[d][p];
      load(buf[d], it0 + d + DEPTH);  // wraps around: a few extra loads at the end
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) { sink[0] = 1; pad[t] = 1; }
}

template <int DEPTH, int LDS_KB, bool BUF = false>
void run(const char* name, const char* a, long m, int row_bytes, unsigned* sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((walk<DEPTH, LDS_KB, BUF>), dim3(m / 128), dim3(256), 0, 0, a, m, row_bytes, sink);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((walk<DEPTH, LDS_KB, BUF>), dim3(m / 128), dim3(256), 0, 0, a, m, row_bytes, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-36s %.3f ms  %.2f TB/s\n", name, ms, (double)m * row_bytes / ms / 1e9);
}

__global__ void fill_random(unsigned* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = x;
  }
}

int main(int argc, char** argv) {
  const long m = 131072; const int row_bytes = 8192;
  char* a; unsigned* sink;
  (void)hipMalloc(&a, m * row_bytes + (1 << 20)); (void)hipMalloc(&sink, 4);
  (void)hipMemset(a, 1, m * row_bytes);
  run<4, 56>("constant data: depth 4, 2 blocks/CU", a, m, row_bytes, sink);
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned*)a, m * row_bytes / 4);
  run<4, 56>("random data:   depth 4, 2 blocks/CU", a, m, row_bytes, sink);
  run<1, 1>("random data:   depth 1, 8 blocks/CU", a, m, row_bytes, sink);
  run<4, 56, true>("buffer loads:  depth 4, 2 blocks/CU", a, m, row_bytes, sink);
  run<6, 56, true>("buffer loads:  depth 6, 2 blocks/CU", a, m, row_bytes, sink);
  run<1, 1, true>("buffer loads:  depth 1, 8 blocks/CU", a, m, row_bytes, sink);
  return 0;
}
