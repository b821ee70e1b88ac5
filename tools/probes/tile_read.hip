// Probe: HBM read bandwidth of a [M, 2048] fp32 matrix walked in (BM rows x BKB bytes) tiles per workgroup, the
// access pattern of the NN GEMM's A operand.  hipcc --offload-arch=gfx950 -O3 tile_read.hip -o tile_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int BM, int BKB>  // rows per block, bytes per row per tile
__global__ __launch_bounds__(256) void walk(const char* a, long m, int row_bytes, unsigned* sink) {
  constexpr int LPR = BKB / 16;          // lanes per row
  constexpr int RPP = 256 / LPR;         // rows per pass
  constexpr int PASSES = BM / RPP;
  const int t = threadIdx.x;
  const long row0 = (long)blockIdx.x * BM;
  const int ntile = row_bytes / BKB;
  const int phase = (blockIdx.x * 7) % ntile;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < ntile; ++it) {
    const int kt = (it + phase) % ntile;
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
      const long r = row0 + p * RPP + t / LPR;
      const u32x4 v = *(const u32x4*)(a + r * row_bytes + (long)kt * BKB + (t % LPR) * 16);
      acc ^= v;
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int BM, int BKB>
void run(const char* name, const char* a, long m, int row_bytes, unsigned* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((walk<BM, BKB>), dim3(m / BM), dim3(256), 0, 0, a, m, row_bytes, sink);
  hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((walk<BM, BKB>), dim3(m / BM), dim3(256), 0, 0, a, m, row_bytes, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-28s %.3f ms  %.2f TB/s\n", name, ms, (double)m * row_bytes / ms / 1e9);
}

int main() {
  const long m = 131072; const int row_bytes = 8192;
  char* a; unsigned* sink;
  hipMalloc(&a, m * row_bytes); hipMalloc(&sink, 4);
  hipMemset(a, 1, m * row_bytes);
  run<128, 128>("128 rows x 128 B", a, m, row_bytes, sink);
  run<64, 256>("64 rows x 256 B", a, m, row_bytes, sink);
  run<32, 512>("32 rows x 512 B", a, m, row_bytes, sink);
  run<16, 1024>("16 rows x 1 KB", a, m, row_bytes, sink);
  run<8, 2048>("8 rows x 2 KB", a, m, row_bytes, sink);
  run<128, 256>("128 rows x 256 B", a, m, row_bytes, sink);
  run<128, 512>("128 rows x 512 B", a, m, row_bytes, sink);
  run<64, 512>("64 rows x 512 B", a, m, row_bytes, sink);
  return 0;
}
