// Probe 3: what bounds the NN GEMM's A stream -- bytes or requests?  The same 128-row block walk over rows in the
// packed-word format (8192-byte rows, 128 B per row and k-tile) and in the 3-byte format (6144-byte rows), the latter
// with 32- and 64-element k-tiles and with hi/lo planes either separate or interleaved per k-tile.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// PAT 0: words, 32-k tiles   (4 pieces/thread: 8 lanes x 16 B per row)
// PAT 1: T24,   32-k tiles   (hi 64 B/row: 2 pieces, lo 32 B/row: 1 piece)
// PAT 2: T24,   64-k tiles   (hi 128 B/row: 4 pieces, lo 64 B/row: 2 pieces)
// PAT 3: T24 with [64 B hi | 32 B lo] interleaved per 32-k tile (3 pieces, 6 lanes per row)
// PAT 4: words, 64-k tiles   (8 pieces/thread: 16 lanes x 16 B per row)
// PAT 5: T24 with [128 B hi | 64 B lo] interleaved per 64-k tile (6 pieces, 12 lanes per row)
template <int PAT> struct Pat;
template <> struct Pat<0> { static constexpr int NP = 4, ROW = 8192, NT = 64; };
template <> struct Pat<1> { static constexpr int NP = 3, ROW = 6144, NT = 64; };
template <> struct Pat<2> { static constexpr int NP = 6, ROW = 6144, NT = 32; };
template <> struct Pat<3> { static constexpr int NP = 3, ROW = 6144, NT = 64; };
template <> struct Pat<4> { static constexpr int NP = 8, ROW = 8192, NT = 32; };
template <> struct Pat<5> { static constexpr int NP = 6, ROW = 6144, NT = 32; };

template <int PAT>
__device__ __forceinline__ unsigned piece_offset(long row0, int t, int p, int kt) {
  constexpr int ROW = Pat<PAT>::ROW;
  if constexpr (PAT == 0) return (unsigned)((row0 + p * 32 + t / 8) * ROW + kt * 128 + (t % 8) * 16);
  if constexpr (PAT == 1) {
    if (p < 2) return (unsigned)((row0 + p * 64 + t / 4) * ROW + kt * 64 + (t % 4) * 16);
    return (unsigned)((row0 + t / 2) * ROW + 4096 + kt * 32 + (t % 2) * 16);
  }
  if constexpr (PAT == 2) {
    if (p < 4) return (unsigned)((row0 + p * 32 + t / 8) * ROW + kt * 128 + (t % 8) * 16);
    return (unsigned)((row0 + (p - 4) * 64 + t / 4) * ROW + 4096 + kt * 64 + (t % 4) * 16);
  }
  if constexpr (PAT == 3) {
    const int q = t + 256 * p;
    return (unsigned)((row0 + q / 6) * ROW + kt * 96 + (q % 6) * 16);
  }
  if constexpr (PAT == 4) return (unsigned)((row0 + p * 16 + t / 16) * ROW + kt * 256 + (t % 16) * 16);
  if constexpr (PAT == 5) {
    const int q = t + 256 * p;
    return (unsigned)((row0 + q / 12) * ROW + kt * 192 + (q % 12) * 16);
  }
  return 0;
}

template <int PAT, int DEPTH, int LDS_KB>
__global__ __launch_bounds__(256) void walk(const char* a, long m, unsigned* sink) {
  __shared__ char pad[LDS_KB * 1024];
  constexpr int NP = Pat<PAT>::NP, NT = Pat<PAT>::NT;
  const int t = threadIdx.x;
  const long row0 = (long)blockIdx.x * 128;
  const int phase = (blockIdx.x * 5) % NT;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[DEPTH][NP];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a), (short)0, (int)(unsigned)(m * Pat<PAT>::ROW), 0x00020000);
  auto load = [&](u32x4* dst, int it) {
    const int kt = (it + phase) % NT;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      dst[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, piece_offset<PAT>(row0, t, p, kt), 0, 0));
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(buf[d], d);
  for (int it0 = 0; it0 < NT; it0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int p = 0; p < NP; ++p) acc ^= buf[d][p];
      load(buf[d], it0 + d + DEPTH);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) { sink[0] = 1; pad[t] = 1; }
}

template <int PAT, int DEPTH, int LDS_KB>
void run(const char* name, const char* a, long m, unsigned* sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((walk<PAT, DEPTH, LDS_KB>), dim3(m / 128), dim3(256), 0, 0, a, m, sink);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((walk<PAT, DEPTH, LDS_KB>), dim3(m / 128), dim3(256), 0, 0, a, m, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-52s %.3f ms  %.2f TB/s\n", name, ms, (double)m * Pat<PAT>::ROW / ms / 1e9);
}

__global__ void fill_random(unsigned* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = x;
  }
}

int main() {
  const long m = 131072;
  char* a; unsigned* sink;
  (void)hipMalloc(&a, m * 8192 + (1 << 20)); (void)hipMalloc(&sink, 4);
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned*)a, m * 8192 / 4);
  (void)hipDeviceSynchronize();
  run<0, 4, 56>("words 32-k tiles, depth 4, 2 blocks/CU", a, m, sink);
  run<4, 2, 56>("words 64-k tiles, depth 2, 2 blocks/CU", a, m, sink);
  run<1, 4, 56>("T24 32-k tiles, planes, depth 4, 2 blocks/CU", a, m, sink);
  run<2, 2, 56>("T24 64-k tiles, planes, depth 2, 2 blocks/CU", a, m, sink);
  run<2, 4, 56>("T24 64-k tiles, planes, depth 4, 2 blocks/CU", a, m, sink);
  run<3, 4, 56>("T24 32-k tiles, interleaved, depth 4, 2 blocks/CU", a, m, sink);
  run<5, 2, 56>("T24 64-k tiles, interleaved, depth 2, 2 blocks/CU", a, m, sink);
  run<5, 4, 56>("T24 64-k tiles, interleaved, depth 4, 2 blocks/CU", a, m, sink);
  run<2, 4, 36>("T24 64-k tiles, planes, depth 4, 4 blocks/CU", a, m, sink);
  run<0, 4, 36>("words 32-k tiles, depth 4, 4 blocks/CU", a, m, sink);
  return 0;
}
