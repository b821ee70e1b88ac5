// Probe 4: what the LDS staging of the NN GEMM costs its A stream.  The walk of probe 3 over 3-byte rows in 64-k super
// tiles (6 pieces per thread), with (a) nothing else, (b) a workgroup barrier per 32 k, (c) the pieces written to LDS
// behind the barrier and read back as fragments (2 x b128 + b64 per thread and step), (d) = (c) plus the weight tile's
// LDS traffic (8 KB written, 32 KB read per block and 32 k).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void walk(const char* a, long m, unsigned* sink) {
  __shared__ __attribute__((aligned(16))) char lds[2][128 * 128 + 20480];
  constexpr int ROW = 6144, NT = 32;
  const int t = threadIdx.x;
  const long row0 = (long)blockIdx.x * 128;
  const int phase = (blockIdx.x * 5) % NT;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[DEPTH][6];
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a), (short)0, (int)(unsigned)(m * ROW), 0x00020000);
  auto load = [&](u32x4* dst, int it) {
    const int kt = (it + phase) % NT;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      dst[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((row0 + p * 32 + t / 8) * ROW + kt * 128 + (t % 8) * 16), 0, 0));
#pragma unroll
    for (int p = 0; p < 2; ++p)
      dst[4 + p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)((row0 + p * 64 + t / 4) * ROW + 4096 + kt * 64 + (t % 4) * 16), 0, 0));
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(buf[d], d);
  for (int it0 = 0; it0 < NT; it0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {  // two 32-k steps per super tile
        if (MODE >= 2) {
          char* l = lds[half];
#pragma unroll
          for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(l + (t + 256 * p) * 16) = buf[d][half * 3 + p];
          if (MODE >= 3) *reinterpret_cast<u32x4*>(l + 16384 + t * 16) = buf[d][half], *reinterpret_cast<u32x4*>(l + 16384 + 4096 + t * 16) = buf[d][half + 2];
        }
        if (MODE >= 1) __syncthreads();
        if (MODE >= 2) {
          const char* l = lds[half];
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            acc ^= *reinterpret_cast<const u32x4*>(l + ((t & 127) * 2 + s) * 16);
            acc.x ^= *reinterpret_cast<const unsigned*>(l + 12288 + (t * 2 + s) * 4);
            if (MODE >= 3) {
#pragma unroll
              for (int c = 0; c < 4; ++c) acc ^= *reinterpret_cast<const u32x4*>(l + 16384 + ((c * 64 + (t & 63)) * 2 + s) * 16);
            }
          }
        } else {
#pragma unroll
          for (int p = 0; p < 3; ++p) acc ^= buf[d][half * 3 + p];
        }
      }
      load(buf[d], it0 + d + DEPTH);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int DEPTH>
void run(const char* name, const char* a, long m, unsigned* sink) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((walk<MODE, DEPTH>), dim3(m / 128), dim3(256), 0, 0, a, m, sink);
  (void)hipEventRecord(e0);
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((walk<MODE, DEPTH>), dim3(m / 128), dim3(256), 0, 0, a, m, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  printf("%-64s %.3f ms  %.2f TB/s\n", name, ms, (double)m * 6144 / ms / 1e9);
}

__global__ void fill_random(unsigned* p, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; p[i] = x;
  }
}

int main() {
  const long m = 131072;
  char* a; unsigned* sink;
  (void)hipMalloc(&a, m * 6144 + (1 << 20)); (void)hipMalloc(&sink, 4);
  hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, (unsigned*)a, m * 6144 / 4);
  (void)hipDeviceSynchronize();
  run<0, 2>("loads only (2 super tiles in flight, 2 blocks/CU by LDS)", a, m, sink);
  run<1, 2>("+ workgroup barrier per 32 k", a, m, sink);
  run<2, 2>("+ A pieces through LDS (write, barrier, fragment reads)", a, m, sink);
  run<3, 2>("+ weight-tile LDS traffic", a, m, sink);
  run<3, 3>("the same, 3 super tiles in flight", a, m, sink);
  return 0;
}
