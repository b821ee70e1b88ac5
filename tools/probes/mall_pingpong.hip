// Probe 3: does a tensor that is written by one kernel and read by the next stay in the memory-side cache?
// write kernel (16 B/lane stores) then read kernel (16 B/lane loads) over a buffer of S MB, alternating; per-kernel
// times by HIP events.  If small S reads/writes run well above the HBM rate, chunking producer->consumer pairs of
// the pipeline (T, grad_T, U) through a reused ring buffer would keep those round trips off HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void wr(u32x4* p, long n, unsigned seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    p[i] = u32x4{(unsigned)i ^ seed, seed, (unsigned)i, 7u};
}
__global__ __launch_bounds__(256) void rd(const u32x4* p, long n, unsigned* sink) {
  u32x4 acc = {0, 0, 0, 0};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc ^= p[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
  unsigned* sink; (void)hipMalloc(&sink, 4);
  u32x4* buf; (void)hipMalloc(&buf, 2048l << 20);
  hipEvent_t e[3]; for (auto& x : e) (void)hipEventCreate(&x);
  for (long mb : {16l, 32l, 64l, 128l, 192l, 256l, 384l, 512l, 1024l, 2048l}) {
    const long n = (mb << 20) / 16;
    const int blocks = 4096;
    float tw = 0, tr = 0;
    const int reps = 20;
    for (int i = 0; i < reps + 3; ++i) {
      (void)hipEventRecord(e[0]);
      hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, buf, n, (unsigned)i);
      (void)hipEventRecord(e[1]);
      hipLaunchKernelGGL(rd, dim3(blocks), dim3(256), 0, 0, buf, n, sink);
      (void)hipEventRecord(e[2]);
      (void)hipEventSynchronize(e[2]);
      float a, b; (void)hipEventElapsedTime(&a, e[0], e[1]); (void)hipEventElapsedTime(&b, e[1], e[2]);
      if (i >= 3) { tw += a; tr += b; }
    }
    printf("%5ld MB: write %.3f ms %.2f TB/s | read-after-write %.3f ms %.2f TB/s\n", mb, tw / reps,
           (mb << 20) / (tw / reps) / 1e9, tr / reps, (mb << 20) / (tr / reps) / 1e9);
  }
  return 0;
}
