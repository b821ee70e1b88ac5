// Neighbourhood construction for gfx950: grid keys, radius (ball) query, edge-list transpose.
//
// Ball query (replaces custom_ops/ball_query/*.cu of the reference):
//   keys(src) -> radix sort (hipCUB) -> per sample 9 key windows [(x+dx, y+dy, z-1) .. (.., z+1)]
//   located by binary search in the sorted keys (no dense pencil table, so the grid size never has
//   to travel to the host) -> one wavefront per sample tests the flattened candidate list 64 at a
//   time; hits are compacted with ballot + popcount, so the order inside a sample is deterministic
//   (the reference scatters with atomics, store_neighbors.cu:129-175).
#include <algorithm>

#include <hipcub/hipcub.hpp>

#include <cstdlib>

#include "common.h"

namespace se3 {

namespace {

__device__ __forceinline__ void cell_of(const float* __restrict__ pts, const int32_t* __restrict__ batch_ids,
                                        const float* __restrict__ aabb_min, const int nc[3], const float inv[3],
                                        int64_t i, int cell[3], int& b) {
  b = batch_ids[i];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    // (p - min) * (1/cell), floor, clamp: grid_utils.cuh:57-67
    const float rel = __fmul_rn(__fsub_rn(pts[i * 3 + d], aabb_min[(int64_t)b * 3 + d]), inv[d]);
    int c = (int)floorf(rel);
    cell[d] = min(max(c, 0), nc[d] - 1);
  }
}

__device__ __forceinline__ int64_t key_of(const int cell[3], const int nc[3], int b) {
  // grid_utils.cuh:79-93
  return (((int64_t)b * nc[0] + cell[0]) * nc[1] + cell[1]) * nc[2] + cell[2];
}

__global__ void compute_keys_kernel(const float* __restrict__ pts, const int32_t* __restrict__ batch_ids,
                                    const float* __restrict__ aabb_min, const int32_t* __restrict__ num_cells,
                                    const float* __restrict__ cell_size, float cell_scalar, int64_t n,
                                    int64_t* __restrict__ keys, int32_t* __restrict__ iota) {
  // cell_size == nullptr: the same cell size `cell_scalar` in every dimension (ball query: cell = radius)
  const int nc[3] = {num_cells[0], num_cells[1], num_cells[2]};
  const float inv[3] = {1.0f / (cell_size ? cell_size[0] : cell_scalar), 1.0f / (cell_size ? cell_size[1] : cell_scalar),
                        1.0f / (cell_size ? cell_size[2] : cell_scalar)};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int cell[3], b;
    cell_of(pts, batch_ids, aabb_min, nc, inv, i, cell, b);
    keys[i] = key_of(cell, nc, b);
    if (iota) iota[i] = (int32_t)i;
  }
}

__global__ void gather_sorted_points_kernel(const float* __restrict__ pts, const int32_t* __restrict__ ids, int64_t n,
                                            float4* __restrict__ spts) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int id = ids[i];
    spts[i] = make_float4(pts[(int64_t)id * 3], pts[(int64_t)id * 3 + 1], pts[(int64_t)id * 3 + 2], __int_as_float(id));
  }
}

__device__ __forceinline__ int lower_bound_key(const int64_t* __restrict__ keys, int n, int64_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// ranges[s][o] = [lo, hi) positions in the sorted source order of pencil window o = (dx+1)*3 + (dy+1)
__global__ void find_ranges_kernel(const float* __restrict__ pts_dst, const int32_t* __restrict__ batch_dst,
                                   const float* __restrict__ aabb_min, const int32_t* __restrict__ num_cells,
                                   float radius, const int64_t* __restrict__ skeys, int n_src, int64_t n_dst,
                                   int2* __restrict__ ranges, const int32_t* __restrict__ order) {
  const int nc[3] = {num_cells[0], num_cells[1], num_cells[2]};
  const float inv_r = 1.0f / radius;
  const float inv[3] = {inv_r, inv_r, inv_r};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dst * 9; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = i / 9;
    const int o = (int)(i - j * 9);
    const int64_t s = order ? order[j] : j;  // a cloud against itself: samples in cell order (see ball_query_count_impl)
    int cell[3], b;
    cell_of(pts_dst, batch_dst, aabb_min, nc, inv, s, cell, b);
    const int x = cell[0] + o / 3 - 1, y = cell[1] + o % 3 - 1;
    int2 r = make_int2(0, 0);
    if (x >= 0 && x < nc[0] && y >= 0 && y < nc[1]) {
      const int z0 = max(cell[2] - 1, 0), z1 = min(cell[2] + 1, nc[2] - 1);
      const int64_t base = (((int64_t)b * nc[0] + x) * nc[1] + y) * nc[2];
      r.x = lower_bound_key(skeys, n_src, base + z0);
      r.y = lower_bound_key(skeys, n_src, base + z1 + 1);
    }
    ranges[s * 9 + o] = r;
  }
}

// ---- 32-bit keys for the ball query (one or two batch elements) ---------------------------------------------------
// The reference's key (grid_utils.cuh:79-93) multiplies the true cell counts; how many bits it needs is only known on
// the device, so the radix sort has to walk all 64 (8 passes over 8-byte keys; at 9 k points those passes are latency:
// ~10 us each).  The ball query's own keys need not be the reference's -- only the edge SET is defined -- so for
// n_batches <= 2 they are built with a fixed stride of 1024 cells per dimension: 30 + 1 bits, 4 passes over 4-byte
// keys.  Cell indices beyond 1023 are clamped to 1023 (points of the far cells share one cell: candidates are a
// superset there, the distance test decides), so any extent / radius ratio stays exact.
constexpr int kBq32Cells = 1024;
__device__ __forceinline__ uint32_t key32_of(const int cell[3], int b) {
  return ((((uint32_t)b * kBq32Cells + (uint32_t)cell[0]) * kBq32Cells) + (uint32_t)cell[1]) * kBq32Cells + (uint32_t)cell[2];
}
__global__ void compute_keys32_kernel(const float* __restrict__ pts, const int32_t* __restrict__ batch_ids,
                                      const float* __restrict__ aabb_min, const int32_t* __restrict__ num_cells,
                                      float cell_scalar, int64_t n, uint32_t* __restrict__ keys,
                                      int32_t* __restrict__ iota) {
  const int nc[3] = {min(num_cells[0], kBq32Cells), min(num_cells[1], kBq32Cells), min(num_cells[2], kBq32Cells)};
  const float inv[3] = {1.0f / cell_scalar, 1.0f / cell_scalar, 1.0f / cell_scalar};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int cell[3], b;
    cell_of(pts, batch_ids, aabb_min, nc, inv, i, cell, b);
    keys[i] = key32_of(cell, b);
    iota[i] = (int32_t)i;
  }
}
__device__ __forceinline__ int lower_bound_key32(const uint32_t* __restrict__ keys, int n, uint32_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__global__ void find_ranges32_kernel(const float* __restrict__ pts_dst, const int32_t* __restrict__ batch_dst,
                                     const float* __restrict__ aabb_min, const int32_t* __restrict__ num_cells,
                                     float radius, const uint32_t* __restrict__ skeys, int n_src, int64_t n_dst,
                                     int2* __restrict__ ranges, const int32_t* __restrict__ order) {
  const int nc[3] = {min(num_cells[0], kBq32Cells), min(num_cells[1], kBq32Cells), min(num_cells[2], kBq32Cells)};
  const float inv_r = 1.0f / radius;
  const float inv[3] = {inv_r, inv_r, inv_r};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_dst * 9; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = i / 9;
    const int o = (int)(i - j * 9);
    const int64_t s = order ? order[j] : j;  // a cloud against itself: samples in cell order (see ball_query_count_impl)
    int cell[3], b;
    cell_of(pts_dst, batch_dst, aabb_min, nc, inv, s, cell, b);
    const int x = cell[0] + o / 3 - 1, y = cell[1] + o % 3 - 1;
    int2 r = make_int2(0, 0);
    if (x >= 0 && x < nc[0] && y >= 0 && y < nc[1]) {
      const int z0 = max(cell[2] - 1, 0), z1 = min(cell[2] + 1, nc[2] - 1);
      const int c0[3] = {x, y, z0};
      const uint32_t base = key32_of(c0, b);
      r.x = lower_bound_key32(skeys, n_src, base);
      r.y = lower_bound_key32(skeys, n_src, base + (uint32_t)(z1 - z0) + 1u);
    }
    ranges[s * 9 + o] = r;
  }
}

// One wavefront per sample.  MODE 0: counts[s] = #hits.  MODE 1: neighbors[base + j] = (s, source id) for the j-th hit
// in candidate order, base from the inclusive offsets `ends`.  MODE 2 (capacity-bounded call): the same, slots at or
// beyond `limit` dropped, the sample's own offset clamped to `limit` in place (its neighbour reads ends[s-1] either
// way and clamps what it reads), the last sample records the true total and the overflow flag in `info`, and
// `sources` (optional) receives the source ids alone -- the source-major list of a cloud against itself.
template <int MODE>
__global__ __launch_bounds__(256) void scan_candidates_kernel(const float* __restrict__ pts_dst, float inv_r,
                                                              const float4* __restrict__ spts,
                                                              const int2* __restrict__ ranges, int64_t n_dst,
                                                              int32_t* __restrict__ counts, int32_t* __restrict__ ends,
                                                              int32_t* __restrict__ neighbors, int limit,
                                                              int32_t* __restrict__ sources, int32_t* __restrict__ info,
                                                              const int32_t* __restrict__ order) {
  constexpr bool STORE = MODE != 0;
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= n_dst) return;
  const int64_t s = order ? order[w] : w;  // the wavefronts of a workgroup then share their candidate windows
  const float sx = pts_dst[s * 3], sy = pts_dst[s * 3 + 1], sz = pts_dst[s * 3 + 2];
  int lo[9], pre[10];
  pre[0] = 0;
#pragma unroll
  for (int o = 0; o < 9; ++o) {
    const int2 r = ranges[s * 9 + o];
    lo[o] = r.x;
    pre[o + 1] = pre[o] + (r.y - r.x);
  }
  const int total = pre[9];
  int found = 0;
  int base = 0;
  if (STORE) base = s > 0 ? ends[s - 1] : 0;
  if (MODE == 2) {
    base = min(base, limit);
    if (lane == 0) {
      const int e = ends[s];
      if (s == n_dst - 1) info[0] = e, info[1] = e > limit ? 1 : 0;
      if (e > limit) ends[s] = limit;
    }
  }
  for (int c0 = 0; c0 < total; c0 += 64) {
    const int c = c0 + lane;
    bool hit = false;
    int id = 0;
    if (c < total) {
      int o = 0;
#pragma unroll
      for (int t = 1; t < 9; ++t) o += (c >= pre[t]) ? 1 : 0;
      int pos = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if (t == o) pos = lo[t] + (c - pre[t]);
      const float4 p = spts[pos];
      // length((s - p) * invR) < 1, un-fused so that it is bit-identical to the CPU oracle
      const float dx = __fmul_rn(__fsub_rn(sx, p.x), inv_r);
      const float dy = __fmul_rn(__fsub_rn(sy, p.y), inv_r);
      const float dz = __fmul_rn(__fsub_rn(sz, p.z), inv_r);
      const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      hit = __fsqrt_rn(d2) < 1.0f;
      id = __float_as_int(p.w);
    }
    const unsigned long long mask = __ballot(hit);
    if (STORE && hit) {
      const int slot = base + found + __popcll(mask & ((1ull << lane) - 1ull));
      if (slot < limit) {
        neighbors[(int64_t)slot * 2] = (int32_t)s;
        neighbors[(int64_t)slot * 2 + 1] = id;
        if (MODE == 2 && sources) sources[slot] = id;
      }
    }
    found += __popcll(mask);
  }
  if (!STORE && lane == 0) counts[s] = found;
}

// Small source sets (n_src <= kBqScanAllMax): one wavefront per sample tests every source, 64 at a time -- no boxes,
// keys, sort or windows, i.e. 3 launches instead of 16 where the launches are all there is to the cost.  Same
// predicate, same batch test; hits of a sample come out in ascending source id.
constexpr int64_t kBqScanAllMax = 2048;
// MODE 0 (count) also leaves (x, y, z, batch id) records of the sources in the workspace: the store phase of the C ABI
// is not handed the source arrays again.  MODE 1: store behind the caller's inclusive offsets.  MODE 2 (bounded call):
// the same with clamping / info / sources as scan_candidates_kernel<2>.  MODE 3 (bounded call, few samples): no scan
// launch at all -- every wavefront sums the counts in front of its sample itself (n_dst / 64 loads per lane), writes the
// sample's clamped inclusive offset, stores, and the last sample records total + overflow flag.
constexpr int64_t kBqInlinePrefixMax = 4096;
template <int MODE>
__global__ __launch_bounds__(256) void scan_all_kernel(const float* __restrict__ pts_src, const int32_t* __restrict__ batch_src,
                                                       float4* __restrict__ recs, const float* __restrict__ pts_dst,
                                                       const int32_t* __restrict__ batch_dst, float inv_r, int n_src,
                                                       int64_t n_dst, int32_t* __restrict__ counts,
                                                       int32_t* __restrict__ ends, int32_t* __restrict__ neighbors,
                                                       int limit, int32_t* __restrict__ sources, int32_t* __restrict__ info) {
  constexpr bool STORE = MODE != 0;
  const int lane = threadIdx.x & 63;
  if (!STORE) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_src) recs[t] = make_float4(pts_src[t * 3], pts_src[t * 3 + 1], pts_src[t * 3 + 2], __int_as_float(batch_src[t]));
  }
  const int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= n_dst) return;
  const float sx = pts_dst[s * 3], sy = pts_dst[s * 3 + 1], sz = pts_dst[s * 3 + 2];
  const int sb = batch_dst[s];
  int found = 0;
  int base = 0;
  if (MODE == 1 || MODE == 2) base = s > 0 ? ends[s - 1] : 0;
  if (MODE == 2) {  // offsets from the scan: clamp in place like scan_candidates_kernel<2>
    base = min(base, limit);
    if (lane == 0) {
      const int e = ends[s];
      if (s == n_dst - 1) info[0] = e, info[1] = e > limit ? 1 : 0;
      if (e > limit) ends[s] = limit;
    }
  }
  if (MODE == 3) {
    int acc = 0;
    for (int64_t j = lane; j < s; j += 64) acc += counts[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    const int e = acc + counts[s];
    base = min(acc, limit);
    if (lane == 0) {
      ends[s] = min(e, limit);
      if (s == n_dst - 1) info[0] = e, info[1] = e > limit ? 1 : 0;
    }
  }
  for (int c0 = 0; c0 < n_src; c0 += 64) {
    const int id = c0 + lane;
    bool hit = false;
    if (id < n_src) {
      float4 p;
      if (STORE) p = recs[id];
      else p = make_float4(pts_src[(int64_t)id * 3], pts_src[(int64_t)id * 3 + 1], pts_src[(int64_t)id * 3 + 2],
                           __int_as_float(batch_src[id]));
      const float dx = __fmul_rn(__fsub_rn(sx, p.x), inv_r);
      const float dy = __fmul_rn(__fsub_rn(sy, p.y), inv_r);
      const float dz = __fmul_rn(__fsub_rn(sz, p.z), inv_r);
      const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      hit = __float_as_int(p.w) == sb && __fsqrt_rn(d2) < 1.0f;
    }
    const unsigned long long mask = __ballot(hit);
    if (STORE && hit) {
      const int slot = base + found + __popcll(mask & ((1ull << lane) - 1ull));
      if (slot < limit) {
        neighbors[(int64_t)slot * 2] = (int32_t)s;
        neighbors[(int64_t)slot * 2 + 1] = id;
        if (MODE >= 2 && sources) sources[slot] = id;
      }
    }
    found += __popcll(mask);
  }
  if (!STORE && lane == 0) counts[s] = found;
}

// Per-batch bounding boxes (BallQuery.py:35-36 / BoundingBox.py:17-18 use torch_scatter's scatter_min/max).
// Wave-level reduction first (batch ids are sorted, so a wavefront almost always holds one batch element), then
// one float atomic per wavefront and coordinate -- torch's scatter_reduce on three addresses took 1.3 ms here.
__device__ __forceinline__ void atomic_min_f(float* addr, float v) {
  // order-preserving integer view of a float: positive floats as signed ints, negative ones reversed.  The branch
  // is taken on the SIGN BIT, not on v >= 0: -0.0f compares equal to zero but its bits are INT_MIN, which as a
  // signed operand would overwrite a negative minimum (and never replace the -inf start value of a maximum).
  if (__float_as_int(v) >= 0) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f(float* addr, float v) {
  if (__float_as_int(v) >= 0) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

__global__ void batch_aabb_init_kernel(float* __restrict__ mn, float* __restrict__ mx, int count,
                                       int32_t* __restrict__ num_cells) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) mn[i] = __int_as_float(0x7f800000), mx[i] = __int_as_float(0xff800000);  // +inf / -inf
  if (num_cells && i < 3) num_cells[i] = 0;  // the grid-parameter kernel behind this one takes maxima into it
}

// A wavefront keeps running minima / maxima in its lanes for as long as the points it reads belong to one batch
// element and turns them into 6 atomics when that element changes or its walk ends (batch ids are sorted, so that is
// once or twice per wavefront): one atomic per wavefront and 64 points made 6144 same-address atomics -- 60 us -- out
// of a single 65 k-point cloud.  64-point groups that straddle two elements fall back to per-point atomics.
__global__ __launch_bounds__(256) void batch_aabb_kernel(const float* __restrict__ pts,
                                                         const int32_t* __restrict__ batch_ids, int64_t n,
                                                         float* __restrict__ mn, float* __restrict__ mx) {
  const float inf = __int_as_float(0x7f800000);
  float lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
  int cur = -1;  // wave-uniform: batch element of the running values
  auto flush = [&]() {
    if (cur < 0) return;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float l = lo[d], h = hi[d];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) l = fminf(l, __shfl_xor(l, off)), h = fmaxf(h, __shfl_xor(h, off));
      if ((threadIdx.x & 63) == 0) atomic_min_f(&mn[cur * 3 + d], l), atomic_max_f(&mx[cur * 3 + d], h);
      lo[d] = inf, hi[d] = -inf;
    }
  };
  for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~63ll; i0 < n;
       i0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i0 + (threadIdx.x & 63);
    const bool ok = i < n;
    const int64_t ic = ok ? i : n - 1;  // lanes past the end repeat the last point: harmless for min / max
    const int b = batch_ids[ic];
    const int b0 = __builtin_amdgcn_readfirstlane(b);
    const bool uniform = __all(b == b0);
    if (uniform) {
      if (b0 != cur) {
        flush();
        cur = b0;
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const float v = pts[ic * 3 + d];
        lo[d] = fminf(lo[d], v), hi[d] = fmaxf(hi[d], v);
      }
    } else {
      // a 64-point group that straddles batch elements (ids are sorted: two of them, rarely more): one reduction and six
      // atomics per element present -- per-point atomics here were 56 us on a 32-body batch of 58 k points, all of it on
      // the 31 straddling groups (6 x 64 atomics on one line each)
      flush();
      cur = -1;
      float v[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) v[d] = pts[ic * 3 + d];
      uint64_t todo = __ballot(1);
      while (todo) {
        const int first = __builtin_ctzll(todo);
        const int be = __shfl(b, first);
        const bool mine = b == be;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          float l = mine ? v[d] : inf, h = mine ? v[d] : -inf;
#pragma unroll
          for (int off = 32; off > 0; off >>= 1) l = fminf(l, __shfl_xor(l, off)), h = fmaxf(h, __shfl_xor(h, off));
          if ((threadIdx.x & 63) == 0) atomic_min_f(&mn[be * 3 + d], l), atomic_max_f(&mx[be * 3 + d], h);
        }
        todo &= ~__ballot(mine);
      }
    }
  }
  flush();
}

// (source, edge index) pairs of the list for the merge-sort form of the transposition.  n_valid (device, may be NULL): rows
// from *n_valid on are unset (the tail of a capacity-bounded edge buffer) -- they get the source id n_src, which sorts
// behind every real group and which no offset of group_ends_kernel reaches
__global__ void split_edges_kernel(const int32_t* __restrict__ neighbors, int64_t e, const int32_t* __restrict__ n_valid,
                                   int32_t n_src, int32_t* __restrict__ src, int32_t* __restrict__ ids) {
  const int64_t valid = n_valid ? (int64_t)max(min((int64_t)*n_valid, e), (int64_t)0) : e;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
    ids[i] = (int32_t)i;
    src[i] = i < valid ? neighbors[i * 2 + 1] : n_src;
  }
}

__global__ void group_ends_kernel(const int32_t* __restrict__ sorted_keys, int64_t e, int64_t n_groups,
                                  int32_t* __restrict__ ends) {
  // ends[p] = number of sorted keys <= p
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_groups; p += (int64_t)gridDim.x * blockDim.x) {
    int64_t lo = 0, hi = e;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sorted_keys[mid] <= (int32_t)p) lo = mid + 1; else hi = mid;
    }
    ends[p] = (int32_t)lo;
  }
}

inline unsigned blocks_for(int64_t n, int per = 256) {
  int64_t b = (n + per - 1) / per;
  if (b < 1) b = 1;
  if (b > 65535 * 16) b = 65535 * 16;
  return (unsigned)b;
}

// ---- exact kNN through the cell grid (scope row f-1) -------------------------------------------------------
// One thread per query, taken in cell order (neighbouring threads walk the same candidates).  The 3x3x3 block of
// cells around the query covers every point closer than one cell size c, so the k best of the block are the true
// k nearest iff the k-th of them is closer than c; otherwise the query goes on a list that the all-pairs
// kernel recomputes (launch_knn_bruteforce).  Order: ascending (distance, index), as the all-pairs kernel and the reference's
// sweep (knn_query.cu:68) produce.
constexpr int kKnnGroup = 4;  // lanes that share one query
template <int K>
__global__ __launch_bounds__(256) void knn_grid_kernel(const float4* __restrict__ spts,
                                                       const int64_t* __restrict__ skeys,
                                                       const int32_t* __restrict__ num_cells,
                                                       const float* __restrict__ cell_size, int n, int k_out,
                                                       int32_t* __restrict__ out, int32_t* __restrict__ list,
                                                       int32_t* __restrict__ list_count) {
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = gtid & (kKnnGroup - 1);
  const int t = min(gtid / kKnnGroup, n - 1);  // whole groups stay together (shuffles below); extras repeat the last query
  const bool writer = g == 0 && gtid / kKnnGroup < n;
  const int nc[3] = {num_cells[0], num_cells[1], num_cells[2]};
  const float4 q = spts[t];
  const int qid = __float_as_int(q.w);
  int64_t key = skeys[t];
  const int iz = (int)(key % nc[2]);
  key /= nc[2];
  const int iy = (int)(key % nc[1]);
  key /= nc[1];
  const int ix = (int)(key % nc[0]);
  const int64_t b = key / nc[0];
  // the 9 (x, y) pencils of the 27-cell block: lane g looks up pencils g, g+4, g+8, the group shares the ranges
  int my_lo[3], my_hi[3];
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int o = g + kKnnGroup * u;
    my_lo[u] = my_hi[u] = 0;
    const int x = ix + o / 3 - 1, y = iy + o % 3 - 1;
    if (o < 9 && x >= 0 && x < nc[0] && y >= 0 && y < nc[1]) {
      const int z0 = max(iz - 1, 0), z1 = min(iz + 1, nc[2] - 1);
      const int64_t base = ((b * nc[0] + x) * nc[1] + y) * nc[2];
      my_lo[u] = lower_bound_key(skeys, n, base + z0);  // (the 18 searches of a group are 7 % of the kernel: 202 -> 188 us
      my_hi[u] = lower_bound_key(skeys, n, base + z1 + 1);  //  per 58 k-point query with fixed windows instead, round 6)
    }
  }
  TopK<K> best;
  best.init();
  int found = 0;
  const int lane = threadIdx.x & 63, lane0 = lane & ~(kKnnGroup - 1);
#pragma unroll
  for (int o = 0; o < 9; ++o) {
    const int src = lane0 + o % kKnnGroup;
    const int lo = __shfl(o / kKnnGroup == 0 ? my_lo[0] : (o / kKnnGroup == 1 ? my_lo[1] : my_lo[2]), src);
    const int hi = __shfl(o / kKnnGroup == 0 ? my_hi[0] : (o / kKnnGroup == 1 ? my_hi[1] : my_hi[2]), src);
    found += hi - lo;
    for (int pos = lo + g; pos < hi; pos += kKnnGroup) {
      const float4 p = spts[pos];
      best.insert(knn_dist2(p.x - q.x, p.y - q.y, p.z - q.z), __float_as_int(p.w));
    }
  }
  best.merge_xor(1);
  best.merge_xor(2);
  const float c = cell_size[0] * 0.999f;  // margin for the rounding of the cell assignment
  float kth = 3.0e38f;
#pragma unroll
  for (int e = 0; e < K; ++e)
    if (e == k_out - 1) kth = best.dist(e);
  const bool ok = found >= k_out && kth < c * c;
  if (writer) {
    if (!ok) list[atomicAdd(list_count, 1)] = qid;  // order of the list does not matter: each entry is recomputed alone
    if (ok) {
#pragma unroll
      for (int e = 0; e < K; ++e)
        if (e < k_out) out[(int64_t)qid * k_out + e] = best.idx(e);
    }
  }
}


// hipcub::DeviceRadixSort::SortPairs for the capturable paths.  Up to rocPRIM's own limit of 1 M items that call is a merge
// sort (kernels and device-to-device copies only); above it, it is the one-sweep radix sort, which issues hipMemsetAsync
// per pass and whose kernels use scratch memory -- and a captured graph with memset nodes faults on replay next to a live
// RCCL communicator on the HIP runtime PyTorch 2.10 ships (see the note above se3_csr_transpose_bounded).  Larger inputs
// therefore take the stable merge sort explicitly (keys here are non-negative with zero bits above end_bit, so both
// orders agree).
constexpr int kRadixIsMergeLimit = 1 << 20;
// The guard rests on rocPRIM's dispatch rule (device_radix_sort.hpp: block sort, then merge sort up to
// radix_sort_config<>::merge_sort_limit for keys wider than 2 bytes, one-sweep above): a rocPRIM whose default limit is
// lower would send sizes below kRadixIsMergeLimit to the one-sweep sort again -- that build must fail, not fault on a GPU
// (ADVICE r4).  tests/test_gpu_graph_nodes.py checks the captured graphs themselves for memset nodes.
static_assert(rocprim::radix_sort_config<>::merge_sort_limit >= (size_t)kRadixIsMergeLimit,
              "rocPRIM's radix sort leaves its merge-sort form below kRadixIsMergeLimit: lower the constant to its limit");
struct KeyLess {
  template <class K>
  __device__ __forceinline__ bool operator()(const K& a, const K& b) const { return a < b; }
};
// Clouds of a few thousand to a few hundred thousand points (every grid of a training step): rocPRIM's stable merge sort with
// block-sorted runs of 4 096 pairs (512 threads x 8) instead of the 1 024 its radix-sort front end uses below 1 M items -- two
// merge launches fewer per sort (58 k pairs: 1 + 4 launches instead of 1 + 6), and the grid builds of a step are bound by
// launches (27 sorts per DFaust step).  Stable either way, so equal keys keep their input order: same result, bit for bit.
#ifndef SE3_SORT_RUN
#define SE3_SORT_RUN 4096  // 0: hipcub::DeviceRadixSort's own choice (A/B)
#endif
using GridSortConfig = rocprim::merge_sort_config<512, 512, (SE3_SORT_RUN ? SE3_SORT_RUN : 4096) / 512>;
template <class Key>
hipError_t sort_pairs_no_scratch(void* temp, size_t& temp_bytes, const Key* kin, Key* kout, const int32_t* vin, int32_t* vout,
                                 int n, int begin_bit = 0, int end_bit = (int)sizeof(Key) * 8, hipStream_t stream = nullptr) {
  static_assert(sizeof(Key) > 2, "rocPRIM sends 1- and 2-byte keys to the one-sweep sort from 100 000 items on");
  if (temp == nullptr) {  // size query: the largest of the forms
    size_t a = 0, b = 0, c = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, a, kin, kout, vin, vout, n, begin_bit, end_bit, stream);
    if (e != hipSuccess) return e;
    e = hipcub::DeviceMergeSort::StableSortPairs(nullptr, b, kout, vout, n, KeyLess(), stream);
    if (e != hipSuccess) return e;
    e = rocprim::merge_sort<GridSortConfig>(nullptr, c, kin, kout, vin, vout, (size_t)n, KeyLess(), stream);
    temp_bytes = a > b ? a : b;
    if (c > temp_bytes) temp_bytes = c;
    return e;
  }
  if (SE3_SORT_RUN && n <= kRadixIsMergeLimit)
    return rocprim::merge_sort<GridSortConfig>(temp, temp_bytes, kin, kout, vin, vout, (size_t)n, KeyLess(), stream);
  if (n <= kRadixIsMergeLimit) return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, kin, kout, vin, vout, n, begin_bit, end_bit, stream);
  hipError_t e = hipMemcpyAsync(kout, kin, (size_t)n * sizeof(Key), hipMemcpyDeviceToDevice, stream);
  if (e != hipSuccess) return e;
  e = hipMemcpyAsync(vout, vin, (size_t)n * 4, hipMemcpyDeviceToDevice, stream);
  if (e != hipSuccess) return e;
  return hipcub::DeviceMergeSort::StableSortPairs(temp, temp_bytes, kout, vout, n, KeyLess(), stream);
}

struct KnnLayout {
  size_t keys, skeys, ids, sids, spts, list, count, temp, temp_bytes, total;
};

KnnLayout knn_layout(int64_t n) {
  KnnLayout l{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t ns = (size_t)(n > 0 ? n : 1);
  l.keys = take(ns * 8);
  l.skeys = take(ns * 8);
  l.ids = take(ns * 4);
  l.sids = take(ns * 4);
  l.spts = take(ns * 16);
  l.list = take(ns * 4);
  l.count = take(4);
  size_t t_sort = 0;
  (void)sort_pairs_no_scratch(nullptr, t_sort, (const int64_t*)nullptr, (int64_t*)nullptr,
                                           (const int32_t*)nullptr, (int32_t*)nullptr, (int)ns);
  l.temp_bytes = t_sort;
  l.temp = take(l.temp_bytes);
  l.total = off;
  return l;
}

struct BqLayout {
  size_t keys, skeys, ids, sids, spts, ranges, counts, temp, temp_bytes, total;
};

BqLayout bq_layout(int64_t n_src, int64_t n_dst) {
  BqLayout l{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t ns = (size_t)(n_src > 0 ? n_src : 1), nd = (size_t)(n_dst > 0 ? n_dst : 1);
  l.keys = take(ns * 8);
  l.skeys = take(ns * 8);
  l.ids = take(ns * 4);
  l.sids = take(ns * 4);
  l.spts = take(ns * 16);
  l.ranges = take(nd * 9 * 8);
  l.counts = take(nd * 4);
  size_t t_sort = 0, t_scan = 0;
  (void)sort_pairs_no_scratch(nullptr, t_sort, (const int64_t*)nullptr, (int64_t*)nullptr,
                                     (const int32_t*)nullptr, (int32_t*)nullptr, (int)ns);
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, t_scan, (const int32_t*)nullptr, (int32_t*)nullptr, (int)nd);
  size_t t_sort32 = 0;
  (void)sort_pairs_no_scratch(nullptr, t_sort32, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                           (const int32_t*)nullptr, (int32_t*)nullptr, (int)ns);
  if (t_sort32 > t_sort) t_sort = t_sort32;
  l.temp_bytes = t_sort > t_scan ? t_sort : t_scan;
  l.temp = take(l.temp_bytes);
  l.total = off;
  return l;
}

}  // namespace

}  // namespace se3

using namespace se3;

extern "C" int se3_compute_keys(const float* pts, const int32_t* batch_ids, const float* aabb_min,
                                const int32_t* num_cells, const float* cell_size, int64_t n, int64_t* keys,
                                void* stream) {
  if (n < 0 || (n > 0 && (!pts || !batch_ids || !aabb_min || !num_cells || !cell_size || !keys)))
    return SE3_ERR_INVALID_ARGUMENT;
  if (n == 0) return SE3_OK;
  hipLaunchKernelGGL(compute_keys_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, pts, batch_ids,
                     aabb_min, num_cells, cell_size, 0.f, n, keys, (int32_t*)nullptr);
  return check_launch();
}

static int batch_aabb_impl(const float* pts, const int32_t* batch_ids, int64_t n, int32_t n_batches, float* aabb_min,
                           float* aabb_max, int32_t* num_cells_to_zero, hipStream_t stream) {
  hipLaunchKernelGGL(batch_aabb_init_kernel, dim3((n_batches * 3 + 255) / 256), dim3(256), 0, stream, aabb_min, aabb_max,
                     n_batches * 3, num_cells_to_zero);
  if (n > 0) {
    int64_t blocks = (n + 1023) / 1024;  // 4 groups of 64 points per wavefront before it issues its 6 atomics
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(batch_aabb_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, pts, batch_ids, n, aabb_min,
                       aabb_max);
  }
  return check_launch();
}

extern "C" int se3_batch_aabb(const float* pts, const int32_t* batch_ids, int64_t n, int32_t n_batches, float* aabb_min,
                              float* aabb_max, void* stream_) {
  if (n < 0 || n_batches < 1) return SE3_ERR_INVALID_ARGUMENT;
  if (!aabb_min || !aabb_max || (n > 0 && (!pts || !batch_ids))) return SE3_ERR_INVALID_ARGUMENT;
  return batch_aabb_impl(pts, batch_ids, n, n_batches, aabb_min, aabb_max, nullptr, (hipStream_t)stream_);
}

namespace {
// grid parameters: min' = min - 1e-6, max' = max + max_shift, cells[d] = max over batches of
// int((max' - min') / cell) + 1; empty batches do not vote.  max_shift = -1e-6 in BallQuery.py:34-38,
// +1e-6 in BoundingBox.py:17-18 (grid sub-sampling, Grid.py:28-29).
__global__ void grid_params_kernel(float* __restrict__ mn, const float* __restrict__ mx, int n_batches, float cell,
                                   float max_shift, int32_t* __restrict__ num_cells) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_batches * 3) return;
  const float lo = mn[i], hi = mx[i];
  const float lo_s = __fsub_rn(lo, 1e-6f), hi_s = __fadd_rn(hi, max_shift);
  if (lo <= hi) atomicMax(&num_cells[i % 3], (int)__fdiv_rn(__fsub_rn(hi_s, lo_s), cell) + 1);
  mn[i] = lo_s;
}
// the same in one launch from boxes that are already known (one block: n_batches * 3 values)
__global__ __launch_bounds__(256) void grid_params_from_box_kernel(const float* __restrict__ mn_raw, const float* __restrict__ mx_raw,
                                                                   int n_batches, float cell, float max_shift,
                                                                   float* __restrict__ mn_out, int32_t* __restrict__ num_cells) {
  __shared__ int cells[3];
  if (threadIdx.x < 3) cells[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_batches * 3; i += blockDim.x) {
    const float lo = mn_raw[i], hi = mx_raw[i];
    const float lo_s = __fsub_rn(lo, 1e-6f), hi_s = __fadd_rn(hi, max_shift);
    if (lo <= hi) atomicMax(&cells[i % 3], (int)__fdiv_rn(__fsub_rn(hi_s, lo_s), cell) + 1);
    mn_out[i] = lo_s;
  }
  __syncthreads();
  if (threadIdx.x < 3) num_cells[threadIdx.x] = cells[threadIdx.x];
}
}  // namespace

extern "C" int se3_ball_query_grid_from_box(const float* box_min, const float* box_max, int32_t n_batches, float radius,
                                            float* aabb_min, int32_t* num_cells, void* stream_) {
  if (n_batches < 1 || !(radius > 0.f) || !box_min || !box_max || !aabb_min || !num_cells) return SE3_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(grid_params_from_box_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream_, box_min, box_max, n_batches,
                     radius, -1e-6f, aabb_min, num_cells);
  return check_launch();
}

extern "C" int se3_ball_query_grid(const float* pts_src, const int32_t* batch_src, int64_t n_src, int32_t n_batches,
                                   float radius, float* aabb_min, float* aabb_max_scratch, int32_t* num_cells,
                                   void* stream_) {
  if (n_src < 0 || n_batches < 1 || !(radius > 0.f)) return SE3_ERR_INVALID_ARGUMENT;
  if (!aabb_min || !aabb_max_scratch || !num_cells) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  if (n_src > 0 && (!pts_src || !batch_src)) return SE3_ERR_INVALID_ARGUMENT;
  if (int rc = batch_aabb_impl(pts_src, batch_src, n_src, n_batches, aabb_min, aabb_max_scratch, num_cells, stream)) return rc;
  hipLaunchKernelGGL(grid_params_kernel, dim3((n_batches * 3 + 255) / 256), dim3(256), 0, stream, aabb_min,
                     aabb_max_scratch, n_batches, radius, -1e-6f, num_cells);
  return check_launch();
}


namespace {
// se3_knn_grid_params: one workgroup; thread b takes batch element b, b + 256, ...
__global__ __launch_bounds__(256) void knn_grid_params_kernel(const int32_t* __restrict__ batch_ids, int64_t n,
                                                              const float* __restrict__ box_min, const float* __restrict__ box_max,
                                                              int n_batches, int k, float cell_factor, float* __restrict__ aabb_min,
                                                              int32_t* __restrict__ num_cells, float* __restrict__ cell_size) {
  __shared__ float s_c[256], s_e[256];
  __shared__ int s_cells[3][256];
  __shared__ float s_cell;
  float c_best = 0.f, e_best = 0.f;
  for (int b = threadIdx.x; b < n_batches; b += 256) {
    // points of batch element b (ids sorted): upper bound - lower bound
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (batch_ids[mid] < b) lo = mid + 1; else hi = mid; }
    const int64_t first = lo;
    hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (batch_ids[mid] <= b) lo = mid + 1; else hi = mid; }
    const float cnt = fmaxf((float)(lo - first), 1.0f);
    float e[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) e[d] = fmaxf(box_max[b * 3 + d] - box_min[b * 3 + d], 0.f);
    const float e1 = fmaxf(e[0], fmaxf(e[1], e[2])), e3 = fminf(e[0], fminf(e[1], e[2]));
    const float e2 = e[0] + e[1] + e[2] - e1 - e3;
    const float c_vol = cbrtf((float)k * e1 * e2 * e3 / (4.19f * cnt));
    const float c_area = sqrtf((float)k * e1 * e2 / (3.14f * cnt));
    const float c_len = (float)k * e1 / (2.0f * cnt);
    c_best = fmaxf(c_best, fmaxf(c_vol, fmaxf(c_area, c_len)));
    e_best = fmaxf(e_best, e1);
  }
  s_c[threadIdx.x] = c_best, s_e[threadIdx.x] = e_best;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      s_c[threadIdx.x] = fmaxf(s_c[threadIdx.x], s_c[threadIdx.x + st]);
      s_e[threadIdx.x] = fmaxf(s_e[threadIdx.x], s_e[threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) s_cell = fmaxf(cell_factor * s_c[0], fmaxf(s_e[0] * 1e-6f, 1e-30f));
  __syncthreads();
  const float cell = s_cell;
  int cells[3] = {1, 1, 1};
  for (int b = threadIdx.x; b < n_batches; b += 256) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float mn = box_min[b * 3 + d] - 1e-6f;
      aabb_min[b * 3 + d] = mn;
      cells[d] = max(cells[d], (int)fminf((box_max[b * 3 + d] - mn) / cell, 1048576.0f) + 1);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) s_cells[d][threadIdx.x] = cells[d];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
#pragma unroll
      for (int d = 0; d < 3; ++d) s_cells[d][threadIdx.x] = max(s_cells[d][threadIdx.x], s_cells[d][threadIdx.x + st]);
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) {
    num_cells[threadIdx.x] = s_cells[threadIdx.x][0];
    cell_size[threadIdx.x] = cell;
  }
}
}  // namespace

extern "C" int se3_knn_grid_params(const int32_t* batch_ids, int64_t n, const float* box_min, const float* box_max,
                                   int32_t n_batches, int32_t k, float cell_factor, float* aabb_min, int32_t* num_cells,
                                   float* cell_size, void* stream) {
  if (n < 0 || n_batches < 1 || k < 1 || !(cell_factor > 0.f)) return SE3_ERR_INVALID_ARGUMENT;
  if (!box_min || !box_max || !aabb_min || !num_cells || !cell_size || (n > 0 && !batch_ids)) return SE3_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(knn_grid_params_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, batch_ids, n, box_min, box_max,
                     (int)n_batches, (int)k, cell_factor, aabb_min, num_cells, cell_size);
  return check_launch();
}

extern "C" size_t se3_knn_query_grid_workspace_bytes(int64_t n) { return knn_layout(n).total; }

extern "C" int se3_knn_query_grid(const float* pts, const int32_t* batch_ids, const float* aabb_min,
                                  const int32_t* num_cells, const float* cell_size, int64_t n, int32_t k, int32_t* out,
                                  void* workspace, size_t workspace_bytes, void* stream_) {
  if (n < 0 || k < 1) return SE3_ERR_INVALID_ARGUMENT;
  if (k > 32 || n >= (1ll << 31)) return SE3_ERR_UNSUPPORTED;
  if (n == 0) return SE3_OK;
  if (!pts || !batch_ids || !aabb_min || !num_cells || !cell_size || !out || !workspace) return SE3_ERR_INVALID_ARGUMENT;
  const KnnLayout l = knn_layout(n);
  if (workspace_bytes < l.total) return SE3_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)workspace;
  int64_t* keys = (int64_t*)(ws + l.keys);
  int64_t* skeys = (int64_t*)(ws + l.skeys);
  int32_t* ids = (int32_t*)(ws + l.ids);
  int32_t* sids = (int32_t*)(ws + l.sids);
  float4* spts = (float4*)(ws + l.spts);
  int32_t* list = (int32_t*)(ws + l.list);
  int32_t* list_count = (int32_t*)(ws + l.count);
  size_t temp_bytes = l.temp_bytes;
  if (int rc = se3::launch_fill_words(list_count, 0u, 1, stream)) return rc;
  {
    ProfScope prof("knn_sort", stream);
    hipLaunchKernelGGL(compute_keys_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, pts, batch_ids, aabb_min, num_cells,
                       cell_size, 0.f, n, keys, ids);
    if (sort_pairs_no_scratch(ws + l.temp, temp_bytes, keys, skeys, ids, sids, (int)n, 0, 64, stream) !=
        hipSuccess)
      return SE3_ERR_LAUNCH;
    hipLaunchKernelGGL(gather_sorted_points_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, pts, sids, n, spts);
  }
  {
    ProfScope prof("knn_cells", stream);
    const dim3 grid((unsigned)((n * kKnnGroup + 255) / 256));
    if (k <= 8)
      hipLaunchKernelGGL(knn_grid_kernel<8>, grid, dim3(256), 0, stream, spts, skeys, num_cells, cell_size, (int)n, (int)k, out, list, list_count);
    else if (k <= 16)
      hipLaunchKernelGGL(knn_grid_kernel<16>, grid, dim3(256), 0, stream, spts, skeys, num_cells, cell_size, (int)n, (int)k, out, list, list_count);
    else
      hipLaunchKernelGGL(knn_grid_kernel<32>, grid, dim3(256), 0, stream, spts, skeys, num_cells, cell_size, (int)n, (int)k, out, list, list_count);
    if (int rc = check_launch()) return rc;
  }
  ProfScope prof("knn_fallback", stream);
  // exact fallback for the queries the 27-cell block could not settle (sparse regions, cloud boundary)
  return launch_knn_listed(pts, batch_ids, n, (int)k, out, list, list_count, stream);
}

extern "C" int se3_ball_query_needs_grid(int64_t n_src) { return n_src > kBqScanAllMax ? 1 : 0; }

extern "C" size_t se3_ball_query_workspace_bytes(int64_t n_src, int64_t n_dst) {
  return bq_layout(n_src, n_dst).total;
}

// skip_scan: the all-pairs path of the bounded call with few samples leaves the per-sample counts in the workspace and
// lets the store kernel form the offsets itself (scan_all_kernel<3>)
// grid (may be NULL = inside the workspace): the source cloud's part of the layout -- keys, sorted keys / ids / records --
// in a buffer of its own that outlives the call; grid_valid: it already holds this source cloud's grid for this radius
// (built by an earlier call with the same pts_src / batch_src / aabb_min / num_cells / radius / key width), so the key,
// sort and gather launches are skipped.
static int ball_query_count_impl(const float* pts_src, const float* pts_dst, const int32_t* batch_src,
                                 const int32_t* batch_dst, const float* aabb_min, const int32_t* num_cells,
                                 float radius, int64_t n_src, int64_t n_dst, void* workspace,
                                 size_t workspace_bytes, int32_t* ends, bool skip_scan, int key_bits, void* stream_,
                                 void* grid = nullptr, bool grid_valid = false) {
  if (n_src < 0 || n_dst < 0 || !(radius > 0.f)) return SE3_ERR_INVALID_ARGUMENT;
  if (n_src >= (1ll << 31) || n_dst >= (1ll << 31) / 9) return SE3_ERR_UNSUPPORTED;
  if (n_dst == 0) return SE3_OK;
  const bool scan_all = n_src <= kBqScanAllMax;
  if (!pts_dst || !batch_dst || !workspace || !ends || (n_src > 0 && (!pts_src || !batch_src)) ||
      (!scan_all && (!aabb_min || !num_cells)))
    return SE3_ERR_INVALID_ARGUMENT;
  const BqLayout l = bq_layout(n_src, n_dst);
  if (workspace_bytes < l.total) return SE3_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)workspace;
  char* gws = grid ? (char*)grid : ws;
  if (scan_all) {
    int32_t* counts = (int32_t*)(ws + l.counts);
    size_t temp_bytes = l.temp_bytes;
    const int64_t blocks = std::max((n_dst + 3) / 4, (n_src + 255) / 256);
    hipLaunchKernelGGL(scan_all_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, pts_src, batch_src,
                       (float4*)(gws + l.spts), pts_dst, batch_dst, 1.0f / radius, (int)n_src, n_dst, counts,
                       (int32_t*)nullptr, (int32_t*)nullptr, 0, (int32_t*)nullptr, (int32_t*)nullptr);
    if (skip_scan) return check_launch();
    if (hipcub::DeviceScan::InclusiveSum(ws + l.temp, temp_bytes, counts, ends, (int)n_dst, stream) != hipSuccess)
      return SE3_ERR_LAUNCH;
    return check_launch();
  }
  int64_t* keys = (int64_t*)(gws + l.keys);
  int64_t* skeys = (int64_t*)(gws + l.skeys);
  int32_t* ids = (int32_t*)(gws + l.ids);
  int32_t* sids = (int32_t*)(gws + l.sids);
  float4* spts = (float4*)(gws + l.spts);
  int2* ranges = (int2*)(ws + l.ranges);
  int32_t* counts = (int32_t*)(ws + l.counts);
  size_t temp_bytes = l.temp_bytes;
  const bool build = n_src > 0 && !grid_valid;
  // A cloud against itself: the samples are walked in the cell order the sort just produced (`sids`), so that
  // neighbouring threads search for neighbouring keys and the wavefronts of a workgroup read the same candidate
  // windows.  Results are stored at the sample's own index: nothing changes but the order of the work.
  const int32_t* order = (pts_src == pts_dst && n_src == n_dst && batch_src == batch_dst) ? sids : nullptr;

  // key_bits > 0 (the bounded call, which knows the batch count): 32-bit keys with a fixed cell stride, see key32_of
  if (key_bits > 0) {
    uint32_t* keys32 = (uint32_t*)keys;
    uint32_t* skeys32 = (uint32_t*)skeys;
    if (build) {
      hipLaunchKernelGGL(compute_keys32_kernel, dim3(blocks_for(n_src)), dim3(256), 0, stream, pts_src, batch_src,
                         aabb_min, num_cells, radius, n_src, keys32, ids);
      if (sort_pairs_no_scratch(ws + l.temp, temp_bytes, keys32, skeys32, ids, sids, (int)n_src, 0, key_bits,
                                             stream) != hipSuccess)
        return SE3_ERR_LAUNCH;
      hipLaunchKernelGGL(gather_sorted_points_kernel, dim3(blocks_for(n_src)), dim3(256), 0, stream, pts_src, sids,
                         n_src, spts);
    }
    hipLaunchKernelGGL(find_ranges32_kernel, dim3(blocks_for(n_dst * 9)), dim3(256), 0, stream, pts_dst, batch_dst,
                       aabb_min, num_cells, radius, skeys32, (int)n_src, n_dst, ranges, order);
  } else {
    if (build) {
      // cell size = radius in every dimension (BallQuery.py:39-40)
      hipLaunchKernelGGL(compute_keys_kernel, dim3(blocks_for(n_src)), dim3(256), 0, stream, pts_src, batch_src, aabb_min,
                         num_cells, (const float*)nullptr, radius, n_src, keys, ids);
      if (sort_pairs_no_scratch(ws + l.temp, temp_bytes, keys, skeys, ids, sids, (int)n_src, 0, 64,
                                             stream) != hipSuccess)
        return SE3_ERR_LAUNCH;
      hipLaunchKernelGGL(gather_sorted_points_kernel, dim3(blocks_for(n_src)), dim3(256), 0, stream, pts_src, sids, n_src,
                         spts);
    }
    hipLaunchKernelGGL(find_ranges_kernel, dim3(blocks_for(n_dst * 9)), dim3(256), 0, stream, pts_dst, batch_dst,
                       aabb_min, num_cells, radius, skeys, (int)n_src, n_dst, ranges, order);
  }
  hipLaunchKernelGGL(scan_candidates_kernel<0>, dim3((unsigned)((n_dst + 3) / 4)), dim3(256), 0, stream, pts_dst,
                     1.0f / radius, spts, ranges, n_dst, counts, (int32_t*)nullptr, (int32_t*)nullptr, 0, (int32_t*)nullptr,
                     (int32_t*)nullptr, order);
  temp_bytes = l.temp_bytes;
  if (hipcub::DeviceScan::InclusiveSum(ws + l.temp, temp_bytes, counts, ends, (int)n_dst, stream) != hipSuccess)
    return SE3_ERR_LAUNCH;
  return check_launch();
}

extern "C" int se3_ball_query_count(const float* pts_src, const float* pts_dst, const int32_t* batch_src,
                                    const int32_t* batch_dst, const float* aabb_min, const int32_t* num_cells,
                                    float radius, int64_t n_src, int64_t n_dst, void* workspace,
                                    size_t workspace_bytes, int32_t* ends, void* stream) {
  return ball_query_count_impl(pts_src, pts_dst, batch_src, batch_dst, aabb_min, num_cells, radius, n_src, n_dst,
                               workspace, workspace_bytes, ends, false, 0, stream);
}

// mode 1: two-phase store; 2: bounded (clamp, info, sources); 3: bounded all-pairs with the offsets formed in the kernel
static int ball_query_store_impl(const float* pts_dst, const int32_t* batch_dst, float radius, int64_t n_src,
                                 int64_t n_dst, const void* workspace, size_t workspace_bytes, int32_t* ends,
                                 int32_t* neighbors, int limit, int mode, int32_t* sources, int32_t* info, bool ordered,
                                 void* stream, const void* grid = nullptr) {
  const BqLayout l = bq_layout(n_src, n_dst);
  if (workspace_bytes < l.total) return SE3_ERR_WORKSPACE;
  const char* ws = (const char*)workspace;
  const char* gws = grid ? (const char*)grid : ws;
  const dim3 wgrid((unsigned)((n_dst + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (n_src <= kBqScanAllMax) {  // the count phase took the all-pairs path (and left the source records)
    if (!batch_dst) return SE3_ERR_INVALID_ARGUMENT;
    float4* recs = (float4*)(gws + l.spts);
    int32_t* counts = (int32_t*)(ws + l.counts);
#define SE3_SCAN_ALL(M)                                                                                                 \
  hipLaunchKernelGGL(scan_all_kernel<M>, wgrid, block, 0, st, (const float*)nullptr, (const int32_t*)nullptr, recs, pts_dst, \
                     batch_dst, 1.0f / radius, (int)n_src, n_dst, counts, ends, neighbors, limit, sources, info)
    if (mode == 1) SE3_SCAN_ALL(1);
    else if (mode == 2) SE3_SCAN_ALL(2);
    else SE3_SCAN_ALL(3);
#undef SE3_SCAN_ALL
    return check_launch();
  }
  if (mode == 1)
    hipLaunchKernelGGL(scan_candidates_kernel<1>, wgrid, block, 0, st, pts_dst, 1.0f / radius, (const float4*)(gws + l.spts),
                       (const int2*)(ws + l.ranges), n_dst, (int32_t*)nullptr, ends, neighbors, limit, sources, info,
                       ordered ? (const int32_t*)(gws + l.sids) : (const int32_t*)nullptr);
  else
    hipLaunchKernelGGL(scan_candidates_kernel<2>, wgrid, block, 0, st, pts_dst, 1.0f / radius, (const float4*)(gws + l.spts),
                       (const int2*)(ws + l.ranges), n_dst, (int32_t*)nullptr, ends, neighbors, limit, sources, info,
                       ordered ? (const int32_t*)(gws + l.sids) : (const int32_t*)nullptr);
  return check_launch();
}

extern "C" int se3_ball_query_store(const float* pts_dst, const int32_t* batch_dst, float radius, int64_t n_src,
                                    int64_t n_dst, const void* workspace, size_t workspace_bytes,
                                    const int32_t* ends, int64_t n_edges, int32_t* neighbors, void* stream) {
  if (n_src < 0 || n_dst < 0 || n_edges < 0 || !(radius > 0.f)) return SE3_ERR_INVALID_ARGUMENT;
  if (n_dst == 0 || n_edges == 0) return SE3_OK;
  if (!pts_dst || !workspace || !ends || !neighbors) return SE3_ERR_INVALID_ARGUMENT;
  return ball_query_store_impl(pts_dst, batch_dst, radius, n_src, n_dst, workspace, workspace_bytes,
                               const_cast<int32_t*>(ends), neighbors, 0x7fffffff, 1, nullptr, nullptr, false, stream);
}

static int ball_query_bounded_impl(const float* pts_src, const float* pts_dst, const int32_t* batch_src,
                                   const int32_t* batch_dst, const float* aabb_min, const int32_t* num_cells,
                                   float radius, int64_t n_src, int64_t n_dst, int32_t n_batches, void* workspace,
                                   size_t workspace_bytes, int64_t capacity, int32_t* neighbors, int32_t* sources,
                                   int32_t* ends, int32_t* info, void* stream, void* grid, bool grid_valid) {
  if (capacity < 0 || capacity >= (1ll << 31) || !info) return SE3_ERR_INVALID_ARGUMENT;
  if (n_dst == 0) return se3::launch_fill_words(info, 0u, 2, (hipStream_t)stream);
  if (capacity > 0 && !neighbors) return SE3_ERR_INVALID_ARGUMENT;
  const bool inline_prefix = n_src <= kBqScanAllMax && n_dst <= kBqInlinePrefixMax;
  // one or two batch elements: 30 + 1 key bits (key32_of; the window's upper bound base + 3 then cannot wrap);
  // more: the 64-bit keys of the two-phase path
  int key_bits = 0;
  if (n_batches >= 1 && n_batches <= 2) key_bits = 30 + (n_batches > 1 ? 1 : 0);
  // (count + prefix + store as one launch with a decoupled look-back was measured in round 4: slower, removed --
  // profiles/r04_ball_query_onepass_ab.txt)
  if (int rc = ball_query_count_impl(pts_src, pts_dst, batch_src, batch_dst, aabb_min, num_cells, radius, n_src, n_dst,
                                     workspace, workspace_bytes, ends, inline_prefix, key_bits, stream, grid, grid_valid))
    return rc;
  // one store launch also clamps the offsets to the buffer and records total + overflow flag
  return ball_query_store_impl(pts_dst, batch_dst, radius, n_src, n_dst, workspace, workspace_bytes, ends, neighbors,
                               (int)capacity, inline_prefix ? 3 : 2, sources, info,
                               pts_src == pts_dst && n_src == n_dst && batch_src == batch_dst, stream, grid);
}

extern "C" int se3_ball_query_bounded(const float* pts_src, const float* pts_dst, const int32_t* batch_src,
                                      const int32_t* batch_dst, const float* aabb_min, const int32_t* num_cells,
                                      float radius, int64_t n_src, int64_t n_dst, int32_t n_batches, void* workspace,
                                      size_t workspace_bytes, int64_t capacity, int32_t* neighbors, int32_t* sources,
                                      int32_t* ends, int32_t* info, void* stream) {
  return ball_query_bounded_impl(pts_src, pts_dst, batch_src, batch_dst, aabb_min, num_cells, radius, n_src, n_dst, n_batches,
                                 workspace, workspace_bytes, capacity, neighbors, sources, ends, info, stream, nullptr, false);
}

extern "C" size_t se3_ball_query_grid_bytes(int64_t n_src) { return bq_layout(n_src, 0).ranges; }

extern "C" int se3_ball_query_bounded_shared(const float* pts_src, const float* pts_dst, const int32_t* batch_src,
                                             const int32_t* batch_dst, const float* aabb_min, const int32_t* num_cells,
                                             float radius, int64_t n_src, int64_t n_dst, int32_t n_batches, void* grid,
                                             size_t grid_bytes, int32_t grid_valid, void* workspace, size_t workspace_bytes,
                                             int64_t capacity, int32_t* neighbors, int32_t* sources, int32_t* ends,
                                             int32_t* info, void* stream) {
  if (!grid || grid_bytes < bq_layout(n_src, 0).ranges) return SE3_ERR_WORKSPACE;
  return ball_query_bounded_impl(pts_src, pts_dst, batch_src, batch_dst, aabb_min, num_cells, radius, n_src, n_dst, n_batches,
                                 workspace, workspace_bytes, capacity, neighbors, sources, ends, info, stream, grid,
                                 grid_valid != 0);
}

// ---- source-major copy of an edge list (se3_csr_transpose*) -----------------------------------------------------------
// Nothing here may become a MEMSET NODE of a captured graph: on the HIP runtime PyTorch 2.10 ships (7.0.51831, RCCL 2.26.6)
// a graph with hipMemsetAsync nodes faults on replay once an RCCL collective has run between two replays.  Found in round 4
// with rocPRIM's one-sweep radix sort (hipcub::DeviceRadixSort above 1 M items: three hipMemsetAsync per pass), which this
// transposition used until then -- the captured level 1 -> 0 convolution next to a live communicator faulted in 7 of 7
// runs, at the first replay behind the first barrier; the same graph with a merge sort (no memset) was clean, a counting
// form with two hipMemsetAsync faulted again, the same form zeroing its arrays by a kernel is clean
// (tools/debug_up_graph.py, tools/fault_bisect_memset.sh).  The one-sweep kernels also use scratch
// memory (80 bytes per lane), the only kernels of this file that do; sorts stay on scratch-free forms too.
// The list arrives grouped by sample in ascending sample order and a sample lists a source at most once, so "stable sort
// by source" = per source the ascending list of its samples: count per source (atomics), inclusive scan, scatter into
// the source's segment in arrival order (atomic cursor), then every segment is put in ascending order by ranking its
// entries against each other -- the result does not depend on the order the atomics were served in.
// Cost (rocprofv3, 2 M edges): short segments (a down-convolution, ~4 per source) 70 us in all; few sources with long
// segments (an up-convolution, 9 k sources x 220) 260 us -- scattered atomics onto a few hundred cache lines and the
// ranking of long segments -- against 90 us for the radix sort it replaces.  A variant with per-wavefront counters in
// LDS and atomic-free, stable slot hand-out was built and measured no faster (285 us: an [slices x sources] count matrix
// larger than the list itself).  For ball-query neighbourhoods the host side therefore builds that case's source-major
// list as a second ball query with the clouds' roles swapped (pc.BQNeighborhood.source_major: the predicate is symmetric
// bit for bit), which costs what the forward query costs.
namespace {
struct TrLayout {
  size_t tmp, tmp_ids, cursor, temp, temp_bytes, total;  // counting form (n_src <= 2 * rows)
  size_t src, smp, merge_temp, merge_bytes;           // merge-sort form (sparser graphs): keys / values sorted in place
};
struct TrLess {
  __device__ __forceinline__ bool operator()(const int32_t& a, const int32_t& b) const { return a < b; }
};
TrLayout tr_layout(int64_t e) {
  TrLayout l{};
  const size_t ne = (size_t)(e > 0 ? e : 1);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = se3::align_up(off + bytes, 256); return o; };
  l.tmp = take(ne * 4);
  l.tmp_ids = take(ne * 4);
  l.cursor = take(2 * ne * 4);
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, l.temp_bytes, (const int32_t*)nullptr, (int32_t*)nullptr, (int)(2 * ne));
  l.temp = take(l.temp_bytes);
  const size_t counting_total = off;
  off = 0;
  l.src = take(ne * 4);
  l.smp = take(ne * 4);
  (void)hipcub::DeviceMergeSort::StableSortPairs(nullptr, l.merge_bytes, (int32_t*)nullptr, (int32_t*)nullptr, (int)ne, TrLess());
  l.merge_temp = take(l.merge_bytes);
  l.total = off > counting_total ? off : counting_total;
  return l;
}

__device__ __forceinline__ int64_t valid_rows(const int32_t* n_valid, int64_t e) {
  return n_valid ? (int64_t)max(min((int64_t)*n_valid, e), (int64_t)0) : e;
}

// (a kernel, not hipMemsetAsync: see the note on memset nodes above se3_csr_transpose_bounded)
__global__ void tr_zero_kernel(int32_t* __restrict__ a, int32_t* __restrict__ b, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    a[i] = 0;
    if (b) b[i] = 0;
  }
}

// counts[p] += 1 for every edge into source p (ids outside [0, n_src) are not edges of this graph: skipped everywhere)
__global__ void tr_count_kernel(const int32_t* __restrict__ neighbors, int64_t e, const int32_t* __restrict__ n_valid,
                                int32_t n_src, int32_t* __restrict__ counts) {
  const int64_t valid = valid_rows(n_valid, e);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < valid; i += (int64_t)gridDim.x * blockDim.x) {
    const int32_t p = neighbors[i * 2 + 1];
    if ((uint32_t)p < (uint32_t)n_src) atomicAdd(&counts[p], 1);
  }
}

// edge i goes to the next free slot of its source's segment; rows past the list (the unset tail of a bounded buffer) are zeroed
__global__ void tr_scatter_kernel(const int32_t* __restrict__ neighbors, int64_t e, const int32_t* __restrict__ n_valid,
                                  int32_t n_src, const int32_t* __restrict__ ends, int32_t* __restrict__ cursor,
                                  int32_t* __restrict__ tmp, int32_t* __restrict__ tmp_ids, int32_t* __restrict__ t_samples,
                                  int32_t* __restrict__ t_edge_ids) {
  const int64_t valid = valid_rows(n_valid, e);
  const int64_t total = n_src > 0 ? ends[n_src - 1] : 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
    if (i >= total) {  // nothing of the result is left unset
      t_samples[i] = 0;
      if (t_edge_ids) t_edge_ids[i] = 0;
    }
    if (i >= valid) continue;
    const int32_t p = neighbors[i * 2 + 1];
    if ((uint32_t)p >= (uint32_t)n_src) continue;
    const int32_t base = p > 0 ? ends[p - 1] : 0;
    const int32_t slot = base + atomicAdd(&cursor[p], 1);
    tmp[slot] = neighbors[i * 2];
    tmp_ids[slot] = (int32_t)i;
  }
}

// One wavefront per source: its segment of `tmp` in ascending order -> t_samples.  rank = number of entries that sort in
// front of mine (ties, which a well-formed list does not have, by position).  Segments of up to 64 entries live in
// registers and are compared through shuffles; up to kTrSegLds entries are staged in a wave-private piece of LDS and every
// lane ranks its entries against broadcast reads, four at a time; longer ones are read back from memory (L^2 / 64 steps).
constexpr int kTrSegLds = 2048;
__global__ __launch_bounds__(256) void tr_segment_sort_kernel(const int32_t* __restrict__ tmp, const int32_t* __restrict__ tmp_ids,
                                                              const int32_t* __restrict__ ends, int64_t n_src,
                                                              int32_t* __restrict__ t_samples, int32_t* __restrict__ t_edge_ids) {
  __shared__ __attribute__((aligned(16))) int32_t seg[4][kTrSegLds];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t p = (int64_t)blockIdx.x * 4 + wave; p < n_src; p += (int64_t)gridDim.x * 4) {
    const int start = p > 0 ? ends[p - 1] : 0;
    const int len = ends[p] - start;
    if (len <= 1) {
      if (len == 1 && lane == 0) {
        t_samples[start] = tmp[start];
        if (t_edge_ids) t_edge_ids[start] = tmp_ids[start];
      }
      continue;
    }
    if (len <= 64) {
      const int v = lane < len ? tmp[start + lane] : 0x7fffffff;
      int rank = 0;
      for (int j = 0; j < len; ++j) {
        const int vj = __shfl(v, j);
        rank += (vj < v || (vj == v && j < lane)) ? 1 : 0;
      }
      if (lane < len) {
        t_samples[start + rank] = v;
        if (t_edge_ids) t_edge_ids[start + rank] = tmp_ids[start + lane];
      }
      continue;
    }
    if (len <= kTrSegLds) {
      const int len4 = (len + 3) & ~3;
      for (int i = lane; i < len4; i += 64) seg[wave][i] = i < len ? tmp[start + i] : 0x7fffffff;  // the pad sorts behind everything
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (int i = lane; i < len; i += 64) {
        const int v = seg[wave][i];
        int rank = 0;
        for (int j = 0; j < len4; j += 4) {
          const int4 q = *reinterpret_cast<const int4*>(&seg[wave][j]);
          rank += (q.x < v || (q.x == v && j < i)) ? 1 : 0;
          rank += (q.y < v || (q.y == v && j + 1 < i)) ? 1 : 0;
          rank += (q.z < v || (q.z == v && j + 2 < i)) ? 1 : 0;
          rank += (q.w < v || (q.w == v && j + 3 < i)) ? 1 : 0;
        }
        t_samples[start + rank] = v;
        if (t_edge_ids) t_edge_ids[start + rank] = tmp_ids[start + i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // the next source of this wavefront reuses the piece
      continue;
    }
    for (int i = lane; i < len; i += 64) {
      const int v = tmp[start + i];
      int rank = 0;
      for (int j = 0; j < len; ++j) {
        const int vj = tmp[start + j];
        rank += (vj < v || (vj == v && j < i)) ? 1 : 0;
      }
      t_samples[start + rank] = v;
      if (t_edge_ids) t_edge_ids[start + rank] = tmp_ids[start + i];
    }
  }
}

// merge-sort form: the sorted values are edge indices -> the samples of those edges (rows behind the list: zeros)
__global__ void tr_samples_of_ids_kernel(const int32_t* __restrict__ neighbors, const int32_t* __restrict__ sorted_src,
                                         const int32_t* __restrict__ sorted_ids, int64_t e, int32_t n_src,
                                         int32_t* __restrict__ t_samples, int32_t* __restrict__ t_edge_ids) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
    const bool ok = (uint32_t)sorted_src[i] < (uint32_t)n_src;
    const int32_t id = ok ? sorted_ids[i] : 0;
    t_samples[i] = ok ? neighbors[(int64_t)id * 2] : 0;
    if (t_edge_ids) t_edge_ids[i] = id;
  }
}

}  // namespace

extern "C" size_t se3_csr_transpose_workspace_bytes(int64_t n_edges) { return tr_layout(n_edges).total; }

extern "C" int se3_csr_transpose_bounded(const int32_t* neighbors, int64_t n_rows, const int32_t* n_valid, int64_t n_src,
                                         void* workspace, size_t workspace_bytes, int32_t* t_samples, int32_t* t_ends,
                                         int32_t* t_edge_ids, void* stream_) {
  const int64_t n_edges = n_rows;
  if (n_edges < 0 || n_src < 0) return SE3_ERR_INVALID_ARGUMENT;
  if (n_edges >= (1ll << 30) || n_src >= (1ll << 31) - 1) return SE3_ERR_UNSUPPORTED;
  if (n_src == 0) return SE3_OK;
  if (!t_ends || (n_edges > 0 && (!neighbors || !workspace || !t_samples))) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const TrLayout l = tr_layout(n_edges);
  if (n_edges > 0 && workspace_bytes < l.total) return SE3_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  if (n_edges == 0) {
    hipLaunchKernelGGL(tr_zero_kernel, dim3(blocks_for(n_src)), dim3(256), 0, stream, t_ends, (int32_t*)nullptr, n_src);
    return check_launch();
  }
  static const bool force_merge = getenv("SE3_TR_MERGE_SORT") != nullptr;  // A/B and test switch: the fallback form everywhere
  // The counting form ranks every source's segment by itself: up to kTrSegLds entries per segment in registers / LDS, longer
  // ones from memory at L^2 / 64 steps of ONE wavefront -- fine for the odd long segment, seconds for a list whose segments
  // are all that long (a coarsest-level up-convolution handed over as a plain list: 1e5 .. 1e6 edges per source).  Lists
  // averaging more than kTrSegLds / 2 edges per source take the merge-sort form, whose time does not depend on the segment
  // lengths (the list's LONGEST segment is only known on the device, so the average decides).
  if (n_src <= 2 * n_edges && n_edges <= (int64_t)(kTrSegLds / 2) * n_src && !force_merge) {
    int32_t* tmp = (int32_t*)(ws + l.tmp);
    int32_t* tmp_ids = (int32_t*)(ws + l.tmp_ids);
    int32_t* cursor = (int32_t*)(ws + l.cursor);
    hipLaunchKernelGGL(tr_zero_kernel, dim3(blocks_for(n_src)), dim3(256), 0, stream, t_ends, cursor, n_src);
    hipLaunchKernelGGL(tr_count_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, stream, neighbors, n_edges, n_valid,
                       (int32_t)n_src, t_ends);
    size_t temp_bytes = l.temp_bytes;  // sized for 2 * rows >= n_src items
    if (hipcub::DeviceScan::InclusiveSum(ws + l.temp, temp_bytes, t_ends, t_ends, (int)n_src, stream) != hipSuccess)
      return SE3_ERR_LAUNCH;
    hipLaunchKernelGGL(tr_scatter_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, stream, neighbors, n_edges, n_valid,
                       (int32_t)n_src, t_ends, cursor, tmp, tmp_ids, t_samples, t_edge_ids);
    hipLaunchKernelGGL(tr_segment_sort_kernel, dim3(blocks_for(n_src, 4)), dim3(256), 0, stream, tmp, tmp_ids, t_ends, n_src,
                       t_samples, t_edge_ids);
    return check_launch();
  }
  // more than two sources per row of the list: stable merge sort of (source, sample) pairs, in place (no scratch either)
  int32_t* src = (int32_t*)(ws + l.src);
  int32_t* smp = (int32_t*)(ws + l.smp);
  hipLaunchKernelGGL(split_edges_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, stream, neighbors, n_edges, n_valid,
                     (int32_t)n_src, src, smp);
  size_t merge_bytes = l.merge_bytes;
  if (hipcub::DeviceMergeSort::StableSortPairs(ws + l.merge_temp, merge_bytes, src, smp, (int)n_edges, TrLess(), stream) != hipSuccess)
    return SE3_ERR_LAUNCH;
  hipLaunchKernelGGL(tr_samples_of_ids_kernel, dim3(blocks_for(n_edges)), dim3(256), 0, stream, neighbors, src, smp, n_edges,
                     (int32_t)n_src, t_samples, t_edge_ids);
  hipLaunchKernelGGL(group_ends_kernel, dim3(blocks_for(n_src)), dim3(256), 0, stream, src, n_edges, n_src, t_ends);
  return check_launch();
}

extern "C" int se3_csr_transpose(const int32_t* neighbors, int64_t n_edges, int64_t n_src, void* workspace,
                                 size_t workspace_bytes, int32_t* t_samples, int32_t* t_ends, int32_t* t_edge_ids, void* stream) {
  return se3_csr_transpose_bounded(neighbors, n_edges, nullptr, n_src, workspace, workspace_bytes, t_samples, t_ends, t_edge_ids,
                                   stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Hierarchy build (scope row f-2): grid sub-sampling and the pool / up-sample maps between two levels.
//   reference: pc/Grid.py:20-51 (bounding box, cell counts, ComputeKeys, torch.unique(return_inverse)),
//              pc/GridSubSample.py:63-93 (scatter_mean / scatter_max over the cell ids, gather for up-sampling),
//              pc/PointHierarchy.py:40-51 (level points = cell means, level batch ids = cell max).
// Cells are numbered in ascending key order (what torch.unique returns); inside a cell the points keep their
// input order (stable sort), so every reduction below has a fixed summation order (the reference's is not defined).
// ---------------------------------------------------------------------------------------------------------------
namespace {

__global__ void cell_heads_kernel(const int64_t* __restrict__ skeys, int64_t n, int32_t* __restrict__ flags) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    flags[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1 : 0;
}

// ranks = inclusive scan of the head flags: position i of the sorted order belongs to cell ranks[i] - 1
__global__ void cell_scatter_kernel(const int32_t* __restrict__ sids, const int32_t* __restrict__ ranks,
                                    const int32_t* __restrict__ flags, int64_t n, int32_t* __restrict__ cell_ids,
                                    int32_t* __restrict__ cell_ends, int32_t* __restrict__ n_cells) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = ranks[i] - 1;
    cell_ids[sids[i]] = r;
    if (i == n - 1 || flags[i + 1]) cell_ends[r] = (int32_t)(i + 1);
    if (i == n - 1) *n_cells = r + 1;
  }
}

enum { kPoolAvg = 0, kPoolMax = 1, kPoolMin = 2, kPoolSum = 3 };

// out[cell, ch] = reduce over the cell's rows (sorted_ids[start .. end)) of src[row, ch]; thread = (cell, channel),
// channels fastest, so a cell's row is read with consecutive lanes.  n_cells_dev != nullptr: the cell count is still
// on the device (inside se3_grid_subsample); the grid is sized for the upper bound.
template <int MODE>
__global__ void segment_pool_kernel(const float* __restrict__ src, const int32_t* __restrict__ sorted_ids,
                                    const int32_t* __restrict__ cell_ends, int64_t n_cells, const int32_t* n_cells_dev,
                                    int c, float* __restrict__ out, int32_t* __restrict__ arg) {
  if (n_cells_dev) n_cells = *n_cells_dev;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_cells * c;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = idx / c;
    const int ch = (int)(idx - cell * c);
    const int start = cell > 0 ? cell_ends[cell - 1] : 0, end = cell_ends[cell];
    float acc = 0.f;
    int best = -1;
    for (int j = start; j < end; ++j) {
      const int id = sorted_ids[j];
      const float v = src[(int64_t)id * c + ch];
      if (MODE == kPoolAvg || MODE == kPoolSum) {
        acc += v;
      } else if (best < 0 || (MODE == kPoolMax ? v > acc : v < acc)) {  // first extremum wins (lowest row id)
        acc = v, best = id;
      }
    }
    if (MODE == kPoolAvg && end > start) acc = __fdiv_rn(acc, (float)(end - start));
    out[idx] = acc;
    if (arg) arg[idx] = best;
  }
}

// the maps from cells back to rows: MODE sum = plain gather (the up-sampling itself and the gradient of a sum),
// avg = gather / cell size, max / min = the gradient goes to the row that supplied the extremum
template <int MODE>
__global__ void segment_unpool_kernel(const float* __restrict__ cell_vals, const int32_t* __restrict__ cell_ids,
                                      const int32_t* __restrict__ cell_ends, const int32_t* __restrict__ arg, int64_t n,
                                      int c, float* __restrict__ out) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n * c; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / c;
    const int ch = (int)(idx - row * c);
    const int cell = cell_ids[row];
    float v = cell_vals[(int64_t)cell * c + ch];
    if (MODE == kPoolAvg) v = __fdiv_rn(v, (float)(cell_ends[cell] - (cell > 0 ? cell_ends[cell - 1] : 0)));
    if (MODE == kPoolMax || MODE == kPoolMin) v = arg[(int64_t)cell * c + ch] == (int32_t)row ? v : 0.f;
    out[idx] = v;
  }
}

__global__ void cell_batch_ids_kernel(const int32_t* __restrict__ batch_ids, const int32_t* __restrict__ sorted_ids,
                                      const int32_t* __restrict__ cell_ends, const int32_t* __restrict__ n_cells_dev,
                                      int32_t* __restrict__ out) {
  const int64_t n_cells = *n_cells_dev;
  for (int64_t cell = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; cell < n_cells; cell += (int64_t)gridDim.x * blockDim.x)
    out[cell] = batch_ids[sorted_ids[cell > 0 ? cell_ends[cell - 1] : 0]];  // the batch id is part of the key
}

// Frame pooling (scope row f-3, pc/PointcloudRotEquiv.py:224-251): rows point*F + frame -> one row per point
template <int MODE>
__global__ void frame_pool_kernel(const float* __restrict__ x, int64_t n_pts, int f, int c, float* __restrict__ out,
                                  int32_t* __restrict__ arg) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_pts * c; idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = idx / c;
    const int ch = (int)(idx - p * c);
    float acc = 0.f;
    int best = -1;
    for (int a = 0; a < f; ++a) {
      const float v = x[(p * f + a) * c + ch];
      if (MODE == kPoolAvg || MODE == kPoolSum) acc += v;
      else if (best < 0 || (MODE == kPoolMax ? v > acc : v < acc)) acc = v, best = a;
    }
    if (MODE == kPoolAvg) acc = __fdiv_rn(acc, (float)f);
    out[idx] = acc;
    if (arg) arg[idx] = best;
  }
}

template <int MODE>
__global__ void frame_unpool_kernel(const float* __restrict__ g, const int32_t* __restrict__ arg, int64_t n_pts, int f,
                                    int c, float* __restrict__ out) {
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_pts * f * c;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = idx / c;
    const int ch = (int)(idx - row * c);
    const int64_t p = row / f;
    const int a = (int)(row - p * f);
    float v = g[p * c + ch];
    if (MODE == kPoolAvg) v = __fdiv_rn(v, (float)f);
    if (MODE == kPoolMax || MODE == kPoolMin) v = arg[p * c + ch] == a ? v : 0.f;
    out[idx] = v;
  }
}

struct GsLayout {
  size_t box_min, box_max, num_cells, keys, skeys, ids, flags, ranks, temp, temp_bytes, total;
};

GsLayout gs_layout(int64_t n, int n_batches) {
  GsLayout l{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t nn = (size_t)(n > 0 ? n : 1), nb = (size_t)(n_batches > 0 ? n_batches : 1);
  l.box_min = take(nb * 12), l.box_max = take(nb * 12), l.num_cells = take(16);
  l.keys = take(nn * 8), l.skeys = take(nn * 8), l.ids = take(nn * 4), l.flags = take(nn * 4), l.ranks = take(nn * 4);
  size_t sort_bytes = 0, scan_bytes = 0;
  (void)sort_pairs_no_scratch(nullptr, sort_bytes, (int64_t*)nullptr, (int64_t*)nullptr, (int32_t*)nullptr,
                                           (int32_t*)nullptr, (int)nn);
  (void)hipcub::DeviceScan::InclusiveSum(nullptr, scan_bytes, (int32_t*)nullptr, (int32_t*)nullptr, (int)nn);
  l.temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
  l.temp = take(l.temp_bytes);
  l.total = off;
  return l;
}

}  // namespace

extern "C" size_t se3_grid_subsample_workspace_bytes(int64_t n, int32_t n_batches) { return gs_layout(n, n_batches).total; }

extern "C" int se3_grid_subsample(const float* pts, const int32_t* batch_ids, int64_t n, int32_t n_batches,
                                  float cell_size, void* workspace, size_t workspace_bytes, int32_t* cell_ids,
                                  int32_t* sorted_ids, int32_t* cell_ends, int32_t* n_cells, float* cell_pts,
                                  int32_t* cell_batch_ids, void* stream_) {
  if (n < 0 || n_batches < 1 || !(cell_size > 0.f)) return SE3_ERR_INVALID_ARGUMENT;
  if (n >= (1ll << 31) / 3) return SE3_ERR_UNSUPPORTED;
  if (!n_cells) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0) return se3::launch_fill_words(n_cells, 0u, 1, stream);
  if (!pts || !batch_ids || !workspace || !cell_ids || !sorted_ids || !cell_ends || !cell_pts || !cell_batch_ids)
    return SE3_ERR_INVALID_ARGUMENT;
  const GsLayout l = gs_layout(n, n_batches);
  if (workspace_bytes < l.total) return SE3_ERR_WORKSPACE;
  char* ws = (char*)workspace;
  float* box_min = (float*)(ws + l.box_min);
  float* box_max = (float*)(ws + l.box_max);
  int32_t* num_cells = (int32_t*)(ws + l.num_cells);
  int64_t* keys = (int64_t*)(ws + l.keys);
  int64_t* skeys = (int64_t*)(ws + l.skeys);
  int32_t* ids = (int32_t*)(ws + l.ids);
  int32_t* flags = (int32_t*)(ws + l.flags);
  int32_t* ranks = (int32_t*)(ws + l.ranks);
  if (int rc = batch_aabb_impl(pts, batch_ids, n, n_batches, box_min, box_max, num_cells, stream)) return rc;
  hipLaunchKernelGGL(grid_params_kernel, dim3((n_batches * 3 + 255) / 256), dim3(256), 0, stream, box_min, box_max,
                     n_batches, cell_size, 1e-6f, num_cells);
  hipLaunchKernelGGL(compute_keys_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, pts, batch_ids, box_min, num_cells,
                     (const float*)nullptr, cell_size, n, keys, ids);
  size_t temp_bytes = l.temp_bytes;
  if (sort_pairs_no_scratch(ws + l.temp, temp_bytes, keys, skeys, ids, sorted_ids, (int)n, 0, 64, stream) !=
      hipSuccess)
    return SE3_ERR_LAUNCH;
  hipLaunchKernelGGL(cell_heads_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, skeys, n, flags);
  temp_bytes = l.temp_bytes;
  if (hipcub::DeviceScan::InclusiveSum(ws + l.temp, temp_bytes, flags, ranks, (int)n, stream) != hipSuccess)
    return SE3_ERR_LAUNCH;
  hipLaunchKernelGGL(cell_scatter_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, sorted_ids, ranks, flags, n, cell_ids,
                     cell_ends, n_cells);
  // level points = cell means (first n_cells rows of cell_pts), level batch ids
  hipLaunchKernelGGL(segment_pool_kernel<kPoolAvg>, dim3(blocks_for(n * 3)), dim3(256), 0, stream, pts, sorted_ids,
                     cell_ends, (int64_t)0, n_cells, 3, cell_pts, (int32_t*)nullptr);
  hipLaunchKernelGGL(cell_batch_ids_kernel, dim3(blocks_for(n)), dim3(256), 0, stream, batch_ids, sorted_ids, cell_ends,
                     n_cells, cell_batch_ids);
  return check_launch();
}

extern "C" int se3_segment_pool(const float* src, const int32_t* sorted_ids, const int32_t* cell_ends, int64_t n_cells,
                                int32_t channels, int32_t mode, float* out, int32_t* arg, void* stream_) {
  if (n_cells < 0 || channels < 1 || mode < kPoolAvg || mode > kPoolSum) return SE3_ERR_INVALID_ARGUMENT;
  if (n_cells == 0) return SE3_OK;
  if (!src || !sorted_ids || !cell_ends || !out) return SE3_ERR_INVALID_ARGUMENT;
  if ((mode == kPoolMax || mode == kPoolMin) && !arg) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid(blocks_for(n_cells * channels)), block(256);
#define SE3_POOL(M)                                                                                                   \
  hipLaunchKernelGGL(segment_pool_kernel<M>, grid, block, 0, stream, src, sorted_ids, cell_ends, n_cells,                 \
                     (const int32_t*)nullptr, channels, out, arg)
  if (mode == kPoolAvg) SE3_POOL(kPoolAvg);
  else if (mode == kPoolMax) SE3_POOL(kPoolMax);
  else if (mode == kPoolMin) SE3_POOL(kPoolMin);
  else SE3_POOL(kPoolSum);
#undef SE3_POOL
  return check_launch();
}

extern "C" int se3_segment_unpool(const float* cell_vals, const int32_t* cell_ids, const int32_t* cell_ends,
                                  const int32_t* arg, int64_t n, int32_t channels, int32_t mode, float* out,
                                  void* stream_) {
  if (n < 0 || channels < 1 || mode < kPoolAvg || mode > kPoolSum) return SE3_ERR_INVALID_ARGUMENT;
  if (n == 0) return SE3_OK;
  if (!cell_vals || !cell_ids || !out || (mode == kPoolAvg && !cell_ends)) return SE3_ERR_INVALID_ARGUMENT;
  if ((mode == kPoolMax || mode == kPoolMin) && !arg) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid(blocks_for(n * channels)), block(256);
#define SE3_UNPOOL(M) \
  hipLaunchKernelGGL(segment_unpool_kernel<M>, grid, block, 0, stream, cell_vals, cell_ids, cell_ends, arg, n, channels, out)
  if (mode == kPoolAvg) SE3_UNPOOL(kPoolAvg);
  else if (mode == kPoolMax) SE3_UNPOOL(kPoolMax);
  else if (mode == kPoolMin) SE3_UNPOOL(kPoolMin);
  else SE3_UNPOOL(kPoolSum);
#undef SE3_UNPOOL
  return check_launch();
}

extern "C" int se3_frame_pool(const float* x, int64_t n_points, int32_t frames, int32_t channels, int32_t mode, float* out,
                              int32_t* arg, void* stream_) {
  if (n_points < 0 || frames < 1 || channels < 1 || mode < kPoolAvg || mode > kPoolSum) return SE3_ERR_INVALID_ARGUMENT;
  if (n_points == 0) return SE3_OK;
  if (!x || !out || ((mode == kPoolMax || mode == kPoolMin) && !arg)) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid(blocks_for(n_points * channels)), block(256);
#define SE3_FPOOL(M) hipLaunchKernelGGL(frame_pool_kernel<M>, grid, block, 0, stream, x, n_points, frames, channels, out, arg)
  if (mode == kPoolAvg) SE3_FPOOL(kPoolAvg);
  else if (mode == kPoolMax) SE3_FPOOL(kPoolMax);
  else if (mode == kPoolMin) SE3_FPOOL(kPoolMin);
  else SE3_FPOOL(kPoolSum);
#undef SE3_FPOOL
  return check_launch();
}

extern "C" int se3_frame_unpool(const float* grad_out, const int32_t* arg, int64_t n_points, int32_t frames,
                                int32_t channels, int32_t mode, float* grad_x, void* stream_) {
  if (n_points < 0 || frames < 1 || channels < 1 || mode < kPoolAvg || mode > kPoolSum) return SE3_ERR_INVALID_ARGUMENT;
  if (n_points == 0) return SE3_OK;
  if (!grad_out || !grad_x || ((mode == kPoolMax || mode == kPoolMin) && !arg)) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid(blocks_for(n_points * frames * channels)), block(256);
#define SE3_FUNPOOL(M) \
  hipLaunchKernelGGL(frame_unpool_kernel<M>, grid, block, 0, stream, grad_out, arg, n_points, frames, channels, grad_x)
  if (mode == kPoolAvg) SE3_FUNPOOL(kPoolAvg);
  else if (mode == kPoolMax) SE3_FUNPOOL(kPoolMax);
  else if (mode == kPoolMin) SE3_FUNPOOL(kPoolMin);
  else SE3_FUNPOOL(kPoolSum);
#undef SE3_FUNPOOL
  return check_launch();
}

// ---- random one-point-per-cell sub-sampling (GridSubSample(..., p_rnd_sample=True), pc/GridSubSample.py:43-54) -------
namespace se3 {
namespace {
// ids[c] = start(c) + floor(u[c] * count(c))  (a position in the cell-sorted point list, GridSubSample.py:52-54);
// picked[c] = sorted_ids[ids[c]] (the point that represents the cell, :67).  The product is clamped to count - 1:
// u * count can round up to count in fp32, which in the reference selects the first point of the NEXT cell.
__global__ void grid_pick_kernel(const int32_t* __restrict__ cell_ends, const int32_t* __restrict__ sorted_ids,
                                 const float* __restrict__ u, int64_t n_cells, int32_t* __restrict__ ids,
                                 int32_t* __restrict__ picked) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  const int start = c > 0 ? cell_ends[c - 1] : 0;
  const int count = cell_ends[c] - start;
  int off = (int)floorf(u[c] * (float)count);
  off = min(max(off, 0), count - 1);
  ids[c] = start + off;
  picked[c] = sorted_ids[start + off];
}

// out[r] = src[idx[r]] for rows of `row_words` 4-byte words (gather) / out[idx[r]] = src[r] (scatter; idx unique)
template <bool SCATTER, typename W>
__global__ void rows_move_kernel(const W* __restrict__ src, const int32_t* __restrict__ idx, int64_t n_rows,
                                 int64_t row_words, W* __restrict__ out) {
  const int64_t total = n_rows * row_words;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = t / row_words, w = t - r * row_words;
    const int64_t other = (int64_t)idx[r] * row_words + w;
    if (SCATTER) out[other] = src[t];
    else out[t] = src[other];
  }
}
}  // namespace
}  // namespace se3

extern "C" int se3_grid_pick(const int32_t* cell_ends, const int32_t* sorted_ids, const float* u, int64_t n_cells,
                             int32_t* ids, int32_t* picked, void* stream) {
  if (n_cells < 0) return SE3_ERR_INVALID_ARGUMENT;
  if (n_cells == 0) return SE3_OK;
  if (!cell_ends || !sorted_ids || !u || !ids || !picked) return SE3_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(grid_pick_kernel, dim3(blocks_for(n_cells)), dim3(256), 0, (hipStream_t)stream, cell_ends, sorted_ids,
                     u, n_cells, ids, picked);
  return check_launch();
}

static int rows_move(bool scatter, const void* src, const int32_t* idx, int64_t n_rows, int64_t row_bytes, void* out,
                     void* stream_) {
  if (n_rows < 0 || row_bytes < 1) return SE3_ERR_INVALID_ARGUMENT;
  if (n_rows == 0) return SE3_OK;
  if (!src || !idx || !out) return SE3_ERR_INVALID_ARGUMENT;
  hipStream_t stream = (hipStream_t)stream_;
  const bool words = row_bytes % 4 == 0 && ((uintptr_t)src | (uintptr_t)out) % 4 == 0;
  const int64_t rw = words ? row_bytes / 4 : row_bytes;
  const dim3 grid(blocks_for(n_rows * rw)), block(256);
  if (words) {
    if (scatter) hipLaunchKernelGGL((rows_move_kernel<true, uint32_t>), grid, block, 0, stream, (const uint32_t*)src, idx, n_rows, rw, (uint32_t*)out);
    else hipLaunchKernelGGL((rows_move_kernel<false, uint32_t>), grid, block, 0, stream, (const uint32_t*)src, idx, n_rows, rw, (uint32_t*)out);
  } else {
    if (scatter) hipLaunchKernelGGL((rows_move_kernel<true, uint8_t>), grid, block, 0, stream, (const uint8_t*)src, idx, n_rows, rw, (uint8_t*)out);
    else hipLaunchKernelGGL((rows_move_kernel<false, uint8_t>), grid, block, 0, stream, (const uint8_t*)src, idx, n_rows, rw, (uint8_t*)out);
  }
  return check_launch();
}

extern "C" int se3_rows_gather(const void* src, const int32_t* idx, int64_t n_out, int64_t row_bytes, void* out, void* stream) {
  return rows_move(false, src, idx, n_out, row_bytes, out, stream);
}

extern "C" int se3_rows_scatter(const void* src, const int32_t* idx, int64_t n_src, int64_t row_bytes, void* out, void* stream) {
  return rows_move(true, src, idx, n_src, row_bytes, out, stream);
}
