// Reference-frame construction (scope row f-1): exact self-kNN and per-point PCA frames.
//
//   se3_knn_query  <- point_cloud_lib_ops.knn_query (custom_ops/knn_query/knn_query.cu:18-196): the k nearest
//                     points of the same batch element, the point itself first, ascending distance, -1 padded.
//   se3_pca_frames <- sample_reference_frames_pca (point_cloud_lib/pc/RotationFunctions.py:307-406): covariance
//                     of the k neighbours, symmetric 3x3 eigen-decomposition, orientation fix, sign-flipped copies.
//
// kNN is a tiled all-pairs scan inside the batch segment (batch ids are sorted, as everywhere in the
// reference): one query per thread, candidate tiles staged in LDS, the k best kept in registers by an
// unrolled insertion.  N = 65k: 4.3 G distance tests ~ 1 ms; it runs once per hierarchy level.
#include "common.h"

namespace se3 {

namespace {

constexpr int kKnnQueries = 64;  // queries per block (one per lane)
constexpr int kKnnSlices = 8;    // wavefronts per block, each scanning 1/8 of the candidate range (4 for K = 64: LDS)

// A block owns 64 consecutive queries; its S wavefronts scan disjoint slices of the candidate range (all-pairs
// inside the batch segment), each keeping the K best per query in registers; the S partial lists meet in LDS
// and wavefront 0 merges them.  8x the wavefronts of the one-wave-per-64-queries version (which left 3 of 4
// SIMD slots idle and serialised on LDS latency).  Queries (qpts, qbatch, m) and candidates (pts, batch_ids, n)
// may be different clouds (KnnNeighborhood between two clouds, pc/KnnNeighborhood.py:77-84); both batch-id arrays
// are sorted.  The self query passes the same arrays twice.
template <int K, int S>
__global__ __launch_bounds__(kKnnQueries* S) void knn_kernel(const float* __restrict__ pts,
                                                             const int32_t* __restrict__ batch_ids, int64_t n,
                                                             const float* __restrict__ qpts,
                                                             const int32_t* __restrict__ qbatch, int64_t m, int k_out,
                                                             int32_t* __restrict__ out) {
  __shared__ float4 tile[S][64];  // per wavefront: x, y, z, batch id (as bits)
  __shared__ float m_d[S][K][kKnnQueries];
  __shared__ int m_i[S][K][kKnnQueries];
  __shared__ int64_t s_lo, s_hi;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * kKnnQueries + lane;
  const bool active = i < m;
  const int64_t ic = active ? i : m - 1;
  const float qx = qpts[ic * 3], qy = qpts[ic * 3 + 1], qz = qpts[ic * 3 + 2];
  const int qb = qbatch[ic];
  if (threadIdx.x == 0) {
    // candidate range of the block: batch ids are sorted, so [first candidate of the first query's batch element,
    // last candidate of the last query's batch element]
    const int64_t first = (int64_t)blockIdx.x * kKnnQueries;
    const int64_t last = min(m, first + kKnnQueries) - 1;
    const int b0 = qbatch[first], b1 = qbatch[last];
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (batch_ids[mid] < b0) lo = mid + 1; else hi = mid; }
    s_lo = lo;
    hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (batch_ids[mid] <= b1) lo = mid + 1; else hi = mid; }
    s_hi = lo;
  }
  __syncthreads();
  float best_d[K];
  int best_i[K];
#pragma unroll
  for (int e = 0; e < K; ++e) best_d[e] = 1e10f, best_i[e] = -1;
  auto insert = [&](float d, int j) {
    // strict '<' keeps the earlier index among equal distances (knn_query.cu:68 `best_dist[e1] > tmp_dist`);
    // slices are scanned in ascending index order and merged in slice order, so this holds globally
    best_d[K - 1] = d;
    best_i[K - 1] = j;
#pragma unroll
    for (int e = K - 1; e > 0; --e) {
      const bool sw = best_d[e] < best_d[e - 1];
      const float dl = sw ? best_d[e] : best_d[e - 1], dh = sw ? best_d[e - 1] : best_d[e];
      const int il = sw ? best_i[e] : best_i[e - 1], ih = sw ? best_i[e - 1] : best_i[e];
      best_d[e - 1] = dl, best_d[e] = dh, best_i[e - 1] = il, best_i[e] = ih;
    }
  };
  // slice of this wavefront (multiples of 64 candidates)
  const int64_t total = s_hi - s_lo;
  const int64_t per = ((total + S - 1) / S + 63) / 64 * 64;
  const int64_t w_lo = s_lo + wave * per, w_hi = min(s_hi, w_lo + per);
  for (int64_t t0 = w_lo; t0 < w_hi; t0 += 64) {
    const int64_t j = t0 + lane;
    if (j < w_hi) tile[wave][lane] = make_float4(pts[j * 3], pts[j * 3 + 1], pts[j * 3 + 2], __int_as_float(batch_ids[j]));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int cnt = (int)min((int64_t)64, w_hi - t0);
#pragma unroll 4
    for (int c = 0; c < cnt; ++c) {
      const float4 p = tile[wave][c];
      const float d = knn_dist2(p.x - qx, p.y - qy, p.z - qz);
      if (d < best_d[K - 1] && __float_as_int(p.w) == qb) insert(d, (int)(t0 + c));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int e = 0; e < K; ++e) m_d[wave][e][lane] = best_d[e], m_i[wave][e][lane] = best_i[e];
  __syncthreads();
  if (wave == 0) {
    // merge the other slices' lists in slice order (their indices are larger: ties keep the lower index)
    for (int w = 1; w < S; ++w)
#pragma unroll
      for (int e = 0; e < K; ++e) {
        const float d = m_d[w][e][lane];
        const int j = m_i[w][e][lane];
        if (j >= 0 && d < best_d[K - 1]) insert(d, j);
      }
    if (active) {
#pragma unroll
      for (int e = 0; e < K; ++e)
        if (e < k_out) out[i * k_out + e] = best_i[e];
    }
  }
}

// Cyclic Jacobi for a symmetric 3x3 matrix: a <- V^T a V diagonal, columns of V = eigenvectors.
__device__ __forceinline__ void jacobi3(float a[3][3], float v[3][3]) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) v[i][j] = i == j ? 1.f : 0.f;
  for (int sweep = 0; sweep < 10; ++sweep) {
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      const float apq = a[p][q];
      if (fabsf(apq) > 1e-30f) {
        const float theta = (a[q][q] - a[p][p]) / (2.f * apq);
        const float t = copysignf(1.f, theta) / (fabsf(theta) + sqrtf(theta * theta + 1.f));
        const float c = 1.f / sqrtf(t * t + 1.f), s = t * c;
#pragma unroll
        for (int r = 0; r < 3; ++r) {  // a <- a J
          const float arp = a[r][p], arq = a[r][q];
          a[r][p] = c * arp - s * arq, a[r][q] = s * arp + c * arq;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {  // a <- J^T a
          const float apr = a[p][r], aqr = a[q][r];
          a[p][r] = c * apr - s * aqr, a[q][r] = s * apr + c * aqr;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float vrp = v[r][p], vrq = v[r][q];
          v[r][p] = c * vrp - s * vrq, v[r][q] = s * vrp + c * vrq;
        }
      }
    }
  }
}

// frames_out [n, NF, 9]: NF = 4 (axis_fixed < 0) or 2.
__global__ void pca_frames_kernel(const float* __restrict__ pts, const int32_t* __restrict__ knn, int64_t n, int k,
                                  int axis_fixed, float* __restrict__ frames) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // neighbourhood matrix (missing neighbours -> the point itself, RotationFunctions.py:314-317)
  float mean[3] = {0.f, 0.f, 0.f};
  for (int e = 0; e < k; ++e) {
    int j = knn[i * k + e];
    if (j < 0) j = (int)i;
#pragma unroll
    for (int d = 0; d < 3; ++d) mean[d] += (d == axis_fixed) ? 0.f : pts[(int64_t)j * 3 + d];
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) mean[d] /= (float)k;
  float a[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  for (int e = 0; e < k; ++e) {
    int j = knn[i * k + e];
    if (j < 0) j = (int)i;
    float x[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = ((d == axis_fixed) ? 0.f : pts[(int64_t)j * 3 + d]) - mean[d];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) a[r][c] += x[r] * x[c];
  }
  float v[3][3];
  jacobi3(a, v);
  // order the eigenpairs: ascending like torch.linalg.eigh (descending after the flip of the fixed-axis branch)
  int ord[3] = {0, 1, 2};
  float ev[3] = {a[0][0], a[1][1], a[2][2]};
#pragma unroll
  for (int pass = 0; pass < 2; ++pass)
#pragma unroll
    for (int t = 0; t < 2 - pass; ++t)
      if (ev[ord[t]] > ev[ord[t + 1]]) { const int tmp = ord[t]; ord[t] = ord[t + 1]; ord[t + 1] = tmp; }
  float f[3][3];  // columns = frame axes
  const bool fixed = axis_fixed >= 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int src = fixed ? ord[2 - c] : ord[c];
#pragma unroll
    for (int r = 0; r < 3; ++r) f[r][c] = src == 0 ? v[r][0] : (src == 1 ? v[r][1] : v[r][2]);
  }
  if (fixed) {
    // the zeroed coordinate gives the eigenvalue-0 direction +-e_axis (last column after the flip); its sign is
    // implementation-defined in LAPACK -- here it is made +e_axis before the orientation fix
    if (f[axis_fixed][2] < 0.f) {
#pragma unroll
      for (int r = 0; r < 3; ++r) f[r][2] = -f[r][2];
    }
  }
  const float det = f[0][0] * (f[1][1] * f[2][2] - f[1][2] * f[2][1]) - f[0][1] * (f[1][0] * f[2][2] - f[1][2] * f[2][0]) +
                    f[0][2] * (f[1][0] * f[2][1] - f[1][1] * f[2][0]);
  if (det < 0.f) {
    if (fixed) {  // keep +e_axis: flip one in-plane axis instead of the whole matrix
#pragma unroll
      for (int r = 0; r < 3; ++r) f[r][1] = -f[r][1];
    } else {      // eigenvec[det < 0] *= -1 (RotationFunctions.py:339)
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) f[r][c] = -f[r][c];
    }
  }
  const int nf = fixed ? 2 : 4;
  // sign patterns with product +1 (columns scaled): free (1,1,1),(1,-1,-1),(-1,1,-1),(-1,-1,1); fixed (1,1,1),(-1,-1,1)
  for (int p = 0; p < nf; ++p) {
    float sg[3];
    if (fixed) sg[0] = p ? -1.f : 1.f, sg[1] = p ? -1.f : 1.f, sg[2] = 1.f;
    else sg[0] = (p & 2) ? -1.f : 1.f, sg[1] = (p == 1 || p == 3) ? -1.f : 1.f, sg[2] = (p == 1 || p == 2) ? -1.f : 1.f;
    float* o = frames + (i * nf + p) * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        int cc = c;
        if (fixed && axis_fixed == 1) cc = c == 1 ? 2 : (c == 2 ? 1 : 0);   // ref_frames[..., [0, 2, 1]]
        float val = f[r][cc] * sg[cc];
        if (fixed && fabsf(val) < 1e-6f) val = 0.f;
        o[r * 3 + c] = val;
      }
  }
}

}  // namespace
}  // namespace se3

using namespace se3;

namespace se3 {
namespace {
// One 1024-thread block per listed query: the threads stride over the points of the query's batch element, each
// keeps its K best in registers; six bitonic shuffle merges per wavefront, then the 16 wavefront lists meet in LDS
// and four more merges finish.  Exact fallback behind the grid search (se3_knn_query_grid) for the few queries
// its 27 cells could not settle (typically a few dozen: a single wavefront per query spent ~1 ms in load latency).
constexpr int kListedWaves = 16;
template <int K>
__global__ __launch_bounds__(64 * kListedWaves) void knn_listed_kernel(const float* __restrict__ pts,
                                                                       const int32_t* __restrict__ batch_ids,
                                                                       int64_t n, int k_out, int32_t* __restrict__ out,
                                                                       const int32_t* __restrict__ list,
                                                                       const int32_t* __restrict__ list_count) {
  __shared__ float w_d[kListedWaves][K];
  __shared__ int w_i[kListedWaves][K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n_q = *list_count;
  for (int64_t qi = blockIdx.x; qi < n_q; qi += gridDim.x) {
    const int64_t i = list[qi];
    const float qx = pts[i * 3], qy = pts[i * 3 + 1], qz = pts[i * 3 + 2];
    const int qb = batch_ids[i];
    int64_t lo = 0, hi = i;  // batch ids are sorted: [first, last] point of the query's batch element
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (batch_ids[mid] < qb) lo = mid + 1; else hi = mid; }
    const int64_t first = lo;
    lo = i, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (batch_ids[mid] <= qb) lo = mid + 1; else hi = mid; }
    const int64_t end = lo;
    TopK<K> best;
    best.init();
    constexpr int kStride = 64 * kListedWaves;
    int64_t j = first + threadIdx.x;
    for (; j + kStride < end; j += 2 * kStride) {  // two independent loads per trip
      const int64_t j2 = j + kStride;
      const float ax = pts[j * 3], ay = pts[j * 3 + 1], az = pts[j * 3 + 2];
      const float bx = pts[j2 * 3], by = pts[j2 * 3 + 1], bz = pts[j2 * 3 + 2];
      best.insert(knn_dist2(ax - qx, ay - qy, az - qz), (int)j);
      best.insert(knn_dist2(bx - qx, by - qy, bz - qz), (int)j2);
    }
    if (j < end) best.insert(knn_dist2(pts[j * 3] - qx, pts[j * 3 + 1] - qy, pts[j * 3 + 2] - qz), (int)j);
#pragma unroll 1
    for (int m = 1; m < 64; m <<= 1) best.merge_xor(m);  // runtime mask: one copy of the merge network
    __syncthreads();  // previous query's LDS lists are consumed
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < K; ++e) w_d[wave][e] = best.dist(e), w_i[wave][e] = best.idx(e);
    }
    __syncthreads();
    if (wave == 0) {
      best.init();
      if (lane < kListedWaves) {
#pragma unroll
        for (int e = 0; e < K; ++e) best.set(e, w_d[lane][e], w_i[lane][e]);
      }
#pragma unroll 1
      for (int m = 1; m < kListedWaves; m <<= 1) best.merge_xor(m);
      if (lane == 0) {
#pragma unroll
        for (int e = 0; e < K; ++e)
          if (e < k_out) out[i * k_out + e] = best.idx(e) == 0x7fffffff ? -1 : best.idx(e);
      }
    }
  }
}
}  // namespace

int launch_knn_bruteforce(const float* pts, const int32_t* batch_ids, int64_t n, const float* qpts, const int32_t* qbatch,
                          int64_t m, int k, int32_t* out, hipStream_t s) {
  const dim3 grid((unsigned)((m + kKnnQueries - 1) / kKnnQueries));
#define SE3_KNN(K, S) \
  hipLaunchKernelGGL((knn_kernel<K, S>), grid, dim3(kKnnQueries * S), 0, s, pts, batch_ids, n, qpts, qbatch, m, k, out)
  if (k <= 8) SE3_KNN(8, kKnnSlices);
  else if (k <= 16) SE3_KNN(16, kKnnSlices);
  else if (k <= 32) SE3_KNN(32, kKnnSlices);
  else SE3_KNN(64, 4);  // the partial lists of 8 slices would not fit LDS
#undef SE3_KNN
  return check_launch();
}

int launch_knn_listed(const float* pts, const int32_t* batch_ids, int64_t n, int k, int32_t* out, const int32_t* list,
                      const int32_t* list_count, hipStream_t s) {
  // the list length lives on the device: a fixed grid of blocks strides over it
  const dim3 grid((unsigned)(n < 512 ? (n > 0 ? n : 1) : 512)), block(64 * kListedWaves);
  if (k <= 8) hipLaunchKernelGGL(knn_listed_kernel<8>, grid, block, 0, s, pts, batch_ids, n, k, out, list, list_count);
  else if (k <= 16) hipLaunchKernelGGL(knn_listed_kernel<16>, grid, block, 0, s, pts, batch_ids, n, k, out, list, list_count);
  else hipLaunchKernelGGL(knn_listed_kernel<32>, grid, block, 0, s, pts, batch_ids, n, k, out, list, list_count);
  return check_launch();
}
}  // namespace se3

extern "C" int se3_knn_query(const float* pts, const int32_t* batch_ids, int64_t n, int32_t k, int32_t* out,
                             void* stream) {
  if (n < 0 || k < 1) return SE3_ERR_INVALID_ARGUMENT;
  if (k > 64 || n >= (1ll << 31)) return SE3_ERR_UNSUPPORTED;  // k <= 64 as the reference's kernel (knn_query.cu:167)
  if (n == 0) return SE3_OK;
  if (!pts || !batch_ids || !out) return SE3_ERR_INVALID_ARGUMENT;
  return launch_knn_bruteforce(pts, batch_ids, n, pts, batch_ids, n, (int)k, out, (hipStream_t)stream);
}

extern "C" int se3_knn_query_pair(const float* src_pts, const int32_t* src_batch, int64_t n_src, const float* q_pts,
                                  const int32_t* q_batch, int64_t n_q, int32_t k, int32_t* out, void* stream) {
  if (n_src < 0 || n_q < 0 || k < 1) return SE3_ERR_INVALID_ARGUMENT;
  if (k > 64 || n_src >= (1ll << 31) || n_q >= (1ll << 31)) return SE3_ERR_UNSUPPORTED;
  if (n_q == 0) return SE3_OK;
  if (!q_pts || !q_batch || !out) return SE3_ERR_INVALID_ARGUMENT;
  if (n_src == 0) return se3::launch_fill_words(out, 0xffffffffu, n_q * k, (hipStream_t)stream);
  if (!src_pts || !src_batch) return SE3_ERR_INVALID_ARGUMENT;
  return launch_knn_bruteforce(src_pts, src_batch, n_src, q_pts, q_batch, n_q, (int)k, out, (hipStream_t)stream);
}

namespace se3 {
namespace {
// out[p, j] = all[p, perm_p[j]], perm_p = the order of point p's n_all uniform draws (ascending value, ties to the lower
// index): a uniformly random permutation per point.  One thread per (point, output frame).
__global__ __launch_bounds__(256) void shuffle_frames_kernel(const float* __restrict__ all, const float* __restrict__ rnd,
                                                             int64_t n, int n_all, int n_frames, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * n_frames) return;
  const int64_t p = t / n_frames;
  const int j = (int)(t - p * n_frames);
  // the draw of rank j: the one with exactly j draws in front of it
  int src = 0;
  for (int a = 0; a < n_all; ++a) {
    const float va = rnd[p * n_all + a];
    int rank = 0;
    for (int b = 0; b < n_all; ++b) {
      const float vb = rnd[p * n_all + b];
      rank += (vb < va || (vb == va && b < a)) ? 1 : 0;
    }
    if (rank == j) src = a;
  }
  const float* f = all + (p * n_all + src) * 9;
  float* o = out + t * 9;
#pragma unroll
  for (int i = 0; i < 9; ++i) o[i] = f[i];
}
}  // namespace
}  // namespace se3

extern "C" int se3_shuffle_frames(const float* all_frames, const float* draws, int64_t n, int32_t n_all, int32_t n_frames,
                                  float* out, void* stream) {
  if (n < 0 || n_all < 1 || n_frames < 1 || n_frames > n_all) return SE3_ERR_INVALID_ARGUMENT;
  if (n_all > 8) return SE3_ERR_UNSUPPORTED;  // the reference's PCA frames come in sets of 4 (2 with a fixed axis)
  if (n == 0) return SE3_OK;
  if (!all_frames || !draws || !out) return SE3_ERR_INVALID_ARGUMENT;
  const int64_t threads = n * n_frames;
  hipLaunchKernelGGL(se3::shuffle_frames_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     all_frames, draws, n, (int)n_all, (int)n_frames, out);
  return check_launch();
}

extern "C" int se3_pca_frames(const float* pts, const int32_t* knn, int64_t n, int32_t k, int32_t axis_fixed,
                              float* frames, void* stream) {
  if (n < 0 || k < 1 || axis_fixed > 2) return SE3_ERR_INVALID_ARGUMENT;
  if (axis_fixed == 0) return SE3_ERR_UNSUPPORTED;  // the reference treats fixed_axis = 0 as "not fixed"; pass -1
  if (n == 0) return SE3_OK;
  if (!pts || !knn || !frames) return SE3_ERR_INVALID_ARGUMENT;
  hipLaunchKernelGGL(pca_frames_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, pts, knn,
                     n, (int)k, axis_fixed < 0 ? -1 : (int)axis_fixed, frames);
  return check_launch();
}
