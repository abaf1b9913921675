// datapipe.hip — the cut of the reference's loader (dataset.py:761-775, 1165-1190) as one launch per batch.
//
// `CADDataset.slice` draws a plane (normal = rand(3,1), z = rand(1)/3), splits the raw cloud by the sign of
// points . normal + z and re-draws while a piece holds fewer than N points.  Here the K candidate planes of every sample are
// drawn up front; one workgroup per sample takes the FIRST candidate that leaves >= n_min points on both sides (the
// sequential re-draw's distribution as long as one of the K is valid) and writes both pieces in their original point order
// (a stable partition: the piece's row order decides which point a start index names, dataset.py:1153), padded to `cap`
// rows with copies of the piece's first row (which can never win farthest point sampling), with the piece sizes and the two
// FPS start indices floor(u * size).  The signed distance is evaluated in float64 like numpy does for float32 points times
// float64 draws, every operation individually rounded (no fma): ((x n0 + y n1) + z n2) + offset.
//
// Replaces, per batch: one float64 einsum, two stable sorts of [B, M] keys, two gathers, ~25 element-wise launches.
#include "pzn_common.h"

namespace {

constexpr int DP_T = 1024;
constexpr int DP_W = DP_T / PZN_WAVE;

struct CutArgs {
  const float* raw;        // [B, M, 3]
  const double* normals;   // [B, K, 3]
  const double* zs;        // [B, K]
  const double* u;         // [B, 2]: start fractions (up, down)
  int B, M, K, n_min, cap;
  float* pieces;           // [2B, cap, 3]: rows 0..B-1 the up pieces (distance >= 0), rows B..2B-1 the down pieces
  int64_t* counts;         // [2B]
  int64_t* start;          // [2B]
  double* plane;           // [B, 4]: normal, offset of the plane that was taken
  uint8_t* ok;             // [B]: a candidate was valid (else: the most balanced candidate was taken)
};

__device__ __forceinline__ bool is_up(float x, float y, float z, double n0, double n1, double n2, double off) {
  const double d = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn((double)x, n0), __dmul_rn((double)y, n1)), __dmul_rn((double)z, n2)), off);
  return d >= 0.0;
}

// sum of one int per thread over the workgroup, the same value returned to every thread (two barriers)
__device__ __forceinline__ int block_sum(int v, int* slots) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, PZN_WAVE);
  __syncthreads();          // (slots may still be read from the previous call)
  if ((threadIdx.x & (PZN_WAVE - 1)) == 0) slots[threadIdx.x / PZN_WAVE] = v;
  __syncthreads();
  int t = 0;
#pragma unroll
  for (int w = 0; w < DP_W; ++w) t += slots[w];
  return t;
}

__global__ __launch_bounds__(DP_T) void cut_compact_kernel(CutArgs a) {
  __shared__ int slots[DP_W];
  __shared__ int wave_base[DP_W];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (PZN_WAVE - 1), wave = tid / PZN_WAVE;
  const int M = a.M;
  const float* g = a.raw + (size_t)b * M * 3;
  // a thread owns a CONTIGUOUS run of points, so that the partition keeps the original order with one scan over threads
  const int chunk = (M + DP_T - 1) / DP_T;
  const int lo = tid * chunk < M ? tid * chunk : M, hi = lo + chunk < M ? lo + chunk : M;

  int chosen = -1, best_k = 0, best_bal = -1;
  for (int k = 0; k < a.K; ++k) {
    const double* nk = a.normals + ((size_t)b * a.K + k) * 3;
    const double n0 = nk[0], n1 = nk[1], n2 = nk[2], off = a.zs[(size_t)b * a.K + k];
    int c = 0;
    for (int j = lo; j < hi; ++j) c += is_up(g[3 * j], g[3 * j + 1], g[3 * j + 2], n0, n1, n2, off) ? 1 : 0;
    const int up = block_sum(c, slots);
    const int bal = up < M - up ? up : M - up;
    if (bal > best_bal) best_bal = bal, best_k = k;
    if (up >= a.n_min && M - up >= a.n_min) {      // (uniform: every thread holds the same sum)
      chosen = k;
      break;
    }
  }
  const bool valid = chosen >= 0;
  if (!valid) chosen = best_k;
  const double* nk = a.normals + ((size_t)b * a.K + chosen) * 3;
  const double n0 = nk[0], n1 = nk[1], n2 = nk[2], off = a.zs[(size_t)b * a.K + chosen];

  // stable partition: exclusive scan of the per-thread up counts over the workgroup
  int c = 0;
  for (int j = lo; j < hi; ++j) c += is_up(g[3 * j], g[3 * j + 1], g[3 * j + 2], n0, n1, n2, off) ? 1 : 0;
  int incl = c;
  for (int d = 1; d < PZN_WAVE; d <<= 1) {
    const int o = __shfl_up(incl, d, PZN_WAVE);
    if (lane >= d) incl += o;
  }
  __syncthreads();
  if (lane == PZN_WAVE - 1) slots[wave] = incl;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < DP_W; ++w) wave_base[w] = run, run += slots[w];
    slots[0] = run;      // total
  }
  __syncthreads();
  const int n_up = slots[0], n_down = M - n_up;
  int up_at = wave_base[wave] + incl - c;      // ups in front of this thread's run
  int down_at = lo - up_at;                    // downs in front of it
  float* pu = a.pieces + (size_t)b * a.cap * 3;
  float* pd = a.pieces + (size_t)(a.B + b) * a.cap * 3;
  for (int j = lo; j < hi; ++j) {
    const float x = g[3 * j], y = g[3 * j + 1], z = g[3 * j + 2];
    const bool up = is_up(x, y, z, n0, n1, n2, off);
    const int at = up ? up_at : down_at;
    float* dst = (up ? pu : pd) + (size_t)at * 3;
    if (at < a.cap) dst[0] = x, dst[1] = y, dst[2] = z;
    up_at += up ? 1 : 0;
    down_at += up ? 0 : 1;
  }
  __syncthreads();      // the pieces' first rows are in memory for this workgroup
  // padding: copies of the piece's first row (of the cloud's first row when the piece is empty)
  for (int half = 0; half < 2; ++half) {
    float* p = half ? pd : pu;
    const int cnt = half ? n_down : n_up;
    const float* first = cnt > 0 ? p : g;
    const float fx = first[0], fy = first[1], fz = first[2];
    for (int r = (cnt < a.cap ? cnt : a.cap) + tid; r < a.cap; r += DP_T) p[3 * r] = fx, p[3 * r + 1] = fy, p[3 * r + 2] = fz;
  }
  if (tid == 0) {
    a.counts[b] = n_up;
    a.counts[a.B + b] = n_down;
    for (int half = 0; half < 2; ++half) {
      const int cnt = half ? n_down : n_up;
      long s = (long)floor(a.u[2 * b + half] * (double)cnt);      // np.random.randint(0, n_piece) from a uniform draw
      s = s < 0 ? 0 : (s > cnt - 1 ? cnt - 1 : s);
      a.start[half * a.B + b] = s < 0 ? 0 : s;
    }
    a.plane[4 * b + 0] = n0, a.plane[4 * b + 1] = n1, a.plane[4 * b + 2] = n2, a.plane[4 * b + 3] = off;
    a.ok[b] = (valid && n_up <= a.cap && n_down <= a.cap) ? 1 : 0;
  }
}

// down_mask / up_mask of dataset.py:1357-1367: 1.0 at the k picked rows of each cloud, 0 elsewhere (one launch for both pieces)
__global__ void pick_mask_kernel(const int64_t* __restrict__ idx, int R, int k, int N, float* __restrict__ mask) {
  const int r = blockIdx.x;
  float* m = mask + (size_t)r * N;
  for (int j = threadIdx.x; j < N; j += blockDim.x) m[j] = 0.f;
  __syncthreads();
  for (int j = threadIdx.x; j < k; j += blockDim.x) {
    const int64_t p = idx[(size_t)r * k + j];
    if (p >= 0 && p < N) m[p] = 1.f;
  }
}

}  // namespace

PZN_EXPORT int pzn_cut_compact_f32(const float* raw, const double* normals, const double* zs, const double* u, int B, int M,
                                   int K, int n_min, int cap, float* pieces, int64_t* counts, int64_t* start, double* plane,
                                   uint8_t* ok, pzn_stream_t stream) {
  PZN_CHECK_ARG(raw && normals && zs && u && pieces && counts && start && plane && ok);
  PZN_CHECK_ARG(B > 0 && M > 0 && K > 0 && cap > 0 && n_min >= 0);
  CutArgs a{raw, normals, zs, u, B, M, K, n_min, cap, pieces, counts, start, plane, ok};
  PZN_LAUNCH(cut_compact_kernel, dim3(B), dim3(DP_T), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_pick_mask_f32(const int64_t* idx, int R, int k, int N, float* mask, pzn_stream_t stream) {
  PZN_CHECK_ARG(idx && mask && R > 0 && k > 0 && N > 0);
  PZN_LAUNCH(pick_mask_kernel, dim3(R), dim3(256), 0, pzn_hip_stream(stream), idx, R, k, N, mask);
  PZN_RETURN_LAUNCH_STATUS();
}
