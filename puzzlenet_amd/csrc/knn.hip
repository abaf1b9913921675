// knn.hip — K nearest neighbours (fused distance + selection) and ball query.
//
// Replaces, in pointnet_util.sample_and_group (pointnet_util.py:117-121):
//   dists = square_distance(new_xyz, xyz)      # [B,S,N] fp32 materialised
//   idx   = dists.argsort()[:, :, :K]          # full O(N log N) sort per row, int64 [B,S,N]
// and query_ball_point (pointnet_util.py:76-96).  Neither tensor exists here.
//
// Layout: a workgroup stages ONE cloud as an SoA image (x[N] | y[N] | z[N]) in
// LDS with a single coalesced sweep of the (N,3) rows, then its 4 wavefronts
// each take queries of that cloud.  One 64-lane wavefront = one query:
//   pass 1  every lane scans points lane, lane+64, ... and keeps the minimum
//           key = (dist_bits << 32) | index  (dist >= 0 => bits monotonic;
//           the index in the low word makes keys unique and orders ties by
//           ascending index == a stable ascending sort);
//   sort    the 64 lane-minima are bitonic-sorted across the wave; lanes 0..31
//           are a valid first "best 32" and tau = lane 31's key bounds the
//           answer from above (there are >= 32 keys <= tau);
//   pass 2  rescan, ballot-compact the few keys < tau that are not already a
//           lane minimum into a per-wave LDS buffer, and merge them 32 at a
//           time with another 64-wide bitonic sort (tau tightens as we go).
// Expected work for uniform data at N=2048: ~12 late candidates, i.e. two
// 64-sorts per query; adversarial data only costs more merges, never a wrong
// answer.  K > 32 takes the simple K-round extract-min kernel below.
#include "pzn_common.h"

namespace {

constexpr int KNN_WAVES = 4;
constexpr int KNN_CAND_CAP = 96;  // >= 32 + 63 rounded up

template <int K, int J>
__device__ __forceinline__ uint64_t bitonic_step(uint64_t v, int lane) {
  uint32_t lo = pzn::xor_lane<J>((uint32_t)v), hi = pzn::xor_lane<J>((uint32_t)(v >> 32));
  uint64_t o = ((uint64_t)hi << 32) | lo;
  bool keepmin = ((lane & J) == 0) == ((lane & K) == 0);
  bool olt = o < v;
  return (olt == keepmin) ? o : v;
}

// ascending bitonic sort of 64 keys, one per lane
__device__ __forceinline__ uint64_t bitonic_sort64(uint64_t v, int lane) {
  v = bitonic_step<2, 1>(v, lane);
  v = bitonic_step<4, 2>(v, lane);
  v = bitonic_step<4, 1>(v, lane);
  v = bitonic_step<8, 4>(v, lane);
  v = bitonic_step<8, 2>(v, lane);
  v = bitonic_step<8, 1>(v, lane);
  v = bitonic_step<16, 8>(v, lane);
  v = bitonic_step<16, 4>(v, lane);
  v = bitonic_step<16, 2>(v, lane);
  v = bitonic_step<16, 1>(v, lane);
  v = bitonic_step<32, 16>(v, lane);
  v = bitonic_step<32, 8>(v, lane);
  v = bitonic_step<32, 4>(v, lane);
  v = bitonic_step<32, 2>(v, lane);
  v = bitonic_step<32, 1>(v, lane);
  v = bitonic_step<64, 32>(v, lane);
  v = bitonic_step<64, 16>(v, lane);
  v = bitonic_step<64, 8>(v, lane);
  v = bitonic_step<64, 4>(v, lane);
  v = bitonic_step<64, 2>(v, lane);
  v = bitonic_step<64, 1>(v, lane);
  return v;
}

__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int src) {
  uint32_t lo = __shfl((uint32_t)v, src, PZN_WAVE);
  uint32_t hi = __shfl((uint32_t)(v >> 32), src, PZN_WAVE);
  return ((uint64_t)hi << 32) | lo;
}

struct CloudView {
  const float* x;
  const float* y;
  const float* z;
  int stride;  // 1 for the SoA image in LDS, 3 for the AoS rows in global memory
  __device__ __forceinline__ float X(int j) const { return x[(size_t)j * stride]; }
  __device__ __forceinline__ float Y(int j) const { return y[(size_t)j * stride]; }
  __device__ __forceinline__ float Z(int j) const { return z[(size_t)j * stride]; }
};

__device__ __forceinline__ void stage_cloud(const float* __restrict__ g, int N, float* sx, int T, int tid) {
  float* sy = sx + N;
  float* sz = sy + N;
  for (int i = tid; i < 3 * N; i += T) {
    float v = g[i];
    int p = i / 3, c = i - 3 * p;
    (c == 0 ? sx : (c == 1 ? sy : sz))[p] = v;
  }
}

// ---------------------------------------------------------------- K <= 32 --
__global__ __launch_bounds__(KNN_WAVES* PZN_WAVE) void knn32_kernel(
    const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S, int K,
    int q_per_block, int64_t* __restrict__ idx, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* cand_all = reinterpret_cast<uint64_t*>(smem_raw);  // [KNN_WAVES][KNN_CAND_CAP]
  float* sx = reinterpret_cast<float*>(smem_raw + KNN_WAVES * KNN_CAND_CAP * sizeof(uint64_t));

  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  const float* g = xyz + (size_t)b * N * 3;
  CloudView cv;
  if (use_lds) {
    stage_cloud(g, N, sx, KNN_WAVES * PZN_WAVE, tid);
    __syncthreads();
    cv = CloudView{sx, sx + N, sx + 2 * N, 1};
  } else {
    cv = CloudView{g, g + 1, g + 2, 3};
  }
  uint64_t* cand = cand_all + wave * KNN_CAND_CAP;
  const int rows = (N + PZN_WAVE - 1) / PZN_WAVE;
  const int s_begin = blockIdx.x * q_per_block;
  const int s_end = min(S, s_begin + q_per_block);

  for (int s = s_begin + wave; s < s_end; s += KNN_WAVES) {
    const float* q = new_xyz + ((size_t)b * S + s) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];

    // pass 1: per-lane minimum key
    uint64_t lmin = ~0ull;
    for (int r = 0; r < rows; ++r) {
      int j = r * PZN_WAVE + lane;
      if (j < N) {
        float d = pzn::sqdist3(qx, qy, qz, cv.X(j), cv.Y(j), cv.Z(j));
        uint64_t key = ((uint64_t)__float_as_uint(d) << 32) | (uint32_t)j;
        lmin = key < lmin ? key : lmin;
      }
    }
    uint64_t best = bitonic_sort64(lmin, lane);  // lanes 0..31: current best 32, ascending
    uint64_t tau = bcast_u64(best, 31);

    // pass 2: late candidates
    int cnt = 0;  // wave-uniform
    for (int r = 0; r < rows; ++r) {
      int j = r * PZN_WAVE + lane;
      bool pred = false;
      uint64_t key = 0;
      if (j < N) {
        float d = pzn::sqdist3(qx, qy, qz, cv.X(j), cv.Y(j), cv.Z(j));
        key = ((uint64_t)__float_as_uint(d) << 32) | (uint32_t)j;
        pred = key < tau && key != lmin;
      }
      unsigned long long mask = __ballot(pred);
      if (mask == 0) continue;
      int pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
      if (pred) cand[pos] = key;
      cnt += __popcll(mask);
      while (cnt >= 32) {
        __builtin_amdgcn_wave_barrier();
        uint64_t c = lane >= 32 ? cand[lane - 32] : best;
        best = bitonic_sort64(c, lane);
        tau = bcast_u64(best, 31);
        // shift the (< 64) leftovers down by 32
        uint64_t mv = (lane + 32 < cnt) ? cand[lane + 32] : 0;
        __builtin_amdgcn_wave_barrier();
        if (lane + 32 < cnt) cand[lane] = mv;
        cnt -= 32;
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (cnt > 0) {
      __builtin_amdgcn_wave_barrier();
      uint64_t c = lane >= 32 ? ((lane - 32 < cnt) ? cand[lane - 32] : ~0ull) : best;
      best = bitonic_sort64(c, lane);
      __builtin_amdgcn_wave_barrier();
    }
    if (lane < K) idx[((size_t)b * S + s) * K + lane] = (int64_t)(uint32_t)best;
  }
}

// ------------------------------------------ K <= 32, distances kept in registers --
// Same algorithm as knn32_kernel, for 64 <= N <= 64*R: the R distances a lane owns stay in VGPRs
// between the two passes, so pass 2 is one v_cmp + ballot per row (a float pre-filter d <= tau_d,
// exact key test only on the rare rows that pass) instead of recomputing every distance.
template <int R>
__global__ __launch_bounds__(KNN_WAVES* PZN_WAVE) void knn32_reg_kernel(
    const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S, int K, int q_per_block,
    int64_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* cand_all = reinterpret_cast<uint64_t*>(smem_raw);
  float* sx = reinterpret_cast<float*>(smem_raw + KNN_WAVES * KNN_CAND_CAP * sizeof(uint64_t));
  const float* sy = sx + N;
  const float* sz = sy + N;
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  stage_cloud(xyz + (size_t)b * N * 3, N, sx, KNN_WAVES * PZN_WAVE, tid);
  __syncthreads();
  uint64_t* cand = cand_all + wave * KNN_CAND_CAP;
  const int s_begin = blockIdx.x * q_per_block;
  const int s_end = min(S, s_begin + q_per_block);

  for (int s = s_begin + wave; s < s_end; s += KNN_WAVES) {
    const float* q = new_xyz + ((size_t)b * S + s) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    float d[R];
    float md = INFINITY;
    int mi = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int j = r * PZN_WAVE + lane;
      float v = INFINITY;
      if (j < N) v = pzn::sqdist3(qx, qy, qz, sx[j], sy[j], sz[j]);
      d[r] = v;
      bool lt = v < md;  // strict: the lowest index of equal distances stays (rows ascend with the index)
      md = lt ? v : md;
      mi = lt ? j : mi;
    }
    const uint64_t lmin = md < INFINITY ? (((uint64_t)__float_as_uint(md) << 32) | (uint32_t)mi) : ~0ull;
    uint64_t best = bitonic_sort64(lmin, lane);
    uint64_t tau = bcast_u64(best, 31);
    float tau_d = __uint_as_float((uint32_t)(tau >> 32));

    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (__ballot(d[r] <= tau_d) == 0) continue;  // the common case: nobody in this row can matter
      int j = r * PZN_WAVE + lane;
      uint64_t key = ((uint64_t)__float_as_uint(d[r]) << 32) | (uint32_t)j;
      bool pred = d[r] < INFINITY && key < tau && key != lmin;
      unsigned long long mask = __ballot(pred);
      if (mask == 0) continue;
      int pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
      if (pred) cand[pos] = key;
      cnt += __popcll(mask);
      while (cnt >= 32) {
        __builtin_amdgcn_wave_barrier();
        uint64_t c = lane >= 32 ? cand[lane - 32] : best;
        best = bitonic_sort64(c, lane);
        tau = bcast_u64(best, 31);
        tau_d = __uint_as_float((uint32_t)(tau >> 32));
        uint64_t mv = (lane + 32 < cnt) ? cand[lane + 32] : 0;
        __builtin_amdgcn_wave_barrier();
        if (lane + 32 < cnt) cand[lane] = mv;
        cnt -= 32;
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (cnt > 0) {
      __builtin_amdgcn_wave_barrier();
      uint64_t c = lane >= 32 ? ((lane - 32 < cnt) ? cand[lane - 32] : ~0ull) : best;
      best = bitonic_sort64(c, lane);
      __builtin_amdgcn_wave_barrier();
    }
    if (lane < K) idx[((size_t)b * S + s) * K + lane] = (int64_t)(uint32_t)best;
  }
}

// ------------------------------------- K = 32 neighbours + padded group, one kernel --
// knn32_reg_kernel followed, per query, by the model-internal group write (group.hip,
// group_pad_direct_kernel): rows {dx,dy,dz,0,f_0..f_{D-1}} of the 32 neighbours.  Selection is VALU work,
// the group write is HBM work; in one kernel the wavefronts that are selecting hide behind the ones
// that are streaming rows out, so the stage costs max(select, write) instead of their sum.
// 1-D grid with the XCD-aware order of the group kernel (whole clouds per XCD: the feature table of a
// cloud is re-read S*K/N = 8 times and should hit one L2).
template <int R, int WAVES>
__global__ __launch_bounds__(WAVES* PZN_WAVE) void knn_group_pad_kernel(
    const float* __restrict__ xyz, const float* __restrict__ feat, const float* __restrict__ new_xyz, int N, int S,
    int D, int q_per_block, int blocks_per_cloud, int64_t* __restrict__ idx, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* cand_all = reinterpret_cast<uint64_t*>(smem_raw);
  float* sx = reinterpret_cast<float*>(smem_raw + WAVES * KNN_CAND_CAP * sizeof(uint64_t));
  const float* sy = sx + N;
  const float* sz = sy + N;
  const int nb = gridDim.x;
  const int vb = (nb & 7) == 0 ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int b = vb / blocks_per_cloud;
  const int tid = threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  stage_cloud(xyz + (size_t)b * N * 3, N, sx, WAVES * PZN_WAVE, tid);
  __syncthreads();
  uint64_t* cand = cand_all + wave * KNN_CAND_CAP;
  const int s_begin = (vb % blocks_per_cloud) * q_per_block;
  const int s_end = min(S, s_begin + q_per_block);
  const int V = D >> 2, W4 = 1 + V;
  const float4* cf = reinterpret_cast<const float4*>(feat + (size_t)b * N * D);

  for (int s = s_begin + wave; s < s_end; s += WAVES) {
    const long qi = (long)b * S + s;
    const float* q = new_xyz + qi * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    float d[R];
    float md = INFINITY;
    int mi = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      int j = r * PZN_WAVE + lane;
      float v = INFINITY;
      if (j < N) v = pzn::sqdist3(qx, qy, qz, sx[j], sy[j], sz[j]);
      d[r] = v;
      bool lt = v < md;
      md = lt ? v : md;
      mi = lt ? j : mi;
    }
    const uint64_t lmin = md < INFINITY ? (((uint64_t)__float_as_uint(md) << 32) | (uint32_t)mi) : ~0ull;
    uint64_t best = bitonic_sort64(lmin, lane);
    uint64_t tau = bcast_u64(best, 31);
    float tau_d = __uint_as_float((uint32_t)(tau >> 32));
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (__ballot(d[r] <= tau_d) == 0) continue;
      int j = r * PZN_WAVE + lane;
      uint64_t key = ((uint64_t)__float_as_uint(d[r]) << 32) | (uint32_t)j;
      bool pred = d[r] < INFINITY && key < tau && key != lmin;
      unsigned long long mask = __ballot(pred);
      if (mask == 0) continue;
      int pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
      if (pred) cand[pos] = key;
      cnt += __popcll(mask);
      while (cnt >= 32) {
        __builtin_amdgcn_wave_barrier();
        uint64_t c = lane >= 32 ? cand[lane - 32] : best;
        best = bitonic_sort64(c, lane);
        tau = bcast_u64(best, 31);
        tau_d = __uint_as_float((uint32_t)(tau >> 32));
        uint64_t mv = (lane + 32 < cnt) ? cand[lane + 32] : 0;
        __builtin_amdgcn_wave_barrier();
        if (lane + 32 < cnt) cand[lane] = mv;
        cnt -= 32;
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (cnt > 0) {
      __builtin_amdgcn_wave_barrier();
      uint64_t c = lane >= 32 ? ((lane - 32 < cnt) ? cand[lane - 32] : ~0ull) : best;
      best = bitonic_sort64(c, lane);
      __builtin_amdgcn_wave_barrier();
    }
    // ---- group write for this query (K = 32): lane k < 32 owns neighbour k
    const int myj = lane < 32 ? (int)(uint32_t)best : 0;
    if (lane < 32) idx[qi * 32 + lane] = (int64_t)myj;
    float4* o4 = reinterpret_cast<float4*>(out) + qi * 32 * W4;
    for (int t = lane; t < 32 * V; t += PZN_WAVE) {
      int k = t / V, v = t - k * V;
      int j = __shfl(myj, k, PZN_WAVE);
      o4[k * W4 + 1 + v] = cf[(size_t)j * V + v];
    }
    if (lane < 32)
      o4[lane * W4] = make_float4(__fsub_rn(sx[myj], qx), __fsub_rn(sy[myj], qy), __fsub_rn(sz[myj], qz), 0.f);
  }
}

// ------------------------------------------------------------- any K <= N --
// K rounds of "smallest key greater than the last one taken".
__global__ __launch_bounds__(KNN_WAVES* PZN_WAVE) void knn_any_kernel(
    const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S, int K,
    int q_per_block, int64_t* __restrict__ idx, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sx = reinterpret_cast<float*>(smem_raw);
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  const float* g = xyz + (size_t)b * N * 3;
  CloudView cv;
  if (use_lds) {
    stage_cloud(g, N, sx, KNN_WAVES * PZN_WAVE, tid);
    __syncthreads();
    cv = CloudView{sx, sx + N, sx + 2 * N, 1};
  } else {
    cv = CloudView{g, g + 1, g + 2, 3};
  }
  const int rows = (N + PZN_WAVE - 1) / PZN_WAVE;
  const int s_begin = blockIdx.x * q_per_block;
  const int s_end = min(S, s_begin + q_per_block);
  for (int s = s_begin + wave; s < s_end; s += KNN_WAVES) {
    const float* q = new_xyz + ((size_t)b * S + s) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    uint64_t last = 0;
    bool have_last = false;
    for (int k = 0; k < K; ++k) {
      uint64_t m = ~0ull;
      for (int r = 0; r < rows; ++r) {
        int j = r * PZN_WAVE + lane;
        if (j < N) {
          float d = pzn::sqdist3(qx, qy, qz, cv.X(j), cv.Y(j), cv.Z(j));
          uint64_t key = ((uint64_t)__float_as_uint(d) << 32) | (uint32_t)j;
          bool ok = !have_last || key > last;
          m = (ok && key < m) ? key : m;
        }
      }
      m = pzn::wave_min_u64(m);
      last = m;
      have_last = true;
      if (lane == 0) idx[((size_t)b * S + s) * K + k] = (int64_t)(uint32_t)m;
    }
  }
}

// ------------------------------------------------------------ ball query --
// pointnet_util.py:89-95: indices with d <= r^2 in ascending order, first
// `nsample` of them, the rest padded with the first hit (N when no hit).
__global__ __launch_bounds__(KNN_WAVES* PZN_WAVE) void ball_kernel(
    const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S, int nsample,
    float radius2, int q_per_block, int64_t* __restrict__ idx, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sx = reinterpret_cast<float*>(smem_raw);
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  const float* g = xyz + (size_t)b * N * 3;
  CloudView cv;
  if (use_lds) {
    stage_cloud(g, N, sx, KNN_WAVES * PZN_WAVE, tid);
    __syncthreads();
    cv = CloudView{sx, sx + N, sx + 2 * N, 1};
  } else {
    cv = CloudView{g, g + 1, g + 2, 3};
  }
  const int rows = (N + PZN_WAVE - 1) / PZN_WAVE;
  const int s_begin = blockIdx.x * q_per_block;
  const int s_end = min(S, s_begin + q_per_block);
  for (int s = s_begin + wave; s < s_end; s += KNN_WAVES) {
    const float* q = new_xyz + ((size_t)b * S + s) * 3;
    const float qx = q[0], qy = q[1], qz = q[2];
    int64_t* o = idx + ((size_t)b * S + s) * nsample;
    int cnt = 0;
    int first = N;
    for (int r = 0; r < rows && cnt < nsample; ++r) {
      int j = r * PZN_WAVE + lane;
      bool pred = false;
      if (j < N) {
        float d = pzn::sqdist3(qx, qy, qz, cv.X(j), cv.Y(j), cv.Z(j));
        pred = !(d > radius2);  // :91 group_idx[sqrdists > radius ** 2] = N
      }
      unsigned long long mask = __ballot(pred);
      if (mask == 0) continue;
      if (cnt == 0) first = r * PZN_WAVE + __builtin_ctzll(mask);
      int pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
      if (pred && pos < nsample) o[pos] = j;
      cnt += __popcll(mask);
    }
    cnt = cnt < nsample ? cnt : nsample;
    for (int k = cnt + lane; k < nsample; k += PZN_WAVE) o[k] = first;  // :93-95
  }
}

struct Geometry {
  dim3 grid;
  size_t lds;
  int q_per_block;
  int use_lds;
};

Geometry geometry(int B, int N, int S, size_t extra_lds) {
  Geometry g;
  size_t cloud = (size_t)3 * N * sizeof(float);
  g.use_lds = cloud + extra_lds <= 150 * 1024;
  g.lds = extra_lds + (g.use_lds ? cloud : 0);
  // enough queries per block to amortise staging the cloud, enough blocks to fill 256 CUs
  int qpb = 64;
  while (qpb > KNN_WAVES && (long)B * ((S + qpb - 1) / qpb) < 1024) qpb >>= 1;
  g.q_per_block = qpb;
  g.grid = dim3((S + qpb - 1) / qpb, B);
  return g;
}

template <typename Kern>
int set_lds(Kern k, size_t lds) {
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
          hipSuccess)
    return PZN_ELAUNCH;
  return PZN_OK;
}

}  // namespace

PZN_EXPORT int pzn_knn_f32(const float* xyz, const float* new_xyz, int B, int N, int S, int K, int64_t* idx,
                           pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && new_xyz && idx && B > 0 && N > 0 && S > 0 && K > 0 && K <= N && B <= 65535);
  hipStream_t st = pzn_hip_stream(stream);
  if (K <= 32) {
    Geometry g = geometry(B, N, S, KNN_WAVES * KNN_CAND_CAP * sizeof(uint64_t));
    const int rows = (N + PZN_WAVE - 1) / PZN_WAVE;
#define PZN_KNN_REG(RR)                                                                                          \
  do {                                                                                                           \
    if (set_lds(&knn32_reg_kernel<RR>, g.lds) != PZN_OK) return PZN_ELAUNCH;                                     \
    hipLaunchKernelGGL((knn32_reg_kernel<RR>), g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, K, \
                       g.q_per_block, idx);                                                                      \
  } while (0)
    if (g.use_lds && N >= 64 && rows <= 8)
      PZN_KNN_REG(8);
    else if (g.use_lds && N >= 64 && rows <= 16)
      PZN_KNN_REG(16);
    else if (g.use_lds && N >= 64 && rows <= 32)
      PZN_KNN_REG(32);
    else if (g.use_lds && N >= 64 && rows <= 64)
      PZN_KNN_REG(64);
    else {
      if (set_lds(&knn32_kernel, g.lds) != PZN_OK) return PZN_ELAUNCH;
      hipLaunchKernelGGL(knn32_kernel, g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, K,
                         g.q_per_block, idx, g.use_lds);
    }
#undef PZN_KNN_REG
  } else {
    Geometry g = geometry(B, N, S, 0);
    if (set_lds(&knn_any_kernel, g.lds) != PZN_OK) return PZN_ELAUNCH;
    hipLaunchKernelGGL(knn_any_kernel, g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, K,
                       g.q_per_block, idx, g.use_lds);
  }
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_ball_query_f32(float radius2, int nsample, const float* xyz, const float* new_xyz, int B, int N,
                                  int S, int64_t* idx, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && new_xyz && idx && B > 0 && N > 0 && S > 0 && nsample > 0 && B <= 65535);
  hipStream_t st = pzn_hip_stream(stream);
  Geometry g = geometry(B, N, S, 0);
  if (set_lds(&ball_kernel, g.lds) != PZN_OK) return PZN_ELAUNCH;
  hipLaunchKernelGGL(ball_kernel, g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, nsample, radius2,
                     g.q_per_block, idx, g.use_lds);
  PZN_RETURN_LAUNCH_STATUS();
}

// pointnet_util.py:117-132 for the encoder path in ONE launch: idx[B,S,32] = the 32 nearest neighbours
// (stable (distance, index) order, as pzn_knn_f32) and out[B,S,32,4+D] = the padded grouped rows
// (as pzn_group_pad_fwd_f32).  Needs 64 <= N <= 4096, N*12 B of LDS, D % 4 == 0.
PZN_EXPORT int pzn_knn_group_pad_f32(const float* xyz, const float* feat, const float* new_xyz, int B, int N, int S,
                                     int D, int64_t* idx, float* out, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && feat && new_xyz && idx && out && B > 0 && N >= 64 && S > 0 && D > 0 && (D & 3) == 0);
  PZN_CHECK_ARG((reinterpret_cast<uintptr_t>(feat) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const int rows = (N + PZN_WAVE - 1) / PZN_WAVE;
  if (rows > 64) return PZN_EUNSUPPORTED;
  hipStream_t st = pzn_hip_stream(stream);
  constexpr int KGW = 4;  // wavefronts per workgroup (8 measured the same: the kernel is not occupancy-limited)
  Geometry g = geometry(B, N, S, KGW * KNN_CAND_CAP * sizeof(uint64_t));
  if (!g.use_lds) return PZN_EUNSUPPORTED;
  const int bpc = (S + g.q_per_block - 1) / g.q_per_block;
  const int nb = B * bpc;
#define PZN_KG(RR)                                                                                              \
  do {                                                                                                          \
    if (set_lds(&knn_group_pad_kernel<RR, KGW>, g.lds) != PZN_OK) return PZN_ELAUNCH;                           \
    hipLaunchKernelGGL((knn_group_pad_kernel<RR, KGW>), dim3(nb), dim3(KGW * PZN_WAVE), g.lds, st, xyz, feat,   \
                       new_xyz, N, S, D, g.q_per_block, bpc, idx, out);                                         \
  } while (0)
  if (rows <= 8)
    PZN_KG(8);
  else if (rows <= 16)
    PZN_KG(16);
  else if (rows <= 32)
    PZN_KG(32);
  else
    PZN_KG(64);
#undef PZN_KG
  PZN_RETURN_LAUNCH_STATUS();
}
