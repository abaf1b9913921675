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
#include <stdlib.h>

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
template <bool use_lds>
__global__ __launch_bounds__(KNN_WAVES* PZN_WAVE) void knn32_kernel(
    const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S, int K,
    int q_per_block, int64_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* cand_all = reinterpret_cast<uint64_t*>(smem_raw);  // [KNN_WAVES][KNN_CAND_CAP]
  float* sx = reinterpret_cast<float*>(smem_raw + KNN_WAVES * KNN_CAND_CAP * sizeof(uint64_t));

  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  const float* g = xyz + (size_t)b * N * 3;
  CloudView cv;
  if constexpr (use_lds) {
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
        pzn::wave_lds_sync();
        uint64_t c = lane >= 32 ? cand[lane - 32] : best;
        best = bitonic_sort64(c, lane);
        tau = bcast_u64(best, 31);
        // shift the (< 64) leftovers down by 32
        uint64_t mv = (lane + 32 < cnt) ? cand[lane + 32] : 0;
        pzn::wave_lds_sync();
        if (lane + 32 < cnt) cand[lane] = mv;
        cnt -= 32;
        pzn::wave_lds_sync();
      }
    }
    if (cnt > 0) {
      pzn::wave_lds_sync();
      uint64_t c = lane >= 32 ? ((lane - 32 < cnt) ? cand[lane - 32] : ~0ull) : best;
      best = bitonic_sort64(c, lane);
      pzn::wave_lds_sync();
    }
    if (lane < K) idx[((size_t)b * S + s) * K + lane] = (int64_t)(uint32_t)best;
  }
}

// --------------------------------------- K = 32: threshold selection (+ reference-layout group write) --
// Second-generation selection, exact like the kernels above but with ~40 % fewer vector instructions per query:
//   pass 1   the cloud image in LDS is padded to 64*R points with +INF; a lane owns the point PAIRS
//            (128p + 2 lane, +1): three ds_read_b64 per pair, the distance of both points with packed fp32
//            arithmetic (v_pk_add / v_pk_mul: individually rounded, no fma: bit-identical to sqdist3), one
//            v_min3 for the lane minimum.  No index tracking.
//   tau      the 64 lane minima are sorted as 32-bit keys (distances are >= 0: the uint order is the float order);
//            the 32nd smallest bounds the answer from above: at least 32 points lie at or below it.
//   collect  every point with d <= tau (expected ~44 of 2048 on uniform data) is ballot-compacted into a
//            per-wave LDS buffer as a unique key (dist_bits << 32 | index);
//   sort     ONE 64-wide u64 bitonic sort of the candidates; lanes 0..31 = the answer in stable (distance, index)
//            order.  More than 64 candidates (ties, adversarial layouts): the buffer is reduced to its best 32
//            whenever it exceeds 64 entries, which also tightens tau.
// With GROUP the same wavefront then writes its query's rows of pointnet_util.py:123-132 in the REFERENCE layout
// [B,S,32,3+D]: kp rows at a time are assembled in LDS (feature rows gathered with 16-byte loads from the
// L2-resident table, xyz - centre from the cloud image) and streamed out as one contiguous, 64-byte aligned run of
// 16-byte-per-lane stores (a query's 32 rows are 128 (3+D) bytes: 8,576 B at D = 64) — selection is vector-ALU
// work, the row stream is HBM work, and wavefronts in different phases overlap the two.
typedef float pzn_f2 __attribute__((ext_vector_type(2)));
typedef float pzn_f4v __attribute__((ext_vector_type(4)));
constexpr int SEL_CAND_CAP = 128;  // per wave: SEL_CAND_CAP distance words followed by SEL_CAND_CAP index words

template <int K, int J>
__device__ __forceinline__ uint32_t bitonic_step_u32(uint32_t v, int lane) {
  uint32_t o = pzn::xor_lane<J>(v);
  bool keepmin = ((lane & J) == 0) == ((lane & K) == 0);
  uint32_t lo = v < o ? v : o, hi = v < o ? o : v;
  return keepmin ? lo : hi;
}

__device__ __forceinline__ uint32_t bitonic_sort64_u32(uint32_t v, int lane) {
  v = bitonic_step_u32<2, 1>(v, lane);
  v = bitonic_step_u32<4, 2>(v, lane);
  v = bitonic_step_u32<4, 1>(v, lane);
  v = bitonic_step_u32<8, 4>(v, lane);
  v = bitonic_step_u32<8, 2>(v, lane);
  v = bitonic_step_u32<8, 1>(v, lane);
  v = bitonic_step_u32<16, 8>(v, lane);
  v = bitonic_step_u32<16, 4>(v, lane);
  v = bitonic_step_u32<16, 2>(v, lane);
  v = bitonic_step_u32<16, 1>(v, lane);
  v = bitonic_step_u32<32, 16>(v, lane);
  v = bitonic_step_u32<32, 8>(v, lane);
  v = bitonic_step_u32<32, 4>(v, lane);
  v = bitonic_step_u32<32, 2>(v, lane);
  v = bitonic_step_u32<32, 1>(v, lane);
  v = bitonic_step_u32<64, 32>(v, lane);
  v = bitonic_step_u32<64, 16>(v, lane);
  v = bitonic_step_u32<64, 8>(v, lane);
  v = bitonic_step_u32<64, 4>(v, lane);
  v = bitonic_step_u32<64, 2>(v, lane);
  v = bitonic_step_u32<64, 1>(v, lane);
  return v;
}

// cloud image padded with +INF up to NP = 64 * R points.  A thread moves 4 points per pass: three 16-byte loads (48
// contiguous bytes) and one 16-byte LDS store per coordinate plane (conflict-free); the last N % 4 points and clouds
// that do not start on a 16-byte boundary go point by point.
__device__ __forceinline__ void stage_cloud_padded(const float* __restrict__ g, int N, int NP, float* sx, int T, int tid) {
  float* sy = sx + NP;
  float* sz = sy + NP;
  int done = 0;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    const float4* g4 = reinterpret_cast<const float4*>(g);
    const int quads = N >> 2;
    for (int t = tid; t < quads; t += T) {
      const float4 a = g4[3 * t], b = g4[3 * t + 1], c = g4[3 * t + 2];  // x0 y0 z0 x1 | y1 z1 x2 y2 | z2 x3 y3 z3
      *reinterpret_cast<float4*>(sx + 4 * t) = make_float4(a.x, a.w, b.z, c.y);
      *reinterpret_cast<float4*>(sy + 4 * t) = make_float4(a.y, b.x, b.w, c.z);
      *reinterpret_cast<float4*>(sz + 4 * t) = make_float4(a.z, b.y, c.x, c.w);
    }
    done = quads << 2;
  }
  for (int p = done + tid; p < N; p += T) sx[p] = g[3 * p], sy[p] = g[3 * p + 1], sz[p] = g[3 * p + 2];
  for (int i = N + tid; i < NP; i += T) sx[i] = INFINITY, sy[i] = INFINITY, sz[i] = INFINITY;
}

// -> lanes 0..31: the 32 smallest keys (dist_bits << 32 | index) of the cloud image, ascending.
// cand: SEL_CAND_CAP distance words followed by SEL_CAND_CAP index words (per wave).
__device__ __forceinline__ uint64_t cand_key(const uint32_t* cand, int i) {
  return ((uint64_t)cand[i] << 32) | cand[SEL_CAND_CAP + i];
}
__device__ __forceinline__ void cand_put(uint32_t* cand, int i, uint64_t k) {
  cand[i] = (uint32_t)(k >> 32);
  cand[SEL_CAND_CAP + i] = (uint32_t)k;
}
// buffer of cnt > 64 keys -> its best 32 (sorted) followed by the keys beyond the first 64; returns the new bound
__device__ __forceinline__ float cand_reduce(uint32_t* cand, int& cnt, int lane) {
  pzn::wave_lds_sync();
  const uint64_t c = bitonic_sort64(cand_key(cand, lane), lane);
  const uint64_t mv = (64 + lane < cnt) ? cand_key(cand, 64 + lane) : 0;
  pzn::wave_lds_sync();
  if (lane < 32) cand_put(cand, lane, c);
  if (64 + lane < cnt) cand_put(cand, 32 + lane, mv);
  cnt -= 32;
  const float tau_d = __uint_as_float((uint32_t)(bcast_u64(c, 31) >> 32));
  pzn::wave_lds_sync();
  return tau_d;
}

template <int R>
__device__ __forceinline__ uint64_t select32(const float* __restrict__ sx, const float* __restrict__ sy,
                                             const float* __restrict__ sz, float qx, float qy, float qz,
                                             uint32_t* __restrict__ cand, int lane) {
#pragma clang fp contract(off)
  pzn_f2 d[R / 2];
  const pzn_f2 q_x = {qx, qx}, q_y = {qy, qy}, q_z = {qz, qz};
  float md = INFINITY;
#pragma unroll
  for (int p = 0; p < R / 2; ++p) {
    const int j0 = p * 128 + 2 * lane;
    const pzn_f2 dx = q_x - *reinterpret_cast<const pzn_f2*>(sx + j0);
    const pzn_f2 dy = q_y - *reinterpret_cast<const pzn_f2*>(sy + j0);
    const pzn_f2 dz = q_z - *reinterpret_cast<const pzn_f2*>(sz + j0);
    const pzn_f2 v = (dx * dx + dy * dy) + dz * dz;
    d[p] = v;
    md = fminf(md, fminf(v.x, v.y));
  }
  const uint32_t sm = bitonic_sort64_u32(__float_as_uint(md), lane);
  float tau_d = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)sm, 31));

  int cnt = 0;  // wave-uniform
  bool overflow = false;
  // (opaque to the optimiser: otherwise the R index constants 128 p + 2 lane + h are hoisted out of the query loop as
  // loop invariants, held in R registers across the whole kernel and spilled)
  int lane2 = 2 * lane;
  asm volatile("" : "+v"(lane2));
  {
    // Lane-wise collection: each lane marks its own values <= tau in a bit mask (two instructions per value, no ballot /
    // branch per row), the lanes' counts are prefix-summed with three ballots (counts up to 7: more is the rolled
    // path's business), and every lane then appends its few hits — ~0.7 on average — recomputing their distance from
    // the LDS image with the same arithmetic (a dynamic index into the register array is not available).  Against the
    // row-wise ballot scan this took the search alone from 65 to 41-47 us at N = 2048 (64 clouds x 512 queries).
    constexpr int HW = (R + 31) / 32;
    uint32_t hm[HW];
#pragma unroll
    for (int w = 0; w < HW; ++w) hm[w] = 0;
#pragma unroll
    for (int p = 0; p < R / 2; ++p) {
      hm[(2 * p) / 32] |= (d[p].x <= tau_d ? 1u : 0u) << ((2 * p) & 31);
      hm[(2 * p + 1) / 32] |= (d[p].y <= tau_d ? 1u : 0u) << ((2 * p + 1) & 31);
    }
    int mine = 0;
#pragma unroll
    for (int w = 0; w < HW; ++w) mine += __popc(hm[w]);
    overflow = __ballot(mine > 7) != 0;
    int off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned long long mk = __ballot((mine >> k) & 1);
      off += __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0)) << k;
      total += __popcll(mk) << k;
    }
    if (total > SEL_CAND_CAP) overflow = true;
    if (!overflow) {
      int left = mine;
      while (__ballot(left > 0)) {
        if (left > 0) {
          int b = -1;
#pragma unroll
          for (int w = 0; w < HW; ++w)
            if (b < 0 && hm[w] != 0) {
              b = w * 32 + __builtin_ctz(hm[w]);
              hm[w] &= hm[w] - 1;
            }
          --left;
          const int j = (b >> 1) * 128 + (b & 1) + lane2;
          const float ex = qx - sx[j], ey = qy - sy[j], ez = qz - sz[j];
          const float v = (ex * ex + ey * ey) + ez * ez;
          cand[off] = __float_as_uint(v);
          cand[SEL_CAND_CAP + off] = (uint32_t)j;
          ++off;
        }
      }
      cnt = total;
    }
  }
  if (overflow) {
    // Rare path, rolled (one copy of the code): rescan the image pair row by pair row with the same arithmetic and
    // reduce the buffer to its best 32 whenever it holds more than 64 keys, which also tightens the bound.
    cnt = 0;
    tau_d = INFINITY;
    for (int p = 0; p < R / 2; ++p) {
      const int j0 = p * 128 + 2 * lane;
      const pzn_f2 dx = q_x - *reinterpret_cast<const pzn_f2*>(sx + j0);
      const pzn_f2 dy = q_y - *reinterpret_cast<const pzn_f2*>(sy + j0);
      const pzn_f2 dz = q_z - *reinterpret_cast<const pzn_f2*>(sz + j0);
      const pzn_f2 vv = (dx * dx + dy * dy) + dz * dz;
      for (int h = 0; h < 2; ++h) {
        const float v = h ? vv.y : vv.x;
        const bool pred = v <= tau_d;
        const unsigned long long mask = __ballot(pred);
        if (mask == 0) continue;
        const int pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
        if (pred) {
          cand[pos] = __float_as_uint(v);
          cand[SEL_CAND_CAP + pos] = (uint32_t)(j0 + h);
        }
        cnt += __popcll(mask);
        while (cnt > 64) tau_d = cand_reduce(cand, cnt, lane);
      }
    }
  }
  while (cnt > 64) cand_reduce(cand, cnt, lane);  // 65..128 candidates
  pzn::wave_lds_sync();
  const uint64_t c = lane < cnt ? cand_key(cand, lane) : ~0ull;
  const uint64_t best = bitonic_sort64(c, lane);
  pzn::wave_lds_sync();
  return best;
}

// One piece (KP rows of a query's 32) of the reference-layout group write, compile-time shape: D in {64, 128},
// KP * D = 1024 floats.  ONE float per lane and instruction on the gather side: a wave instruction reads 256
// contiguous bytes of a feature row (row index wave-uniform: v_readlane -> scalar base) and writes 64 consecutive LDS
// words; all 16 loads of the piece are in flight before the first LDS write.  Then the piece leaves as a contiguous,
// 64-byte aligned run of 16-byte-per-lane stores.
template <int D, int KP>
__device__ __forceinline__ void piece_load(float* __restrict__ val, __amdgpu_buffer_rsrc_t rs, int myj, int k0, int lane) {
  constexpr int RPI = D / 64;  // load instructions per row
  // buffer_load_dword: descriptor of the cloud's feature table + SCALAR row offset (v_readlane -> s_lshl) + the lane's
  // fixed offset: no 64-bit address arithmetic on the vector unit
#pragma unroll
  for (int i = 0; i < KP * RPI; ++i) {
    const int j = __builtin_amdgcn_readlane(myj, k0 + i / RPI);
    val[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, j * (D * 4) + (i % RPI) * 256, 0));
  }
}

template <int D, int KP, bool NT>
__device__ __forceinline__ void piece_finish(const float* __restrict__ val, float* __restrict__ chunk, int k0, int lane,
                                             float ex, float ey, float ez, float* __restrict__ outp) {
  constexpr int W = 3 + D;
  constexpr int RPI = D / 64;
#pragma unroll
  for (int i = 0; i < KP * RPI; ++i) chunk[(i / RPI) * W + 3 + (i % RPI) * 64 + lane] = val[i];
  if (lane >= k0 && lane < k0 + KP) {
    float* dst = chunk + (lane - k0) * W;
    dst[0] = ex, dst[1] = ey, dst[2] = ez;
  }
  pzn::wave_lds_sync();
  constexpr int N4 = (KP * W) / 4;
  constexpr int FULL = N4 / 64, TAIL = N4 % 64;
  float4* o4 = reinterpret_cast<float4*>(outp);
  const float4* c4 = reinterpret_cast<const float4*>(chunk);
  float4 r[FULL + 1];
#pragma unroll
  for (int i = 0; i < FULL; ++i) r[i] = c4[i * 64 + lane];
  if (TAIL) r[FULL] = c4[FULL * 64 + (lane < TAIL ? lane : 0)];
  if constexpr (NT) {
#pragma unroll
    for (int i = 0; i < FULL; ++i)
      __builtin_nontemporal_store(*reinterpret_cast<const pzn_f4v*>(&r[i]), reinterpret_cast<pzn_f4v*>(&o4[i * 64 + lane]));
    if (TAIL && lane < TAIL)
      __builtin_nontemporal_store(*reinterpret_cast<const pzn_f4v*>(&r[FULL]),
                                  reinterpret_cast<pzn_f4v*>(&o4[FULL * 64 + lane]));
  } else {
#pragma unroll
    for (int i = 0; i < FULL; ++i) o4[i * 64 + lane] = r[i];
    if (TAIL && lane < TAIL) o4[FULL * 64 + lane] = r[FULL];
  }
  pzn::wave_lds_sync();
}

// The same for any D % 4 == 0 and piece height kp (runtime shape: guarded loops).
__device__ __forceinline__ void group_piece_any(float* __restrict__ chunk, const float* __restrict__ cfs, int myj, int k0,
                                                int kp, int D, int dshift, int lane, float ex, float ey, float ez,
                                                float* __restrict__ outp) {
  const int W = 3 + D;
  const int nfl = kp * D;
  for (int e = lane; e < nfl; e += PZN_WAVE) {
    const int kk = dshift >= 0 ? (e >> dshift) : (e / D);
    const int c = e - kk * D;
    const int j = __shfl(myj, k0 + kk, PZN_WAVE);
    chunk[kk * W + 3 + c] = cfs[(size_t)j * D + c];
  }
  if (lane >= k0 && lane < k0 + kp) {
    float* dst = chunk + (lane - k0) * W;
    dst[0] = ex, dst[1] = ey, dst[2] = ez;
  }
  pzn::wave_lds_sync();
  float4* o4 = reinterpret_cast<float4*>(outp);
  const float4* c4 = reinterpret_cast<const float4*>(chunk);
  const int n4 = (kp * W) >> 2;
  for (int t = lane; t < n4; t += PZN_WAVE) o4[t] = c4[t];
  pzn::wave_lds_sync();
}

template <int R, int WAVES, bool GROUP, int DT, int KPT, bool NT>
__global__ __launch_bounds__(WAVES* PZN_WAVE) __attribute__((amdgpu_waves_per_eu(R <= 32 ? 4 : (R <= 64 ? 3 : 2), 8))) void knn_select_kernel(
    const float* __restrict__ xyz, const float* __restrict__ feat, const float* __restrict__ new_xyz, int N, int S, int K,
    int D, int dshift, int kp, int q_per_block, int blocks_per_cloud, int64_t* __restrict__ idx,
    float* __restrict__ out, float* __restrict__ grouped_xyz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int NP = 64 * R;
  const int W = 3 + D;
  uint32_t* cand_all = reinterpret_cast<uint32_t*>(smem_raw);
  float* chunk_all = reinterpret_cast<float*>(smem_raw + WAVES * SEL_CAND_CAP * sizeof(uint64_t));
  float* sx = chunk_all + (GROUP ? (size_t)WAVES * kp * W : 0);
  const float* sy = sx + NP;
  const float* sz = sy + NP;
  const int nb = gridDim.x;
  // XCD-aware order (workgroups are dealt round-robin over the 8 XCDs): whole clouds per XCD, so that a cloud's
  // feature table, re-read S*32/N times by its own queries, stays in ONE 4 MB L2
  const int vb = (nb & 7) == 0 ? (blockIdx.x & 7) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int b = vb / blocks_per_cloud;
  const int tid = threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  stage_cloud_padded(xyz + (size_t)b * N * 3, N, NP, sx, WAVES * PZN_WAVE, tid);
  __syncthreads();
  uint32_t* cand = cand_all + wave * 2 * SEL_CAND_CAP;
  float* chunk = chunk_all + (size_t)wave * kp * W;
  const int s_begin = (vb % blocks_per_cloud) * q_per_block;
  const int s_end = min(S, s_begin + q_per_block);
  const float* cfs = GROUP ? feat + (size_t)b * N * D : nullptr;

  if constexpr (GROUP && DT > 0) {
    // Software pipeline per wavefront: the feature rows of query i are requested (PF pieces = 32 loads per lane in
    // flight), the selection of query i+1 runs while they travel, then query i is assembled and streamed out.  The
    // gather latency disappears behind vector work, and the wavefronts of a CU drift apart instead of all selecting
    // and then all storing in lock step.
    constexpr int PIECES = 32 / KPT;
    constexpr int PF = (KPT * (DT / 64) * PIECES <= 32) ? PIECES : 32 / (KPT * (DT / 64));   // prefetched pieces
    constexpr int NV = KPT * (DT / 64);                                                       // loads per piece
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cfs), 0, N * DT * 4, 0x00020000);
    int s = s_begin + wave;
    int myj = 0;
    float ex = 0.f, ey = 0.f, ez = 0.f;
    auto run_select = [&](int sq) {
      const long qn = (long)b * S + sq;
      const float* q = new_xyz + qn * 3;
      const float qx = q[0], qy = q[1], qz = q[2];
      const uint64_t best = select32<R>(sx, sy, sz, qx, qy, qz, cand, lane);
      const int j = lane < 32 ? (int)(uint32_t)best : 0;
      if (lane < 32) idx[qn * 32 + lane] = (int64_t)j;
      const float px = sx[j], py = sy[j], pz = sz[j];
      if (grouped_xyz && lane < 32) {
        float* gx = grouped_xyz + (qn * 32 + lane) * 3;
        gx[0] = px, gx[1] = py, gx[2] = pz;
      }
      myj = j;
      ex = __fsub_rn(px, qx), ey = __fsub_rn(py, qy), ez = __fsub_rn(pz, qz);  // pointnet_util.py:125
    };
    if (s < s_end) run_select(s);
    while (s < s_end) {
      const long qi = (long)b * S + s;
      float val[PF * NV];
#pragma unroll
      for (int pc = 0; pc < PF; ++pc) piece_load<DT, KPT>(val + pc * NV, rs, myj, pc * KPT, lane);
      const int cj = myj;
      const float cx = ex, cy = ey, cz = ez;
      s += WAVES;
      if (s < s_end) run_select(s);      // (overwrites myj / ex / ey / ez with the NEXT query's)
      float* oq = out + qi * 32 * (3 + DT);
#pragma unroll
      for (int pc = 0; pc < PF; ++pc)
        piece_finish<DT, KPT, NT>(val + pc * NV, chunk, pc * KPT, lane, cx, cy, cz, oq + pc * KPT * (3 + DT));
#pragma unroll
      for (int pc = PF; pc < PIECES; ++pc) {
        float v2[NV];
        piece_load<DT, KPT>(v2, rs, cj, pc * KPT, lane);
        piece_finish<DT, KPT, NT>(v2, chunk, pc * KPT, lane, cx, cy, cz, oq + pc * KPT * (3 + DT));
      }
    }
  } else {
    for (int s = s_begin + wave; s < s_end; s += WAVES) {
      const long qi = (long)b * S + s;
      const float* q = new_xyz + qi * 3;
      const float qx = q[0], qy = q[1], qz = q[2];
      const uint64_t best = select32<R>(sx, sy, sz, qx, qy, qz, cand, lane);
      const int myj = lane < 32 ? (int)(uint32_t)best : 0;
      if (lane < K) idx[qi * K + lane] = (int64_t)myj;
      if (GROUP) {
        const float px = sx[myj], py = sy[myj], pz = sz[myj];
        if (grouped_xyz && lane < 32) {
          float* gx = grouped_xyz + (qi * 32 + lane) * 3;
          gx[0] = px, gx[1] = py, gx[2] = pz;
        }
        const float ex = __fsub_rn(px, qx), ey = __fsub_rn(py, qy), ez = __fsub_rn(pz, qz);  // pointnet_util.py:125
        for (int k0 = 0; k0 < 32; k0 += kp)
          group_piece_any(chunk, cfs, myj, k0, kp, D, dshift, lane, ex, ey, ez, out + (qi * 32 + k0) * W);
      }
    }
  }
}

// ------------------------------------------------------------- any K <= N --
// K rounds of "smallest key greater than the last one taken".
template <bool use_lds>
__global__ __launch_bounds__(KNN_WAVES* PZN_WAVE) void knn_any_kernel(
    const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S, int K,
    int q_per_block, int64_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sx = reinterpret_cast<float*>(smem_raw);
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  const float* g = xyz + (size_t)b * N * 3;
  CloudView cv;
  if constexpr (use_lds) {
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
template <bool use_lds>
__global__ __launch_bounds__(KNN_WAVES* PZN_WAVE) void ball_kernel(
    const float* __restrict__ xyz, const float* __restrict__ new_xyz, int N, int S, int nsample,
    float radius2, int q_per_block, int64_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* sx = reinterpret_cast<float*>(smem_raw);
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  const float* g = xyz + (size_t)b * N * 3;
  CloudView cv;
  if constexpr (use_lds) {
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

namespace {

constexpr int SEL_WAVES = 8;

// Launch of knn_select_kernel: R = 2 * ceil(N / 128) rounded up to a power of two (<= 128: N <= 8192), 8 wavefronts per
// workgroup, ~512 workgroups (two per CU, one round), a multiple of 8 of them when possible (XCD-aware order).  GROUP: the piece
// height kp (rows assembled in LDS per pass) is the largest of 32 / 16 / 8 / 4 that lets two workgroups share a CU.
template <bool GROUP>
int launch_select(const float* xyz, const float* feat, const float* new_xyz, int B, int N, int S, int K, int D,
                  int64_t* idx, float* out, float* grouped_xyz, hipStream_t st) {
  const int rows = (N + PZN_WAVE - 1) / PZN_WAVE;
  if (N < 64 || rows > 128) return PZN_EUNSUPPORTED;      // (N <= 8192: the lane's R distances stay in registers)
  int R = 2;
  while (R < rows) R <<= 1;
  const size_t cloud = (size_t)3 * 64 * R * sizeof(float);
  const size_t fixed = (size_t)SEL_WAVES * SEL_CAND_CAP * sizeof(uint64_t) + cloud;
  int kp = 0;
  const int W = 3 + D;
  if (GROUP) {
    for (kp = 32; kp >= 4; kp >>= 1)
      if (fixed + (size_t)SEL_WAVES * kp * W * sizeof(float) <= 78 * 1024) break;
    if (kp < 4)  // wide rows: one workgroup per CU
      for (kp = 32; kp >= 4; kp >>= 1)
        if (fixed + (size_t)SEL_WAVES * kp * W * sizeof(float) <= 150 * 1024) break;
    if (kp < 4) return PZN_EUNSUPPORTED;
  }
  // compile-time piece shapes of the encoder's two levels (D = 64: 16 rows, D = 128: 8 rows per piece)
  int dt = 0;
  if (GROUP && D == 64 && fixed + (size_t)SEL_WAVES * 16 * W * sizeof(float) <= 150 * 1024) dt = 64, kp = 16;
  if (GROUP && D == 128 && fixed + (size_t)SEL_WAVES * 8 * W * sizeof(float) <= 150 * 1024) dt = 128, kp = 8;
  const size_t lds = fixed + (GROUP ? (size_t)SEL_WAVES * kp * W * sizeof(float) : 0);
  // ~2 workgroups per CU in ONE round (512 measured better than 1024 for the fused launch); PZN_KG_BLOCKS: tuning aid
  constexpr int target = 512;
  int qpb = 64;
  while (qpb > SEL_WAVES && (long)B * ((S + qpb - 1) / qpb) < target) qpb >>= 1;
  const int bpc = (S + qpb - 1) / qpb;
  const long nb = (long)B * bpc;
  if (nb > 0x7fffffffL) return PZN_EINVAL;
  // Streaming (non-temporal) row stores when the feature tables of the clouds resident on one XCD (32 CUs x 2
  // workgroups, whole clouds per XCD) fill its 4 MB L2: the write stream then evicts the tables it is gathering from
  // (measured at N = 2048, D = 64: 0.082 -> 0.075 ms per launch; at N = 512, D = 128 the tables fit and plain stores win)
  const long clouds_per_xcd = (64 + bpc - 1) / bpc;
  constexpr int nt_force = -1;  // tuning aid
  const bool nt = GROUP && (nt_force >= 0 ? nt_force != 0 : clouds_per_xcd * (long)N * D * 4 > 3L * 1024 * 1024);
  int dshift = -1;  // log2(D) when D is a power of two
  if (GROUP && D > 0 && (D & (D - 1)) == 0) dshift = __builtin_ctz((unsigned)D);
#define PZN_SEL_K(RR, DTT, KPP, NTT)                                                                                      \
  do {                                                                                                                \
    if (set_lds(&knn_select_kernel<RR, SEL_WAVES, GROUP, DTT, KPP, NTT>, lds) != PZN_OK) return PZN_ELAUNCH;               \
    PZN_LAUNCH((knn_select_kernel<RR, SEL_WAVES, GROUP, DTT, KPP, NTT>), dim3((unsigned)nb),                       \
                       dim3(SEL_WAVES * PZN_WAVE), lds, st, xyz, feat, new_xyz, N, S, K, D, dshift, kp, qpb, bpc, idx, \
                       out, grouped_xyz);                                                                             \
  } while (0)
#define PZN_SEL(RR)                    \
  do {                                 \
    if (GROUP && dt == 64 && nt)       \
      PZN_SEL_K(RR, (GROUP ? 64 : 0), (GROUP ? 16 : 0), GROUP);  \
    else if (GROUP && dt == 64)        \
      PZN_SEL_K(RR, (GROUP ? 64 : 0), (GROUP ? 16 : 0), false);  \
    else if (GROUP && dt == 128)       \
      PZN_SEL_K(RR, (GROUP ? 128 : 0), (GROUP ? 8 : 0), false);  \
    else                               \
      PZN_SEL_K(RR, 0, 0, false);      \
  } while (0)
  switch (R) {
    case 2: PZN_SEL(2); break;
    case 4: PZN_SEL(4); break;
    case 8: PZN_SEL(8); break;
    case 16: PZN_SEL(16); break;
    case 32: PZN_SEL(32); break;
    case 64: PZN_SEL(64); break;
    default: PZN_SEL(128); break;
  }
#undef PZN_SEL_K
#undef PZN_SEL
  PZN_RETURN_LAUNCH_STATUS();
}

}  // namespace

PZN_EXPORT int pzn_knn_f32(const float* xyz, const float* new_xyz, int B, int N, int S, int K, int64_t* idx,
                           pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && new_xyz && idx && B > 0 && N > 0 && S > 0 && K > 0 && K <= N && B <= 65535);
  hipStream_t st = pzn_hip_stream(stream);
  if (K <= 32) {
    if (K <= N && N >= 64 && N <= 8192) {
      int rc = launch_select<false>(xyz, nullptr, new_xyz, B, N, S, K, 0, idx, nullptr, nullptr, st);
      if (rc != PZN_EUNSUPPORTED) return rc;
    }
    Geometry g = geometry(B, N, S, KNN_WAVES * KNN_CAND_CAP * sizeof(uint64_t));
    // any other N (clouds below 64 or beyond 8192 points): the general kernel, distances recomputed per round
    {
      if (g.use_lds) {
        if (set_lds(&knn32_kernel<true>, g.lds) != PZN_OK) return PZN_ELAUNCH;
        PZN_LAUNCH(knn32_kernel<true>, g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, K,
                           g.q_per_block, idx);
      } else {
        PZN_LAUNCH(knn32_kernel<false>, g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, K,
                           g.q_per_block, idx);
      }
    }
  } else {
    Geometry g = geometry(B, N, S, 0);
    if (g.use_lds) {
      if (set_lds(&knn_any_kernel<true>, g.lds) != PZN_OK) return PZN_ELAUNCH;
      PZN_LAUNCH(knn_any_kernel<true>, g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, K,
                         g.q_per_block, idx);
    } else {
      PZN_LAUNCH(knn_any_kernel<false>, g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, K,
                         g.q_per_block, idx);
    }
  }
  PZN_RETURN_LAUNCH_STATUS();
}

// pointnet_util.py:117-132 with knn=True, K = 32, in ONE launch and in the reference's layout: idx[B,S,32] (stable
// (distance, index) order, as pzn_knn_f32) and out[B,S,32,3+D] = cat(xyz[idx] - new_xyz, feat[idx]); grouped_xyz
// [B,S,32,3] (returnfps) when non-NULL.  64 <= N <= 8192, D > 0, D % 4 == 0, feat / out 16-byte aligned; other
// shapes return PZN_EUNSUPPORTED (compose pzn_knn_f32 + pzn_group_fwd_f32 then).
PZN_EXPORT int pzn_knn_group_f32(const float* xyz, const float* feat, const float* new_xyz, int B, int N, int S, int D,
                                 int64_t* idx, float* out, float* grouped_xyz, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && feat && new_xyz && idx && out && B > 0 && N >= 32 && S > 0 && D > 0);
  if ((D & 3) != 0 || (reinterpret_cast<uintptr_t>(feat) & 15) != 0 || (reinterpret_cast<uintptr_t>(out) & 15) != 0)
    return PZN_EUNSUPPORTED;
  return launch_select<true>(xyz, feat, new_xyz, B, N, S, 32, D, idx, out, grouped_xyz, pzn_hip_stream(stream));
}

PZN_EXPORT int pzn_ball_query_f32(float radius2, int nsample, const float* xyz, const float* new_xyz, int B, int N,
                                  int S, int64_t* idx, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && new_xyz && idx && B > 0 && N > 0 && S > 0 && nsample > 0 && B <= 65535);
  hipStream_t st = pzn_hip_stream(stream);
  Geometry g = geometry(B, N, S, 0);
  if (g.use_lds) {      // (compile-time variants: a run-time choice of address space turns every point fetch into a flat load)
    if (set_lds(&ball_kernel<true>, g.lds) != PZN_OK) return PZN_ELAUNCH;
    PZN_LAUNCH(ball_kernel<true>, g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, nsample,
                       radius2, g.q_per_block, idx);
  } else {
    PZN_LAUNCH(ball_kernel<false>, g.grid, dim3(KNN_WAVES * PZN_WAVE), g.lds, st, xyz, new_xyz, N, S, nsample,
                       radius2, g.q_per_block, idx);
  }
  PZN_RETURN_LAUNCH_STATUS();
}

