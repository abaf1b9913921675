// fps.hip — farthest point sampling for gfx950.
//
// Replaces pointnet_util.farthest_point_sample (pointnet_util.py:53-73): a
// Python loop of npoint iterations x 5 torch ops.  Here one workgroup owns one
// cloud for the whole loop:
//   * the cloud is read from HBM exactly once (coalesced flat copy of the
//     (N,3) AoS rows), transposed into an SoA image in LDS (centroid fetch)
//     and into registers (each thread keeps PPT points + their running
//     min-distance for all iterations);
//   * per iteration: distance update in registers, arg-max as ONE u64 max of
//     key = (dist_bits << 32) | ~index  (dist >= 0 so its bit pattern is
//     monotonic; ~index makes the LOWEST index win ties, as torch.max does on
//     CPU), wavefront (64-lane) shuffle reduction, one LDS slot per wave, one
//     barrier, every wave re-reduces the <=16 slots redundantly;
//   * slots are double-buffered on the iteration parity, so one barrier per
//     iteration is enough.
// The loop is latency-bound by construction (npoint dependent rounds); the
// launch is B workgroups, i.e. parallel over clouds only.
#include <stdlib.h>

#include <type_traits>

#include "pzn_common.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

// f(0), f(1), ... while q < n, q < Q1, as NESTED ifs on compile-time q: the first failing guard jumps past all the rest (an
// unrolled loop with one guard per body jumps over every unused body in turn; with `break` the loop is not unrolled at all and
// the register arrays are indexed dynamically)
template <int Q, int Q1, typename F>
__device__ __forceinline__ void nested_while_below(int n, F&& f) {
  if constexpr (Q < Q1) {
    if (Q < n) {
      f(std::integral_constant<int, Q>{});
      nested_while_below<Q + 1, Q1>(n, f);
    }
  }
}

// f(q) for the one compile-time q in [Q, Q1) that equals the (wavefront-uniform) n
template <int Q, int Q1, typename F>
__device__ __forceinline__ void uniform_pick(int n, F&& f) {
  if constexpr (Q < Q1) {
    if (n == Q)
      f(std::integral_constant<int, Q>{});
    else
      uniform_pick<Q + 1, Q1>(n, f);
  }
}

constexpr int FPS_OUT_CHUNK = 256;      // picks buffered in LDS between write-outs (power of two)

// LDS: the cloud's SoA image fits in LDS (compile-time: with a run-time choice the centroid fetch became three flat_load
// instructions waited for with vmcnt(0) lgkmcnt(0) in the middle of every round's dependent chain)
// G clouds per workgroup (T threads each; cloud blockIdx.x + g gridDim.x): the background form packs the two pieces of a cut
// into one workgroup - they share nothing but the barrier of a round -, so that the loader holds half as many CUs beside a
// training step (two or four such 4-wavefront groups on a CU run as fast as one: the round is a latency chain).
template <int T, int PPT, bool use_lds, int G = 1>
__global__ __launch_bounds__(T * G) void fps_kernel(const float* __restrict__ xyz, int N, int npoint,
                                                    const int64_t* __restrict__ start,
                                                    int64_t* __restrict__ out, const int64_t* __restrict__ counts, int lds_per_cloud) {
  constexpr int W = T / PZN_WAVE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int grp = G > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / T) : 0;
  unsigned char* smem_raw = smem_all + (size_t)grp * lds_per_cloud;
  uint64_t* slots = reinterpret_cast<uint64_t*>(smem_raw);             // [2][W]
  int* sout = reinterpret_cast<int*>(smem_raw + 2 * W * sizeof(uint64_t));      // [FPS_OUT_CHUNK] picks not yet written out
  float* sx = reinterpret_cast<float*>(smem_raw + 2 * W * sizeof(uint64_t) + FPS_OUT_CHUNK * sizeof(int));
  float* sy = sx + N;
  float* sz = sy + N;
  // without the image (PPT > 4): every wavefront publishes its best point's coordinates beside its key, [2][W] x 16 bytes
  // where the image would start (the pick's owner has them in registers: no fetch from memory in the round's dependent chain)
  float4* scoord = reinterpret_cast<float4*>(sx);
  constexpr bool publish = !use_lds && PPT > 4;

  const int b = blockIdx.x + grp * gridDim.x;
  const int tid = G > 1 ? (int)threadIdx.x - grp * T : (int)threadIdx.x;
  const int lane = tid & (PZN_WAVE - 1);
  const int wave = tid / PZN_WAVE;
  const float* g = xyz + (size_t)b * N * 3;

  float px[PPT], py[PPT], pz[PPT], dist[PPT];
  // the no-image form keeps its points as PAIRS of register slots (2q, 2q + 1): the distance update is packed fp32 - two points
  // per instruction, every operation individually rounded as in sqdist3 (this file is built with -ffp-contract=off) -: with
  // up to 16 slots per thread and 8 wavefronts the round is bound by vector issue, not by its dependent latencies
  constexpr int PQ = (!use_lds && PPT > 4) ? PPT / 2 : 1;
  v2f qx[PQ], qy[PQ], qz[PQ], qd[PQ];
  if (use_lds) {
    for (int i = tid; i < 3 * N; i += T) {
      float v = g[i];
      int p = i / 3, c = i - 3 * p;
      (c == 0 ? sx : (c == 1 ? sy : sz))[p] = v;
    }
    __syncthreads();
  }
  if constexpr (!use_lds && PPT > 4) {
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const int j = tid + p * T;
      const bool ok = j < N;
      // (a slot without a point: distance 0 for ever - it can only tie at 0, and ties go to the lower, i.e. a real, index)
      qx[p >> 1][p & 1] = ok ? g[(size_t)j * 3 + 0] : 0.f;
      qy[p >> 1][p & 1] = ok ? g[(size_t)j * 3 + 1] : 0.f;
      qz[p >> 1][p & 1] = ok ? g[(size_t)j * 3 + 2] : 0.f;
      qd[p >> 1][p & 1] = ok ? 1e10f : 0.f;  // pointnet_util.py:64
    }
  }
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    if (!use_lds && PPT > 4) break;
    int j = tid + p * T;
    bool ok = j < N;
    if (use_lds) {
      px[p] = ok ? sx[j] : 0.f;
      py[p] = ok ? sy[j] : 0.f;
      pz[p] = ok ? sz[j] : 0.f;
    } else {
      px[p] = ok ? g[(size_t)j * 3 + 0] : 0.f;
      py[p] = ok ? g[(size_t)j * 3 + 1] : 0.f;
      pz[p] = ok ? g[(size_t)j * 3 + 2] : 0.f;
    }
    dist[p] = 1e10f;  // pointnet_util.py:64
  }

  // counts (the data pipeline's padded pieces): only the first counts[b] rows of the cloud are real, the rest are copies of
  // row 0, which can never be picked (distance 0 after the first round at the latest, and any tie goes to the lower index);
  // the rounds then leave out every register slot that holds padding only (a workgroup-uniform bound)
  constexpr int CAP = T * PPT;      // (the background form may hold fewer slots than the buffer has rows: max_count)
  const int nmax = N < CAP ? N : CAP;
  const int nreal = counts ? (int)(counts[b] < 1 ? 1 : (counts[b] > nmax ? nmax : counts[b])) : nmax;
  const int pmax = (nreal + T - 1) / T;
  const bool full = N == T * PPT && nreal == N;      // every thread's every point exists: no bounds tests in the rounds
  int far = (int)start[b];  // pointnet_util.py:65 (the caller's randint draw)
  far = far < 0 ? 0 : (far >= N ? N - 1 : far);
  int64_t* o = out + (size_t)b * npoint;
  float ncx = 0.f, ncy = 0.f, ncz = 0.f;      // (publish) the next round's centroid

  for (int i = 0; i < npoint; ++i) {
    // :68 — the pick goes to LDS and leaves in chunks: a global store inside the loop keeps a vector-memory operation
    // outstanding at every barrier (__syncthreads waits for it: several hundred cycles per round on wave 0)
    if (tid == 0) sout[i & (FPS_OUT_CHUNK - 1)] = far;
    if ((i & (FPS_OUT_CHUNK - 1)) == FPS_OUT_CHUNK - 1 || i == npoint - 1) {
      __syncthreads();
      const int base = i & ~(FPS_OUT_CHUNK - 1);
      for (int t = tid; t <= i - base; t += T) o[base + t] = (int64_t)sout[t];
      __syncthreads();      // (sout is rewritten next round)
    }
    float cx, cy, cz;          // :69
    if (use_lds) {
      cx = sx[far];
      cy = sy[far];
      cz = sz[far];
    } else if (publish && i > 0) {
      cx = ncx, cy = ncy, cz = ncz;      // published by the owner's wavefront in the round before
    } else {
      cx = g[(size_t)far * 3 + 0];
      cy = g[(size_t)far * 3 + 1];
      cz = g[(size_t)far * 3 + 2];
    }
    uint64_t best;
    if constexpr (PPT <= 4) {
      // arg-max in two parts: the 32-bit distance pattern (>= 0, so monotonic as an integer) goes through the wave
      // reduction alone — six v_max_u32 with DPP operands instead of six 64-bit compare-and-select steps — and the index
      // is resolved afterwards: one lane holds the maximum almost always (ballot + readlane); on a tie the lowest index
      // wins (as torch.max on CPU), found by a second reduction only then.
      uint32_t bd = 0, bj = 0x7fffffffu;      // (a thread without a valid point keeps the sentinel and never ties)
  #pragma unroll
      for (int p = 0; p < PPT; ++p) {
        int j = tid + p * T;
        float d = pzn::sqdist3(px[p], py[p], pz[p], cx, cy, cz);  // :70
        float nd = d < dist[p] ? d : dist[p];                     // :71
        dist[p] = nd;
        const uint32_t nb = __float_as_uint(nd);
        // strict >: the lower index of equal distances stays (j ascends with p); the thread's first point is always taken
        const bool take = p == 0 ? (full || j < N) : ((full || j < N) && nb > bd);
        bd = take ? nb : bd;
        bj = take ? (uint32_t)j : bj;
      }
      const uint32_t wm = pzn::wave_max_u32_dpp(bd);
      const bool tied = bd == wm && (full || bj != 0x7fffffffu);
      const unsigned long long tmask = __ballot(tied);
      uint32_t wj;
      if (__popcll(tmask) == 1)
        wj = (uint32_t)__builtin_amdgcn_readlane((int)bj, __builtin_ctzll(tmask));
      else
        wj = pzn::wave_min_u32_dpp(tied ? bj : 0xffffffffu);
      best = ((uint64_t)wm << 32) | (uint32_t)(~wj);
    } else if constexpr (publish) {
      // packed distances, 32-bit keys with the slot number deferred (as the small-cloud form above): 8 vector instructions
      // per point instead of 13
      // the centroid as three real register pairs (the asm is empty: it only keeps the compiler from broadcasting one half of a
      // pair with op_sel on src1 - the packed form that returns wrong results beside AGPR-accumulator MFMAs, DESIGN.md section 4,
      // tests/test_isa_forms.py)
      v2f c2x = v2f{cx, cx}, c2y = v2f{cy, cy}, c2z = v2f{cz, cz};
      asm volatile("" : "+v"(c2x), "+v"(c2y), "+v"(c2z));
      const int qmax = (pmax + 1) >> 1;      // (workgroup-uniform; an odd pmax evaluates one slot of padding: harmless, see above)
      uint32_t bd = 0, bp = 0;
      nested_while_below<0, PQ>(qmax, [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const v2f dx = qx[q] - c2x, dy = qy[q] - c2y, dz = qz[q] - c2z;      // :70
        const v2f d = (dx * dx + dy * dy) + dz * dz;
        v2f nd;                                                             // :71
        nd.x = d.x < qd[q].x ? d.x : qd[q].x;
        nd.y = d.y < qd[q].y ? d.y : qd[q].y;
        qd[q] = nd;
        const uint32_t n0 = __float_as_uint(nd.x), n1 = __float_as_uint(nd.y);
        // strict >: the lower slot (= lower index) of equal distances stays; the thread's first slot is always taken
        const bool t0 = q == 0 ? true : n0 > bd;
        bd = t0 ? n0 : bd;
        bp = t0 ? (uint32_t)(2 * q) : bp;
        const bool t1 = n1 > bd;
        bd = t1 ? n1 : bd;
        bp = t1 ? (uint32_t)(2 * q + 1) : bp;
      });
      const uint32_t bj = (uint32_t)tid + bp * (uint32_t)T;
      const uint32_t wm = pzn::wave_max_u32_dpp(bd);
      const bool tied = bd == wm;
      const unsigned long long tmask = __ballot(tied);
      uint32_t wj;
      if (__popcll(tmask) == 1)
        wj = (uint32_t)__builtin_amdgcn_readlane((int)bj, __builtin_ctzll(tmask));
      else
        wj = pzn::wave_min_u32_dpp(tied ? bj : 0xffffffffu);
      best = ((uint64_t)wm << 32) | (uint32_t)(~wj);
    } else {      // many points per thread: the 64-bit key (distance, ~index) per point measured faster there
      best = 0;  // below every real key: real keys have ~j >= 1
#pragma unroll
      for (int p = 0; p < PPT; ++p) {
        if (p < pmax) {      // (workgroup-uniform: slots at and beyond pmax hold padding only)
          int j = tid + p * T;
          float d = pzn::sqdist3(px[p], py[p], pz[p], cx, cy, cz);  // :70
          float nd = d < dist[p] ? d : dist[p];                     // :71
          dist[p] = nd;
          uint64_t key = ((uint64_t)__float_as_uint(nd) << 32) | (uint32_t)(~(uint32_t)j);
          key = j < N ? key : 0ull;
          best = key > best ? key : best;
        }
      }
      best = pzn::wave_max_u64_dpp(best);
    }
    uint64_t* sl = slots + (i & 1) * W;
    if constexpr (publish) {
      // the wavefront's best point j = tid' + ps T sits in register slot ps (wavefront-uniform) of lane j & 63
      const uint32_t wj = ~(uint32_t)best;
      const int ps = __builtin_amdgcn_readfirstlane((int)(wj / T));
      const int qs = ps >> 1;
      // slot pair qs is wavefront-uniform: a chain of scalar compares with ONE taken body instead of six selects per pair on
      // every lane (the empty asm keeps the compiler from turning the bodies back into selects)
      v2f sx2 = qx[0], sy2 = qy[0], sz2 = qz[0];
      uniform_pick<1, PQ>(qs, [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        sx2 = qx[q], sy2 = qy[q], sz2 = qz[q];
        asm volatile("" : "+v"(sx2), "+v"(sy2), "+v"(sz2));
      });
      const bool hi = (ps & 1) != 0;
      const float bx = hi ? sx2.y : sx2.x, by = hi ? sy2.y : sy2.x, bz = hi ? sz2.y : sz2.x;
      const int ol = (int)(wj & (PZN_WAVE - 1));
      const float ox = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bx), ol));
      const float oy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, by), ol));
      const float oz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, bz), ol));
      if (lane == 0) {
        sl[wave] = best;
        scoord[(i & 1) * W + wave] = make_float4(ox, oy, oz, 0.f);
      }
    } else {
      if (lane == 0) sl[wave] = best;
    }
    __syncthreads();
    uint64_t m = sl[0];
    int mw = 0;
#pragma unroll
    for (int w = 1; w < W; ++w) {
      uint64_t v = sl[w];
      mw = v > m ? w : mw;
      m = v > m ? v : m;
    }
    far = (int)(~(uint32_t)m);  // :72 first (lowest-index) maximum
    if constexpr (publish) {
      const float4 c = scoord[(i & 1) * W + mw];
      ncx = c.x, ncy = c.y, ncz = c.z;
    }
  }
}

template <int T, int PPT>
int launch(const float* xyz, int B, int N, int npoint, const int64_t* start, int64_t* out, hipStream_t st, bool image = true,
           const int64_t* counts = nullptr) {
  constexpr int W = T / PZN_WAVE;
  size_t lds_xyz = (size_t)3 * N * sizeof(float);
  size_t lds = 2 * W * sizeof(uint64_t) + FPS_OUT_CHUNK * sizeof(int);
  int use_lds = image && lds + lds_xyz <= 150 * 1024;
  lds += use_lds ? lds_xyz : 2 * W * sizeof(float4);      // the image, or the wavefronts' published coordinates
  if (use_lds) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&fps_kernel<T, PPT, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return PZN_ELAUNCH;
    PZN_LAUNCH((fps_kernel<T, PPT, true>), dim3(B), dim3(T), lds, st, xyz, N, npoint, start, out, counts, (int)lds);
  } else if constexpr (T == 256 && PPT > 4) {      // the background form: clouds b and b + B / 2 (the two pieces of a cut) per workgroup
    if (B % 2 == 0)
      PZN_LAUNCH((fps_kernel<T, PPT, false, 2>), dim3(B / 2), dim3(2 * T), 2 * lds, st, xyz, N, npoint, start, out, counts, (int)lds);
    else
      PZN_LAUNCH((fps_kernel<T, PPT, false>), dim3(B), dim3(T), lds, st, xyz, N, npoint, start, out, counts, (int)lds);
  } else {
    PZN_LAUNCH((fps_kernel<T, PPT, false>), dim3(B), dim3(T), lds, st, xyz, N, npoint, start, out, counts, (int)lds);
  }
  PZN_RETURN_LAUNCH_STATUS();
}

}  // namespace

// The same sampling as a BACKGROUND job (datapipe.PairFeeder: pieces of up to 32768 raw points sampled on a side stream while
// a training step owns the chip): no LDS image of the cloud - every wavefront publishes its best point's coordinates beside its
// key, so the next round's centroid comes from the owner's registers through 16 bytes of LDS, not from memory -, so
// a workgroup holds 1.5 KB of LDS instead of up to 150 KB and the step's LDS-tiled kernels keep their CUs; 512 threads for
// N <= 16384.  counts (may be NULL): int64 [B], the number of REAL rows of each cloud when the rest is padding with copies of
// row 0 (datapipe._compact): the rounds skip the padding.  Same picks bit for bit (the arithmetic is the same code).
PZN_EXPORT int pzn_fps_background_f32(const float* xyz, int B, int N, int npoint, const int64_t* start_idx,
                                      int64_t* out_idx, const int64_t* counts, int max_count, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && start_idx && out_idx && B > 0 && N > 0 && npoint > 0 && max_count >= 0);
  hipStream_t st = pzn_hip_stream(stream);
  // Threads x register slots by the number of rows that can be REAL (max_count: the caller's promise counts[b] <= max_count;
  // 0 or no counts: N): four wavefronts wherever the points fit 32 slots - the round's fixed part (wave reductions, barrier,
  // re-reduction of the per-wave slots) grows with the wavefront count: per round on 2048 real points 1.16 us with eight
  // wavefronts against 0.77 with four, and 0.07 us per 512 points either way (tools/bench_fps.py).  Rows at and beyond
  // threads x slots are never looked at (padding by the promise).
  const int cap = (counts && max_count > 0 && max_count < N) ? max_count : N;
  if (cap <= 2048) return launch<256, 8>(xyz, B, N, npoint, start_idx, out_idx, st, false, counts);
  if (cap <= 4096) return launch<256, 16>(xyz, B, N, npoint, start_idx, out_idx, st, false, counts);
  if (cap <= 8192) return launch<256, 32>(xyz, B, N, npoint, start_idx, out_idx, st, false, counts);
  if (cap <= 16384) return launch<512, 32>(xyz, B, N, npoint, start_idx, out_idx, st, false, counts);
  if (cap <= 32768) return launch<1024, 32>(xyz, B, N, npoint, start_idx, out_idx, st, false, counts);
  return PZN_EUNSUPPORTED;
}

PZN_EXPORT int pzn_fps_f32(const float* xyz, int B, int N, int npoint, const int64_t* start_idx,
                           int64_t* out_idx, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && start_idx && out_idx && B > 0 && N > 0 && npoint > 0);
  hipStream_t st = pzn_hip_stream(stream);
  // Fewer, fatter wavefronts: the round is a dependent chain (fetch the pick, update, wave reduction, barrier, re-reduce
  // the per-wave slots), and the cross-wave part grows with the wave count while the per-thread update is cheap.  Measured
  // per round at N = 2048: 1024 threads 1.35 us, 512 0.75, 256 0.62 (tools/bench_fps.py)
  if (N <= 64) return launch<64, 1>(xyz, B, N, npoint, start_idx, out_idx, st);
  if (N <= 128) return launch<128, 1>(xyz, B, N, npoint, start_idx, out_idx, st);
  if (N <= 256) return launch<256, 1>(xyz, B, N, npoint, start_idx, out_idx, st);
  if (N <= 512) return launch<256, 2>(xyz, B, N, npoint, start_idx, out_idx, st);
  if (N <= 1024) return launch<256, 4>(xyz, B, N, npoint, start_idx, out_idx, st);
  if (N <= 2048) return launch<256, 8>(xyz, B, N, npoint, start_idx, out_idx, st);
  if (N <= 4096) return launch<256, 16>(xyz, B, N, npoint, start_idx, out_idx, st);
  if (N <= 8192) return launch<512, 16>(xyz, B, N, npoint, start_idx, out_idx, st);
  if (N <= 16384) return launch<1024, 16>(xyz, B, N, npoint, start_idx, out_idx, st);
  if (N <= 32768) return launch<1024, 32>(xyz, B, N, npoint, start_idx, out_idx, st);
  return PZN_EUNSUPPORTED;
}
