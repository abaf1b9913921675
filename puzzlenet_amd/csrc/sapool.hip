// sapool.hip — input-gradient side of a set-abstraction level's backward, BY POINT (round 5).
//
// Level: rows h[(g,k),:] = relu(P'[idx[g,k],:] + Q[g,:]) (csrc/sapoint.hip), out[g,c] = max_k relu(h[(g,k),:] . W2[c,:] + b2[c])
// (model5_b.py:452-454 / 459-461).  The gradient that reaches a row is sparse: one (row, channel) hit per live channel,
//     dh[(g,k),:] = gate(g,k,:) * sum over the channels c with argmax[g,c] == k of dout[g,c] W2[c,:],
// and all the caller needs of dh are its per-point sums dP[j,:] = sum over the rows (g,k) that gathered point j of dh[(g,k),:]
// and three small reductions (dW1[:,0:3] += dh^T (xyz[j] - centre_g), db1 += column sums of dh).
//
// Rounds 2-4 ran this in two launches with dh in memory between them: pool_dgrad_kernel (csrc/poolbwd.hip) walked the groups,
// found the channels of every row with 32 x C2/64 ballots per group AND column slice, regenerated 32 gate rows per group and
// wrote the rows that exist (0.36-0.54 GB per launch); sa_point_l1_bwd_kernel (csrc/sapoint.hip) read them back through the
// inverse neighbour lists and summed them per point: 0.52 + 0.37 ms of a 7.5 ms step, 1 GB of HBM traffic, both kernels
// latency-bound (profiles/r5_pool_bwd_stamps.txt).  Here:
//   * pool_hits_kernel: ONCE per group (a wavefront each), the live (channel, gradient) pairs sorted by their arg-max row,
//     with the 33 row offsets — the ballots are paid once, not per column slice, and nothing else happens in that kernel;
//   * pool_point_kernel: the walk of sa_point_l1_bwd_kernel — (row, point) pairs sorted by point, 64 per wavefront batch,
//     running sum stored when the point changes — but the row is COMPUTED where it was loaded: the hit list of (g,k) (one
//     coalesced 8-byte load per hit, several rows in flight), one W2 row from LDS per hit (the W2 column slice of the
//     workgroup, [C2][128] floats), the gate from the P' row of the point and the Q row of the group.  dh is never written.
// Arithmetic per element is that of the two kernels it replaces (sum of the hits in ascending channel order, gate, sum of the
// rows of a point in list order).
#include <stdlib.h>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int PH_T = 256;        // pool_hits_kernel: 4 wavefronts, a group each
constexpr int PP_COLS = 128;     // columns of C1 per workgroup of pool_point_kernel
constexpr int RS_LD = 34;        // row offsets per group: 33 used (uint16)

// hits[g][0 .. rstart[g][32]) = {channel, gradient bits} of the channels with a non-zero gradient (out > 0: the ReLU of
// the pooled layer), sorted by (arg-max row, channel); rstart[g][k] = first hit of row k.
template <int NQ>  // C2 = 64 NQ
__global__ __launch_bounds__(PH_T) void pool_hits_kernel(const float* __restrict__ dout, const int32_t* __restrict__ argmax,
                                                         const float* __restrict__ out, int G, uint2* __restrict__ hits,
                                                         uint16_t* __restrict__ rstart) {
  constexpr int C2 = 64 * NQ;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * (PH_T / 64) + (threadIdx.x >> 6), nw = gridDim.x * (PH_T / 64);
  const uint64_t lt = (1ull << lane) - 1ull;
  for (int g = gw; g < G; g += nw) {
    int a[NQ];
    float gv[NQ];
    uint32_t rows = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const size_t o = (size_t)g * C2 + q * 64 + lane;
      const float go = out[o], gd = dout[o];
      gv[q] = go > 0.f ? gd : 0.f;
      a[q] = gv[q] != 0.f ? (argmax[o] & 31) : -1;      // a dead channel is no hit of any row
      rows |= a[q] >= 0 ? 1u << a[q] : 0u;
    }
    rows = (uint32_t)__builtin_amdgcn_readfirstlane((int)pzn::wave_or_u32_dpp(rows));      // rows with at least one hit
    uint2* hg = hits + (size_t)g * C2;
    int base = 0;          // wave-uniform
    int mystart = 0;       // lane k (and lane 32): rstart[k]
    int prev = 0;
    uint32_t todo = rows;
    while (todo) {
      const int k = __builtin_ctz(todo);
      todo &= todo - 1;
      // rows prev .. k start here (rows without hits are empty ranges)
      mystart = (lane >= prev && lane <= k) ? base : mystart;
      prev = k + 1;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bool on = a[q] == k;
        const uint64_t bal = __ballot(on);
        if (on) hg[base + __builtin_popcountll(bal & lt)] = make_uint2((uint32_t)(q * 64 + lane), __float_as_uint(gv[q]));
        base += __builtin_popcountll(bal);
      }
    }
    mystart = (lane >= prev && lane <= 32) ? base : mystart;
    if (lane <= 32) rstart[(size_t)g * RS_LD + lane] = (uint16_t)mystart;
  }
}

__device__ __forceinline__ float bcastf(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

struct PointArgs {
  const uint2* hits;        // [G][C2]
  const uint16_t* rstart;   // [G][RS_LD]
  const float* W2;          // [C2][C1]
  const float* Pp;          // [B*N][C1]
  const float* Q;           // [G][C1]
  const float* xyz;         // [B*N][3]
  const float* new_xyz;     // [G][3]
  const int32_t* off;       // [B][N+1] first list entry of every point of its cloud (pzn_knn_inverse_lists)
  const int32_t* rows;      // [B*S*32] row of its cloud, sorted by point (rows of a point in ascending order)
  const int32_t* pts;       // [B*S*32] point of its cloud
  float* dP;                // [B*N][C1]: every row written exactly once, by the wavefront that owns the point
  float* dW1;               // [C1][ldw]: columns 0..2 += dh^T (xyz - centre)
  float* db1;               // [C1] += column sums of dh (may be NULL)
  int B, N, S, C1, C2, ldw;
  int pc;                   // points per chunk (a wavefront's unit of work: whole points, ~64 list entries)
};

// NW wavefronts; lane l owns columns col0 + 2 l, 2 l + 1 of the workgroup's 128-column slice; GF rows in flight.
// A wavefront owns WHOLE points (round 6): chunks of `pc` consecutive points of a cloud, their list entries walked in
// order - the running sum of a point is stored once, by plain stores, when the walk moves on (points without a hit get their
// zeros from the same wavefront): no atomics on dP, no zero fill in front of the launch, and - the lists being sorted -
// the same summation order in every run.  (Round 5 walked 64-entry batches: a point whose list crossed a batch boundary
// was added atomically by two wavefronts into a zero-filled row, three writes per such row.)
template <int NW, int GF>
__global__ __launch_bounds__(NW * 64) void pool_point_kernel(PointArgs p) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];      // [C2][PP_COLS], then reused for the final sums
  __shared__ __attribute__((aligned(16))) uint2 hslot[NW][64];      // per wavefront: the hit pairs of the row it is summing
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col0 = blockIdx.y * PP_COLS, C1 = p.C1, C2 = p.C2;
  for (int f = tid; f < C2 * (PP_COLS / 4); f += NW * 64) {
    const int c = f / (PP_COLS / 4), q4 = (f % (PP_COLS / 4)) * 4;
    *reinterpret_cast<float4*>(&wlds[c * PP_COLS + q4]) = *reinterpret_cast<const float4*>(p.W2 + (size_t)c * C1 + col0 + q4);
  }
  __syncthreads();
  const int SK = p.S * 32;
  const int cpc = (p.N + p.pc - 1) / p.pc;                  // chunks per cloud
  const long nchunk = (long)p.B * cpc;
  // XCD-aware walk: workgroups x, x + 8, ... share an XCD; XCD x takes the x-th contiguous eighth of the chunks (whole
  // clouds: their P' / Q tables and hit lists are gathered through ONE L2)
  long ck = (long)blockIdx.x * NW + wave, nw = (long)gridDim.x * NW, ck_end = nchunk;
  if ((gridDim.x & 7) == 0) {
    const long per = (nchunk + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    ck = xcd * per + (long)(blockIdx.x >> 3) * NW + wave, nw = (long)(gridDim.x >> 3) * NW;
    ck_end = (xcd + 1) * per < nchunk ? (xcd + 1) * per : nchunk;
  }
  v2f ax = {0.f, 0.f}, ay = ax, az = ax, ab = ax;
  const float* wl = wlds + 2 * lane;
  for (; ck < ck_end; ck += nw) {
    const int b = (int)(ck / cpc);
    const int q0 = (int)(ck - (long)b * cpc) * p.pc, q1 = q0 + p.pc < p.N ? q0 + p.pc : p.N;
    const int32_t* ob = p.off + (size_t)b * (p.N + 1);
    const int e0 = ob[q0], e1 = ob[q1];                     // (scalar loads: the chunk's entries, in-cloud numbering)
    const int gq0 = b * p.N + q0, gq1 = b * p.N + q1;
    int cur = gq0;                                          // the point whose sum `acc` holds; every point < cur is written
    v2f acc = {0.f, 0.f};
    float* dcol = p.dP + col0 + 2 * lane;
    for (int eb = e0; eb < e1; eb += 64) {                  // usually one trip
      const int e = eb + lane;
      int gp = -1, grp = 0, nh = 0, hb = 0;      // global point, global group, hits of the row, first hit
      float dx = 0.f, dy = 0.f, dz = 0.f;
      if (e < e1) {
        const int rid = p.rows[(size_t)b * SK + e];
        grp = b * p.S + (rid >> 5);
        gp = b * p.N + p.pts[(size_t)b * SK + e];
        const uint16_t* rs = p.rstart + (size_t)grp * RS_LD + (rid & 31);
        const int r0 = rs[0], r1 = rs[1];
        nh = r1 - r0, hb = grp * C2 + r0;
        const float* qp = p.xyz + (size_t)gp * 3;
        const float* c = p.new_xyz + (size_t)grp * 3;
        dx = qp[0] - c[0], dy = qp[1] - c[1], dz = qp[2] - c[2];
      }
      // rows with hits, GF of them in flight: hit list (lane h holds hit h), Q row of the group and P' row of the point
      uint64_t todo = __ballot(nh > 0);
      while (todo) {      // wave-uniform
        int ru[GF];
        uint2 hv[GF];
        v2f qv[GF], pv[GF];
#pragma unroll
        for (int u = 0; u < GF; ++u) {
          ru[u] = todo ? __builtin_ctzll(todo) : -1;
          todo &= todo - 1;
        }
#pragma unroll
        for (int u = 0; u < GF; ++u) {
          hv[u] = make_uint2(0u, 0u);
          qv[u] = pv[u] = v2f{0.f, 0.f};
          if (ru[u] >= 0) {
            const int n_ = __builtin_amdgcn_readlane(nh, ru[u]), b_ = __builtin_amdgcn_readlane(hb, ru[u]);
            if (lane < n_) hv[u] = p.hits[(size_t)b_ + lane];
            qv[u] = *reinterpret_cast<const v2f*>(p.Q + (size_t)__builtin_amdgcn_readlane(grp, ru[u]) * C1 + col0 + 2 * lane);
            pv[u] = *reinterpret_cast<const v2f*>(p.Pp + (size_t)__builtin_amdgcn_readlane(gp, ru[u]) * C1 + col0 + 2 * lane);
          }
        }
#pragma unroll
        for (int u = 0; u < GF; ++u) {
          if (ru[u] < 0) continue;      // wave-uniform
          const int r = ru[u];
          const int n_ = __builtin_amdgcn_readlane(nh, r), b_ = __builtin_amdgcn_readlane(hb, r);
          v2f s0 = {0.f, 0.f}, s1 = s0;
          uint2 cur_h = hv[u];
          for (int h0 = 0; h0 < n_; h0 += 64) {      // (more than 64 channels on one row: rare)
            if (h0 > 0) cur_h = (h0 + lane < n_) ? p.hits[(size_t)b_ + h0 + lane] : make_uint2(0u, 0u);
            const int m_ = min(64, n_ - h0);
            // the row's (channel, gradient) pairs go through a wavefront-private 512 bytes of LDS and come back as BROADCAST
            // reads (every lane the same address): the two v_readlane per hit and their wait states were half the hit's issue
            // slots (same wavefront writes and reads: in order, no barrier)
            hslot[wave][lane] = cur_h;
            int h = 0;
            for (; h + 1 < m_; h += 2) {      // two hits per trip: two LDS reads in flight
              const uint4 two = *reinterpret_cast<const uint4*>(&hslot[wave][h]);
              const v2f w0 = *reinterpret_cast<const v2f*>(wl + (int)two.x * PP_COLS);
              const v2f w1 = *reinterpret_cast<const v2f*>(wl + (int)two.z * PP_COLS);
              // (the gradients as registers of their own: taken straight from the high halves of the loaded pairs the packed fma
              // gets op_sel on src1 - the form that is wrong beside AGPR-accumulator MFMAs, tests/test_isa_forms.py)
              float g0 = __uint_as_float(two.y), g1 = __uint_as_float(two.w);
              asm volatile("" : "+v"(g0), "+v"(g1));
              s0 += g0 * w0;
              s1 += g1 * w1;
            }
            if (h < m_) {
              const uint2 one = hslot[wave][h];
              float g0 = __uint_as_float(one.y);
              asm volatile("" : "+v"(g0));
              s0 += g0 * *reinterpret_cast<const v2f*>(wl + (int)one.x * PP_COLS);
            }
          }
          v2f row = s0 + s1;
          const v2f gate = pv[u] + qv[u];      // the forward's own expression: relu(P'[j] + Q[g]) is on where this is > 0
          row.x = gate.x > 0.f ? row.x : 0.f, row.y = gate.y > 0.f ? row.y : 0.f;
          const int pj = __builtin_amdgcn_readlane(gp, r);
          if (pj != cur) {  // wave-uniform: the walk has left `cur` (and every point up to pj - 1 has no further row)
            *reinterpret_cast<v2f*>(dcol + (size_t)cur * C1) = acc;
            ab += acc;
            acc = v2f{0.f, 0.f};
            for (int t = cur + 1; t < pj; ++t) *reinterpret_cast<v2f*>(dcol + (size_t)t * C1) = acc;
            cur = pj;
          }
          const float rx = bcastf(dx, r), ry = bcastf(dy, r), rz = bcastf(dz, r);
          acc += row;
          ax += rx * row, ay += ry * row, az += rz * row;
        }
      }
    }
    // the chunk's last points: the running sum, then zeros for the points nobody gathered or that won no channel
    *reinterpret_cast<v2f*>(dcol + (size_t)cur * C1) = acc;
    ab += acc;
    acc = v2f{0.f, 0.f};
    for (int t = cur + 1; t < gq1; ++t) *reinterpret_cast<v2f*>(dcol + (size_t)t * C1) = acc;
  }
  // dW1[:, 0:3] and db1: the four per-column sums of the workgroup's wavefronts meet in LDS (the W2 slice is not needed
  // any more), one set of atomics per workgroup
  __syncthreads();
  float* red = wlds;      // [NW][4][PP_COLS]
  {
    float* r = red + (size_t)wave * 4 * PP_COLS + 2 * lane;
    *reinterpret_cast<v2f*>(r) = ax;
    *reinterpret_cast<v2f*>(r + PP_COLS) = ay;
    *reinterpret_cast<v2f*>(r + 2 * PP_COLS) = az;
    *reinterpret_cast<v2f*>(r + 3 * PP_COLS) = ab;
  }
  __syncthreads();
  for (int f = tid; f < 4 * PP_COLS; f += NW * 64) {
    const int q = f / PP_COLS, c = f - q * PP_COLS;
    float t = 0.f;
    for (int w = 0; w < NW; ++w) t += red[(size_t)w * 4 * PP_COLS + q * PP_COLS + c];
    if (q < 3)
      atomicAdd(p.dW1 + (size_t)(col0 + c) * p.ldw + q, t);
    else if (p.db1)
      atomicAdd(p.db1 + col0 + c, t);
  }
}

}  // namespace

// workspace of pzn_sa_level_bwd_pt_f32: hit lists [B*S][C2] x 8 bytes + row offsets [B*S][34] x 2 bytes; the weight-gradient
// pass, which runs first, parks its workgroups' partial tiles in the same bytes (at B = 64 the two are the same 33.5 MB)
static size_t sa_bwd_pt_hits_bytes(size_t G, int C2) {
  return ((G * C2 * sizeof(uint2) + 255) / 256) * 256 + ((G * RS_LD * sizeof(uint16_t) + 255) / 256) * 256;
}
static size_t sa_bwd_pt_ws_bytes(size_t G, int C2) {
  const size_t hits = sa_bwd_pt_hits_bytes(G, C2);
  const size_t parts = (size_t)C2 * (256 * 128 + 256) * sizeof(float);      // >= pzn_pool_wgrad_ws_bytes for every C1: <= 256 tiles of [C2][128] + bias sums
  return hits > parts ? hits : parts;
}
PZN_EXPORT size_t pzn_sa_level_bwd_pt_workspace_bytes(int B, int S, int C2) {
  if (B <= 0 || S <= 0 || C2 <= 0) return 0;
  return sa_bwd_pt_ws_bytes((size_t)B * S, C2);
}

// Backward of the pooled level behind pzn_sa_level_fwd_*: dW2, db2 (overwritten, or added to when accumulate), the per-point
// sums dP[B*N, C1] of the rows' gradient (overwritten; dh itself is never written), dW1[:, 0:3] += dh^T (xyz[idx] - centre)
// and db1 += column sums of dh (dW1[C1, 3+D] and db1[C1] are ADDED to; db1 may be NULL).  rows / pts: the inverse neighbour
// lists of idx (pzn_knn_inverse_lists).  PZN_EUNSUPPORTED for shapes the kernels do not take (C1 % 128, C2 not 64 / 128 / 256).
PZN_EXPORT int pzn_sa_level_bwd_pt_f32(const float* dout, const int32_t* argmax, const float* out, const float* W2,
                                       const float* Pp, const float* Q, const int64_t* idx, const float* xyz,
                                       const float* new_xyz, const int32_t* off, const int32_t* rows, const int32_t* pts, int B, int N,
                                       int S, int D,
                                       int C1, int C2, float* dP, float* dW2, float* db2, float* dW1, float* db1, int accumulate,
                                       void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(dout && argmax && out && W2 && Pp && Q && idx && xyz && new_xyz && off && rows && pts && dP && dW2 && db2 && dW1 && workspace);
  PZN_CHECK_ARG(B > 0 && N > 0 && S > 0 && D >= 0 && C1 > 0 && C2 > 0 && (long)B * N < 2147483647L);
  PZN_CHECK_ARG((long)B * S * 32 < 2147483647L && (long)B * S * C2 < 2147483647L);
  if (C1 % PP_COLS != 0 || !(C2 == 64 || C2 == 128 || C2 == 256)) return PZN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(W2) & 15) || (reinterpret_cast<uintptr_t>(Pp) & 7) || (reinterpret_cast<uintptr_t>(Q) & 7) ||
      (reinterpret_cast<uintptr_t>(dP) & 7) || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return PZN_EUNSUPPORTED;
  hipStream_t st = pzn_hip_stream(stream);
  if (!accumulate) {
    if (pzn_zero_async(dW2, (size_t)C2 * C1, st) != PZN_OK) return PZN_ELAUNCH;
    if (pzn_zero_async(db2, (size_t)C2, st) != PZN_OK) return PZN_ELAUNCH;
  }
  const int G = B * S;
  // weight gradient of the pooled layer: the sparse pass of csrc/poolbwd.hip on regenerated rows
  PznGateSource gs{Pp, idx, Q, N, S};
  int rc = pzn_pool_wgrad_sparse(dout, argmax, out, nullptr, dW2, db2, G, C1, C2, st, &gs, workspace, sa_bwd_pt_ws_bytes((size_t)G, C2));
  if (rc != PZN_OK) return rc;
  uint2* hits = static_cast<uint2*>(workspace);
  uint16_t* rstart = reinterpret_cast<uint16_t*>(static_cast<unsigned char*>(workspace) + (((size_t)G * C2 * sizeof(uint2) + 255) / 256) * 256);
  const int hb = (G + PH_T / 64 - 1) / (PH_T / 64);
  const dim3 hgrid((unsigned)(hb < 4096 ? hb : 4096));
  if (C2 == 64)
    PZN_LAUNCH((pool_hits_kernel<1>), hgrid, dim3(PH_T), 0, st, dout, argmax, out, G, hits, rstart);
  else if (C2 == 128)
    PZN_LAUNCH((pool_hits_kernel<2>), hgrid, dim3(PH_T), 0, st, dout, argmax, out, G, hits, rstart);
  else
    PZN_LAUNCH((pool_hits_kernel<4>), hgrid, dim3(PH_T), 0, st, dout, argmax, out, G, hits, rstart);
  // points per chunk: ~64 list entries (S * 32 / N per point on average)
  int pc = (int)(64L * N / ((long)S * 32));
  pc = pc < 1 ? 1 : (pc > 32 ? 32 : pc);
  PointArgs a{hits, rstart, W2, Pp, Q, xyz, new_xyz, off, rows, pts, dP, dW1, db1, B, N, S, C1, C2, 3 + D, pc};
  const int ny = C1 / PP_COLS;
  const size_t lds = (size_t)C2 * PP_COLS * sizeof(float);      // >= the final sums' 16 x 4 x 128 floats for C2 >= 64
  const int per_cu = 1;      // 16 wavefronts of 92 registers: one workgroup per CU whatever the slice's size
  int gx = 256 * per_cu / ny;
  if (gx < 8) gx = 8;
  gx &= ~7;
  const dim3 grid((unsigned)gx, (unsigned)ny);
  // eight rows in flight per wavefront (4 and 16 measured the same or slower)
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_point_kernel<16, 8>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PZN_ELAUNCH;
  PZN_LAUNCH((pool_point_kernel<16, 8>), grid, dim3(1024), lds, st, a);
  PZN_RETURN_LAUNCH_STATUS();
}
