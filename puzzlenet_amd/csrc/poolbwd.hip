// poolbwd.hip — backward of "linear + ReLU + max over the K = 32 neighbours" (model5_b.py:452-454 /
// 459-461) as the sparse problem it is.
//
// The gradient that reaches the pre-pool activations has ONE non-zero per (centroid g, channel c):
//     dy[g*32 + k, c] = (argmax[g,c] == k && out[g,c] > 0) ? dOut[g,c] : 0.
// Written as dense GEMMs (dh = dy W, dW = dy^T h) 31 of every 32 multiply-adds are by zero.  Here
//     dh[g*32 + a(g,c), :] += dOut[g,c] * W[c, :]          (C2 row-axpys per group instead of 32*C2)
//     dW[c, :]             += dOut[g,c] * h[g*32 + a(g,c), :]
//     db[c]                += dOut[g,c]
// which is 32x fewer flops, so the pass is bound by streaming h in (and dh out) once.
//
// Two kernels, both walking groups g = blockIdx.x, += gridDim.x over a 128-column slice (blockIdx.y) of the
// C1 hidden columns, lane l holding columns 2l, 2l+1:
//  * pool_wgrad_kernel: 16 (or 8) wavefronts, wave w owns CPW = C2/16 channels whose dW accumulators stay in registers
//    for the whole walk.  Per group the 32 x 128 tile of h is staged in LDS (double-buffered, next tile
//    prefetched into registers), the wave's CPW (argmax, gradient) pairs are read as one vector and
//    broadcast with v_readlane, then per channel: one ds_read_b64 of the arg-max row + one packed fma.
//  * pool_dgrad_kernel: 16 wavefronts, each walking whole groups on its own (no barrier in the walk, no LDS
//    float atomics — those measured ~200 cycles per wave-instruction here).  The W slice [C2][128] sits in
//    LDS; each lane holds the (argmax, gradient) pairs of channels lane, lane+64, ...; for a row k the
//    channels that selected it come out of a v_cmp ballot and are visited two bits at a time: v_readlane of
//    the gradient, ds_read_b64 of the W row, packed fma.  The finished row is ReLU-masked with h and stored
//    straight from registers.  Work per group is exactly C2 hits however skewed the arg-max rows are (in the
//    model the nearest neighbours win most channels; a rows-per-wave split was 3x slower there).
#include <stdlib.h>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int PB_T = 512;     // threads per workgroup
constexpr int PB_W = 8;       // wavefronts
constexpr int PB_COLS = 128;  // hidden columns per workgroup

struct PoolBwdArgs {
  const float* dout;      // [G, C2]
  const int32_t* argmax;  // [G, C2], values 0..31
  const float* out;       // [G, C2] pooled output (gradient flows only where it is > 0)
  const float* W;         // [C2, C1]           (DGRAD)
  const float* h;         // [G*32, C1]         (WGRAD operand; DGRAD ReLU mask, may be NULL there)
  float* dh;              // [G*32, C1]         (DGRAD)
  float* dW;              // [C2, C1] += ...    (WGRAD)
  float* db;              // [C2] += ...        (WGRAD, may be NULL)
  int G, C1, C2;
  // DGRAD, per-point first layer (csrc/sapoint.hip): h[r,:] = relu(W1[:,0:3] (xyz[idx[r]] - centre) + P[idx[r],:] + b1) is
  // a pure function of L2-resident data, so the ReLU gate is REGENERATED with the forward's own expression instead
  // of streaming h from HBM (537 MB per launch).  gP == NULL: the gate comes from h.
  const float* gP;        // [B*N, C1]
  const int64_t* gidx;    // [G*32]
  const float* gxyz;      // [B*N, 3]
  const float* gnew;      // [G, 3]
  const float* gW1;       // [C1, gldw] (columns 0..2)
  const float* gb1;       // [C1] (may be NULL)
  int gldw, gN, gS;
  // second form (pzn_sa_prep_f32): h[(g,k),:] = relu(gP[(g / gS) * gN + gidx[g*32+k], :] + gQ[g,:]).  WGRAD regenerates its
  // rows from it (h == NULL); DGRAD its gate, and accumulates what flows through Q: gdW1x[c,0:3] -= dq[g,c] centre_g,
  // gdb1[c] += dq[g,c], dq[g,:] = sum_k dh[(g,k),:]
  const float* gQ;        // [G, C1]
  float* gdW1x;           // [C1, gldw] or NULL
  float* gdb1;            // [C1] or NULL
  // DGRAD, optional: rowmask[g] bit k = some channel with a non-zero gradient selected row k.  About half the rows of a
  // level-1 group (a third at level 2) win no channel: their dh row is exactly zero.  With a mask the kernel writes only
  // row pairs that hold a non-zero row (and fetches gates / runs hit loops only for those), and the list sum reads only
  // rows whose bit is set: ~40 % of the 2.1 GB of dh per step are neither written nor read.
  uint32_t* rowmask;
#ifdef POOL_STAMPS
  unsigned long long* stamps;   // diagnostic build (tools/pool_stamps.py): [launch % 16][workgroup][8] phase sums of wavefront 0
#endif
};

#ifdef POOL_STAMPS
// phase sums in core clocks (s_memtime), wavefront 0 of every workgroup: PS_MARK(i) adds the time since the previous mark to
// phase i.  Record: 8 uint64 per workgroup {phases 0..5, groups walked, total}.
unsigned long long* g_pool_stamps = nullptr;
int g_pool_launch = 0;
constexpr int PS_WGS = 1024, PS_LAUNCHES = 16;
int g_pool_log[PS_LAUNCHES][4];      // {kind: 0 wgrad / 1 dgrad, G, C1, C2} of the launch that filled a slot
#define PS_BEGIN() unsigned long long ps_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ps_last_ = __builtin_amdgcn_s_memtime(); const unsigned long long ps_t0_ = ps_last_
#define PS_MARK(I) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ps_[I] += t_ - ps_last_; ps_last_ = t_; } while (0)
#define PS_COUNT() ps_[6] += 1
#define PS_END(P)                                                                                     \
  do {                                                                                                \
    const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                         \
    if (threadIdx.x == 0 && (P).stamps && wg_ < PS_WGS) {                                             \
      ps_[7] = __builtin_amdgcn_s_memtime() - ps_t0_;                                                 \
      for (int i_ = 0; i_ < 8; ++i_) (P).stamps[(size_t)wg_ * 8 + i_] = ps_[i_];                      \
    }                                                                                                 \
  } while (0)
#else
#define PS_BEGIN() do {} while (0)
#define PS_MARK(I) do {} while (0)
#define PS_COUNT() do {} while (0)
#define PS_END(P) do {} while (0)
#endif

__device__ __forceinline__ float4 pb_add_relu(float4 a, float4 q) {
  return make_float4(fmaxf(a.x + q.x, 0.f), fmaxf(a.y + q.y, 0.f), fmaxf(a.z + q.z, 0.f), fmaxf(a.w + q.w, 0.f));
}

// NWV wavefronts per workgroup (8 or 16), wave w owns CPW = C2 / NWV channels.  16 wavefronts: the whole 32 x 128 tile is
// staged in one pass (one 16-byte piece per thread) and four wavefronts per SIMD hide the LDS latency of the hit loop
// (8 wavefronts = 2 per SIMD waited on a dependent ds_read 37 % of the time); the workgroup count, and with it the
// number of atomic adds at the end, stays the same.
template <int CPW, int NWV>
__global__ __launch_bounds__(NWV * 64) void pool_wgrad_kernel(PoolBwdArgs p) {
  constexpr int T = NWV * 64;
  constexpr bool TWO = NWV == 8;      // two staging rows per thread
  __shared__ __attribute__((aligned(16))) float hbuf[2][32][PB_COLS];
  PS_BEGIN();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col0 = blockIdx.y * PB_COLS;
  const int ch = wave * CPW + (lane < CPW ? lane : 0);  // this lane's channel in the per-group vector loads

  v2f acc[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) acc[c] = v2f{0.f, 0.f};
  float dbacc = 0.f;

  // staging map for the 32 x 128 tile: row srow (and srow + 16 with 8 wavefronts), 16-byte column scol
  const int srow = tid >> 5, scol = (tid & 31) * 4;
  // Two register sets = two groups in flight.  The loads are unconditional (group index clamped) and the
  // loop is unrolled by the two sets so that the compiler's vmcnt bookkeeping stays exact (a branch around
  // a load makes it fall back to vmcnt(0), i.e. to a prefetch distance of nothing).
  float4 pa0, pa1 = make_float4(0.f, 0.f, 0.f, 0.f), pb0, pb1 = pa1, qa = pa1, qb = pa1;
  int ava, avb;
  float gva, gvb;
  // regenerated rows (p.gQ): the point indices of this thread's rows, looked up one round ahead of the row loads
  // they address (a dependent idx -> row chain inside one round would cost a memory latency per group)
  int ja0 = 0, ja1 = 0, jb0 = 0, jb1 = 0;
  const bool regen = p.gQ != nullptr;
#define PB_IDX(gg, j0_, j1_)                                                         \
  do {                                                                              \
    const int g_ = (gg) < p.G ? (gg) : p.G - 1;                                     \
    const int64_t* ip = p.gidx + (size_t)g_ * 32 + srow;                            \
    j0_ = (int)ip[0];                                                               \
    if (TWO) j1_ = (int)ip[16];                                                     \
  } while (0)
#define PB_ISSUE(gg, x0, x1, q_, av_, gv_, j0_, j1_)                                \
  do {                                                                              \
    const int g_ = (gg) < p.G ? (gg) : p.G - 1;                                     \
    if (regen) {                                                                    \
      const size_t pb_ = (size_t)(g_ / p.gS) * p.gN;                                \
      x0 = *reinterpret_cast<const float4*>(p.gP + (pb_ + j0_) * p.C1 + col0 + scol); \
      if (TWO) x1 = *reinterpret_cast<const float4*>(p.gP + (pb_ + j1_) * p.C1 + col0 + scol); \
      q_ = *reinterpret_cast<const float4*>(p.gQ + (size_t)g_ * p.C1 + col0 + scol); \
    } else {                                                                        \
      const float* hp = p.h + ((size_t)g_ * 32 + srow) * p.C1 + col0 + scol;        \
      x0 = *reinterpret_cast<const float4*>(hp);                                    \
      if (TWO) x1 = *reinterpret_cast<const float4*>(hp + (size_t)16 * p.C1);       \
    }                                                                               \
    const size_t o = (size_t)g_ * p.C2 + ch;                                        \
    const int a = p.argmax[o];                                                      \
    const float go = p.out[o], gd = p.dout[o];                                      \
    av_ = a & 31;                                                                   \
    gv_ = (lane < CPW && go > 0.f) ? gd : 0.f;                                      \
  } while (0)
#define PB_GROUP(buf, x0, x1, q_, av_, gv_, j0_, j1_, gnext)                        \
  do {                                                                              \
    PS_MARK(0); /* (loop control, previous group's tail) */                        \
    if (regen) {                                                                    \
      x0 = pb_add_relu(x0, q_);                                                     \
      if (TWO) x1 = pb_add_relu(x1, q_);                                            \
    }                                                                               \
    *reinterpret_cast<float4*>(&hbuf[buf][srow][scol]) = x0;                        \
    if (TWO) *reinterpret_cast<float4*>(&hbuf[buf][(srow + 16) & 31][scol]) = x1;   \
    const int av = av_;                                                             \
    const float gv = gv_;                                                           \
    PS_MARK(1); /* rows landed (vmcnt), add + ReLU, tile write */                   \
    __syncthreads(); /* one barrier per group: the tile two groups back is free again by construction */ \
    PS_MARK(2); /* barrier */                                                       \
    PB_ISSUE(gnext, x0, x1, q_, av_, gv_, j0_, j1_);  /* rows of gnext: their indices arrived a round ago */ \
    if (regen) PB_IDX((gnext) + 2 * gs_, j0_, j1_);   /* indices for the round after */ \
    PS_MARK(3); /* issue of the next loads (address arithmetic; waits for the indices) */ \
    PS_COUNT();                                                                     \
    dbacc += gv;                                                                    \
    _Pragma("unroll") for (int c = 0; c < CPW; ++c) {                               \
      const int a = __builtin_amdgcn_readlane(av, c);                               \
      const float gs = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gv), c)); \
      v2f hr = v2f{0.f, 0.f};                                                       \
      /* wave-uniform: a dead channel (ReLU off, ~45 % of them) reads nothing; with 16 channels per wavefront the branches */ \
      /* cost more registers than the 128 a 1024-thread workgroup has (the accumulators went to scratch: 30x slower) */      \
      if (CPW > 8 || gs != 0.f) hr = *reinterpret_cast<const v2f*>(&hbuf[buf][a][2 * lane]); \
      acc[c] += gs * hr;                                                            \
    }                                                                               \
    PS_MARK(4); /* hit loop */                                                      \
  } while (0)
  // XCD-aware walk (round 3): workgroups x, x + 8, ... share an XCD; XCD x takes the contiguous eighth [x Gx, (x+1) Gx) of
  // the groups - whole clouds - so that a cloud's per-point table is gathered through ONE L2 (PMC: the strided walk fetched
  // 290-400 MB per launch for tables of 34-67 MB)
  int g_first = blockIdx.x, gs_ = gridDim.x, g_end = p.G;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, gx8 = (p.G + 7) >> 3;
    g_first = xcd * gx8 + (int)(blockIdx.x >> 3), gs_ = gridDim.x >> 3;
    g_end = (xcd + 1) * gx8 < p.G ? (xcd + 1) * gx8 : p.G;
  }
  if (regen) {
    PB_IDX(g_first, ja0, ja1);
    PB_IDX(g_first + gs_, jb0, jb1);
  }
  PB_ISSUE(g_first, pa0, pa1, qa, ava, gva, ja0, ja1);
  PB_ISSUE(g_first + gs_, pb0, pb1, qb, avb, gvb, jb0, jb1);
  if (regen) {
    PB_IDX(g_first + 2 * gs_, ja0, ja1);
    PB_IDX(g_first + 3 * gs_, jb0, jb1);
  }
  for (int g = g_first; g < g_end; g += 2 * gs_) {
    PB_GROUP(0, pa0, pa1, qa, ava, gva, ja0, ja1, g + 2 * gs_);
    if (g + gs_ >= g_end) break;
    PB_GROUP(1, pb0, pb1, qb, avb, gvb, jb0, jb1, g + 3 * gs_);
  }
#undef PB_GROUP
#undef PB_ISSUE
#undef PB_IDX

#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    float* o = p.dW + (size_t)(wave * CPW + c) * p.C1 + col0 + 2 * lane;
    atomicAdd(o, acc[c].x);
    atomicAdd(o + 1, acc[c].y);
  }
  if (p.db && blockIdx.y == 0 && lane < CPW) atomicAdd(p.db + wave * CPW + lane, dbacc);
  PS_MARK(5); /* atomics */
  PS_END(p);
  (void)T;
}

constexpr int PD_T = 1024;  // dgrad: 16 wavefronts, each walking whole groups
#ifndef PD_ROWS_N
#define PD_ROWS_N 4
#endif
constexpr int PD_ROWS = PD_ROWS_N;  // rows of a group per trip of the row loop (gate rows in flight, independent hit loops)

template <int NQ>  // C2 = 64 * NQ
__global__ __launch_bounds__(PD_T) void pool_dgrad_kernel(PoolBwdArgs p) {
  extern __shared__ __attribute__((aligned(16))) float wlds[];  // [C2][PB_COLS]
  PS_BEGIN();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col0 = blockIdx.y * PB_COLS;
  const int C2 = NQ * 64;
  for (int f = tid; f < C2 * (PB_COLS / 4); f += PD_T) {
    const int c = f / (PB_COLS / 4), q4 = (f % (PB_COLS / 4)) * 4;
    *reinterpret_cast<float4*>(&wlds[c * PB_COLS + q4]) =
        *reinterpret_cast<const float4*>(p.W + (size_t)c * p.C1 + col0 + q4);
  }
  __syncthreads();  // the only barrier: from here on every wavefront walks its own groups
  PS_MARK(5); /* W slice into LDS */

  // XCD-aware walk (see pool_wgrad_kernel): the wavefronts of XCD x stride through the x-th contiguous eighth of the groups
  int gw = blockIdx.x * (PD_T / 64) + wave, nw = gridDim.x * (PD_T / 64), g_end = p.G;
  if ((gridDim.x & 7) == 0) {
    const int xcd = blockIdx.x & 7, gx8 = (p.G + 7) >> 3;
    gw = xcd * gx8 + (int)(blockIdx.x >> 3) * (PD_T / 64) + wave, nw = (int)(gridDim.x >> 3) * (PD_T / 64);
    g_end = (xcd + 1) * gx8 < p.G ? (xcd + 1) * gx8 : p.G;
  }
  const bool qform = p.gQ != nullptr;      // h = relu(P'[idx] + Q[g]) (pzn_sa_prep_f32); else the round-1 expression
  float gwx0 = 0.f, gwy0 = 0.f, gwz0 = 0.f, gbb0 = 0.f, gwx1 = 0.f, gwy1 = 0.f, gwz1 = 0.f, gbb1 = 0.f;
  if (p.gP && !qform) {  // first-layer xyz weights and bias of this lane's two columns
    const int c = col0 + 2 * lane;
    gwx0 = p.gW1[(size_t)c * p.gldw], gwy0 = p.gW1[(size_t)c * p.gldw + 1], gwz0 = p.gW1[(size_t)c * p.gldw + 2];
    gwx1 = p.gW1[(size_t)(c + 1) * p.gldw], gwy1 = p.gW1[(size_t)(c + 1) * p.gldw + 1];
    gwz1 = p.gW1[(size_t)(c + 1) * p.gldw + 2];
    if (p.gb1) gbb0 = p.gb1[c], gbb1 = p.gb1[c + 1];
  }
  // what flows through Q = b1 - W1x centre: sums over this wave's groups, met in LDS at the end (one set of atomics per
  // workgroup): dW1x[c, :] -= dq[g, c] centre_g, db1[c] += dq[g, c]
  v2f sq_b = v2f{0.f, 0.f}, sq_x = sq_b, sq_y = sq_b, sq_z = sq_b;
  int av_n[NQ];
  float gv_n[NQ];
#define PD_PREFETCH(gg)                                             \
  do {                                                              \
    _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                \
      size_t o = (size_t)(gg) * C2 + q * 64 + lane;                 \
      float go = p.out[o], gd = p.dout[o];                          \
      gv_n[q] = go > 0.f ? gd : 0.f;                                \
      av_n[q] = gv_n[q] != 0.f ? p.argmax[o] : -1; /* (a dead channel - ReLU off, ~45 % of them - is no hit of any row) */ \
    }                                                               \
  } while (0)
  if (gw < g_end) PD_PREFETCH(gw);

  for (int g = gw; g < g_end; g += nw) {
    int av[NQ];
    float gv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) av[q] = av_n[q], gv[q] = gv_n[q];
    PS_MARK(0); /* this group's (arg-max, gradient) vectors landed */
    PS_COUNT();
    if (g + nw < g_end) PD_PREFETCH(g + nw);
    uint32_t rmask = 0xffffffffu;          // wave-uniform
    if (p.rowmask) {
      uint32_t mm = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) mm |= gv[q] != 0.f ? 1u << (av[q] & 31) : 0u;
      rmask = (uint32_t)__builtin_amdgcn_readfirstlane((int)pzn::wave_or_u32_dpp(mm));
      if (blockIdx.y == 0 && lane == 0) p.rowmask[g] = rmask;
    }
    const size_t row0 = ((size_t)g * 32) * p.C1 + col0 + 2 * lane;
    // regenerated gate: lane l (mod 32) fetches row l's point (and centre offset) once per group
    int gprow = 0;
    float gdx = 0.f, gdy = 0.f, gdz = 0.f;
    v2f qv = v2f{0.f, 0.f};
    float cgx = 0.f, cgy = 0.f, cgz = 0.f;
    if (p.gP) {
      const int rl = lane & 31;
      const long b = g / p.gS;
      const int j = (int)p.gidx[(size_t)g * 32 + rl];
      gprow = (int)(b * p.gN + j);
      if (qform) {
        qv = *reinterpret_cast<const v2f*>(p.gQ + (size_t)g * p.C1 + col0 + 2 * lane);
        const float* c = p.gnew + (size_t)g * 3;
        cgx = c[0], cgy = c[1], cgz = c[2];
      } else {
        const float* pq = p.gxyz + ((size_t)b * p.gN + j) * 3;
        const float* c = p.gnew + (size_t)g * 3;
        gdx = pq[0] - c[0], gdy = pq[1] - c[1], gdz = pq[2] - c[2];
      }
    }
    auto gate_src = [&](int k) {  // the two P values of row k for this lane's columns
      const int pr = __builtin_amdgcn_readlane(gprow, k);
      return *reinterpret_cast<const v2f*>(p.gP + (size_t)pr * p.C1 + col0 + 2 * lane);
    };
    auto gate_of = [&](v2f pv, int k) {  // same expression as the forward: the sign is the forward's
      if (qform) return pv + qv;
      const float rx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gdx), k));
      const float ry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gdy), k));
      const float rz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gdz), k));
      const float t0 = fmaf(gwz0, rz, fmaf(gwy0, ry, gwx0 * rx)) + pv.x + gbb0;
      const float t1 = fmaf(gwz1, rz, fmaf(gwy1, ry, gwx1 * rx)) + pv.y + gbb1;
      return v2f{t0, t1};
    };
    v2f dq = v2f{0.f, 0.f};
    auto gate_raw = [&](int k) {  // what the gate of row k is computed from (its P row, or its h row, or nothing)
      if (p.gP) return gate_src(k);
      if (p.h) return *reinterpret_cast<const v2f*>(p.h + row0 + (size_t)k * p.C1);
      return v2f{1.f, 1.f};
    };
    auto row_acc = [&](int k) {  // sum of the hits of row k: exactly C2 hits per group over the 32 rows, whatever the skew
      v2f acc0 = v2f{0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        uint64_t m = __ballot(av[q] == k);
        while (m) {  // two hits per trip so that two LDS reads are in flight
          const int b0 = __builtin_ctzll(m);
          m &= m - 1;
          const bool two = m != 0;
          const int b1 = two ? __builtin_ctzll(m) : b0;
          m &= m - 1;
          const int gi = __builtin_bit_cast(int, gv[q]);
          const float g0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(gi, b0));
          float g1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(gi, b1));
          g1 = two ? g1 : 0.f;
          v2f w0 = *reinterpret_cast<const v2f*>(&wlds[(q * 64 + b0) * PB_COLS + 2 * lane]);
          v2f w1 = *reinterpret_cast<const v2f*>(&wlds[(q * 64 + b1) * PB_COLS + 2 * lane]);
          acc0 += g0 * w0;
          acc1 += g1 * w1;
        }
      }
      return acc0 + acc1;
    };
    // Rows in fours, stored in pairs: lane l owns columns 2l, 2l+1 of every row; after two rows the lanes of a pair (2i, 2i+1) swap one
    // half each, so that the even lane holds four consecutive columns of row k and the odd lane those of row k+1 — the
    // row stores are 16 bytes per lane instead of 8 (8-byte stores run at 0.54-0.70 of the 16-byte rate; without the
    // store the level-1 launch takes 0.122 of its 0.192 ms).
    const int odd = lane & 1;
    float* drow = p.dh + ((size_t)g * 32 + odd) * p.C1 + col0 + 4 * (lane >> 1);
    PS_MARK(1); /* row mask, the group's indices / Q row / centre requested (and the index waited for) */
    for (int k = 0; k < 32; k += PD_ROWS) {  // PD_ROWS rows per trip: their gate rows are requested first and arrive under the hit loops
      const uint32_t m4 = (rmask >> k) & ((1u << PD_ROWS) - 1u);
      if (m4 == 0) continue;                 // no row of the trip won a channel: nothing fetched, nothing written
      v2f hg[PD_ROWS], ar[PD_ROWS];
#pragma unroll
      for (int u = 0; u < PD_ROWS; ++u) hg[u] = (m4 >> u) & 1u ? gate_raw(k + u) : v2f{0.f, 0.f};
      PS_MARK(2); /* gate rows requested */
#pragma unroll
      for (int u = 0; u < PD_ROWS; ++u) ar[u] = (m4 >> u) & 1u ? row_acc(k + u) : v2f{0.f, 0.f};
      PS_MARK(3); /* hit loops */
#pragma unroll
      for (int u = 0; u < PD_ROWS; ++u) {
        if (p.gP) hg[u] = gate_of(hg[u], k + u);
        ar[u].x = hg[u].x > 0.f ? ar[u].x : 0.f, ar[u].y = hg[u].y > 0.f ? ar[u].y : 0.f;
        dq += ar[u];
      }
#pragma unroll
      for (int u = 0; u < PD_ROWS; u += 2) {
        if (((m4 >> u) & 3u) == 0) continue;   // (a pair is one store: written when either of its rows is non-zero)
        const v2f a0 = ar[u], a1 = ar[u + 1];
        // the half the partner lane stores goes across; the other half stays
        const float sx_ = odd ? a0.x : a1.x, sy_ = odd ? a0.y : a1.y;
        const float rx_ = __builtin_bit_cast(float, pzn::xor_lane<1>(__builtin_bit_cast(uint32_t, sx_)));
        const float ry_ = __builtin_bit_cast(float, pzn::xor_lane<1>(__builtin_bit_cast(uint32_t, sy_)));
        float4 o4;
        o4.x = odd ? rx_ : a0.x, o4.y = odd ? ry_ : a0.y;      // even lane: row k, columns 4i..4i+3 = own pair + partner's
        o4.z = odd ? a1.x : rx_, o4.w = odd ? a1.y : ry_;      // odd lane: row k+1, partner's pair + own
        *reinterpret_cast<float4*>(drow + (size_t)(k + u) * p.C1) = o4;
      }
      PS_MARK(4); /* gate rows landed, mask, lane-pair exchange, stores issued */
    }
    if (qform) sq_b += dq, sq_x += cgx * dq, sq_y += cgy * dq, sq_z += cgz * dq;
  }
#undef PD_PREFETCH
  if (qform && (p.gdW1x || p.gdb1)) {  // the W slice in LDS is not needed any more: its first 4 x 128 floats take the sums
    __syncthreads();
    for (int f = tid; f < 4 * PB_COLS; f += PD_T) wlds[f] = 0.f;
    __syncthreads();
    atomicAdd(&wlds[0 * PB_COLS + 2 * lane], sq_b.x), atomicAdd(&wlds[0 * PB_COLS + 2 * lane + 1], sq_b.y);
    atomicAdd(&wlds[1 * PB_COLS + 2 * lane], sq_x.x), atomicAdd(&wlds[1 * PB_COLS + 2 * lane + 1], sq_x.y);
    atomicAdd(&wlds[2 * PB_COLS + 2 * lane], sq_y.x), atomicAdd(&wlds[2 * PB_COLS + 2 * lane + 1], sq_y.y);
    atomicAdd(&wlds[3 * PB_COLS + 2 * lane], sq_z.x), atomicAdd(&wlds[3 * PB_COLS + 2 * lane + 1], sq_z.y);
    __syncthreads();
    for (int f = tid; f < 4 * PB_COLS; f += PD_T) {
      const int q = f / PB_COLS, c = col0 + f % PB_COLS;
      const float v = wlds[f];
      if (q == 0) {
        if (p.gdb1) atomicAdd(p.gdb1 + c, v);
      } else if (p.gdW1x) {
        atomicAdd(p.gdW1x + (size_t)c * p.gldw + (q - 1), -v);
      }
    }
  }
  PS_END(p);
}

int launch_wgrad(const PoolBwdArgs& p, hipStream_t st) {
  const int ny = p.C1 / PB_COLS;
  static const int total = [] { const char* e = getenv("PZN_POOL_WGRAD_GRID"); return e ? atoi(e) : 256; }();  // tuning aid (one workgroup per CU measured best)
  int gx = total / ny;  // workgroups in flight; each ends with C2*128 atomic adds into dW
  if (gx > p.G) gx = p.G;
  static const int w16 = [] { const char* e = getenv("PZN_POOL_WGRAD_W16"); return e ? atoi(e) : 1; }();  // tuning aid
  const dim3 grid((unsigned)gx, (unsigned)ny);
  if (w16 && p.C2 % 16 == 0 && p.C2 / 16 >= 4) {      // 16 wavefronts, C2 / 16 channels each
    const int cpw = p.C2 / 16;
    if (cpw == 4)
      PZN_LAUNCH((pool_wgrad_kernel<4, 16>), grid, dim3(1024), 0, st, p);
    else if (cpw == 8)
      PZN_LAUNCH((pool_wgrad_kernel<8, 16>), grid, dim3(1024), 0, st, p);
    else
      PZN_LAUNCH((pool_wgrad_kernel<16, 16>), grid, dim3(1024), 0, st, p);
    PZN_RETURN_LAUNCH_STATUS();
  }
  const dim3 block(PB_T);
  const int cpw = p.C2 / PB_W;
  if (cpw == 8)
    PZN_LAUNCH((pool_wgrad_kernel<8, 8>), grid, block, 0, st, p);
  else if (cpw == 16)
    PZN_LAUNCH((pool_wgrad_kernel<16, 8>), grid, block, 0, st, p);
  else
    PZN_LAUNCH((pool_wgrad_kernel<32, 8>), grid, block, 0, st, p);
  PZN_RETURN_LAUNCH_STATUS();
}

template <int NQ>
int launch_dgrad_nq(const PoolBwdArgs& p, hipStream_t st) {
  const int ny = p.C1 / PB_COLS;
  const size_t lds = (size_t)p.C2 * PB_COLS * sizeof(float);
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void*>(&pool_dgrad_kernel<NQ>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PZN_ELAUNCH;
  const int per_cu = lds <= 64 * 1024 ? 2 : 1;  // 16 waves per workgroup, 32 per CU
  int gx = 256 * per_cu / ny;
  if (gx < 1) gx = 1;
  if (gx > p.G) gx = p.G;
  PZN_LAUNCH((pool_dgrad_kernel<NQ>), dim3((unsigned)gx, (unsigned)ny), dim3(PD_T), lds, st, p);
  PZN_RETURN_LAUNCH_STATUS();
}

int launch_dgrad(const PoolBwdArgs& p, hipStream_t st) {
  if (p.C2 == 64) return launch_dgrad_nq<1>(p, st);
  if (p.C2 == 128) return launch_dgrad_nq<2>(p, st);
  return launch_dgrad_nq<4>(p, st);
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

bool pzn_pool_bwd_supported(int C1, int C2, const float* W, const float* h, const float* dh) {
  return C1 > 0 && C1 % PB_COLS == 0 && (C2 == 64 || C2 == 128 || C2 == 256) && aligned16(W) && aligned16(h) &&
         aligned16(dh);      // (h may be NULL: rows regenerated from the gate source)
}

int pzn_pool_bwd_sparse(const float* dout, const int32_t* argmax, const float* out, const float* W, const float* h,
                        float* dh, float* dW, float* db, int G, int C1, int C2, hipStream_t st, const PznGateSource* gs) {
  PoolBwdArgs p{dout, argmax, out, W, h, dh, dW, db, G, C1, C2, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0,
                nullptr, nullptr, nullptr};
  if (gs && gs->P) {
    p.gP = gs->P, p.gidx = gs->idx, p.gxyz = gs->xyz, p.gnew = gs->new_xyz, p.gW1 = gs->W1, p.gb1 = gs->b1;
    p.gldw = gs->ldw, p.gN = gs->N, p.gS = gs->S;
    p.gQ = gs->Q, p.gdW1x = gs->dW1x, p.gdb1 = gs->db1;
    p.rowmask = gs->rowmask;
  }
#ifdef POOL_STAMPS
  p.stamps = nullptr;
#endif
  if (dW && !h && !p.gQ) return PZN_EINVAL;      // the weight-gradient pass needs the rows or their source
  if (!dh && !dW) return PZN_EINVAL;
#ifdef POOL_STAMPS
  auto next_slot = [&](int kind) {
    p.stamps = nullptr;
    if (!g_pool_stamps) return;
    const int slot = g_pool_launch++ % PS_LAUNCHES;
    p.stamps = g_pool_stamps + (size_t)slot * PS_WGS * 8;
    g_pool_log[slot][0] = kind, g_pool_log[slot][1] = G, g_pool_log[slot][2] = C1, g_pool_log[slot][3] = C2;
  };
#else
  auto next_slot = [](int) {};
#endif
  if (dW) {
    next_slot(0);
    int rc = launch_wgrad(p, st);
    if (rc != PZN_OK) return rc;
  }
  if (!dh) return PZN_OK;
  next_slot(1);
  return launch_dgrad(p, st);
}

#ifdef POOL_STAMPS
// diagnostic build only: a device buffer of 16 x 1024 x 8 uint64 that the next launches fill in turn (wgrad, dgrad, ...)
PZN_EXPORT void pzn_pool_bwd_set_stamps(void* buf) {
  g_pool_stamps = static_cast<unsigned long long*>(buf);
  if (buf) g_pool_launch = 0;
}
PZN_EXPORT int pzn_pool_bwd_stamp_log(int slot, int* out4) {
  for (int i = 0; i < 4; ++i) out4[i] = g_pool_log[slot % PS_LAUNCHES][i];
  return g_pool_launch;
}
#endif
