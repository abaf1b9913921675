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
// This file: the WEIGHT-gradient pass, walking groups g = blockIdx.x, += gridDim.x over a 128-column slice (blockIdx.y) of
// the C1 hidden columns, lane l holding columns 2l, 2l+1:
//  * pool_wgrad_kernel: 16 (or 8) wavefronts, wave w owns CPW = C2/16 channels whose dW accumulators stay in registers
//    for the whole walk.  Per group the 32 x 128 tile of h is staged in LDS (double-buffered, next tile
//    prefetched into registers), the wave's CPW (argmax, gradient) pairs are read as one vector and
//    broadcast with v_readlane, then per channel: one ds_read_b64 of the arg-max row + one packed fma.
// The input-gradient side of the encoder's levels is the walk by point of csrc/sapool.hip (the rows' gradient is never in
// memory); the grouped-row composition (pzn_sharedmlp_max_bwd_f32, h in memory) takes its input gradient from the
// generated-operand GEMM of gemm.hip.  (Rounds 2-4 had a second sparse kernel here, pool_dgrad_kernel, that wrote dh.)
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
  const float* h;         // [G*32, C1] the rows, or NULL: regenerated from gP / gQ
  float* dW;              // [C2, C1] += ...
  float* db;              // [C2] += ...        (may be NULL)
  int G, C1, C2;
  // regenerated rows (per-point first layer, csrc/sapoint.hip): h[(g,k),:] = relu(gP[(g / gS) * gN + gidx[g*32+k], :] + gQ[g,:])
  const float* gP;        // [B*N, C1]
  const int64_t* gidx;    // [G*32]
  const float* gQ;        // [G, C1]
  int gN, gS;
  // partial results of the workgroups ([gridDim.y][gridDim.x][C2][128] floats, then [gridDim.x][C2] for db), summed in a fixed
  // order by pool_wgrad_reduce_kernel; NULL: the workgroups add into dW / db with atomics
  float* partials;
};

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
    if (regen) {                                                                    \
      x0 = pb_add_relu(x0, q_);                                                     \
      if (TWO) x1 = pb_add_relu(x1, q_);                                            \
    }                                                                               \
    *reinterpret_cast<float4*>(&hbuf[buf][srow][scol]) = x0;                        \
    if (TWO) *reinterpret_cast<float4*>(&hbuf[buf][(srow + 16) & 31][scol]) = x1;   \
    const int av = av_;                                                             \
    const float gv = gv_;                                                           \
    __syncthreads(); /* one barrier per group: the tile two groups back is free again by construction */ \
    PB_ISSUE(gnext, x0, x1, q_, av_, gv_, j0_, j1_);  /* rows of gnext: their indices arrived a round ago */ \
    if (regen) PB_IDX((gnext) + 2 * gs_, j0_, j1_);   /* indices for the round after */ \
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

  if (p.partials) {
    float* part = p.partials + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * p.C2 * PB_COLS;
#pragma unroll
    for (int c = 0; c < CPW; ++c) *reinterpret_cast<v2f*>(part + (size_t)(wave * CPW + c) * PB_COLS + 2 * lane) = acc[c];
    if (blockIdx.y == 0 && lane < CPW)
      p.partials[(size_t)gridDim.y * gridDim.x * p.C2 * PB_COLS + (size_t)blockIdx.x * p.C2 + wave * CPW + lane] = dbacc;
  } else {
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      float* o = p.dW + (size_t)(wave * CPW + c) * p.C1 + col0 + 2 * lane;
      atomicAdd(o, acc[c].x);
      atomicAdd(o + 1, acc[c].y);
    }
    if (p.db && blockIdx.y == 0 && lane < CPW) atomicAdd(p.db + wave * CPW + lane, dbacc);
  }
  (void)T;
}

// dW[c, :] += the workgroups' partial tiles, db[c] += their partial sums, in workgroup order (the same bits in every run; the
// atomics they replace were 33.5 MB per level-2 launch, all at the end of the kernel: ~1 us per 0.65 MB as measured on the
// attention weight gradients).  A workgroup of 16 wavefronts owns 64 float4 outputs; wavefront w sums partials w, w + 16, ...
// - all its loads in flight - and the sixteen meet in LDS.  The last C2 / 64 workgroups: 64 bias entries each.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void pool_wgrad_reduce_kernel(const float* __restrict__ partials, int nx, int ny, int C1, int C2,
                                                                 float* __restrict__ dW, float* __restrict__ db) {
  __shared__ f32x4 red[15][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n4 = C2 * C1 / 4, nwg = n4 / 64;
  if ((int)blockIdx.x < nwg) {
    const int o = blockIdx.x * 64 + lane;
    const int c = o / (C1 / 4), rem = o - c * (C1 / 4), y = rem >> 5, col4 = rem & 31;
    const size_t stride4 = (size_t)C2 * PB_COLS / 4;
    const f32x4* src = reinterpret_cast<const f32x4*>(partials) + (size_t)y * nx * stride4 + (size_t)c * (PB_COLS / 4) + col4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    int x = w;
    for (; x + 48 < nx; x += 64) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + (size_t)(x + 16 * u) * stride4);
#pragma unroll
      for (int u = 0; u < 4; ++u) sum += v[u];
    }
    for (; x < nx; x += 16) sum += __builtin_nontemporal_load(src + (size_t)x * stride4);
    if (w) red[w - 1][lane] = sum;
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int u = 0; u < 15; ++u) sum += red[u][lane];
      f32x4* dst = reinterpret_cast<f32x4*>(dW + (size_t)c * C1 + y * PB_COLS) + col4;
      *dst += sum;
    }
  } else if (db) {
    const int c = ((int)blockIdx.x - nwg) * 64 + lane;
    const float* src = partials + (size_t)ny * nx * C2 * PB_COLS + c;
    float sum = 0.f;
    for (int x = w; x < nx; x += 16) sum += src[(size_t)x * C2];
    float* redf = reinterpret_cast<float*>(&red[0][0]);
    if (w) redf[(w - 1) * 64 + lane] = sum;
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int u = 0; u < 15; ++u) sum += redf[u * 64 + lane];
      db[c] += sum;
    }
  }
}

int wgrad_grid_x(int G, int C1) {
  const int ny = C1 / PB_COLS;
  int gx = 256 / ny;  // workgroups in flight (one per CU measured best)
  return gx > G ? G : gx;
}

int launch_wgrad(PoolBwdArgs p, hipStream_t st, void* ws, size_t ws_bytes) {
  const int ny = p.C1 / PB_COLS;
  const int gx = wgrad_grid_x(p.G, p.C1);
  const dim3 grid((unsigned)gx, (unsigned)ny);
  p.partials = (ws && ws_bytes >= pzn_pool_wgrad_ws_bytes(p.G, p.C1, p.C2) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(p.dW) & 15) == 0)
                   ? static_cast<float*>(ws)
                   : nullptr;
  if (p.C2 % 16 == 0 && p.C2 / 16 >= 4) {      // 16 wavefronts, C2 / 16 channels each
    const int cpw = p.C2 / 16;
    if (cpw == 4)
      PZN_LAUNCH((pool_wgrad_kernel<4, 16>), grid, dim3(1024), 0, st, p);
    else if (cpw == 8)
      PZN_LAUNCH((pool_wgrad_kernel<8, 16>), grid, dim3(1024), 0, st, p);
    else
      PZN_LAUNCH((pool_wgrad_kernel<16, 16>), grid, dim3(1024), 0, st, p);
  } else {
    const dim3 block(PB_T);
    const int cpw = p.C2 / PB_W;
    if (cpw == 8)
      PZN_LAUNCH((pool_wgrad_kernel<8, 8>), grid, block, 0, st, p);
    else if (cpw == 16)
      PZN_LAUNCH((pool_wgrad_kernel<16, 8>), grid, block, 0, st, p);
    else
      PZN_LAUNCH((pool_wgrad_kernel<32, 8>), grid, block, 0, st, p);
  }
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  if (p.partials) {
    const int nwg = p.C2 * p.C1 / 256;
    PZN_LAUNCH(pool_wgrad_reduce_kernel, dim3((unsigned)(nwg + (p.db ? p.C2 / 64 : 0))), dim3(1024), 0, st, p.partials, gx, ny,
               p.C1, p.C2, p.dW, p.db);
    if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  }
  return PZN_OK;
}

bool aligned16(const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; }

}  // namespace

bool pzn_pool_wgrad_supported(int C1, int C2, const float* h) {
  return C1 > 0 && C1 % PB_COLS == 0 && (C2 == 64 || C2 == 128 || C2 == 256) && aligned16(h);      // (h may be NULL: regenerated rows)
}

// bytes of the optional workspace: one [C2][128] tile (+ C2 bias sums) per workgroup
size_t pzn_pool_wgrad_ws_bytes(int G, int C1, int C2) {
  if (G <= 0 || C1 <= 0 || C1 % PB_COLS != 0 || C2 % 64 != 0) return 0;
  const size_t nx = (size_t)wgrad_grid_x(G, C1), ny = (size_t)(C1 / PB_COLS);
  return (ny * nx * C2 * PB_COLS + nx * C2) * sizeof(float);
}

// dW / db are ADDED to.  ws (pzn_pool_wgrad_ws_bytes, 16-byte aligned; may be NULL): the workgroups' partial results, summed in
// a fixed order; without it they meet in fp32 atomics.
int pzn_pool_wgrad_sparse(const float* dout, const int32_t* argmax, const float* out, const float* h, float* dW, float* db,
                          int G, int C1, int C2, hipStream_t st, const PznGateSource* gs, void* ws, size_t ws_bytes) {
  PoolBwdArgs p{dout, argmax, out, h, dW, db, G, C1, C2, nullptr, nullptr, nullptr, 0, 0, nullptr};
  if (gs && gs->P) p.gP = gs->P, p.gidx = gs->idx, p.gQ = gs->Q, p.gN = gs->N, p.gS = gs->S;
  if (!dW || (!h && !p.gQ)) return PZN_EINVAL;      // the pass needs the rows or their source
  return launch_wgrad(p, st, ws, ws_bytes);
}
