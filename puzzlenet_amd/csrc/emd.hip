// emd.hip — approximate Earth Mover's Distance (auction matching) for gfx950.
//
// Replaces PyTorchEMD/cuda/emd_kernel.cu: approxmatch (:25-158), matchcost
// (:200-243), matchcostgrad1 (:333-355), matchcostgrad2 (:286-327).
//
// The reference runs <<<32,512>>>: one workgroup per cloud pair, <= 32 CUs busy,
// and read-modify-writes the (B,m,n) `match` tensor ten times.  Here every one
// of the 10 levels x 3 passes is its own launch over ALL (pair, row-tile)
// workgroups, so a B=64, n=m=2048 call fills the chip, and the fused entry
// point (pzn_emd_fused_f32) never writes `match`:
//     cost  = sum_kl d_kl match_lk,   grad1_k = 2 sum_l match_lk (x1_k - x2_l),
//     grad2_l = 2 sum_k match_lk (x2_l - x1_k)
// are linear in match = sum_levels w, so they are accumulated level by level
// inside the passes that compute w anyway.
//
// Pass structure per level (emd_kernel.cu line numbers):
//   A (:51-84)   ratioL_k  = remainL_k / (1e-9 + sum_l e_kl remainR_l)
//   B (:86-119)  s_l = remainR_l sum_k e_kl ratioL_k;
//                ratioR_l = min(remainR_l/(s_l+1e-9), 1) remainR_l;  remainR_l = max(0, remainR_l - s_l)
//   C (:121-154) w_kl = e_kl ratioL_k ratioR_l;  match_lk += w_kl;  remainL_k = max(0, remainL_k - sum_l w_kl)
// with e_kl = exp(level * d_kl), level = -4^j for j = 7..-1 and 0 for the last.
//
// Common to every walk: a workgroup = 4 wavefronts that own the SAME 64 rows; the OTHER cloud is staged through LDS
// in tiles as a pair-SoA image {x0,x1,y0,y1}{z0,z1,w0,w1}, every wavefront walks a quarter of each tile with a
// wave-uniform index (broadcast ds_read_b128) and packed fp32 arithmetic (v_pk_add / v_pk_mul / v_pk_fma: two (k,l)
// pairs per instruction), and the four partial sums meet through LDS.  (The first version fed the walked points to the
// VALU as SGPR operands through the scalar cache; that cache thrashed at ~10 cycles per VALU issue.)
//
// Three paths:
//   * pzn_emd_approxmatch_f32 (API parity: writes `match`): plain A / B / C launches per level (emd_pass_{a,b,c}_kernel);
//   * n, m <= 256: the whole auction of a pair in one workgroup (emd_small_fused_kernel);
//   * the fused entry point on large clouds (section "fused path" below): x-sorted clouds, active lists, x windows,
//     pass C fused with the next level's pass A.
#include <stdlib.h>

#include "pzn_common.h"

namespace {

constexpr int EMD_T = 256;  // threads per workgroup (4 wavefronts)
constexpr int EMD_SMALL_MAX = 256;  // n, m up to here: the whole auction of a pair in one workgroup (emd_small_fused_kernel)
// counters of executed evaluations (measurement only, pzn_emd_walk_counter_offset): 32 launch ids x 16 slots, each slot on a
// 128-byte line of its own — one atomic per WORKGROUP, and the workgroups of a launch spread over 16 lines (a first
// version added once per wavefront into 32 adjacent counters: 8192 atomics of a launch on two lines, which serialise at
// ~12 ns each; that was a 50 us floor under every launch, found with the phase stamps of tools/emd_stamps.py)
constexpr int EMD_WALK_LIDS = 32, EMD_WALK_PER_LID = 16, EMD_WALK_STRIDE = 16;
constexpr int EMD_WALK_SLOTS = EMD_WALK_LIDS * EMD_WALK_PER_LID * EMD_WALK_STRIDE;
typedef float v2f __attribute__((ext_vector_type(2)));

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------------------------------
// three-call path (pzn_emd_approxmatch_f32): workspace and kernels

struct EmdWs {
  float4* pk1;     // [B*n] {x1,y1,z1, ratioL}
  float4* pk2a;    // [B*m] {x2,y2,z2, remainR}
  float4* pk2b;    // [B*m] {x2,y2,z2, ratioR}
  float* remainL;  // [B*n]
};

size_t ws3_bytes(int B, int n, int m) {
  return align_up(sizeof(float4) * (size_t)B * n, 256) + 2 * align_up(sizeof(float4) * (size_t)B * m, 256) +
         align_up(sizeof(float) * (size_t)B * n, 256);
}

EmdWs carve(void* ws, int B, int n, int m) {
  unsigned char* p = static_cast<unsigned char*>(ws);
  EmdWs w;
  w.pk1 = reinterpret_cast<float4*>(p);
  p += align_up(sizeof(float4) * (size_t)B * n, 256);
  w.pk2a = reinterpret_cast<float4*>(p);
  p += align_up(sizeof(float4) * (size_t)B * m, 256);
  w.pk2b = reinterpret_cast<float4*>(p);
  p += align_up(sizeof(float4) * (size_t)B * m, 256);
  w.remainL = reinterpret_cast<float*>(p);
  return w;
}

__device__ __forceinline__ float sq3(float dx, float dy, float dz) { return dx * dx + dy * dy + dz * dz; }

// Walk `cnt` packed points of the other cloud (see the file header).  That quadruples the wavefronts of a pass (8 per
// SIMD at B=64, n=2048, where one thread per row gives only 2) — the pass is a long dependent VALU/exp chain per
// lane, and resident waves are the only latency cover.  512-point tiles, double-buffered, one barrier per tile.
// The tile holds point PAIRS in SoA form — {x0,x1,y0,y1} {z0,z1,w0,w1} — so that one lane evaluates two
// walked points per instruction with the packed fp32 ALU ops: ~8 issues per (row, point) instead of ~21 for the
// float4-per-point form (whose pairs the compiler had to assemble with v_mov).  An odd tail point is paired with a
// zero-weight copy of itself.
constexpr int EMD_TL = 512;
constexpr int EMD_ROWS = 64;  // rows per workgroup
#define EMD_WALK(PTR, CNT, EVAL2)                                             \
  do {                                                                        \
    __shared__ float4 emd_tile_[2][EMD_TL];                                   \
    const int cnt_ = (CNT);                                                   \
    const int wq_ = threadIdx.x >> 6;                                         \
    const int ntile_ = (cnt_ + EMD_TL - 1) / EMD_TL;                          \
    auto stage_ = [&](int buf, int base) {                                    \
      const int q = threadIdx.x; /* pair index inside the tile: EMD_T == EMD_TL / 2 */ \
      const int i0 = base + 2 * q;                                            \
      if (i0 < cnt_) {                                                        \
        float4 a = (PTR)[i0];                                                 \
        float4 b = i0 + 1 < cnt_ ? (PTR)[i0 + 1] : make_float4(a.x, a.y, a.z, 0.f); \
        emd_tile_[buf][2 * q] = make_float4(a.x, b.x, a.y, b.y);              \
        emd_tile_[buf][2 * q + 1] = make_float4(a.z, b.z, a.w, b.w);          \
      }                                                                       \
    };                                                                        \
    stage_(0, 0);                                                             \
    __syncthreads();                                                          \
    for (int t_ = 0; t_ < ntile_; ++t_) {                                     \
      const int base_ = t_ * EMD_TL;                                          \
      const int npair_ = (min(EMD_TL, cnt_ - base_) + 1) >> 1;                \
      if (t_ + 1 < ntile_) stage_((t_ + 1) & 1, base_ + EMD_TL);              \
      const float4* tp_ = emd_tile_[t_ & 1];                                  \
      const int per_ = (npair_ + 3) >> 2;                                     \
      int q_ = min(npair_, wq_ * per_);                                       \
      const int end_ = min(npair_, q_ + per_);                                \
      for (; q_ < end_; ++q_) {                                               \
        float4 a0_ = tp_[2 * q_], b0_ = tp_[2 * q_ + 1];                      \
        EVAL2((v2f){a0_.x, a0_.y}, (v2f){a0_.z, a0_.w}, (v2f){b0_.x, b0_.y}, (v2f){b0_.z, b0_.w}, base_ + 2 * q_); \
      }                                                                       \
      __syncthreads();                                                        \
    }                                                                         \
  } while (0)

__device__ __forceinline__ v2f exp2_pair(v2f t) { return (v2f){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)}; }

// sum over the 4 wavefronts of a workgroup, lane by lane (all threads get the total)
__device__ __forceinline__ float cross_wave_sum(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  const int l = threadIdx.x & 63;
  float t = (red[l] + red[64 + l]) + (red[128 + l] + red[192 + l]);
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(EMD_T) void emd_init_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2, int n,
                                                         int m, float multiL, float multiR, EmdWs w) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * EMD_T + threadIdx.x;
  if (i < n) {
    const float* p = xyz1 + ((size_t)b * n + i) * 3;
    w.pk1[(size_t)b * n + i] = make_float4(p[0], p[1], p[2], 0.f);
    w.remainL[(size_t)b * n + i] = multiL;  // :41-42
  }
  if (i < m) {
    const float* p = xyz2 + ((size_t)b * m + i) * 3;
    w.pk2a[(size_t)b * m + i] = make_float4(p[0], p[1], p[2], multiR);  // :43-44
    w.pk2b[(size_t)b * m + i] = make_float4(p[0], p[1], p[2], 0.f);
  }
}

// Pass A: rows = points k of xyz1.
__global__ __launch_bounds__(EMD_T) void emd_pass_a_kernel(int n, int m, float c, EmdWs w) {
  __shared__ float red[EMD_T];
  const int b = blockIdx.y;
  const int k = blockIdx.x * EMD_ROWS + (threadIdx.x & 63);
  const float4* __restrict__ other = w.pk2a + (size_t)b * m;
  float4 me = k < n ? w.pk1[(size_t)b * n + k] : make_float4(0, 0, 0, 0);
  v2f acc = {0.f, 0.f};
  auto eval = [&](v2f X, v2f Y, v2f Z, v2f Wt, int) {
    v2f dx = X - me.x, dy = Y - me.y, dz = Z - me.z;
    v2f d = dx * dx + dy * dy + dz * dz;       // :76
    acc += exp2_pair(d * c) * Wt;              // :77-78   (__expf(level d) = exp2(level log2(e) d), folded on the host)
  };
  EMD_WALK(other, m, eval);
  float suml = 1e-9f + cross_wave_sum(acc.x + acc.y, red);  // :59
  if (k < n && threadIdx.x < 64) {
    me.w = w.remainL[(size_t)b * n + k] / suml;  // :83
    w.pk1[(size_t)b * n + k] = me;
  }
}

// Pass B: rows = points l of xyz2.
__global__ __launch_bounds__(EMD_T) void emd_pass_b_kernel(int n, int m, float c, EmdWs w) {
  __shared__ float red[EMD_T];
  const int b = blockIdx.y;
  const int l = blockIdx.x * EMD_ROWS + (threadIdx.x & 63);
  const float4* __restrict__ other = w.pk1 + (size_t)b * n;
  float4 me = l < m ? w.pk2a[(size_t)b * m + l] : make_float4(0, 0, 0, 0);
  v2f ar = {0.f, 0.f};
  auto eval = [&](v2f X, v2f Y, v2f Z, v2f Wt, int) {
    v2f dx = me.x - X, dy = me.y - Y, dz = me.z - Z;
    ar += exp2_pair((dx * dx + dy * dy + dz * dz) * c) * Wt;  // :108-109
  };
  EMD_WALK(other, n, eval);
  float sumr = cross_wave_sum(ar.x + ar.y, red);
  if (l < m && threadIdx.x < 64) {
    float remainR = me.w;
    sumr *= remainR;                                              // :114
    float consumption = fminf(remainR / (sumr + 1e-9f), 1.0f);    // :115
    float ratioR = consumption * remainR;                         // :116
    me.w = fmaxf(0.0f, remainR - sumr);                           // :117
    w.pk2a[(size_t)b * m + l] = me;
    w.pk2b[(size_t)b * m + l].w = ratioR;
  }
}

// Pass C: rows = points k of xyz1; match[b][l][k] += w.
__global__ __launch_bounds__(EMD_T) void emd_pass_c_kernel(int n, int m, float c, EmdWs w, float* __restrict__ match) {
  __shared__ float red[EMD_T];
  const int b = blockIdx.y;
  const int k = blockIdx.x * EMD_ROWS + (threadIdx.x & 63);
  const float4* __restrict__ other = w.pk2b + (size_t)b * m;
  float4 me = k < n ? w.pk1[(size_t)b * n + k] : make_float4(0, 0, 0, 0);
  const float rl = me.w;  // :139
  v2f al = {0.f, 0.f};
  float* mt = match + (size_t)b * n * m + k;
  auto eval = [&](v2f X, v2f Y, v2f Z, v2f Wt, int l) {
    v2f dx = me.x - X, dy = me.y - Y, dz = me.z - Z;
    v2f d = dx * dx + dy * dy + dz * dz;
    v2f wv = exp2_pair(d * c) * rl * Wt;  // :145
    if (k < n) {
      mt[(size_t)l * n] += wv.x;  // :146
      if (l + 1 < m) mt[(size_t)(l + 1) * n] += wv.y;
    }
    al += wv;  // :147
  };
  EMD_WALK(other, m, eval);
  float suml = cross_wave_sum(al.x + al.y, red);
  if (k < n && threadIdx.x < 64) {
    float* r = w.remainL + (size_t)b * n + k;
    *r = fmaxf(0.0f, *r - suml);  // :153
  }
}

// matchcost (:200-243): cost_b = sum_kl d_kl match[b][l][k].  Thread per k, a
// workgroup takes a slab of l; partial sums meet in one atomic per workgroup.
constexpr int MC_LSLAB = 64;
__global__ __launch_bounds__(EMD_T) void emd_matchcost_kernel(const float* __restrict__ xyz1,
                                                              const float* __restrict__ xyz2,
                                                              const float* __restrict__ match, int n, int m,
                                                              float* __restrict__ cost) {
  const int b = blockIdx.z;
  const int k = blockIdx.x * EMD_T + threadIdx.x;
  const int l0 = blockIdx.y * MC_LSLAB;
  const int l1 = min(m, l0 + MC_LSLAB);
  float x1 = 0, y1 = 0, z1 = 0;
  if (k < n) {
    const float* p = xyz1 + ((size_t)b * n + k) * 3;
    x1 = p[0], y1 = p[1], z1 = p[2];
  }
  const float* p2 = xyz2 + (size_t)b * m * 3;
  const float* mt = match + (size_t)b * n * m + k;
  float s = 0.f;
  if (k < n)
    for (int l = l0; l < l1; ++l) {
      float dx = p2[l * 3] - x1, dy = p2[l * 3 + 1] - y1, dz = p2[l * 3 + 2] - z1;
      s += sq3(dx, dy, dz) * mt[(size_t)l * n];  // :225-226
    }
  __shared__ float red[EMD_T / PZN_WAVE];
  s = pzn::wave_sum_f32(s);
  if ((threadIdx.x & (PZN_WAVE - 1)) == 0) red[threadIdx.x / PZN_WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < EMD_T / PZN_WAVE; ++i) t += red[i];
    atomicAdd(cost + b, t);
  }
}

// matchcostgrad1 (:333-355): thread per point of xyz1, serial over xyz2.
__global__ __launch_bounds__(EMD_T) void emd_grad1_kernel(const float* __restrict__ grad_cost,
                                                          const float* __restrict__ xyz1,
                                                          const float* __restrict__ xyz2,
                                                          const float* __restrict__ match, int n, int m,
                                                          float* __restrict__ grad1) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * EMD_T + threadIdx.x;
  if (k >= n) return;
  const float* p = xyz1 + ((size_t)b * n + k) * 3;
  const float x1 = p[0], y1 = p[1], z1 = p[2];
  const float* p2 = xyz2 + (size_t)b * m * 3;
  const float* mt = match + (size_t)b * n * m + k;
  float dx = 0, dy = 0, dz = 0;
  for (int l = 0; l < m; ++l) {
    float d = mt[(size_t)l * n] * 2;  // :345
    dx += (x1 - p2[l * 3]) * d;
    dy += (y1 - p2[l * 3 + 1]) * d;
    dz += (z1 - p2[l * 3 + 2]) * d;
  }
  const float gc = grad_cost[b];
  float* g = grad1 + ((size_t)b * n + k) * 3;
  g[0] = dx * gc;
  g[1] = dy * gc;
  g[2] = dz * gc;
}

// matchcostgrad2 (:286-327): one wavefront per point of xyz2, lanes stride over xyz1.
__global__ __launch_bounds__(EMD_T) void emd_grad2_kernel(const float* __restrict__ grad_cost,
                                                          const float* __restrict__ xyz1,
                                                          const float* __restrict__ xyz2,
                                                          const float* __restrict__ match, int n, int m,
                                                          float* __restrict__ grad2) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & (PZN_WAVE - 1);
  const int l = blockIdx.x * (EMD_T / PZN_WAVE) + threadIdx.x / PZN_WAVE;
  if (l >= m) return;
  const float* p = xyz2 + ((size_t)b * m + l) * 3;
  const float x2 = p[0], y2 = p[1], z2 = p[2];
  const float* p1 = xyz1 + (size_t)b * n * 3;
  const float* mt = match + (size_t)b * n * m + (size_t)l * n;
  float sx = 0, sy = 0, sz = 0;
  for (int j = lane; j < n; j += PZN_WAVE) {
    float d = mt[j] * 2;  // :301
    sx += (x2 - p1[j * 3]) * d;
    sy += (y2 - p1[j * 3 + 1]) * d;
    sz += (z2 - p1[j * 3 + 2]) * d;
  }
  sx = pzn::wave_sum_f32(sx);
  sy = pzn::wave_sum_f32(sy);
  sz = pzn::wave_sum_f32(sz);
  if (lane == 0) {
    const float gc = grad_cost[b];
    float* g = grad2 + ((size_t)b * m + l) * 3;
    g[0] = sx * gc;
    g[1] = sy * gc;
    g[2] = sz * gc;
  }
}

// ---- small problems (n, m <= 256: the boundary and key-point terms of the loss, 64 pairs of 128 x 128 or
// 64 x 64 points): the whole auction of one pair in ONE workgroup, all 10 levels x 3 passes inside the
// kernel.  Through the general path such a call is 31 launches (+ 3 zero fills) of a few microseconds
// each; here both clouds and the per-point auction state live in LDS, a row is shared by TPR = 1024 / R
// adjacent lanes (R = rows rounded up to a power of two) that walk interleaved parts of the other cloud,
// and cost / gradients accumulate in registers over the levels.  Same formulas as passes A / B / C above.

constexpr int EMD_SMALL_T = 1024;  // 16 wavefronts per pair: the walk is a dependent VALU / exp chain per lane

// Up to EMD_SMALL_NP independent calls in one launch (the three small terms of the loss: model5_b.py:1012, :1123-1125): workgroup
// blockIdx.x belongs to the problem whose range [first, first + B) holds it.
constexpr int EMD_SMALL_NP = 4;
struct EmdSmallProblem {
  const float* xyz1;
  const float* xyz2;
  float* cost;
  float* g1;
  float* g2;
  int n, m, tpr_shift, first;      // first: the problem's first workgroup
  float multiL, multiR;
};
struct EmdSmallArgs {
  EmdSmallProblem pr[EMD_SMALL_NP];
  int count;
};

__global__ __launch_bounds__(EMD_SMALL_T) void emd_small_fused_kernel(EmdSmallArgs args) {
  int which = 0;
#pragma unroll
  for (int i = 1; i < EMD_SMALL_NP; ++i) which = (i < args.count && (int)blockIdx.x >= args.pr[i].first) ? i : which;
  const EmdSmallProblem& pr = args.pr[which];
  const float* __restrict__ xyz1 = pr.xyz1;
  const float* __restrict__ xyz2 = pr.xyz2;
  float* __restrict__ cost = pr.cost;
  float* __restrict__ g1 = pr.g1;
  float* __restrict__ g2 = pr.g2;
  const int n = pr.n, m = pr.m, tpr_shift = pr.tpr_shift;
  const float multiL = pr.multiL, multiR = pr.multiR;
  __shared__ float4 p1[EMD_SMALL_MAX];   // {x1, y1, z1, ratioL}
  __shared__ float4 p2[EMD_SMALL_MAX];   // {x2, y2, z2, remainR}
  __shared__ float rr[EMD_SMALL_MAX];    // ratioR
  __shared__ float rl[EMD_SMALL_MAX];    // remainL
  __shared__ float csum[EMD_SMALL_T / 64];
  const int b = (int)blockIdx.x - pr.first, tid = threadIdx.x;
  const int tpr = 1 << tpr_shift, row = tid >> tpr_shift, part = tid & (tpr - 1);
  for (int i = tid; i < n; i += EMD_SMALL_T) {
    const float* q = xyz1 + ((size_t)b * n + i) * 3;
    p1[i] = make_float4(q[0], q[1], q[2], 0.f);
    rl[i] = multiL;
  }
  for (int i = tid; i < m; i += EMD_SMALL_T) {
    const float* q = xyz2 + ((size_t)b * m + i) * 3;
    p2[i] = make_float4(q[0], q[1], q[2], multiR);
    rr[i] = 0.f;
  }
  __syncthreads();
  // sum over the tpr adjacent lanes that share a row
  auto row_sum = [&](float v) {
    for (int o = 1; o < tpr; o <<= 1) v += __shfl_xor(v, o, PZN_WAVE);
    return v;
  };
  float g1x = 0.f, g1y = 0.f, g1z = 0.f, g2x = 0.f, g2y = 0.f, g2z = 0.f, cacc = 0.f;
  for (int j = 7; j >= -2; --j) {
    const float level = j == -2 ? 0.f : -powf(4.0f, (float)j);
    const float c = level * 1.44269504088896340736f;
    {  // pass A: rows k of xyz1
      float acc = 0.f;
      if (row < n) {
        const float4 me = p1[row];
#pragma unroll 4
        for (int l = part; l < m; l += tpr) {  // (unrolled: the loads and exps of four points are in flight together)
          const float4 o = p2[l];
          const float dx = o.x - me.x, dy = o.y - me.y, dz = o.z - me.z;
          acc += __builtin_amdgcn_exp2f((dx * dx + dy * dy + dz * dz) * c) * o.w;
        }
      }
      acc = row_sum(acc);
      __syncthreads();
      if (row < n && part == 0) p1[row].w = rl[row] / (1e-9f + acc);
      __syncthreads();
    }
    {  // pass B: rows l of xyz2
      float ar = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
      float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < m) {
        me = p2[row];
#pragma unroll 4
        for (int k = part; k < n; k += tpr) {
          const float4 o = p1[k];
          const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
          const float e = __builtin_amdgcn_exp2f((dx * dx + dy * dy + dz * dz) * c) * o.w;
          ar += e, ax += e * dx, ay += e * dy, az += e * dz;
        }
      }
      ar = row_sum(ar), ax = row_sum(ax), ay = row_sum(ay), az = row_sum(az);
      __syncthreads();
      if (row < m && part == 0) {
        const float remainR = me.w;
        const float sumr = ar * remainR;
        const float ratioR = fminf(remainR / (sumr + 1e-9f), 1.0f) * remainR;
        p2[row].w = fmaxf(0.0f, remainR - sumr);
        rr[row] = ratioR;
        const float s2 = 2.f * ratioR;
        g2x += s2 * ax, g2y += s2 * ay, g2z += s2 * az;
      }
      __syncthreads();
    }
    {  // pass C: rows k of xyz1
      float al = 0.f, ax = 0.f, ay = 0.f, az = 0.f, ac = 0.f;
      if (row < n) {
        const float4 me = p1[row];
#pragma unroll 4
        for (int l = part; l < m; l += tpr) {
          const float4 o = p2[l];
          const float dx = me.x - o.x, dy = me.y - o.y, dz = me.z - o.z;
          const float d = dx * dx + dy * dy + dz * dz;
          const float wv = __builtin_amdgcn_exp2f(d * c) * me.w * rr[l];
          al += wv, ax += wv * dx, ay += wv * dy, az += wv * dz, ac += wv * d;
        }
      }
      al = row_sum(al), ax = row_sum(ax), ay = row_sum(ay), az = row_sum(az);
      __syncthreads();
      if (row < n && part == 0) {
        rl[row] = fmaxf(0.0f, rl[row] - al);
        g1x += 2.f * ax, g1y += 2.f * ay, g1z += 2.f * az;
      }
      cacc += ac;
      __syncthreads();
    }
  }
  if (row < n && part == 0) {
    float* g = g1 + ((size_t)b * n + row) * 3;
    g[0] = g1x, g[1] = g1y, g[2] = g1z;
  }
  if (row < m && part == 0) {
    float* g = g2 + ((size_t)b * m + row) * 3;
    g[0] = g2x, g[1] = g2y, g[2] = g2z;
  }
  cacc = pzn::wave_sum_f32(cacc);
  if ((tid & 63) == 0) csum[tid >> 6] = cacc;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int i = 0; i < EMD_SMALL_T / 64; ++i) t += csum[i];
    cost[b] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fused path (pzn_emd_fused_f32 with n or m > 256): cost and both gradients, `match` never written
//
//   * both clouds are walked in ascending x (emd_sort_x_kernel; results go back through the permutation at the end);
//   * ACTIVE LIST of cloud 2: remainR_l == 0 is absorbing (pass B: s_l *= remainR_l -> 0, ratioR_l = 0, remainR_l
//     stays 0, :114-117) and such a point contributes exactly +0 to every sum of passes A and C, so a level only needs
//     the points that still hold mass — 57 % of them at level 5, 21 % at level 3, 2 % at level 0 on independent uniform
//     clouds, 91 % / 81 % / 42 % on a cloud against its rigidly moved copy (the loss term under an untrained pose head).
//     The list is COMPACTED: lst[r] = {x, y, z, remainR} of the r-th active point, ascending x, so that the walks stage
//     contiguous records (no index indirection) and pass B writes its results in place; an extra workgroup of the
//     C + A launch builds the next level's list from this level's (emdf_compact_wg);
//   * X WINDOW: at a level with scale c (in exp2 units) a point farther than sqrt(150 / -c) along x from every row of a
//     workgroup contributes exp2(c d^2) = exactly +0 in fp32, so a wavefront skips a quarter tile that lies wholly
//     outside the window of its rows.  Finer tests do not pay: on the loss term's clouds a window per 8 points instead
//     of per quarter tile skips 3 % more, and even a perfect 3-D sphere test would skip only 26 % more (measured with
//     the per-launch counters below; tools/emd_levels.py) — the soft levels, where nothing can be skipped, dominate;
//   * pass C of a level and pass A of the next level are ONE walk (emdf_k_kernel<1>): differences and squared distance
//     are computed once, and since c_j = 4 c_{j-1} EXACTLY the sharper exponential is the softer one squared twice
//     (two packed multiplies instead of two v_exp_f32; the error of either form is dominated by the rounding of
//     c d^2, |c d^2| 2^-24, which is the same for both);
//   * the row factor ratioL_k of pass C multiplies the finished sums, not every term;
//   * per-row state travels as 16-byte records {g.x, g.y, g.z, remainL} in sorted order (one load at the head of the
//     kernel, one store at the end; the gradients are un-permuted once, by emdf_finish_kernel);
//   * the partial sums of the four wavefronts meet in LDS behind ONE barrier (they were 2 barriers per sum).
// 256-point tiles (4 - 6 KB per buffer): eight workgroups per CU fit whatever the pass.

constexpr int EF_T = 256;          // threads per workgroup: 4 wavefronts on the same 64 rows
constexpr int EF_TL = 256;         // points per LDS tile
constexpr int EF_NP = EF_TL / 2;   // point pairs per tile; the first EF_NP threads stage one pair each

struct EmdF {
  float4* pk1;     // [B*n]  {x1,y1,z1, ratioL}, x-sorted order
  float4* st1;     // [B*n]  {g1x,g1y,g1z, remainL}: gradient accumulator and remaining mass of the rows of cloud 1
  float4* st2;     // [B*m]  {g2x,g2y,g2z, -} by x-sorted index of cloud 2
  float4* lst[2];  // [B*m]  compacted active list of cloud 2: {x2,y2,z2, remainR}, ascending x (two lists alternate)
  int* lidx[2];    // [B*m]  x-sorted index of the list entries
  float* rr;       // [B*m]  ratioR of the current level by list position
  int* cnt[2];     // [B]    list lengths
  int* perm[2];    // [B*n], [B*m]: x-sorted position -> original index
  float* costp;    // [B*gk] cost by (pair, row block of cloud 1): summed by emdf_finish_kernel (64 adjacent floats of cost[]
                   //        are two cache lines: 2048 atomics of a launch on them serialise)
  unsigned long long* walk;  // EMD_WALK_SLOTS counters (measurement only): units of 64 (row, point) evaluations executed by
                             // the last call, counter (16 * launch id + s) * 16 for s = 0..15 (launch ids: 0 = A(7),
                             // 1 + 2i = pass B and 2 + 2i = pass C (+ next A) of level 7 - i): also a per-launch histogram
#ifdef EMD_STAMPS
  unsigned long long* stamps;  // diagnostic build (tools/emd_stamps.py): per (launch id, workgroup) {start, end, HW_ID, XCC_ID}
#endif
};
#ifdef EMD_STAMPS
constexpr int EMD_STAMP_WGS = 4096, EMD_STAMP_LIDS = 21;
constexpr size_t EMD_STAMP_BYTES = sizeof(unsigned long long) * 8 * EMD_STAMP_WGS * EMD_STAMP_LIDS;
// record: {start, staged, walked, summed, end, HW_ID, XCC_ID, -} of wavefront 0
#define EMD_STAMP_BEGIN() unsigned long long stamp_[5] = {__builtin_amdgcn_s_memtime(), 0, 0, 0, 0}
#define EMD_STAMP(I) stamp_[I] = __builtin_amdgcn_s_memtime()
#define EMD_STAMP_END(W, LID)                                                                         \
  do {                                                                                                \
    const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                         \
    if (threadIdx.x == 0 && wg_ < EMD_STAMP_WGS && (LID) < EMD_STAMP_LIDS) {                          \
      unsigned long long* s_ = (W).stamps + ((size_t)(LID) * EMD_STAMP_WGS + wg_) * 8;                \
      stamp_[4] = __builtin_amdgcn_s_memtime();                                                       \
      for (int i_ = 0; i_ < 5; ++i_) s_[i_] = stamp_[i_];                                             \
      s_[5] = __builtin_amdgcn_s_getreg((31 << 11) | 4), s_[6] = __builtin_amdgcn_s_getreg((31 << 11) | 20); \
    }                                                                                                 \
  } while (0)
#else
constexpr size_t EMD_STAMP_BYTES = 0;
#define EMD_STAMP_BEGIN() do {} while (0)
#define EMD_STAMP(I) do {} while (0)
#define EMD_STAMP_END(W, LID) do {} while (0)
#endif

size_t wsf_bytes(int B, int n, int m) {
  return 2 * align_up(sizeof(float4) * (size_t)B * n, 256) + 3 * align_up(sizeof(float4) * (size_t)B * m, 256) +
         3 * align_up(sizeof(int) * (size_t)B * m, 256) + 2 * align_up(sizeof(int) * (size_t)B, 256) +
         align_up(sizeof(int) * (size_t)B * n, 256) + align_up(sizeof(int) * (size_t)B * m, 256) +
         align_up(sizeof(float) * (size_t)B * ((n + 63) / 64), 256);
}

EmdF carve_f(void* ws, size_t total, int B, int n, int m) {
  unsigned char* p = static_cast<unsigned char*>(ws);
  EmdF w;
  auto take = [&](size_t bytes) {
    void* r = p;
    p += align_up(bytes, 256);
    return r;
  };
  w.pk1 = static_cast<float4*>(take(sizeof(float4) * (size_t)B * n));
  w.st1 = static_cast<float4*>(take(sizeof(float4) * (size_t)B * n));
  w.st2 = static_cast<float4*>(take(sizeof(float4) * (size_t)B * m));
  for (int i = 0; i < 2; ++i) w.lst[i] = static_cast<float4*>(take(sizeof(float4) * (size_t)B * m));
  for (int i = 0; i < 2; ++i) w.lidx[i] = static_cast<int*>(take(sizeof(int) * (size_t)B * m));
  w.rr = static_cast<float*>(take(sizeof(float) * (size_t)B * m));
  for (int i = 0; i < 2; ++i) w.cnt[i] = static_cast<int*>(take(sizeof(int) * (size_t)B));
  w.perm[0] = static_cast<int*>(take(sizeof(int) * (size_t)B * n));
  w.perm[1] = static_cast<int*>(take(sizeof(int) * (size_t)B * m));
  w.costp = static_cast<float*>(take(sizeof(float) * (size_t)B * ((n + 63) / 64)));
  // the counters (and the diagnostic stamps) sit at the END of the caller's workspace: pzn_emd_walk_counter_offset
  unsigned char* end = static_cast<unsigned char*>(ws) + total;
  w.walk = reinterpret_cast<unsigned long long*>(end - EMD_STAMP_BYTES - sizeof(unsigned long long) * EMD_WALK_SLOTS);
#ifdef EMD_STAMPS
  w.stamps = w.walk + EMD_WALK_SLOTS;
#endif
  return w;
}

__device__ __forceinline__ void count_walk(unsigned long long* walk, int units, int lid) {  // one thread per workgroup calls
  if (units > 0) {
    const unsigned slot = ((unsigned)lid * EMD_WALK_PER_LID + ((blockIdx.y * gridDim.x + blockIdx.x) & (EMD_WALK_PER_LID - 1))) *
                          EMD_WALK_STRIDE;
    atomicAdd(walk + slot, (unsigned long long)units);
  }
}

__device__ __forceinline__ float uniform_f32(float v) {  // the value of the first active lane, as a scalar
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// x range of the workgroup's rows +- win (rows are consecutive x-sorted points: lanes `first` and `last` hold the extremes)
__device__ __forceinline__ void row_window(float myx, int last_lane, float win, float& xlo, float& xhi) {
  const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myx), 0));
  const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myx), last_lane & 63));
  xlo = x0 - win, xhi = x1 + win;
}

// Ascending-x order of one cloud of one pair per workgroup: perm[b][i] = original index of the i-th point by
// (x, index) — a bitonic sort of u64 keys (orderable x bits << 32 | index) in LDS.  grid (2, B): cloud 1 / cloud 2.
constexpr int EMD_ST = 1024;
__global__ __launch_bounds__(EMD_ST) void emd_sort_x_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                            int n, int m, int npow, int mpow, int* __restrict__ perm0,
                                                            int* __restrict__ perm1) {
  extern __shared__ unsigned long long skeys[];
  const int which = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int cnt = which ? m : n, pw = which ? mpow : npow;
  const float* src = (which ? xyz2 : xyz1) + (size_t)b * cnt * 3;
  for (int i = tid; i < pw; i += EMD_ST) {
    unsigned long long k = ~0ull;
    if (i < cnt) {
      uint32_t u = __float_as_uint(src[(size_t)i * 3]);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
      k = ((unsigned long long)u << 32) | (uint32_t)i;
    }
    skeys[i] = k;
  }
  __syncthreads();
  // A thread handles elements tid, tid + 1024, ...: for j < 64 an element's partner i ^ j belongs to the same wavefront
  // (same 64-aligned block), so those steps need no workgroup barrier - 15 of the 66 steps of a 2048-key sort do
  // (j >= 64), and one more in front of each run of them.
  for (int k2 = 2; k2 <= pw; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      if (j >= 64) __syncthreads();       // partners in another wavefront's block: its previous step must be visible
      for (int i = tid; i < pw; i += EMD_ST) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = skeys[i], c = skeys[p];
          const bool up = (i & k2) == 0;
          if ((a > c) == up) skeys[i] = c, skeys[p] = a;
        }
      }
      if (j >= 64) __syncthreads(); else pzn::wave_lds_sync();
    }
  __syncthreads();
  int* perm = (which ? perm1 : perm0) + (size_t)b * cnt;
  for (int i = tid; i < cnt; i += EMD_ST) perm[i] = (int)(uint32_t)skeys[i];
}

__global__ __launch_bounds__(EF_T) void emdf_init_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2, int n,
                                                         int m, float multiL, float multiR, EmdF w) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * EF_T + threadIdx.x;
  if (i == 0) w.cnt[0][b] = m;
  const int gk = (n + 63) / 64;
  if (i < gk) w.costp[(size_t)b * gk + i] = 0.f;
  if (b == 0 && blockIdx.x == 0)
    for (int s_ = threadIdx.x; s_ < EMD_WALK_SLOTS; s_ += EF_T) w.walk[s_] = 0ull;
  if (i < n) {
    const float* p = xyz1 + ((size_t)b * n + w.perm[0][(size_t)b * n + i]) * 3;
    w.pk1[(size_t)b * n + i] = make_float4(p[0], p[1], p[2], 0.f);
    w.st1[(size_t)b * n + i] = make_float4(0.f, 0.f, 0.f, multiL);  // :41-42
  }
  if (i < m) {
    const float* p = xyz2 + ((size_t)b * m + w.perm[1][(size_t)b * m + i]) * 3;
    w.lst[0][(size_t)b * m + i] = make_float4(p[0], p[1], p[2], multiR);  // :43-44: every point of cloud 2 starts with mass
    w.lidx[0][(size_t)b * m + i] = i;
    w.st2[(size_t)b * m + i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// The next level's list: the entries of list `from` that still hold mass, in order.  One workgroup per pair — the EXTRA
// workgroup (blockIdx.x == gridDim.x - 1) of the C + A launch of the level: that walk reads list `from` and nothing of the
// new list, pass B of the next level is the first to need it, so the 5 us launch of a compaction kernel per level is gone
// and the compaction itself (3 us of latency) disappears beside the walk.
__device__ __forceinline__ void emdf_compact_wg(int m, const EmdF& w, int from, int* scratch) {
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, to = from ^ 1;
  const int cnt = w.cnt[from][b];
  const float4* __restrict__ src = w.lst[from] + (size_t)b * m;
  const int* __restrict__ sidx = w.lidx[from] + (size_t)b * m;
  float4* __restrict__ dst = w.lst[to] + (size_t)b * m;
  int* __restrict__ didx = w.lidx[to] + (size_t)b * m;
  // thread t owns the contiguous run [t E, (t + 1) E) of the list: count, ONE prefix sum over the 256 threads, write in order
  const int E = (cnt + EF_T - 1) / EF_T;
  const int r0 = tid * E, r1 = min(cnt, r0 + E);
  int mine = 0;
  for (int r = r0; r < r1; ++r) mine += src[r].w > 0.f ? 1 : 0;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, PZN_WAVE);
    incl += lane >= o ? t : 0;
  }
  if (lane == 63) scratch[wv] = incl;
  __syncthreads();
  int off = incl - mine, total = 0;
#pragma unroll
  for (int q = 0; q < EF_T / 64; ++q) {
    const int c = scratch[q];
    off += q < wv ? c : 0;
    total += c;
  }
  for (int r = r0; r < r1; ++r) {
    const float4 v = src[r];
    if (v.w > 0.f) dst[off] = v, didx[off] = sidx[r], ++off;
  }
  if (tid == 0) w.cnt[to][b] = total;
}

// Rows = points k of cloud 1, walking list `buf` of cloud 2.
//   MODE 0: pass A of the first level            (c_next = its scale)
//   MODE 1: pass C of level j (scale c) + pass A of level j - 1 (scale c_next); SQ: c == 4 c_next, c_next != 0
//   MODE 2: pass C of the last level (scale c)
// Per pair of walked points: 3 differences, 3 for the squared distance, 1 scale, 2 v_exp_f32, (SQ: 2 squarings), the two
// weights, and the sums = 16 packed instructions + 2 exponentials in MODE 1 (round 4: 16 + 4 + three vector instructions of
// loop control per pair; the loop is scalar now).
template <int MODE, bool SQ>
__global__ __launch_bounds__(EF_T, 8) void emdf_k_kernel(int n, int m, float c, float c_next, EmdF w, int buf, float win,
                                                      int lid) {
  constexpr int NF4 = MODE == 1 ? 3 : 2;                       // float4 per point pair of the tile
  constexpr int NV = MODE == 0 ? 1 : (MODE == 1 ? 6 : 5);      // sums per row
  __shared__ float4 tile[2][EF_NP * NF4];
  static_assert(sizeof(float4) * 2 * EF_NP * NF4 >= sizeof(float) * (NV * EF_T + 4), "the partial sums reuse the tiles");
  EMD_STAMP_BEGIN();
  const int gk = (n + 63) / 64;        // row blocks; MODE 1 launches gk + 1 workgroups per pair
  if (MODE == 1 && (int)blockIdx.x == gk) {      // the extra one builds the next level's list beside the walk
    emdf_compact_wg(m, w, buf, reinterpret_cast<int*>(&tile[0][0]));
    EMD_STAMP_END(w, lid);
    return;
  }
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = blockIdx.x * 64 + lane;
  const size_t row = (size_t)b * n + k;
  float4 me = make_float4(0.f, 0.f, 0.f, 0.f), st = me;
  if (k < n) me = w.pk1[row];
  const int cnt = w.cnt[buf][b];
  const float4* __restrict__ L = w.lst[buf] + (size_t)b * m;
  const float* __restrict__ RR = w.rr + (size_t)b * m;
  float xlo, xhi;
  row_window(me.x, min(64, n - (int)blockIdx.x * 64) - 1, win, xlo, xhi);
  v2f al = {0.f, 0.f}, ax = al, ay = al, az = al, ac = al, aa = al;
  const int ntile = (cnt + EF_TL - 1) / EF_TL;
  float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;      // staging registers: the next tile's records are in flight while this one is walked
  float r0 = 0.f, r1 = 0.f;
  auto fetch = [&](int base) {
    const int i0 = base + 2 * tid;
    if (tid < EF_NP && i0 < cnt) {
      sa = L[i0];
      if (MODE != 0) r0 = RR[i0];
      if (i0 + 1 < cnt) {
        sb = L[i0 + 1];
        if (MODE != 0) r1 = RR[i0 + 1];
      } else {
        sb.x = sa.x, sb.y = sa.y, sb.z = sa.z, sb.w = 0.f, r1 = 0.f;   // odd tail: a zero-weight copy
      }
    }
  };
  auto put = [&](int tb, int base) {
    if (tid < EF_NP && base + 2 * tid < cnt) {
      float4* t = tile[tb] + NF4 * tid;
      t[0] = make_float4(sa.x, sb.x, sa.y, sb.y);
      if (MODE == 2) {
        t[1] = make_float4(sa.z, sb.z, r0, r1);
      } else {
        t[1] = make_float4(sa.z, sb.z, sa.w, sb.w);
        if (MODE == 1) t[2] = make_float4(r0, r1, 0.f, 0.f);
      }
    }
  };
  int nev = 0;
  if (ntile > 0) fetch(0), put(0, 0);
  __syncthreads();
  EMD_STAMP(1);
  for (int t = 0; t < ntile; ++t) {
    const int base = t * EF_TL;
    const int npair = (min(EF_TL, cnt - base) + 1) >> 1;
    const bool more = t + 1 < ntile;
    if (more) fetch(base + EF_TL);
    const float4* tp = tile[t & 1];
    const int per = (npair + 3) >> 2;
    const int q0 = min(npair, wq * per), q1 = min(npair, q0 + per);
    if (q0 < q1) {
      const float xfirst = uniform_f32(tp[NF4 * q0].x), xlast = uniform_f32(tp[NF4 * (q1 - 1)].y);
      if (!(xlast < xlo || xfirst > xhi)) {     // else: every exponential of this quarter tile is exactly +0
        nev += q1 - q0;
#pragma unroll 2
        for (int q = q0; q < q1; ++q) {
          const float4 p0 = tp[NF4 * q], p1 = tp[NF4 * q + 1];
          const v2f dx = me.x - (v2f){p0.x, p0.y}, dy = me.y - (v2f){p0.z, p0.w}, dz = me.z - (v2f){p1.x, p1.y};
          const v2f d = dx * dx + dy * dy + dz * dz;                      // :76 / :136-138
          if (MODE == 0) {
            aa += exp2_pair(d * c_next) * (v2f){p1.z, p1.w};              // :77-78
          } else {
            v2f eC, WB;
            if (MODE == 1) {
              const float2 p2 = *reinterpret_cast<const float2*>(tp + NF4 * q + 2);
              WB = (v2f){p2.x, p2.y};
              const v2f eA = exp2_pair(d * c_next);
              if (SQ) {
                const v2f e2 = eA * eA;
                eC = e2 * e2;                                             // exp2(c d) with c = 4 c_next
              } else {
                eC = exp2_pair(d * c);
              }
              aa += eA * (v2f){p1.z, p1.w};                               // :77-78 of the next level
            } else {
              WB = (v2f){p1.z, p1.w};
              eC = exp2_pair(d * c);
            }
            const v2f wv = eC * WB;                                       // :145 without the row's factor ratioL_k
            al += wv;                                                     // :147
            ax += wv * dx;
            ay += wv * dy;
            az += wv * dz;
            ac += wv * d;
          }
        }
      }
    }
    if (more) put((t + 1) & 1, base + EF_TL);
    __syncthreads();
  }
  EMD_STAMP(2);
  if (k < n && tid < 64) st = w.st1[row];                  // in flight while the partial sums meet
  float* red = reinterpret_cast<float*>(&tile[0][0]);      // every wavefront is past its last tile read
  if (lane == 0) reinterpret_cast<int*>(red)[NV * EF_T + wq] = 2 * nev;
  if (MODE != 0) {
    red[0 * EF_T + tid] = al.x + al.y;
    red[1 * EF_T + tid] = ax.x + ax.y;
    red[2 * EF_T + tid] = ay.x + ay.y;
    red[3 * EF_T + tid] = az.x + az.y;
    red[4 * EF_T + tid] = ac.x + ac.y;
  }
  if (MODE != 2) red[(NV - 1) * EF_T + tid] = aa.x + aa.y;
  __syncthreads();
  EMD_STAMP(3);
  if (tid < 64) {
    auto total = [&](int v) {
      const float* r = red + v * EF_T + tid;
      return (r[0] + r[64]) + (r[128] + r[192]);
    };
    float sc = 0.f;
    if (k < n) {
      if (MODE != 0) {
        const float rl = me.w;                                            // :139
        st.w = fmaxf(0.0f, st.w - rl * total(0));                         // :153
        const float g = 2.f * rl;
        st.x += g * total(1), st.y += g * total(2), st.z += g * total(3);
        sc = rl * total(4);
        w.st1[row] = st;
      }
      if (MODE != 2) w.pk1[row].w = st.w / (1e-9f + total(NV - 1));       // :59, :83
    }
    if (MODE != 0) {  // this wavefront holds the 64 row totals; the cell is this workgroup's own
      sc = pzn::wave_sum_f32(sc);
      if (tid == 0) w.costp[(size_t)b * gk + blockIdx.x] += sc;
    }
    if (tid == 0) {
      const int* nv = reinterpret_cast<const int*>(red) + NV * EF_T;
      count_walk(w.walk, nv[0] + nv[1] + nv[2] + nv[3], lid);
    }
  }
  EMD_STAMP_END(w, lid);
}

// Pass B: the rows are the entries r of list `buf`; the walk covers all of cloud 1 inside the level's x window.  SUB = 1 ... 16
// adjacent lanes share a row (64 ... 4 rows per workgroup), each taking every SUB-th point pair of its wavefront's
// quarter tile; they meet by lane shuffles, the four wavefronts through LDS.  SUB is chosen per pair so that a launch
// has at least 4 workgroups per CU however short the lists are (emdf_b_kernel; 2 / 4 / 8 per CU measured equal within noise).
template <int SUB>
__device__ __forceinline__ void emdf_b_rows(int n, int m, float c, const EmdF& w, int buf, int cnt, int rb,
                                            float4 (*tile)[EF_NP * 2], float win, int lid) {
  constexpr int RPB = 64 / SUB;
  if (rb * RPB >= cnt) return;  // workgroup-uniform
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wq = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = lane & (SUB - 1), r = rb * RPB + lane / SUB;
  const bool valid = r < cnt;
  const size_t lr = (size_t)b * m + r;
  float4 me = make_float4(0.f, 0.f, 0.f, 0.f), s2 = me;
  size_t li = 0;
  if (valid) me = w.lst[buf][lr];
  float xlo, xhi;      // the workgroup's rows are consecutive list entries (ascending x)
  row_window(me.x, (min(RPB, cnt - rb * RPB) - 1) * SUB, win, xlo, xhi);
  const float4* __restrict__ P = w.pk1 + (size_t)b * n;
  v2f ar = {0.f, 0.f}, ax = ar, ay = ar, az = ar;
  const int ntile = (n + EF_TL - 1) / EF_TL;
  float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
  auto fetch = [&](int base) {
    const int i0 = base + 2 * tid;
    if (tid < EF_NP && i0 < n) {
      sa = P[i0];
      if (i0 + 1 < n) {
        sb = P[i0 + 1];
      } else {                 // odd tail: a zero-weight copy  (written out: a select between the two would go through memory)
        sb.x = sa.x, sb.y = sa.y, sb.z = sa.z, sb.w = 0.f;
      }
    }
  };
  auto put = [&](int tb, int base) {
    if (tid < EF_NP && base + 2 * tid < n) {
      tile[tb][2 * tid] = make_float4(sa.x, sb.x, sa.y, sb.y);
      tile[tb][2 * tid + 1] = make_float4(sa.z, sb.z, sa.w, sb.w);
    }
  };
  int nev = 0;
  fetch(0), put(0, 0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int base = t * EF_TL;
    const int npair = (min(EF_TL, n - base) + 1) >> 1;
    const bool more = t + 1 < ntile;
    if (more) fetch(base + EF_TL);
    const float4* tp = tile[t & 1];
    const int per = (npair + 3) >> 2;
    const int q0 = min(npair, wq * per), q1 = min(npair, q0 + per);
    if (q0 < q1) {
      const float xfirst = uniform_f32(tp[2 * q0].x), xlast = uniform_f32(tp[2 * (q1 - 1)].y);
      if (!(xlast < xlo || xfirst > xhi)) {  // else: every product underflows to +0
        nev += q1 - q0;
#pragma unroll 2
        for (int q = q0 + sub; q < q1; q += SUB) {
          const float4 p0 = tp[2 * q], p1 = tp[2 * q + 1];
          const v2f dx = me.x - (v2f){p0.x, p0.y}, dy = me.y - (v2f){p0.z, p0.w}, dz = me.z - (v2f){p1.x, p1.y};
          const v2f e = exp2_pair((dx * dx + dy * dy + dz * dz) * c) * (v2f){p1.z, p1.w};  // :108
          ar += e;                                                                          // :109
          ax += e * dx;
          ay += e * dy;
          az += e * dz;
        }
      }
    }
    if (more) put((t + 1) & 1, base + EF_TL);
    __syncthreads();
  }
  if (valid && tid < 64 && sub == 0) {                   // in flight while the partial sums meet
    li = (size_t)b * m + w.lidx[buf][lr];
    s2 = w.st2[li];
  }
  float sr = ar.x + ar.y, sx = ax.x + ax.y, sy = ay.x + ay.y, sz = az.x + az.y;
#pragma unroll
  for (int o = 1; o < SUB; o <<= 1) {
    sr += __shfl_xor(sr, o, PZN_WAVE);
    sx += __shfl_xor(sx, o, PZN_WAVE);
    sy += __shfl_xor(sy, o, PZN_WAVE);
    sz += __shfl_xor(sz, o, PZN_WAVE);
  }
  float* red = reinterpret_cast<float*>(&tile[0][0]);
  red[0 * EF_T + tid] = sr;
  red[1 * EF_T + tid] = sx;
  red[2 * EF_T + tid] = sy;
  red[3 * EF_T + tid] = sz;
  if (lane == 0) reinterpret_cast<int*>(red)[4 * EF_T + wq] = (2 * nev + SUB - 1) / SUB;   // RPB rows x 2 nev points, in units of 64 evaluations
  __syncthreads();
  if (tid == 0) {
    const int* nv = reinterpret_cast<const int*>(red) + 4 * EF_T;
    count_walk(w.walk, nv[0] + nv[1] + nv[2] + nv[3], lid);
  }
  if (valid && tid < 64 && sub == 0) {
    auto total = [&](int v) {
      const float* q = red + v * EF_T + tid;
      return (q[0] + q[64]) + (q[128] + q[192]);
    };
    const float remainR = me.w;
    const float sumr = remainR * total(0);                              // :114
    const float consumption = fminf(remainR / (sumr + 1e-9f), 1.0f);    // :115
    const float ratioR = consumption * remainR;                         // :116
    w.lst[buf][lr].w = fmaxf(0.0f, remainR - sumr);                     // :117
    w.rr[lr] = ratioR;
    const float s = 2.f * ratioR;
    s2.x += s * total(1), s2.y += s * total(2), s2.z += s * total(3);
    w.st2[li] = s2;
  }
}

__global__ __launch_bounds__(EF_T, 8) void emdf_b_kernel(int n, int m, float c, EmdF w, int buf, float win, int lid) {
  __shared__ float4 tile[2][EF_NP * 2];
  static_assert(sizeof(float4) * 2 * EF_NP * 2 >= sizeof(float) * (4 * EF_T + 4), "the partial sums reuse the tiles");
  EMD_STAMP_BEGIN();
  const int cnt = w.cnt[buf][blockIdx.y];
  const int gx = gridDim.x;
  const int target = max(1, 1024 / (int)gridDim.y);     // workgroups per pair: 4 - 8 per CU over the launch however short the lists (a workgroup is
                                                        // then at most an eighth of a CU's share: the launch ends with its most loaded CU)
  int sub = 1;
  while (sub < 16 && ((cnt * sub + 63) >> 6) < target && ((cnt * sub * 2 + 63) >> 6) <= gx) sub <<= 1;
  // Workgroups are dealt round-robin over the XCDs and only the first few row blocks of a pair have rows: rotate the
  // blocks by the pair so that the surplus does not always land on the same XCDs.
  const int rb = (int)((blockIdx.x + 3u * blockIdx.y) % (unsigned)gx);
  if (sub == 1)
    emdf_b_rows<1>(n, m, c, w, buf, cnt, rb, tile, win, lid);
  else if (sub == 2)
    emdf_b_rows<2>(n, m, c, w, buf, cnt, rb, tile, win, lid);
  else if (sub == 4)
    emdf_b_rows<4>(n, m, c, w, buf, cnt, rb, tile, win, lid);
  else if (sub == 8)
    emdf_b_rows<8>(n, m, c, w, buf, cnt, rb, tile, win, lid);
  else
    emdf_b_rows<16>(n, m, c, w, buf, cnt, rb, tile, win, lid);
  EMD_STAMP_END(w, lid);
}

// gradients back in the caller's point order
__global__ __launch_bounds__(EF_T) void emdf_finish_kernel(int n, int m, EmdF w, float* __restrict__ cost, float* __restrict__ g1,
                                                           float* __restrict__ g2) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * EF_T + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 64) {      // cost of the pair: its row blocks' cells in a fixed order
    const int gk = (n + 63) / 64;
    float s = 0.f;
    for (int q = threadIdx.x; q < gk; q += 64) s += w.costp[(size_t)b * gk + q];
    s = pzn::wave_sum_f32(s);
    if (threadIdx.x == 0) cost[b] = s;
  }
  if (i < n) {
    const float4 s = w.st1[(size_t)b * n + i];
    float* g = g1 + ((size_t)b * n + w.perm[0][(size_t)b * n + i]) * 3;
    g[0] = s.x, g[1] = s.y, g[2] = s.z;
  }
  if (i < m) {
    const float4 s = w.st2[(size_t)b * m + i];
    float* g = g2 + ((size_t)b * m + w.perm[1][(size_t)b * m + i]) * 3;
    g[0] = s.x, g[1] = s.y, g[2] = s.z;
  }
}

int run_fused(const float* xyz1, const float* xyz2, int B, int n, int m, float* cost, float* g1, float* g2, void* workspace,
              hipStream_t st) {
  EmdF w = carve_f(workspace, pzn_emd_workspace_bytes(B, n, m), B, n, m);
  const float multiL = n >= m ? 1.f : (float)(m / n), multiR = n >= m ? (float)(n / m) : 1.f;  // :29-35 (integer division)
  int npow = 1, mpow = 1;
  while (npow < n) npow <<= 1;
  while (mpow < m) mpow <<= 1;
  const size_t lds = sizeof(unsigned long long) * (size_t)(npow > mpow ? npow : mpow);
  if (lds > 150 * 1024) return PZN_EUNSUPPORTED;      // (> 16384 points per cloud)
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(emd_sort_x_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PZN_ELAUNCH;
  PZN_LAUNCH(emd_sort_x_kernel, dim3(2, B), dim3(EMD_ST), lds, st, xyz1, xyz2, n, m, npow, mpow, w.perm[0], w.perm[1]);
  const int mx = n > m ? n : m;
  const dim3 gi((mx + EF_T - 1) / EF_T, B), gk((n + 63) / 64, B), gk1((n + 63) / 64 + 1, B);   // gk1: + the compaction workgroup
  PZN_LAUNCH(emdf_init_kernel, gi, dim3(EF_T), 0, st, xyz1, xyz2, n, m, multiL, multiR, w);
  auto cof = [](int j) {                                         // :47-50, * log2(e): __expf(level d) = exp2(level log2(e) d)
    const float level = j == -2 ? 0.f : -powf(4.0f, (float)j);
    return level * 1.44269504088896340736f;
  };
  // x window of a level: exp2(c d^2) with c d^2 <= -150 is exactly +0 in fp32 (below the smallest denormal), and
  // d^2 >= (x distance)^2: points farther than sqrt(150 / -c) along x are not walked (level 0: no window)
  constexpr float win_bits = 150.f;  // tuning aid
  auto winf = [&](int j) {
    const float c = cof(j);
    return c < 0.f ? sqrtf(win_bits / -c) : INFINITY;
  };
  const int target = 1024 / B > 1 ? 1024 / B : 1;
  const dim3 gb((unsigned)((m + 63) / 64 > target ? (m + 63) / 64 : target), B);
  // A(7); then per level B, and C fused with the next level's A (+ the next list); the last level ends with a plain C
  PZN_LAUNCH((emdf_k_kernel<0, false>), gk, dim3(EF_T), 0, st, n, m, 0.f, cof(7), w, 0, winf(7), 0);
  for (int j = 7, buf = 0; j >= -2; --j, buf ^= 1) {  // list `buf` = points of cloud 2 with mass at the start of level j
    const int lb = 1 + 2 * (7 - j);
    PZN_LAUNCH(emdf_b_kernel, gb, dim3(EF_T), 0, st, n, m, cof(j), w, buf, winf(j), lb);
    if (j > -2) {
      // the walk covers the list of THIS level: ratioR of pass C is non-zero exactly there, the points pass B has just
      // exhausted carry remainR = 0 into the next level's sum; window of the SOFTER level: outside it both terms are +0
      if (j >= 0)
        PZN_LAUNCH((emdf_k_kernel<1, true>), gk1, dim3(EF_T), 0, st, n, m, cof(j), cof(j - 1), w, buf,
                           winf(j - 1), lb + 1);
      else       // the next scale is 0: its exponential is 1, nothing to square
        PZN_LAUNCH((emdf_k_kernel<1, false>), gk1, dim3(EF_T), 0, st, n, m, cof(j), cof(j - 1), w, buf,
                           winf(j - 1), lb + 1);
    } else {
      PZN_LAUNCH((emdf_k_kernel<2, false>), gk, dim3(EF_T), 0, st, n, m, cof(j), 0.f, w, buf, winf(j), lb + 1);
    }
  }
  PZN_LAUNCH(emdf_finish_kernel, gi, dim3(EF_T), 0, st, n, m, w, cost, g1, g2);
  PZN_RETURN_LAUNCH_STATUS();
}

int run_match(const float* xyz1, const float* xyz2, int B, int n, int m, float* match, void* workspace, hipStream_t st) {
  EmdWs w = carve(workspace, B, n, m);
  const float multiL = n >= m ? 1.f : (float)(m / n), multiR = n >= m ? (float)(n / m) : 1.f;  // :29-35 (integer division)
  const int mx = n > m ? n : m;
  dim3 gi((mx + EMD_T - 1) / EMD_T, B), gk((n + EMD_ROWS - 1) / EMD_ROWS, B), gl((m + EMD_ROWS - 1) / EMD_ROWS, B);
  PZN_LAUNCH(emd_init_kernel, gi, dim3(EMD_T), 0, st, xyz1, xyz2, n, m, multiL, multiR, w);
  if (pzn_zero_async(match, (size_t)B * n * m, st) != PZN_OK) return PZN_ELAUNCH;  // :39-40
  for (int j = 7; j >= -2; --j) {                                // :46
    const float level = j == -2 ? 0.f : -powf(4.0f, (float)j);   // :47-50
    const float c = level * 1.44269504088896340736f;
    PZN_LAUNCH(emd_pass_a_kernel, gk, dim3(EMD_T), 0, st, n, m, c, w);
    PZN_LAUNCH(emd_pass_b_kernel, gl, dim3(EMD_T), 0, st, n, m, c, w);
    PZN_LAUNCH(emd_pass_c_kernel, gk, dim3(EMD_T), 0, st, n, m, c, w, match);
  }
  PZN_RETURN_LAUNCH_STATUS();
}

}  // namespace

PZN_EXPORT size_t pzn_emd_workspace_bytes(int B, int n, int m) {
  if (B <= 0 || n <= 0 || m <= 0) return 0;
  const size_t a = ws3_bytes(B, n, m), f = wsf_bytes(B, n, m);
  return (a > f ? a : f) + sizeof(unsigned long long) * EMD_WALK_SLOTS + EMD_STAMP_BYTES;
}

// Byte offset, inside the workspace, of pzn_emd_walk_counter_count() uint64 counters that the fused entry point leaves
// behind: their sum x 64 is the number of (row, point) pair evaluations its passes executed (points of exhausted mass and
// points outside the level's x window are not walked); counters (16 i + s) * 16, s = 0..15, belong to launch i
// (EmdF::walk), the others stay 0; (size_t)-1 when the call takes the single-workgroup path (n, m <= 256), which
// evaluates all 30 n m.
PZN_EXPORT size_t pzn_emd_walk_counter_offset(int B, int n, int m) {
  if (B <= 0 || n <= 0 || m <= 0 || (n <= EMD_SMALL_MAX && m <= EMD_SMALL_MAX)) return (size_t)-1;
  return pzn_emd_workspace_bytes(B, n, m) - sizeof(unsigned long long) * EMD_WALK_SLOTS - EMD_STAMP_BYTES;
}

PZN_EXPORT int pzn_emd_walk_counter_count(void) { return EMD_WALK_SLOTS; }

PZN_EXPORT int pzn_emd_approxmatch_f32(const float* xyz1, const float* xyz2, int B, int n, int m, float* match,
                                       void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz1 && xyz2 && match && workspace && B > 0 && n > 0 && m > 0 && B <= 65535);
  PZN_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
  return run_match(xyz1, xyz2, B, n, m, match, workspace, pzn_hip_stream(stream));
}

static EmdSmallProblem emd_small_problem(const float* xyz1, const float* xyz2, int n, int m, float* cost, float* g1, float* g2,
                                         int first) {
  EmdSmallProblem p;
  p.xyz1 = xyz1, p.xyz2 = xyz2, p.cost = cost, p.g1 = g1, p.g2 = g2, p.n = n, p.m = m, p.first = first;
  p.multiL = n >= m ? 1.f : (float)(m / n), p.multiR = n >= m ? (float)(n / m) : 1.f;  // :29-35
  int rows = 1;
  while (rows < (n > m ? n : m)) rows <<= 1;
  int shift = 0;
  while ((rows << shift) < EMD_SMALL_T && shift < 6) ++shift;  // lanes sharing a row stay inside one wavefront
  p.tpr_shift = shift;
  return p;
}

// Several small calls (n, m <= 256 each; 1 <= count <= 4) of pzn_emd_fused_f32 as ONE launch: problem i is xyz1[i][B[i], n[i], 3]
// against xyz2[i][B[i], m[i], 3] -> cost[i][B[i]], g1[i], g2[i] (HOST arrays of device pointers / sizes).  The results are those
// of count separate calls, bit for bit.  PZN_EUNSUPPORTED when a problem is larger (callers then make the calls one by one).
PZN_EXPORT int pzn_emd_fused_small_multi_f32(int count, const float* const* xyz1, const float* const* xyz2, const int* B,
                                             const int* n, const int* m, float* const* cost, float* const* g1,
                                             float* const* g2, pzn_stream_t stream) {
  PZN_CHECK_ARG(count >= 1 && count <= EMD_SMALL_NP && xyz1 && xyz2 && B && n && m && cost && g1 && g2);
  EmdSmallArgs a;
  a.count = count;
  int first = 0;
  for (int i = 0; i < count; ++i) {
    PZN_CHECK_ARG(xyz1[i] && xyz2[i] && cost[i] && g1[i] && g2[i] && B[i] > 0 && n[i] > 0 && m[i] > 0 && B[i] <= 65535);
    if (n[i] > EMD_SMALL_MAX || m[i] > EMD_SMALL_MAX) return PZN_EUNSUPPORTED;
    a.pr[i] = emd_small_problem(xyz1[i], xyz2[i], n[i], m[i], cost[i], g1[i], g2[i], first);
    first += B[i];
  }
  for (int i = count; i < EMD_SMALL_NP; ++i) a.pr[i] = a.pr[0];
  PZN_LAUNCH(emd_small_fused_kernel, dim3((unsigned)first), dim3(EMD_SMALL_T), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_emd_fused_f32(const float* xyz1, const float* xyz2, int B, int n, int m, float* cost, float* g1,
                                 float* g2, void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz1 && xyz2 && cost && g1 && g2 && workspace && B > 0 && n > 0 && m > 0 && B <= 65535);
  PZN_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
  if (n <= EMD_SMALL_MAX && m <= EMD_SMALL_MAX) {  // one workgroup per pair, one launch for the whole auction
    EmdSmallArgs a;
    a.count = 1;
    a.pr[0] = emd_small_problem(xyz1, xyz2, n, m, cost, g1, g2, 0);
    PZN_LAUNCH(emd_small_fused_kernel, dim3((unsigned)B), dim3(EMD_SMALL_T), 0, pzn_hip_stream(stream), a);
    PZN_RETURN_LAUNCH_STATUS();
  }
  return run_fused(xyz1, xyz2, B, n, m, cost, g1, g2, workspace, pzn_hip_stream(stream));
}

PZN_EXPORT int pzn_emd_matchcost_f32(const float* xyz1, const float* xyz2, const float* match, int B, int n, int m,
                                     float* cost, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz1 && xyz2 && match && cost && B > 0 && n > 0 && m > 0 && B <= 65535);
  hipStream_t st = pzn_hip_stream(stream);
  if (pzn_zero_async(cost, (size_t)B, st) != PZN_OK) return PZN_ELAUNCH;
  dim3 grid((n + EMD_T - 1) / EMD_T, (m + MC_LSLAB - 1) / MC_LSLAB, B);
  PZN_CHECK_ARG(grid.y <= 65535);
  PZN_LAUNCH(emd_matchcost_kernel, grid, dim3(EMD_T), 0, st, xyz1, xyz2, match, n, m, cost);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_emd_matchcost_grad_f32(const float* grad_cost, const float* xyz1, const float* xyz2,
                                          const float* match, int B, int n, int m, float* grad1, float* grad2,
                                          pzn_stream_t stream) {
  PZN_CHECK_ARG(grad_cost && xyz1 && xyz2 && match && grad1 && grad2 && B > 0 && n > 0 && m > 0 && B <= 65535);
  hipStream_t st = pzn_hip_stream(stream);
  PZN_LAUNCH(emd_grad1_kernel, dim3((n + EMD_T - 1) / EMD_T, B), dim3(EMD_T), 0, st, grad_cost, xyz1, xyz2,
                     match, n, m, grad1);
  constexpr int LPB = EMD_T / PZN_WAVE;
  PZN_LAUNCH(emd_grad2_kernel, dim3((m + LPB - 1) / LPB, B), dim3(EMD_T), 0, st, grad_cost, xyz1, xyz2, match,
                     n, m, grad2);
  PZN_RETURN_LAUNCH_STATUS();
}
