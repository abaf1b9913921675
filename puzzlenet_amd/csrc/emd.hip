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
// Data layout: points are repacked once per call to float4 {x, y, z, weight}
// where `weight` is the per-point state the NEXT pass multiplies by
// (remainR / ratioL / ratioR).  A workgroup = 4 wavefronts that own the SAME 64
// rows; the OTHER cloud is staged through LDS in 512-point tiles as a pair-SoA
// image {x0,x1,y0,y1}{z0,z1,w0,w1}, every wavefront walks a quarter of each tile
// with a wave-uniform index (broadcast ds_read_b128) and packed fp32 arithmetic
// (v_pk_add / v_pk_mul / v_pk_fma: two (k,l) pairs per instruction), and the four
// partial sums meet through LDS.  (The first version fed the walked points to the
// VALU as SGPR operands through the scalar cache; that cache thrashed at ~10
// cycles per VALU issue.)
//
// Pass structure per level (emd_kernel.cu line numbers):
//   A (:51-84)   ratioL_k  = remainL_k / (1e-9 + sum_l e_kl remainR_l)
//   B (:86-119)  s_l = remainR_l sum_k e_kl ratioL_k;
//                ratioR_l = min(remainR_l/(s_l+1e-9), 1) remainR_l;  remainR_l = max(0, remainR_l - s_l)
//   C (:121-154) w_kl = e_kl ratioL_k ratioR_l;  match_lk += w_kl;  remainL_k = max(0, remainL_k - sum_l w_kl)
// with e_kl = exp(level * d_kl), level = -4^j for j = 7..-1 and 0 for the last.
//
// Fused entry point, large clouds: A(7); per level B over the ACTIVE points of cloud 2 (emd_pass_b_list_kernel), the
// next level's active list (emd_compact_kernel), then C fused with the next level's A, walking the active list
// (emd_pass_ca_kernel).  A point with remainR_l == 0 stays at 0 (:114-117) and contributes exactly +0 to the sums
// of A and C, so leaving it out changes nothing but the summation order.
#include <stdlib.h>

#include "pzn_common.h"

namespace {

constexpr int EMD_T = 256;  // threads per workgroup (4 wavefronts)
constexpr int EMD_SMALL_MAX = 256;  // n, m up to here: the whole auction of a pair in one workgroup (emd_small_fused_kernel)

struct EmdWs {
  float4* pk1;     // [B*n] {x1,y1,z1, ratioL}
  float4* pk2a;    // [B*m] {x2,y2,z2, remainR}
  float4* pk2b;    // [B*m] {x2,y2,z2, ratioR}
  float* remainL;  // [B*n]
  int* act[2];     // [B*m] each: ascending indices l of the points of cloud 2 that still hold mass (remainR_l > 0)
  int* cnt[2];     // [B] each: how many
  int* perm[2];    // [B*n], [B*m]: x-sorted position -> original index (fused entry point: clouds are walked in x order)
  unsigned long long* walk;  // EMD_WALK_SLOTS counters (measurement only): units of 64 (row, point) evaluations executed
                             // by the last fused call, one slot per (workgroup, wavefront) modulo the slot count
};
constexpr int EMD_WALK_SLOTS = 1024;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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
  p += align_up(sizeof(float) * (size_t)B * n, 256);
  for (int i = 0; i < 2; ++i) {
    w.act[i] = reinterpret_cast<int*>(p);
    p += align_up(sizeof(int) * (size_t)B * m, 256);
  }
  for (int i = 0; i < 2; ++i) {
    w.cnt[i] = reinterpret_cast<int*>(p);
    p += align_up(sizeof(int) * (size_t)B, 256);
  }
  w.perm[0] = reinterpret_cast<int*>(p);
  p += align_up(sizeof(int) * (size_t)B * n, 256);
  w.perm[1] = reinterpret_cast<int*>(p);
  p += align_up(sizeof(int) * (size_t)B * m, 256);
  w.walk = reinterpret_cast<unsigned long long*>(p);
  return w;
}

__device__ __forceinline__ float fast_exp_scaled(float c_log2e, float d) {
  // __expf(level*d) == exp2(level*d*log2e); level*log2e is folded on the host.
#ifdef PZN_EMD_EXACT_EXP
  return expf((c_log2e * 0.69314718055994530942f) * d);
#else
  return __builtin_amdgcn_exp2f(c_log2e * d);
#endif
}

__device__ __forceinline__ float sq3(float dx, float dy, float dz) { return dx * dx + dy * dy + dz * dz; }

// Walk `cnt` packed points of the other cloud.  A workgroup = 4 wavefronts that own the SAME 64 rows
// (k or l = blockIdx.x*64 + lane); each wavefront walks one quarter of every tile and the four partial
// sums meet through LDS at the end.  That quadruples the wavefronts of a pass (8 per SIMD at B=64,
// n=2048, where one thread per row gives only 2) — the pass is a long dependent VALU/exp chain per
// lane, and resident waves are the only latency cover.  The walked points are staged through LDS in
// 512-point tiles (double-buffered, one barrier per tile) and read back with a wave-uniform index: a
// broadcast ds_read_b128 per point.  (A first version fed them through the scalar cache with
// s_load_dwordx16: every wave of a pair streams the same 32 KB through a small scalar cache shared
// between CUs, which thrashes — ~10 cycles per VALU instruction measured.)
constexpr int EMD_TL = 512;
constexpr int EMD_ROWS = 64;  // rows per workgroup
typedef float v2f __attribute__((ext_vector_type(2)));

// The tile holds point PAIRS in SoA form — {x0,x1,y0,y1} {z0,z1,w0,w1} — so that one lane evaluates two
// walked points per instruction with the packed fp32 ALU ops (v_pk_add/mul/fma_f32): ~8 issues per
// (row, point) instead of ~21 for the float4-per-point form (whose pairs the compiler had to assemble
// with v_mov).  An odd tail point is paired with a zero-weight copy of itself.
// X window: the fused entry point walks both clouds in ascending x (emd_sort_x_kernel), and a walked point farther
// than `win` from the workgroup's rows along x contributes exp2(c d^2) with c d^2 <= -150, i.e. exactly +0 in fp32: a
// wavefront whose quarter of the tile lies wholly outside [XLO, XHI] skips it (half the time of a 2048 x 2048 call is
// spent in the two sharpest levels, where > 98 % of the exponentials underflow).  XLO > XHI never happens; pass
// (-INFINITY, INFINITY) for "no window".  NEV counts the point pairs this wavefront really evaluated.
#define EMD_WALK(PTR, IX, CNT, EVAL2, XLO, XHI, NEV)                            \
  do {                                                                        \
    __shared__ float4 emd_tile_[2][EMD_TL];                                   \
    const int cnt_ = (CNT);                                                   \
    const int wq_ = threadIdx.x >> 6;                                         \
    const int ntile_ = (cnt_ + EMD_TL - 1) / EMD_TL;                          \
    auto stage_ = [&](int buf, int base) {                                    \
      const int q = threadIdx.x; /* pair index inside the tile: EMD_T == EMD_TL / 2 */ \
      const int i0 = base + 2 * q;                                            \
      if (i0 < cnt_) {                                                        \
        float4 a = (PTR)[IX(i0)];                                             \
        float4 b = i0 + 1 < cnt_ ? (PTR)[IX(i0 + 1)] : make_float4(a.x, a.y, a.z, 0.f); \
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
      if (q_ < end_ && !(tp_[2 * (end_ - 1)].y < (XLO) || tp_[2 * q_].x > (XHI))) { \
        NEV += end_ - q_;                                                     \
        for (; q_ + 1 < end_; q_ += 2) {                                      \
          float4 a0_ = tp_[2 * q_], b0_ = tp_[2 * q_ + 1], a1_ = tp_[2 * q_ + 2], b1_ = tp_[2 * q_ + 3]; \
          EVAL2((v2f){a0_.x, a0_.y}, (v2f){a0_.z, a0_.w}, (v2f){b0_.x, b0_.y}, (v2f){b0_.z, b0_.w}, base_ + 2 * q_); \
          EVAL2((v2f){a1_.x, a1_.y}, (v2f){a1_.z, a1_.w}, (v2f){b1_.x, b1_.y}, (v2f){b1_.z, b1_.w}, base_ + 2 * q_ + 2); \
        }                                                                     \
        for (; q_ < end_; ++q_) {                                             \
          float4 a0_ = tp_[2 * q_], b0_ = tp_[2 * q_ + 1];                    \
          EVAL2((v2f){a0_.x, a0_.y}, (v2f){a0_.z, a0_.w}, (v2f){b0_.x, b0_.y}, (v2f){b0_.z, b0_.w}, base_ + 2 * q_); \
        }                                                                     \
      }                                                                       \
      __syncthreads();                                                        \
    }                                                                         \
  } while (0)

// Same walk over TWO packed arrays of the same points that differ in their weight ({x,y,z,wa} and {x,y,z,wb}):
// tile of three float4 per point pair — {x0,x1,y0,y1} {z0,z1,wa0,wa1} {wb0,wb1,-,-} — for the fused C + next-A pass.
#define EMD_WALK2(PTRA, PTRB, IX, CNT, EVAL2, XLO, XHI, NEV)                    \
  do {                                                                        \
    __shared__ float4 emd_tile2_[2][EMD_TL / 2 * 3];                          \
    const int cnt_ = (CNT);                                                   \
    const int wq_ = threadIdx.x >> 6;                                         \
    const int ntile_ = (cnt_ + EMD_TL - 1) / EMD_TL;                          \
    auto stage_ = [&](int buf, int base) {                                    \
      const int q = threadIdx.x;                                              \
      const int i0 = base + 2 * q;                                            \
      if (i0 < cnt_) {                                                        \
        const int j0 = IX(i0);                                                \
        float4 a = (PTRA)[j0];                                                \
        float wb0 = (PTRB)[j0].w;                                             \
        float4 b = make_float4(a.x, a.y, a.z, 0.f);                           \
        float wb1 = 0.f;                                                      \
        if (i0 + 1 < cnt_) {                                                  \
          const int j1 = IX(i0 + 1);                                          \
          b = (PTRA)[j1], wb1 = (PTRB)[j1].w;                                 \
        }                                                                     \
        emd_tile2_[buf][3 * q] = make_float4(a.x, b.x, a.y, b.y);             \
        emd_tile2_[buf][3 * q + 1] = make_float4(a.z, b.z, a.w, b.w);         \
        emd_tile2_[buf][3 * q + 2] = make_float4(wb0, wb1, 0.f, 0.f);         \
      }                                                                       \
    };                                                                        \
    stage_(0, 0);                                                             \
    __syncthreads();                                                          \
    for (int t_ = 0; t_ < ntile_; ++t_) {                                     \
      const int base_ = t_ * EMD_TL;                                          \
      const int npair_ = (min(EMD_TL, cnt_ - base_) + 1) >> 1;                \
      if (t_ + 1 < ntile_) stage_((t_ + 1) & 1, base_ + EMD_TL);              \
      const float4* tp_ = emd_tile2_[t_ & 1];                                 \
      const int per_ = (npair_ + 3) >> 2;                                     \
      int q_ = min(npair_, wq_ * per_);                                       \
      const int end_ = min(npair_, q_ + per_);                                \
      if (q_ < end_ && !(tp_[3 * (end_ - 1)].y < (XLO) || tp_[3 * q_].x > (XHI))) { \
        NEV += end_ - q_;                                                     \
        for (; q_ < end_; ++q_) {                                             \
          float4 a0_ = tp_[3 * q_], b0_ = tp_[3 * q_ + 1], c0_ = tp_[3 * q_ + 2]; \
          EVAL2((v2f){a0_.x, a0_.y}, (v2f){a0_.z, a0_.w}, (v2f){b0_.x, b0_.y}, (v2f){b0_.z, b0_.w}, (v2f){c0_.x, c0_.y}); \
        }                                                                     \
      }                                                                       \
      __syncthreads();                                                        \
    }                                                                         \
  } while (0)

__device__ __forceinline__ v2f exp2_pair(v2f t) { return (v2f){__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)}; }

// x range of the workgroup's rows +- win (rows are consecutive x-sorted points: the first and the last valid lane hold
// the extremes); win = INFINITY or unsorted clouds (sorted == 0): no window
__device__ __forceinline__ void row_window(float myx, int nvalid, float win, float& xlo, float& xhi) {
  const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myx), 0));
  const int last = nvalid > 0 ? nvalid - 1 : 0;
  const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, myx), last & 63));
  xlo = x0 - win, xhi = x1 + win;
}

__device__ __forceinline__ void count_walk(const EmdWs& w, int pairs) {  // `pairs` point pairs x 64 rows evaluated by this wave
  if ((threadIdx.x & 63) == 0 && pairs > 0) {
    const unsigned slot = ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) & (EMD_WALK_SLOTS - 1);
    atomicAdd(w.walk + slot, (unsigned long long)(2 * pairs));
  }
}

// Ascending-x order of one cloud of one pair per workgroup: perm[b][i] = original index of the i-th point by
// (x, index) — a bitonic sort of u64 keys (orderable x bits << 32 | index) in LDS.  grid (2, B): cloud 1 / cloud 2.
constexpr int EMD_ST = 1024;
__global__ __launch_bounds__(EMD_ST) void emd_sort_x_kernel(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                            int n, int m, int npow, int mpow, EmdWs w) {
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
  for (int k2 = 2; k2 <= pw; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < pw; i += EMD_ST) {
        const int p = i ^ j;
        if (p > i) {
          const unsigned long long a = skeys[i], c = skeys[p];
          const bool up = (i & k2) == 0;
          if ((a > c) == up) skeys[i] = c, skeys[p] = a;
        }
      }
      __syncthreads();
    }
  int* perm = w.perm[which] + (size_t)b * cnt;
  for (int i = tid; i < cnt; i += EMD_ST) perm[i] = (int)(uint32_t)skeys[i];
}

// sum over the 4 wavefronts of a workgroup, lane by lane (all threads get the total)
__device__ __forceinline__ float cross_wave_sum(float v, float* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  const int l = threadIdx.x & 63;
  float t = (red[l] + red[64 + l]) + (red[128 + l] + red[192 + l]);
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(EMD_T) void emd_init_kernel(const float* __restrict__ xyz1,
                                                         const float* __restrict__ xyz2, int n, int m,
                                                         float multiL, float multiR, EmdWs w,
                                                         float* __restrict__ cost, float* __restrict__ g1,
                                                         float* __restrict__ g2, int sorted) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * EMD_T + threadIdx.x;
  if (cost) {  // fused entry point: the accumulators start at zero (no separate fills)
    if (i == 0) cost[b] = 0.f;
    if (b == 0 && blockIdx.x == 0)
      for (int s_ = threadIdx.x; s_ < EMD_WALK_SLOTS; s_ += EMD_T) w.walk[s_] = 0ull;
    if (i < n) {
      float* g = g1 + ((size_t)b * n + i) * 3;
      g[0] = 0.f, g[1] = 0.f, g[2] = 0.f;
    }
    if (i < m) {
      float* g = g2 + ((size_t)b * m + i) * 3;
      g[0] = 0.f, g[1] = 0.f, g[2] = 0.f;
    }
  }
  if (i < n) {
    const float* p = xyz1 + ((size_t)b * n + (sorted ? w.perm[0][(size_t)b * n + i] : i)) * 3;
    w.pk1[(size_t)b * n + i] = make_float4(p[0], p[1], p[2], 0.f);
    w.remainL[(size_t)b * n + i] = multiL;  // :41-42
  }
  if (i < m) {
    const float* p = xyz2 + ((size_t)b * m + (sorted ? w.perm[1][(size_t)b * m + i] : i)) * 3;
    w.pk2a[(size_t)b * m + i] = make_float4(p[0], p[1], p[2], multiR);  // :43-44
    w.pk2b[(size_t)b * m + i] = make_float4(p[0], p[1], p[2], 0.f);
    w.act[0][(size_t)b * m + i] = i;  // every point of cloud 2 starts with mass
    if (i == 0) w.cnt[0][b] = m;
  }
}

// Active list of cloud 2 for the next level.  remainR_l == 0 is absorbing (pass B: s_l *= remainR_l -> 0, ratioR_l = 0,
// remainR_l stays 0) and such a point contributes exactly +0 to every sum of passes A and C, so the three passes of
// a level only need the points that still hold mass: 57 % of them at level 5, 21 % at level 3, 2 % at level 0 on
// uniform clouds — the ten levels together cost about 3.2 full ones.  One workgroup per pair; the list is in
// ascending index order (deterministic), two lists alternate between levels.
constexpr int EMD_CT = 1024;
__global__ __launch_bounds__(EMD_CT) void emd_compact_kernel(int m, EmdWs w, int buf) {
  __shared__ int wsum[EMD_CT / 64];
  __shared__ int base_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float4* __restrict__ src = w.pk2a + (size_t)b * m;
  int* __restrict__ dst = w.act[buf] + (size_t)b * m;
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int i0 = 0; i0 < m; i0 += EMD_CT) {
    const int l = i0 + tid;
    const bool on = l < m && src[l].w > 0.f;
    const uint64_t bal = __ballot(on);
    const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wv] = __builtin_popcountll(bal);
    __syncthreads();
    int off = base_s, tot = 0;
    for (int q = 0; q < EMD_CT / 64; ++q) {
      const int c = wsum[q];
      off += q < wv ? c : 0;
      tot += c;
    }
    if (on) dst[off + before] = l;
    __syncthreads();
    if (tid == 0) base_s += tot;
    __syncthreads();
  }
  if (tid == 0) w.cnt[buf][b] = base_s;
}

// Pass A: rows = points k of xyz1.
__global__ __launch_bounds__(EMD_T) void emd_pass_a_kernel(int n, int m, float c, EmdWs w, float win) {
  __shared__ float red[EMD_T];
  const int b = blockIdx.y;
  const int k = blockIdx.x * EMD_ROWS + (threadIdx.x & 63);
  const float4* __restrict__ other = w.pk2a + (size_t)b * m;
  float4 me = k < n ? w.pk1[(size_t)b * n + k] : make_float4(0, 0, 0, 0);
  float xlo, xhi;
  row_window(me.x, min(EMD_ROWS, n - (int)blockIdx.x * EMD_ROWS), win, xlo, xhi);
  int nev = 0;
  v2f acc = {0.f, 0.f};
  auto eval = [&](v2f X, v2f Y, v2f Z, v2f Wt, int) {
    v2f dx = X - me.x, dy = Y - me.y, dz = Z - me.z;
    v2f d = dx * dx + dy * dy + dz * dz;       // :76
    acc += exp2_pair(d * c) * Wt;              // :77-78
  };
  auto ix = [](int i) { return i; };
  EMD_WALK(other, ix, m, eval, xlo, xhi, nev);
  count_walk(w, nev);
  float suml = 1e-9f + cross_wave_sum(acc.x + acc.y, red);  // :59
  if (k < n && threadIdx.x < 64) {
    me.w = w.remainL[(size_t)b * n + k] / suml;  // :83
    w.pk1[(size_t)b * n + k] = me;
  }
}

// Pass B: rows = points l of xyz2.  FUSED additionally accumulates
// grad2_l += 2 ratioR_l sum_k e_kl ratioL_k (x2_l - x1_k).
template <bool FUSED>
__global__ __launch_bounds__(EMD_T) void emd_pass_b_kernel(int n, int m, float c, EmdWs w, float* __restrict__ g2) {
  __shared__ float red[EMD_T];
  const int b = blockIdx.y;
  const int l = blockIdx.x * EMD_ROWS + (threadIdx.x & 63);
  const float4* __restrict__ other = w.pk1 + (size_t)b * n;
  float4 me = l < m ? w.pk2a[(size_t)b * m + l] : make_float4(0, 0, 0, 0);
  v2f ar = {0.f, 0.f}, ax = ar, ay = ar, az = ar;
  auto eval = [&](v2f X, v2f Y, v2f Z, v2f Wt, int) {
    v2f dx = me.x - X, dy = me.y - Y, dz = me.z - Z;
    v2f e = exp2_pair((dx * dx + dy * dy + dz * dz) * c) * Wt;  // :108
    ar += e;                                                    // :109
    if (FUSED) {
      ax += e * dx;
      ay += e * dy;
      az += e * dz;
    }
  };
  auto ix = [](int i) { return i; };
  int nev = 0;
  EMD_WALK(other, ix, n, eval, -INFINITY, INFINITY, nev);
  float sumr = cross_wave_sum(ar.x + ar.y, red), sx = 0.f, sy = 0.f, sz = 0.f;
  if (FUSED) {
    sx = cross_wave_sum(ax.x + ax.y, red);
    sy = cross_wave_sum(ay.x + ay.y, red);
    sz = cross_wave_sum(az.x + az.y, red);
  }
  if (l < m && threadIdx.x < 64) {
    float remainR = me.w;
    sumr *= remainR;                                              // :114
    float consumption = fminf(remainR / (sumr + 1e-9f), 1.0f);    // :115
    float ratioR = consumption * remainR;                         // :116
    me.w = fmaxf(0.0f, remainR - sumr);                           // :117
    w.pk2a[(size_t)b * m + l] = me;
    w.pk2b[(size_t)b * m + l].w = ratioR;
    if (FUSED) {
      float* g = g2 + ((size_t)b * m + l) * 3;
      float s = 2.f * ratioR;
      g[0] += s * sx;
      g[1] += s * sy;
      g[2] += s * sz;
    }
  }
}

// Pass B of the fused entry point: the rows are the ACTIVE points of cloud 2 (list `buf`, see emd_compact_kernel).
// A workgroup that walks all of cloud 1 for its rows takes ~70 us however few rows are left, so the fewer rows a
// pair has, the more lanes share one: SUB = 1 ... 16 adjacent lanes per row (64 ... 4 rows per workgroup), each
// taking every SUB-th point pair of its wavefront's quarter tile; they meet by lane shuffles, the four wavefronts
// through LDS as before.  SUB = 1 / 2 / 4 / 8 / 16 for cnt <= m, 3m/4, m/2, m/4, m/8 (measured at n = m = 2048, 64
// pairs: with 64 rows per workgroup the pass took 80 us at 35 % of the rows as at 100 %).  The grid is (m+31)/32 workgroups per
// pair (cnt*SUB <= 2m always fits); workgroups past the end of the list leave at once.  Same formulas as
// emd_pass_b_kernel<true>.
template <int SUB>
__device__ __forceinline__ void emd_pass_b_rows(int n, int m, float c, const EmdWs& w, float* __restrict__ g2, int buf,
                                                int cnt, float4 (*tile)[EMD_TL], float* red, float win) {
  constexpr int RPB = EMD_ROWS / SUB;
  if ((int)blockIdx.x * RPB >= cnt) return;  // workgroup-uniform
  const int b = blockIdx.y, lane = threadIdx.x & 63, wq = threadIdx.x >> 6;
  const int sub = lane & (SUB - 1), r = blockIdx.x * RPB + lane / SUB;
  const int l = r < cnt ? w.act[buf][(size_t)b * m + r] : m;
  const float4* __restrict__ other = w.pk1 + (size_t)b * n;
  float4 me = l < m ? w.pk2a[(size_t)b * m + l] : make_float4(0, 0, 0, 0);
  float xlo, xhi;      // the workgroup's rows are consecutive list entries (ascending x): lanes 0 and (last row) * SUB
  row_window(me.x, min(RPB, cnt - (int)blockIdx.x * RPB) * SUB - (SUB - 1), win, xlo, xhi);
  int nev = 0;
  v2f ar = {0.f, 0.f}, ax = ar, ay = ar, az = ar;
  auto eval = [&](float4 a, float4 bb) {  // a = {x0,x1,y0,y1}, bb = {z0,z1,w0,w1}
    v2f dx = me.x - (v2f){a.x, a.y}, dy = me.y - (v2f){a.z, a.w}, dz = me.z - (v2f){bb.x, bb.y};
    v2f e = exp2_pair((dx * dx + dy * dy + dz * dz) * c) * (v2f){bb.z, bb.w};  // :108
    ar += e;                                                                    // :109
    ax += e * dx;
    ay += e * dy;
    az += e * dz;
  };
  auto stage = [&](int tb, int base) {
    const int q = threadIdx.x, i0 = base + 2 * q;
    if (i0 < n) {
      const float4 a = other[i0];
      const float4 bb = i0 + 1 < n ? other[i0 + 1] : make_float4(a.x, a.y, a.z, 0.f);
      tile[tb][2 * q] = make_float4(a.x, bb.x, a.y, bb.y);
      tile[tb][2 * q + 1] = make_float4(a.z, bb.z, a.w, bb.w);
    }
  };
  const int ntile = (n + EMD_TL - 1) / EMD_TL;
  stage(0, 0);
  __syncthreads();
  for (int t = 0; t < ntile; ++t) {
    const int base = t * EMD_TL;
    const int npair = (min(EMD_TL, n - base) + 1) >> 1;
    if (t + 1 < ntile) stage((t + 1) & 1, base + EMD_TL);
    const float4* tp = tile[t & 1];
    const int per = (npair + 3) >> 2;
    const int q0 = min(npair, wq * per), end = min(npair, q0 + per);
    if (q0 < end && !(tp[2 * (end - 1)].y < xlo || tp[2 * q0].x > xhi)) {  // else: every product underflows to +0
      nev += end - q0;
      int q = q0 + sub;
      for (; q + SUB < end; q += 2 * SUB) {
        const float4 a0 = tp[2 * q], b0 = tp[2 * q + 1], a1 = tp[2 * (q + SUB)], b1 = tp[2 * (q + SUB) + 1];
        eval(a0, b0);
        eval(a1, b1);
      }
      if (q < end) eval(tp[2 * q], tp[2 * q + 1]);
    }
    __syncthreads();
  }
  if ((threadIdx.x & 63) == 0 && nev > 0) {  // units of 64 (row, point) evaluations: RPB rows x 2 nev points
    const unsigned slot = ((blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) & (EMD_WALK_SLOTS - 1);
    atomicAdd(w.walk + slot, (unsigned long long)((2 * nev + SUB - 1) / SUB));
  }
  float sr = ar.x + ar.y, sx = ax.x + ax.y, sy = ay.x + ay.y, sz = az.x + az.y;
#pragma unroll
  for (int o = 1; o < SUB; o <<= 1) {
    sr += __shfl_xor(sr, o, PZN_WAVE);
    sx += __shfl_xor(sx, o, PZN_WAVE);
    sy += __shfl_xor(sy, o, PZN_WAVE);
    sz += __shfl_xor(sz, o, PZN_WAVE);
  }
  float sumr = cross_wave_sum(sr, red);
  sx = cross_wave_sum(sx, red);
  sy = cross_wave_sum(sy, red);
  sz = cross_wave_sum(sz, red);
  if (l < m && threadIdx.x < 64 && sub == 0) {
    const float remainR = me.w;
    sumr *= remainR;                                                    // :114
    const float consumption = fminf(remainR / (sumr + 1e-9f), 1.0f);    // :115
    const float ratioR = consumption * remainR;                         // :116
    me.w = fmaxf(0.0f, remainR - sumr);                                 // :117
    w.pk2a[(size_t)b * m + l] = me;
    w.pk2b[(size_t)b * m + l].w = ratioR;
    float* g = g2 + ((size_t)b * m + w.perm[1][(size_t)b * m + l]) * 3;      // (the list path runs on x-sorted clouds)
    const float s = 2.f * ratioR;
    g[0] += s * sx;
    g[1] += s * sy;
    g[2] += s * sz;
  }
}

__global__ __launch_bounds__(EMD_T) void emd_pass_b_list_kernel(int n, int m, float c, EmdWs w, float* __restrict__ g2,
                                                                int buf, float win) {
  __shared__ float4 tile[2][EMD_TL];
  __shared__ float red[EMD_T];
  const int cnt = w.cnt[buf][blockIdx.y];
  if (cnt * 8 <= m)
    emd_pass_b_rows<16>(n, m, c, w, g2, buf, cnt, tile, red, win);
  else if (cnt * 4 <= m)
    emd_pass_b_rows<8>(n, m, c, w, g2, buf, cnt, tile, red, win);
  else if (cnt * 2 <= m)
    emd_pass_b_rows<4>(n, m, c, w, g2, buf, cnt, tile, red, win);
  else if (cnt * 4 <= m * 3)
    emd_pass_b_rows<2>(n, m, c, w, g2, buf, cnt, tile, red, win);
  else  // (nearly) every row: the pass is bound by the vector ALU, sharing rows only adds workgroups
    emd_pass_b_rows<1>(n, m, c, w, g2, buf, cnt, tile, red, win);
}

// Pass C: rows = points k of xyz1.  MATCH writes match[b][l][k] += w (API-parity path);
// FUSED accumulates cost_b += sum_l d_kl w_kl and grad1_k += 2 sum_l w_kl (x1_k - x2_l).
// LIST (not with MATCH): walk only the active points of cloud 2 (list `buf`).
template <bool MATCH, bool FUSED, bool LIST>
__global__ __launch_bounds__(EMD_T) void emd_pass_c_kernel(int n, int m, float c, EmdWs w, float* __restrict__ match,
                                                           float* __restrict__ cost, float* __restrict__ g1, int buf,
                                                           float win) {
  __shared__ float red[EMD_T];
  const int b = blockIdx.y;
  const int k = blockIdx.x * EMD_ROWS + (threadIdx.x & 63);
  const float4* __restrict__ other = w.pk2b + (size_t)b * m;
  float4 me = k < n ? w.pk1[(size_t)b * n + k] : make_float4(0, 0, 0, 0);
  const float rl = me.w;  // :139
  v2f al = {0.f, 0.f}, ax = al, ay = al, az = al, ac = al;
  float* mt = MATCH ? match + (size_t)b * n * m + k : nullptr;
  auto eval = [&](v2f X, v2f Y, v2f Z, v2f Wt, int l) {
    v2f dx = me.x - X, dy = me.y - Y, dz = me.z - Z;
    v2f d = dx * dx + dy * dy + dz * dz;
    v2f wv = exp2_pair(d * c) * rl * Wt;  // :145
    if (MATCH) {
      if (k < n) {
        mt[(size_t)l * n] += wv.x;  // :146
        if (l + 1 < m) mt[(size_t)(l + 1) * n] += wv.y;
      }
    }
    al += wv;  // :147
    if (FUSED) {
      ax += wv * dx;
      ay += wv * dy;
      az += wv * dz;
      ac += wv * d;
    }
  };
  const int* __restrict__ act = w.act[buf] + (size_t)b * m;
  auto ix = [&](int i) { return LIST ? act[i] : i; };
  float xlo = -INFINITY, xhi = INFINITY;
  if (LIST) row_window(me.x, min(EMD_ROWS, n - (int)blockIdx.x * EMD_ROWS), win, xlo, xhi);   // (the list path is x-sorted)
  int nev = 0;
  EMD_WALK(other, ix, LIST ? w.cnt[buf][b] : m, eval, xlo, xhi, nev);
  if (LIST) count_walk(w, nev);
  float suml = cross_wave_sum(al.x + al.y, red), sx = 0.f, sy = 0.f, sz = 0.f, sc = 0.f;
  if (FUSED) {
    sx = cross_wave_sum(ax.x + ax.y, red);
    sy = cross_wave_sum(ay.x + ay.y, red);
    sz = cross_wave_sum(az.x + az.y, red);
    sc = cross_wave_sum(ac.x + ac.y, red);
  }
  if (k < n && threadIdx.x < 64) {
    float* r = w.remainL + (size_t)b * n + k;
    *r = fmaxf(0.0f, *r - suml);  // :153
    if (FUSED) {
      float* g = g1 + ((size_t)b * n + (LIST ? w.perm[0][(size_t)b * n + k] : k)) * 3;
      g[0] += 2.f * sx;
      g[1] += 2.f * sy;
      g[2] += 2.f * sz;
    }
  }
  if (FUSED && threadIdx.x < 64) {  // one wavefront holds the 64 row totals
    sc = k < n ? sc : 0.f;
    sc = pzn::wave_sum_f32(sc);
    if (threadIdx.x == 0) atomicAdd(cost + b, sc);
  }
}

// Pass C of one level fused with pass A of the NEXT level (fused entry point only): both walk cloud 2 for the rows k
// of cloud 1, so the differences and the squared distance of a pair are computed once for the two exponentials
// (6 of the 20 packed instructions of the pair of passes), one launch and one tile staging are saved per level.
// Pass A needs nothing of pass C but the row's own remainL_k, which this thread has just updated.  Arithmetic per
// pass is unchanged, so the results are those of the separate kernels.
// The walk covers the active list of THIS level (`buf`): ratioR of pass C is non-zero exactly there, and the points that
// pass B has just exhausted carry remainR = 0 into the next level's sum.
__global__ __launch_bounds__(EMD_T) void emd_pass_ca_kernel(int n, int m, float c, float c_next, EmdWs w,
                                                            float* __restrict__ cost, float* __restrict__ g1, int buf,
                                                            float win) {
  __shared__ float red[EMD_T];
  const int b = blockIdx.y;
  const int k = blockIdx.x * EMD_ROWS + (threadIdx.x & 63);
  const float4* __restrict__ oa = w.pk2a + (size_t)b * m;  // {x2, y2, z2, remainR}   (pass A of the next level)
  const float4* __restrict__ ob = w.pk2b + (size_t)b * m;  // {.., ratioR}            (pass C of this level)
  float4 me = k < n ? w.pk1[(size_t)b * n + k] : make_float4(0, 0, 0, 0);
  const float rl = me.w;  // :139
  v2f al = {0.f, 0.f}, ax = al, ay = al, az = al, ac = al, aa = al;
  auto eval = [&](v2f X, v2f Y, v2f Z, v2f WA, v2f WB) {
    v2f dx = me.x - X, dy = me.y - Y, dz = me.z - Z;
    v2f d = dx * dx + dy * dy + dz * dz;
    v2f wv = exp2_pair(d * c) * rl * WB;  // :145
    al += wv;                             // :147
    ax += wv * dx;
    ay += wv * dy;
    az += wv * dz;
    ac += wv * d;
    aa += exp2_pair(d * c_next) * WA;     // :77-78 of the next level (same d: (x-y)^2 == (y-x)^2 exactly)
  };
  const int* __restrict__ act = w.act[buf] + (size_t)b * m;
  auto ix = [&](int i) { return act[i]; };
  float xlo, xhi;      // window of the SOFTER of the two levels (c_next): outside it both exponentials are +0
  row_window(me.x, min(EMD_ROWS, n - (int)blockIdx.x * EMD_ROWS), win, xlo, xhi);
  int nev = 0;
  EMD_WALK2(oa, ob, ix, w.cnt[buf][b], eval, xlo, xhi, nev);
  count_walk(w, nev);
  const float suml = cross_wave_sum(al.x + al.y, red);
  const float sx = cross_wave_sum(ax.x + ax.y, red), sy = cross_wave_sum(ay.x + ay.y, red);
  const float sz = cross_wave_sum(az.x + az.y, red);
  float sc = cross_wave_sum(ac.x + ac.y, red);
  const float sa = 1e-9f + cross_wave_sum(aa.x + aa.y, red);  // :59
  if (k < n && threadIdx.x < 64) {
    float* r = w.remainL + (size_t)b * n + k;
    const float rem = fmaxf(0.0f, *r - suml);  // :153
    *r = rem;
    float* g = g1 + ((size_t)b * n + w.perm[0][(size_t)b * n + k]) * 3;
    g[0] += 2.f * sx;
    g[1] += 2.f * sy;
    g[2] += 2.f * sz;
    me.w = rem / sa;  // :83 of the next level
    w.pk1[(size_t)b * n + k] = me;
  }
  if (threadIdx.x < 64) {  // one wavefront holds the 64 row totals
    sc = k < n ? sc : 0.f;
    sc = pzn::wave_sum_f32(sc);
    if (threadIdx.x == 0) atomicAdd(cost + b, sc);
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

__global__ __launch_bounds__(EMD_SMALL_T) void emd_small_fused_kernel(const float* __restrict__ xyz1,
                                                              const float* __restrict__ xyz2, int n, int m,
                                                              float multiL, float multiR, int tpr_shift,
                                                              float* __restrict__ cost, float* __restrict__ g1,
                                                              float* __restrict__ g2) {
  __shared__ float4 p1[EMD_SMALL_MAX];   // {x1, y1, z1, ratioL}
  __shared__ float4 p2[EMD_SMALL_MAX];   // {x2, y2, z2, remainR}
  __shared__ float rr[EMD_SMALL_MAX];    // ratioR
  __shared__ float rl[EMD_SMALL_MAX];    // remainL
  __shared__ float csum[EMD_SMALL_T / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
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

template <bool MATCH, bool FUSED>
int run_levels(const float* xyz1, const float* xyz2, int B, int n, int m, float* match, float* cost, float* g1,
               float* g2, void* workspace, hipStream_t st) {
  EmdWs w = carve(workspace, B, n, m);
  float multiL, multiR;  // :29-35 (integer division)
  if (n >= m) {
    multiL = 1.f;
    multiR = (float)(n / m);
  } else {
    multiL = (float)(m / n);
    multiR = 1.f;
  }
  const int mx = n > m ? n : m;
  dim3 gi((mx + EMD_T - 1) / EMD_T, B), gk((n + EMD_ROWS - 1) / EMD_ROWS, B), gl((m + EMD_ROWS - 1) / EMD_ROWS, B);
  constexpr bool LISTED = FUSED && !MATCH;      // the fused entry point: x-sorted clouds, active lists, x windows
  if (LISTED) {
    int npow = 1, mpow = 1;
    while (npow < n) npow <<= 1;
    while (mpow < m) mpow <<= 1;
    const size_t lds = sizeof(unsigned long long) * (size_t)(npow > mpow ? npow : mpow);
    if (lds > 150 * 1024) return PZN_EUNSUPPORTED;      // (> 16384 points per cloud)
    if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(emd_sort_x_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return PZN_ELAUNCH;
    hipLaunchKernelGGL(emd_sort_x_kernel, dim3(2, B), dim3(EMD_ST), lds, st, xyz1, xyz2, n, m, npow, mpow, w);
  }
  hipLaunchKernelGGL(emd_init_kernel, gi, dim3(EMD_T), 0, st, xyz1, xyz2, n, m, multiL, multiR, w, FUSED ? cost : nullptr,
                     g1, g2, LISTED ? 1 : 0);
  if (MATCH && pzn_zero_async(match, (size_t)B * n * m, st) != PZN_OK) return PZN_ELAUNCH;  // :39-40
  auto cof = [](int j) {                                         // :47-50, * log2(e)
    const float level = j == -2 ? 0.f : -powf(4.0f, (float)j);
    return level * 1.44269504088896340736f;
  };
  // x window of a level: exp2(c d^2) with c d^2 <= -150 is exactly +0 in fp32 (below the smallest denormal), and
  // d^2 >= (x distance)^2: points farther than sqrt(150 / -c) along x are not walked (level 0: no window)
  static const float win_bits = [] { const char* e = getenv("PZN_EMD_WIN_BITS"); return e ? (float)atof(e) : 150.f; }();  // tuning aid
  auto winf = [&](int j) {
    const float c = cof(j);
    return c < 0.f ? sqrtf(win_bits / -c) : INFINITY;
  };
  if (LISTED) {  // A(7); then per level B, and C fused with the next level's A; the last level ends with a plain C
    hipLaunchKernelGGL(emd_pass_a_kernel, gk, dim3(EMD_T), 0, st, n, m, cof(7), w, winf(7));
    for (int j = 7, buf = 0; j >= -2; --j, buf ^= 1) {  // list `buf` = points of cloud 2 with mass at the start of level j
      hipLaunchKernelGGL(emd_pass_b_list_kernel, dim3((m + 31) / 32, B), dim3(EMD_T), 0, st, n, m, cof(j), w, g2, buf,
                         winf(j));
      if (j > -2) {
        hipLaunchKernelGGL(emd_compact_kernel, dim3(B), dim3(EMD_CT), 0, st, m, w, buf ^ 1);  // for level j - 1
        hipLaunchKernelGGL(emd_pass_ca_kernel, gk, dim3(EMD_T), 0, st, n, m, cof(j), cof(j - 1), w, cost, g1, buf,
                           winf(j - 1));
      } else {
        hipLaunchKernelGGL((emd_pass_c_kernel<false, FUSED, true>), gk, dim3(EMD_T), 0, st, n, m, cof(j), w, match, cost,
                           g1, buf, winf(j));
      }
    }
    PZN_RETURN_LAUNCH_STATUS();
  }
  for (int j = 7; j >= -2; --j) {                                // :46
    const float c = cof(j);
    hipLaunchKernelGGL(emd_pass_a_kernel, gk, dim3(EMD_T), 0, st, n, m, c, w, INFINITY);
    hipLaunchKernelGGL((emd_pass_b_kernel<FUSED>), gl, dim3(EMD_T), 0, st, n, m, c, w, g2);
    hipLaunchKernelGGL((emd_pass_c_kernel<MATCH, FUSED, false>), gk, dim3(EMD_T), 0, st, n, m, c, w, match, cost, g1, 0,
                       INFINITY);
  }
  PZN_RETURN_LAUNCH_STATUS();
}

}  // namespace

PZN_EXPORT size_t pzn_emd_workspace_bytes(int B, int n, int m) {
  if (B <= 0 || n <= 0 || m <= 0) return 0;
  return align_up(sizeof(float4) * (size_t)B * n, 256) + 2 * align_up(sizeof(float4) * (size_t)B * m, 256) +
         align_up(sizeof(float) * (size_t)B * n, 256) + 2 * align_up(sizeof(int) * (size_t)B * m, 256) +
         2 * align_up(sizeof(int) * (size_t)B, 256) + align_up(sizeof(int) * (size_t)B * n, 256) +
         align_up(sizeof(int) * (size_t)B * m, 256) + sizeof(unsigned long long) * EMD_WALK_SLOTS;
}

// Byte offset, inside the workspace, of 1024 uint64 counters that the fused entry point leaves behind: their sum x 64 is
// the number of (row, point) pair evaluations its passes executed (points of exhausted mass and points outside the
// level's x window are not walked); (size_t)-1 when the call takes the single-workgroup path (n, m <= 256), which
// evaluates all 30 n m.
PZN_EXPORT size_t pzn_emd_walk_counter_offset(int B, int n, int m) {
  if (B <= 0 || n <= 0 || m <= 0 || (n <= EMD_SMALL_MAX && m <= EMD_SMALL_MAX)) return (size_t)-1;
  return pzn_emd_workspace_bytes(B, n, m) - sizeof(unsigned long long) * EMD_WALK_SLOTS;
}

PZN_EXPORT int pzn_emd_approxmatch_f32(const float* xyz1, const float* xyz2, int B, int n, int m, float* match,
                                       void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz1 && xyz2 && match && workspace && B > 0 && n > 0 && m > 0 && B <= 65535);
  PZN_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
  return run_levels<true, false>(xyz1, xyz2, B, n, m, match, nullptr, nullptr, nullptr, workspace,
                                 pzn_hip_stream(stream));
}

PZN_EXPORT int pzn_emd_fused_f32(const float* xyz1, const float* xyz2, int B, int n, int m, float* cost, float* g1,
                                 float* g2, void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz1 && xyz2 && cost && g1 && g2 && workspace && B > 0 && n > 0 && m > 0 && B <= 65535);
  PZN_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
  if (n <= EMD_SMALL_MAX && m <= EMD_SMALL_MAX) {  // one workgroup per pair, one launch for the whole auction
    const float multiL = n >= m ? 1.f : (float)(m / n), multiR = n >= m ? (float)(n / m) : 1.f;  // :29-35
    int rows = 1;
    while (rows < (n > m ? n : m)) rows <<= 1;
    int shift = 0;
    while ((rows << shift) < EMD_SMALL_T && shift < 6) ++shift;  // lanes sharing a row stay inside one wavefront
    hipLaunchKernelGGL(emd_small_fused_kernel, dim3((unsigned)B), dim3(EMD_SMALL_T), 0, pzn_hip_stream(stream), xyz1, xyz2,
                       n, m, multiL, multiR, shift, cost, g1, g2);
    PZN_RETURN_LAUNCH_STATUS();
  }
  return run_levels<false, true>(xyz1, xyz2, B, n, m, nullptr, cost, g1, g2, workspace, pzn_hip_stream(stream));
}

PZN_EXPORT int pzn_emd_matchcost_f32(const float* xyz1, const float* xyz2, const float* match, int B, int n, int m,
                                     float* cost, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz1 && xyz2 && match && cost && B > 0 && n > 0 && m > 0 && B <= 65535);
  hipStream_t st = pzn_hip_stream(stream);
  if (pzn_zero_async(cost, (size_t)B, st) != PZN_OK) return PZN_ELAUNCH;
  dim3 grid((n + EMD_T - 1) / EMD_T, (m + MC_LSLAB - 1) / MC_LSLAB, B);
  PZN_CHECK_ARG(grid.y <= 65535);
  hipLaunchKernelGGL(emd_matchcost_kernel, grid, dim3(EMD_T), 0, st, xyz1, xyz2, match, n, m, cost);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_emd_matchcost_grad_f32(const float* grad_cost, const float* xyz1, const float* xyz2,
                                          const float* match, int B, int n, int m, float* grad1, float* grad2,
                                          pzn_stream_t stream) {
  PZN_CHECK_ARG(grad_cost && xyz1 && xyz2 && match && grad1 && grad2 && B > 0 && n > 0 && m > 0 && B <= 65535);
  hipStream_t st = pzn_hip_stream(stream);
  hipLaunchKernelGGL(emd_grad1_kernel, dim3((n + EMD_T - 1) / EMD_T, B), dim3(EMD_T), 0, st, grad_cost, xyz1, xyz2,
                     match, n, m, grad1);
  constexpr int LPB = EMD_T / PZN_WAVE;
  hipLaunchKernelGGL(emd_grad2_kernel, dim3((m + LPB - 1) / LPB, B), dim3(EMD_T), 0, st, grad_cost, xyz1, xyz2, match,
                     n, m, grad2);
  PZN_RETURN_LAUNCH_STATUS();
}
