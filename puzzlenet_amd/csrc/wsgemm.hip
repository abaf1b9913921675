// wsgemm.hip — weight-stationary bf16x3 GEMM for the skinny layers (K, N <= a few hundred, M huge).
//
//   C[M, N] = epilogue( A[M, K] * W^T )       A row-major (k contiguous), W = nn.Linear weight [N, K] ("NT",
//   forward) or [K, N] ("NN", input gradient dX = dY W with K = Nout).
//
// Most of the step's dense time is in products whose weight matrix is tiny (<= 256 x 256) while the
// activation matrix has 10^5..10^6 rows (shared MLPs of the set-abstraction levels, attention
// projections, point-wise MLPs).  The general tile engine (gemm.hip) stages BOTH operands through LDS every
// 16 k and synchronises its four waves at every step; on these shapes its waves sit in s_waitcnt /
// s_barrier 50-70 % of the time (SQ_WAIT_ANY, profiles/).  Here instead
//   * the weight slice (NT*32 columns x K) is split ONCE per workgroup into its three bf16 planes, stored
//     in LDS in MFMA-fragment order, and stays there: workgroups are persistent and walk row tiles;
//   * the activation rows never touch LDS: for v_mfma_f32_32x32x16_bf16 lane l holds A[row l&31][8
//     consecutive k], which is exactly what a lane can load from global memory itself (16 consecutive
//     floats per two MFMA steps; the reduction index is permuted identically on the weight side so that
//     a lane's 64 bytes are contiguous) and split in registers;
//   * so a wavefront owns a 32-row tile end to end: no barrier after the prologue, every wave runs its
//     own software pipeline (next loads in flight while the current 16 floats are split and multiplied),
//     and the 32 rows of a tile are one (centroid, K = 32 neighbours) group for the max-pool epilogue.
// Split precision as in gemm.hip: x = x1 + x2 + x3 (bf16 each, exact), six products per tile and 16 k.
#include <stdlib.h>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int WS_NW = 12;        // wavefronts per workgroup, each on its own row tiles
constexpr int WS_STAGE_LD = 36;  // floats per row of a wave's 32 x 32 output patch in LDS (16-B aligned, conflict-free)
constexpr int WS_LDS_MAX = 150 * 1024;

struct WsArgs {
  const float* A;
  int lda;
  const float* W;
  int ldw;
  int w_kmajor;  // 0: W[n*ldw + k]   1: W[k*ldw + n]
  float* C;
  int ldc;
  int M, N, K;
  int nd;               // double steps of 32 k: ceil(K / 32)
  const float* bias;    // per column or NULL
  int relu;
  const float* genY;    // A(m,k) *= genY(m,k) > 0   (same layout as A) or NULL
  const float* maskH;   // C(m,n) *= maskH(m,n) > 0  (same layout as C) or NULL
  int32_t* argmax;      // max-pool epilogue: C is [M/32, N]
  const int64_t* scat;  // scatter-add epilogue: row m goes to C row (m / scat_in) * scat_out + scat[m], atomically
  int scat_in, scat_out;
  const float* residual;  // store epilogue: + residual(m,n) (same layout as C) ...
  float* C2;              // ... into C2 when given (C keeps the value without it), else into C itself
  int accumulate;         // store epilogue: C(m,n) += value instead of = value
  // up to three (W, bias, C) triples that share A (the q, k, v projections of an attention block): blockIdx.y runs
  // over the column slices of all of them, one launch instead of three
  // GATHER (max-pool variant only): the activation stream is never in memory.  Row (g, k) of group g is
  //   relu(A[gbase(g) + gidx[g*32 + k], :] + gQ[g, :])   with A = the per-point table P' and gbase(g) = (g / gS) * gN
  // (the first set-abstraction layer per point, see pzn_sa_level_fwd_f32): the loads of a row group carry per-lane row
  // offsets looked up one tile ahead, the group's Q slice rides along as a fifth load, add + ReLU happen in registers
  // on the way into the wave's LDS patch.
  const int64_t* gidx;
  const float* gQ;
  int gN, gS;
  int nseg;
  const float* sW[3];
  const float* sbias[3];
  float* sC[3];
  int sN[3];
  // store epilogue: C(m, n) += x_m wx[n] + y_m wy[n] + z_m wz[n], the three coordinate columns of a layer whose input is
  // [xyz | features] (first set-abstraction layer per point, csrc/sachain.hip): r3_xyz[M, 3], r3_w = {wx[N], wy[N], wz[N]};
  // the arithmetic of sa_prep_kernel, which made a pass of its own over the table for it: fmaf(wz, z, fmaf(wy, y, wx x))
  const float* r3_xyz;
  const float* r3_w;
};

// two floats -> their three bf16 planes, packed (lo = first element)
__device__ __forceinline__ void split_pair(v2f x, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  bf16x2 a = __builtin_convertvector(x, bf16x2);
  p1 = __builtin_bit_cast(uint32_t, a);
  v2f fa = v2f{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  v2f r = x - fa;
  bf16x2 b = __builtin_convertvector(r, bf16x2);
  p2 = __builtin_bit_cast(uint32_t, b);
  v2f fb = v2f{__uint_as_float(p2 << 16), __uint_as_float(p2 & 0xffff0000u)};
  v2f r2 = r - fb;
  bf16x2 c = __builtin_convertvector(r2, bf16x2);
  p3 = __builtin_bit_cast(uint32_t, c);
}

__device__ __forceinline__ void split8(float4 lo, float4 hi, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
  uint32_t a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
  split_pair(v2f{lo.x, lo.y}, a0, b0, c0);
  split_pair(v2f{lo.z, lo.w}, a1, b1, c1);
  split_pair(v2f{hi.x, hi.y}, a2, b2, c2);
  split_pair(v2f{hi.z, hi.w}, a3, b3, c3);
  const u32x4 a = {a0, a1, a2, a3}, b = {b0, b1, b2, b3}, c = {c0, c1, c2, c3};
  p1 = __builtin_bit_cast(bf16x8, a);
  p2 = __builtin_bit_cast(bf16x8, b);
  p3 = __builtin_bit_cast(bf16x8, c);
}

__device__ __forceinline__ float4 add_relu(float4 x, float4 q) {  // relu(x + q): the first layer's row from P' and Q
  return make_float4(fmaxf(x.x + q.x, 0.f), fmaxf(x.y + q.y, 0.f), fmaxf(x.z + q.z, 0.f), fmaxf(x.w + q.w, 0.f));
}

__device__ __forceinline__ float4 relu_mask(float4 x, float4 y) {
  return make_float4(y.x > 0.f ? x.x : 0.f, y.y > 0.f ? x.y : 0.f, y.z > 0.f ? x.z : 0.f, y.w > 0.f ? x.w : 0.f);
}

// NT = 32-column tiles per wavefront (weight slice = NT*32 columns), MAXPOOL = max over the tile's 32 rows,
// GENY = ReLU-mask the activation stream with genY.
// FULL: M % 32 == 0 and N % (NT*32) == 0: every store of the walk is unconditional too.
// D2: two register sets = two 32 x 32 blocks of the stream in flight per wave (needs an even double-step count).
template <int NT, bool MAXPOOL, bool GENY, int NW, bool FULL, bool D2, bool GATH = false>
__global__ __launch_bounds__(NW * 64) void ws_gemm_kernel(WsArgs p) {
  constexpr int WS_T = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];  // [nd*2][NT][3][64 lanes][16 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: tile offsets must be provably uniform
  const int l31 = lane & 31, half = lane >> 5;
  int slice = blockIdx.y;
  if (p.nseg) {  // workgroup-uniform choice of the segment this column slice belongs to
    const int c0 = (p.sN[0] + NT * 32 - 1) / (NT * 32), c1 = c0 + (p.sN[1] + NT * 32 - 1) / (NT * 32);
    const int sg = slice < c0 ? 0 : (slice < c1 ? 1 : 2);
    slice -= sg == 0 ? 0 : (sg == 1 ? c0 : c1);
    p.W = sg == 0 ? p.sW[0] : (sg == 1 ? p.sW[1] : p.sW[2]);
    p.bias = sg == 0 ? p.sbias[0] : (sg == 1 ? p.sbias[1] : p.sbias[2]);
    p.C = sg == 0 ? p.sC[0] : (sg == 1 ? p.sC[1] : p.sC[2]);
    p.N = sg == 0 ? p.sN[0] : (sg == 1 ? p.sN[1] : p.sN[2]);
    p.ldc = p.N;
  }
  const int n0 = slice * (NT * 32);

  // ---- prologue: this slice of W -> three bf16 planes in fragment order (k permuted: lane half h of
  //      double step d holds k = 32d + 16h + 8s + 0..7 in MFMA step s)
  const int nfrag = p.nd * 2 * NT * 64;
  for (int f = tid; f < nfrag; f += WS_T) {
    const int l = f & 63, j = (f >> 6) % NT, ks = (f >> 6) / NT;
    const int n = n0 + j * 32 + (l & 31);
    const int k = (ks >> 1) * 32 + (l >> 5) * 16 + (ks & 1) * 8;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = n < p.N && k + i < p.K;
      const size_t off = p.w_kmajor ? (size_t)(k + i) * p.ldw + n : (size_t)n * p.ldw + k + i;
      v[i] = ok ? p.W[off] : 0.f;
    }
    bf16x8 w1, w2, w3;
    split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), w1, w2, w3);
    unsigned char* dst = wlds + ((size_t)(ks * NT + j) * 3) * 1024 + l * 16;
    *reinterpret_cast<bf16x8*>(dst) = w1;
    *reinterpret_cast<bf16x8*>(dst + 1024) = w2;
    *reinterpret_cast<bf16x8*>(dst + 2048) = w3;
  }
  __syncthreads();  // the only barrier

  const int ntiles = (p.M + 31) >> 5;
  const int tstride = gridDim.x * NW;
  int t = blockIdx.x * NW + wave;
  if (t >= ntiles) return;

  // Activation stream.  A double step needs the wave's 32 rows x 32 k (128 bytes per row).  Fetching them in the
  // MFMA operand layout (lane = row) makes every load instruction touch 32 different 128-byte lines (measured:
  // 2.4 TB/s with everything else removed), so the tile is fetched COALESCED — lane l takes 16 bytes at k-offset
  // (l&7)*4 of rows (l>>3) + 8i: eight lanes per full line — and goes through the wave's private LDS patch
  // (4 ds_write_b128, 4 ds_read_b128, wave-local: no barrier) to reach the operand layout (row l&31, 16 k at
  // half*16).  Loads are buffer loads: descriptor + wave-uniform tile / row-group offset (scalar) + a lane offset
  // that never changes, so there is no address arithmetic on the vector ALU, rows past M read as zero through the
  // descriptor's size, and every load is unconditional (the compiler's vmcnt bookkeeping stays exact).
  // GATH: A is the per-point table (its row count is not M): the descriptor spans what the caller says it holds
  const size_t a_rows = GATH ? (size_t)p.gN * ((size_t)(p.M / 32 + p.gS - 1) / p.gS) : (size_t)p.M;
  const __amdgpu_buffer_rsrc_t rsA =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)(a_rows * p.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(GENY ? p.genY : (GATH ? p.gQ : p.A)), 0,
      GATH ? (int)((size_t)(p.M / 32) * p.lda * 4) : (int)((size_t)p.M * p.lda * 4), 0x00020000);
  const int prow = lane >> 3, pcol = (lane & 7) * 4;
  const int voff = (prow * p.lda + pcol) * 4;
  const int rstep = 8 * p.lda * 4;  // bytes between the row groups of consecutive load instructions
  float4 g0, g1, g2, g3, y0, y1, y2, y3;      // set A
  float4 h0, h1, h2, h3, z0, z1, z2, z3;      // set B (D2)
  int it = t, id = 0;  // issue cursor (tile, double step)
  // GATH: byte offsets of this lane's four rows (prow + 8i) of the tile the issue cursor is in, and the point indices
  // of the NEXT tile of this wave (fetched a whole tile ahead, so that the offsets are there when the cursor wraps)
  int vo0 = 0, vo1 = 0, vo2 = 0, vo3 = 0, jn0 = 0, jn1 = 0, jn2 = 0, jn3 = 0;
  auto gath_fetch_idx = [&](int tile) {
    const int tc = tile < ntiles ? tile : ntiles - 1;
    const int64_t* ip = p.gidx + (size_t)tc * 32 + prow;
    jn0 = (int)ip[0], jn1 = (int)ip[8], jn2 = (int)ip[16], jn3 = (int)ip[24];
  };
  auto gath_offsets = [&](int tile) {
    const int tc = tile < ntiles ? tile : ntiles - 1;
    const int base = (tc / p.gS) * p.gN;
    vo0 = ((base + jn0) * p.lda + pcol) * 4, vo1 = ((base + jn1) * p.lda + pcol) * 4;
    vo2 = ((base + jn2) * p.lda + pcol) * 4, vo3 = ((base + jn3) * p.lda + pcol) * 4;
  };
  if (GATH) {
    gath_fetch_idx(t);
    gath_offsets(t);
    gath_fetch_idx(t + tstride);
  }
#define WS_BLOAD(rs, so) __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so, 0))
#define WS_GLOAD(vo, so) __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsA, vo, so, 0))
#define WS_ISSUE(a0, a1, a2, a3, b0, b1, b2, b3)                                                    \
  do {                                                                                               \
    if (GATH) {                                                                                      \
      const int sd_ = id * 32 * 4;                                                                   \
      a0 = WS_GLOAD(vo0, sd_), a1 = WS_GLOAD(vo1, sd_), a2 = WS_GLOAD(vo2, sd_), a3 = WS_GLOAD(vo3, sd_); \
      const int tq_ = it < ntiles ? it : ntiles - 1;                                                 \
      b0 = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsY, pcol * 4, (tq_ * p.lda + id * 32) * 4, 0)); \
      if (++id == p.nd) {                                                                            \
        id = 0, it += tstride;                                                                       \
        gath_offsets(it);            /* from the indices fetched a tile ago */                       \
        gath_fetch_idx(it + tstride);                                                                \
      }                                                                                              \
    } else {                                                                                         \
      const int so_ = (it * 32 * p.lda + id * 32) * 4;                                               \
      a0 = WS_BLOAD(rsA, so_), a1 = WS_BLOAD(rsA, so_ + rstep);                                      \
      a2 = WS_BLOAD(rsA, so_ + 2 * rstep), a3 = WS_BLOAD(rsA, so_ + 3 * rstep);                      \
      if (GENY) {                                                                                    \
        b0 = WS_BLOAD(rsY, so_), b1 = WS_BLOAD(rsY, so_ + rstep);                                    \
        b2 = WS_BLOAD(rsY, so_ + 2 * rstep), b3 = WS_BLOAD(rsY, so_ + 3 * rstep);                    \
      }                                                                                              \
      if (++id == p.nd) id = 0, it += tstride;                                                       \
    }                                                                                                \
  } while (0)
  WS_ISSUE(g0, g1, g2, g3, y0, y1, y2, y3);
  if (D2) WS_ISSUE(h0, h1, h2, h3, z0, z1, z2, z3);

  floatx16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float* stage = reinterpret_cast<float*>(wlds + (size_t)p.nd * 2 * NT * 3 * 1024) + wave * (32 * WS_STAGE_LD);
  const bool ktail = (p.K & 31) != 0;  // D2 walks are launched for K % 32 == 0 only
  float* pw = stage + prow * WS_STAGE_LD + pcol;             // patch write position (+ 8*i rows)
  const float* pr = stage + l31 * WS_STAGE_LD + half * 16;   // patch read position (+ 4*q floats)

  int d = 0;
  // one double step: 2 x NT x 6 MFMAs on the 16 floats of this lane's row
  auto step = [&](float4 c0, float4 c1, float4 c2, float4 c3) {
    const unsigned char* wb = wlds + (size_t)(d * 2) * (NT * 3 * 1024) + lane * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 a1, a2, a3;
      split8(s ? c2 : c0, s ? c3 : c1, a1, a2, a3);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const unsigned char* wj = wb + (s * NT + j) * (3 * 1024);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wj);
        const bf16x8 b2 = *reinterpret_cast<const bf16x8*>(wj + 1024);
        const bf16x8 b3 = *reinterpret_cast<const bf16x8*>(wj + 2048);
        floatx16 c = acc[j];
        if (MAXPOOL) {  // C[row][col]: the 32 rows of the group sit in one lane's registers (+ the other half)
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, c, 0, 0, 0);  // small terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
        } else {  // transposed product C^T[col][row]: a lane ends up with 4 consecutive columns of one row
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a3, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b3, a1, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a2, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a1, c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, c, 0, 0, 0);
        }
        acc[j] = c;
      }
    }
  };

  auto epilogue = [&]() {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (MAXPOOL) {  // element r of lane l = row (r&3) + 8*(r>>2) + 4*half, column l31
        const int col = n0 + j * 32 + l31;
        const bool col_ok = col < p.N;
        const float bv = (p.bias && col_ok) ? p.bias[col] : 0.f;
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = acc[j][r] + bv;
          v = v > 0.f ? v : 0.f;  // ReLU before the max (model5_b.py:453-454)
          const bool gt = v > best;
          best = gt ? v : best;
          bi = gt ? rl : bi;
        }
        const float ob = __shfl_xor(best, 32, PZN_WAVE);
        const int oi = __shfl_xor(bi, 32, PZN_WAVE);
        const bool take = ob > best || (ob == best && oi < bi);
        best = take ? ob : best;
        bi = take ? oi : bi;
        if (FULL) {  // one unconditional store per lane: lower half the value, upper half the index
          float* dst = half ? reinterpret_cast<float*>(p.argmax) : p.C;
          dst[(size_t)t * p.ldc + col] = half ? __int_as_float(bi) : best;
        } else if (half == 0 && col_ok) {
          p.C[(size_t)t * p.ldc + col] = best;
          p.argmax[(size_t)t * p.ldc + col] = bi;
        }
      } else {  // element 4q+e of lane l = column 8q + 4*half + e, row l31: through the wave's LDS patch -> 16-B row stores
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cl = 8 * q + 4 * half, col = n0 + j * 32 + cl;
          float4 v = make_float4(acc[j][4 * q], acc[j][4 * q + 1], acc[j][4 * q + 2], acc[j][4 * q + 3]);
          if (p.bias && col + 3 < p.N) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
            v.x += b.x, v.y += b.y, v.z += b.z, v.w += b.w;
          }
          if (p.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
          *reinterpret_cast<float4*>(stage + l31 * WS_STAGE_LD + cl) = v;
        }
        __builtin_amdgcn_wave_barrier();
        if (p.scat) {  // grad_feat[b, idx[b,s,k], :] += row (index_points backward): 128-B contiguous atomic runs
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int rl = 2 * i + half, row = t * 32 + rl, col = n0 + j * 32 + l31;
            const float v = stage[rl * WS_STAGE_LD + l31];
            if (FULL || (row < p.M && col < p.N)) {
              const long dst = (long)(row / p.scat_in) * p.scat_out + p.scat[row];
              atomicAdd(p.C + (size_t)dst * p.ldc + col, v);
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rl = (lane >> 3) + 8 * i, c4 = (lane & 7) * 4;
            float4 v = *reinterpret_cast<const float4*>(stage + rl * WS_STAGE_LD + c4);
            const int row = t * 32 + rl, col = n0 + j * 32 + c4;
            if (FULL || (row < p.M && col + 3 < p.N)) {
              const size_t o = (size_t)row * p.ldc + col;
              if (p.maskH) v = relu_mask(v, *reinterpret_cast<const float4*>(p.maskH + o));
              if (p.residual) {
                const float4 r = *reinterpret_cast<const float4*>(p.residual + o);
                const float4 w = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
                if (p.C2)
                  *reinterpret_cast<float4*>(p.C2 + o) = w;  // two outputs: C without, C2 with the residual
                else
                  v = w;
              }
              if (p.r3_xyz) {
                const float* q = p.r3_xyz + (size_t)row * 3;
                float x = q[0], y = q[1], z = q[2];
                // (registers of their own: loaded as a pair, y or z is broadcast into a packed multiply with op_sel on src1 -
                // the form that is wrong beside AGPR-accumulator MFMAs, tests/test_isa_forms.py)
                asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
                const float4 wx = *reinterpret_cast<const float4*>(p.r3_w + col);
                const float4 wy = *reinterpret_cast<const float4*>(p.r3_w + p.N + col);
                const float4 wz = *reinterpret_cast<const float4*>(p.r3_w + 2 * p.N + col);
                v.x = __fadd_rn(v.x, __fmaf_rn(wz.x, z, __fmaf_rn(wy.x, y, __fmul_rn(wx.x, x))));
                v.y = __fadd_rn(v.y, __fmaf_rn(wz.y, z, __fmaf_rn(wy.y, y, __fmul_rn(wx.y, x))));
                v.z = __fadd_rn(v.z, __fmaf_rn(wz.z, z, __fmaf_rn(wy.z, y, __fmul_rn(wx.z, x))));
                v.w = __fadd_rn(v.w, __fmaf_rn(wz.w, z, __fmaf_rn(wy.w, y, __fmul_rn(wx.w, x))));
              }
              if (p.accumulate) {
                const float4 c = *reinterpret_cast<const float4*>(p.C + o);
                v = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + c.w);
              }
              *reinterpret_cast<float4*>(p.C + o) = v;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
  };

#define WS_CONSUME(a0, a1, a2, a3, b0, b1, b2, b3)                                                            \
  do {                                                                                                         \
    float4 x0 = a0, x1 = a1, x2 = a2, x3 = a3;                                                                 \
    if (GENY) x0 = relu_mask(x0, b0), x1 = relu_mask(x1, b1), x2 = relu_mask(x2, b2), x3 = relu_mask(x3, b3);  \
    if (GATH) x0 = add_relu(x0, b0), x1 = add_relu(x1, b0), x2 = add_relu(x2, b0), x3 = add_relu(x3, b0);      \
    if (!D2 && ktail && d * 32 + pcol >= p.K) x0 = x1 = x2 = x3 = make_float4(0.f, 0.f, 0.f, 0.f); /* k past K */ \
    *reinterpret_cast<float4*>(pw) = x0;                                                                       \
    *reinterpret_cast<float4*>(pw + 8 * WS_STAGE_LD) = x1;                                                     \
    *reinterpret_cast<float4*>(pw + 16 * WS_STAGE_LD) = x2;                                                    \
    *reinterpret_cast<float4*>(pw + 24 * WS_STAGE_LD) = x3;                                                    \
    WS_ISSUE(a0, a1, a2, a3, b0, b1, b2, b3); /* refill this set: in flight for one (two with D2) double steps */ \
    __builtin_amdgcn_wave_barrier();                                                                           \
    const float4 c0 = *reinterpret_cast<const float4*>(pr), c1 = *reinterpret_cast<const float4*>(pr + 4);     \
    const float4 c2 = *reinterpret_cast<const float4*>(pr + 8), c3 = *reinterpret_cast<const float4*>(pr + 12); \
    __builtin_amdgcn_wave_barrier();                                                                           \
    step(c0, c1, c2, c3);                                                                                      \
  } while (0)
  for (; t < ntiles; t += tstride) {
    if (D2) {
      for (d = 0; d < p.nd; ++d) {
        WS_CONSUME(g0, g1, g2, g3, y0, y1, y2, y3);
        ++d;
        WS_CONSUME(h0, h1, h2, h3, z0, z1, z2, z3);
      }
    } else {
      for (d = 0; d < p.nd; ++d) WS_CONSUME(g0, g1, g2, g3, y0, y1, y2, y3);
    }
    epilogue();
  }
#undef WS_CONSUME
#undef WS_ISSUE
#undef WS_BLOAD
#undef WS_GLOAD
}

size_t stage_bytes(bool, int nw = WS_NW) { return (size_t)nw * 32 * WS_STAGE_LD * sizeof(float); }  // every wave's 32 x 32 patch

template <int NT, bool MAXPOOL, bool GENY, bool FULL, bool D2, int NW, bool GATH = false>
int launch_nt_fw(const WsArgs& p, hipStream_t st) {
  auto kern = ws_gemm_kernel<NT, MAXPOOL, GENY, NW, FULL, D2, GATH>;
  const size_t lds = (size_t)p.nd * 2 * NT * 3 * 1024 + stage_bytes(MAXPOOL, NW);
  if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PZN_ELAUNCH;
  const int nslices = (p.N + NT * 32 - 1) / (NT * 32);
  const int per_cu = lds * 2 <= 160 * 1024 ? 2 : 1;
  int gx = (256 * per_cu) / nslices;
  gx = gx < 8 ? 8 : (gx / 8) * 8;  // multiple of 8: the slices of one row range land on one XCD (shared L2)
  const int ntiles = (p.M + 31) / 32, need = (ntiles + NW - 1) / NW;
  if (gx > need) gx = need;
  PZN_LAUNCH(kern, dim3((unsigned)gx, (unsigned)nslices), dim3(NW * 64), lds, st, p);
  PZN_RETURN_LAUNCH_STATUS();
}

// 12 wavefronts per workgroup for the long streams; 8 when there are fewer row tiles than 12 x 256 (more workgroups,
// i.e. more CUs busy: 16384 x 256 x 256 runs 29 -> 25 us forward, 27 -> 20 us input gradient)
template <int NT, bool MAXPOOL, bool GENY, bool FULL, bool D2>
int launch_nt_f(const WsArgs& p, hipStream_t st) {
  return (p.M + 31) / 32 < 12 * 256 ? launch_nt_fw<NT, MAXPOOL, GENY, FULL, D2, 8>(p, st)
                                     : launch_nt_fw<NT, MAXPOOL, GENY, FULL, D2, 12>(p, st);
}

template <int NT, bool MAXPOOL, bool GENY>
int launch_nt(const WsArgs& p, hipStream_t st) {
  constexpr bool d2_on = true;  // tuning aid
  const bool full = (p.M & 31) == 0 && p.N % (NT * 32) == 0;
  if constexpr (!GENY) {  // the mask stream doubles the staging registers: one set there
    if (full && d2_on && (p.nd & 1) == 0 && (p.K & 31) == 0) return launch_nt_f<NT, MAXPOOL, GENY, true, true>(p, st);
  }
  return full ? launch_nt_f<NT, MAXPOOL, GENY, true, false>(p, st) : launch_nt_f<NT, MAXPOOL, GENY, false, false>(p, st);
}

// max-pool variant with the gathered / generated activation stream (WsArgs::gidx): full tiles only
template <int NT>
int launch_gather_nt(const WsArgs& p, hipStream_t st) {
  const bool small = (p.M + 31) / 32 < 12 * 256;
  if ((p.nd & 1) == 0)
    return small ? launch_nt_fw<NT, true, false, true, true, 8, true>(p, st) : launch_nt_fw<NT, true, false, true, true, 12, true>(p, st);
  return small ? launch_nt_fw<NT, true, false, true, false, 8, true>(p, st) : launch_nt_fw<NT, true, false, true, false, 12, true>(p, st);
}

template <bool MAXPOOL, bool GENY>
int launch_mg(const WsArgs& p, int nt, hipStream_t st) {
  if (nt == 4) return launch_nt<4, MAXPOOL, GENY>(p, st);
  if (nt == 2) return launch_nt<2, MAXPOOL, GENY>(p, st);
  return launch_nt<1, MAXPOOL, GENY>(p, st);
}

// columns per workgroup (in 32-column tiles) such that the weight slice fits in LDS; 0 = does not fit
int pick_nt(int N, int nd, bool maxpool) {
  const size_t per_tile = (size_t)nd * 2 * 3 * 1024, room = (size_t)WS_LDS_MAX - stage_bytes(maxpool);
  int nt = N > 64 ? 4 : (N > 32 ? 2 : 1);
  while (nt > 1 && per_tile * nt > room) nt >>= 1;
  return per_tile * nt <= room ? nt : 0;
}

bool ws_enabled() {
  constexpr bool on = true;  // tuning aid
  return on;
}

}  // namespace

bool pzn_ws_gemm_supported(int M, int N, int K, const float* A, int lda, const float* genY, bool maxpool) {
  if (!ws_enabled()) return false;
  if (M < 4096 || N < 32 || (N & 3) || K < 16 || (K & 3) || (lda & 3)) return false;  // small / unaligned: general engine
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(genY) & 15)) return false;
  if ((double)M * lda * 4.0 >= 2147483648.0 - 4.0e8) return false;  // 32-bit buffer offsets incl. the prefetch overrun
  return pick_nt(N, (K + 31) / 32, maxpool) != 0;
}

int pzn_ws_gemm_ex(const float* A, int lda, const float* W, int ldw, int w_kmajor, float* C, int ldc, int M, int N, int K,
                   const float* bias, int relu, const float* genY, const float* maskH, int32_t* argmax,
                   const int64_t* scat, int scat_in, int scat_out, const float* residual, float* C2, int accumulate,
                   hipStream_t st) {
  WsArgs p{A, lda, W, ldw, w_kmajor, C, ldc, M, N, K, (K + 31) / 32, bias, relu, genY, maskH, argmax, scat, scat_in, scat_out,
           residual, C2, accumulate, nullptr, nullptr, 0, 0, 0, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr},
           {nullptr, nullptr, nullptr}, {0, 0, 0}};
  const int nt = pick_nt(N, p.nd, argmax != nullptr);
  if (!nt) return PZN_EUNSUPPORTED;
  if (argmax) return (genY || scat || residual || accumulate) ? PZN_EUNSUPPORTED : launch_mg<true, false>(p, nt, st);
  if ((ldc & 3) || (reinterpret_cast<uintptr_t>(C) & 15) || (reinterpret_cast<uintptr_t>(maskH) & 15) ||
      (reinterpret_cast<uintptr_t>(bias) & 15) || (reinterpret_cast<uintptr_t>(residual) & 15) ||
      (reinterpret_cast<uintptr_t>(C2) & 15))
    return PZN_EUNSUPPORTED;
  return genY ? launch_mg<false, true>(p, nt, st) : launch_mg<false, false>(p, nt, st);
}

// C[M, N] = A[M, K] W[N, K]^T + xyz[M, 3] (wx | wy | wz)[N]: a linear layer over [xyz | features] with the coordinate columns in
// the store epilogue (planes: wx[N], wy[N], wz[N] contiguous).  PZN_EUNSUPPORTED for shapes / alignments the kernel does not take.
int pzn_ws_gemm_r3(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, const float* xyz,
                   const float* planes, hipStream_t st) {
  if (!pzn_ws_gemm_supported(M, N, K, A, lda, nullptr, false) || !xyz || !planes) return PZN_EUNSUPPORTED;
  if ((ldc & 3) || (reinterpret_cast<uintptr_t>(C) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15)) return PZN_EUNSUPPORTED;
  WsArgs p{A, lda, W, ldw, 0, C, ldc, M, N, K, (K + 31) / 32, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0,
           nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 0, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr},
           {nullptr, nullptr, nullptr}, {0, 0, 0}, xyz, planes};
  const int nt = pick_nt(N, p.nd, false);
  if (!nt) return PZN_EUNSUPPORTED;
  return launch_mg<false, false>(p, nt, st);
}

int pzn_ws_gemm(const float* A, int lda, const float* W, int ldw, int w_kmajor, float* C, int ldc, int M, int N, int K,
                const float* bias, int relu, const float* genY, const float* maskH, int32_t* argmax,
                const int64_t* scat, int scat_in, int scat_out, hipStream_t st) {
  return pzn_ws_gemm_ex(A, lda, W, ldw, w_kmajor, C, ldc, M, N, K, bias, relu, genY, maskH, argmax, scat, scat_in, scat_out,
                        nullptr, nullptr, 0, st);
}

// Second shared-MLP layer + ReLU + max over the 32 neighbours on a GENERATED activation stream: row (g, k) =
// relu(Pp[(g / S) * N + idx[g*32 + k], :] + Q[g, :]) (the first layer per point, csrc/sapoint.hip), never written to
// memory.  out[G, C2], argmax[G, C2].  C1 % 32 == 0, C2 % 32 == 0 (full tiles), 16-byte aligned tables.
int pzn_ws_gemm_gather_maxpool(const float* Pp, const float* Q, const int64_t* idx, const float* W2, const float* b2, int G,
                               int N, int S, int C1, int C2, float* out, int32_t* argmax, hipStream_t st) {
  if (!ws_enabled() || (C1 & 31) || (C2 & 31) || G <= 0 || (reinterpret_cast<uintptr_t>(Pp) & 15) ||
      (reinterpret_cast<uintptr_t>(Q) & 15))
    return PZN_EUNSUPPORTED;
  const size_t prow = (size_t)N * ((size_t)(G + S - 1) / S);
  if ((double)prow * C1 * 4.0 >= 2147483648.0 - 4.0e8 || (double)G * C1 * 4.0 >= 2147483648.0 - 4.0e8) return PZN_EUNSUPPORTED;
  WsArgs p{Pp, C1, W2, C1, 0, out, C2, G * 32, C2, C1, C1 / 32, b2, 1, nullptr, nullptr, argmax, nullptr, 0, 0,
           nullptr, nullptr, 0, idx, Q, N, S, 0, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr},
           {nullptr, nullptr, nullptr}, {0, 0, 0}};
  int nt = pick_nt(C2, p.nd, true);
  while (nt > 1 && C2 % (nt * 32)) nt >>= 1;
  if (!nt || C2 % (nt * 32)) return PZN_EUNSUPPORTED;
  if (nt == 4) return launch_gather_nt<4>(p, st);
  if (nt == 2) return launch_gather_nt<2>(p, st);
  return launch_gather_nt<1>(p, st);
}

// Three linear layers that share their input in one launch (the q, k, v projections): C_i[M, N_i] = A W_i^T + b_i,
// W_i[N_i, K] row-major, C_i dense; every N_i a multiple of 64 and 16-byte aligned pointers.
int pzn_ws_gemm3(const float* A, int lda, const float* const W[3], const float* const bias[3], float* const C[3],
                 const int N[3], int M, int K, hipStream_t st) {
  const int ntot = N[0] + N[1] + N[2];
  if (!pzn_ws_gemm_supported(M, 64, K, A, lda, nullptr, false) || (N[0] & 63) || (N[1] & 63) || (N[2] & 63)) return PZN_EUNSUPPORTED;
  for (int i = 0; i < 3; ++i)
    if ((reinterpret_cast<uintptr_t>(C[i]) & 15) || (reinterpret_cast<uintptr_t>(bias[i]) & 15)) return PZN_EUNSUPPORTED;
  WsArgs p{A, lda, W[0], K, 0, C[0], N[0], M, ntot, K, (K + 31) / 32, bias[0], 0, nullptr, nullptr, nullptr, nullptr, 0, 0,
           nullptr, nullptr, 0, nullptr, nullptr, 0, 0, 3, {W[0], W[1], W[2]}, {bias[0], bias[1], bias[2]}, {C[0], C[1], C[2]},
           {N[0], N[1], N[2]}};
  if (pick_nt(64, p.nd, false) < 2) return PZN_EUNSUPPORTED;
  return launch_nt<2, false, false>(p, st);  // 64-column slices: q and k are one slice each
}
