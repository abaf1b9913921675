// attnfused.hip — layerAttention (model5_b.py:67-75, 83-101) as chained matrix-core kernels, gfx950 only.
//
// Shapes are the model's: L = 256 points per cloud, E = 256 channels, dk = 64 (model5_b.py:436-439: embed_dim 256,
// q/k of embed_dim / 4).  Everything is computed TRANSPOSED, with the point (query or key) on the MFMA lane:
//
//   v_mfma_f32_16x16x32_bf16: D[i][n] += sum_k A[i][k] B[k][n]; lane (c = l & 15, g = l >> 4) holds A[c][8g..8g+7],
//   B[8g..8g+7][c] and D[4g + r][c].  A 16x16 result therefore has its COLUMN on the lane and four of its rows in
//   registers; two such tiles are exactly the B operand of a following product that sums over their 32 rows.
//   With the shared operand (keys, values, weights) as A and the wavefront's own 16 points as the columns, every product
//   of the block takes the previous accumulators as its B operand straight from registers:
//
//     S^T = K q^T   ->  P^T = softmax over keys (registers + two cross-lane exchanges)
//     A^T = V^T P^T ->  t^T = x^T - A^T  ->  z^T = Wo t^T  ->  r^T = x^T + relu(z^T + bo)
//
//   No score matrix, no LDS transposes; a wavefront owns 16 points end to end and never talks to another one.
//
// Split precision: every fp32 operand is x = x1 + x2 + x3 (three bf16 "planes"); a product is the six MFMAs of
// magnitude >= 2^-16 (fp32-GEMM accuracy, see gemm.hip).  Shared operands are split ONCE by their producer and kept in
// memory as plane images in MFMA-fragment order; the per-wavefront operand is split in registers as it is consumed.
// NPL = 1 (the opt-in bf16 attention mode): one plane, one MFMA per product.
//
// Plane images (bf16, 16-byte chunks), ONE per operand: [k-step of 32][plane][row tile of 16][lane][8], the chunk of lane
// (c, g) holding M[16 rt + c][32 ks + perm16(g, j)], perm16(g, j) = 16 (j >> 2) + 4g + (j & 3): the order in which two
// accumulator tiles come out as a B fragment.  Consumed three ways (pzn_mfma16.h):
//   - k = column index, operand through LDS: plain half-slabs (8 row tiles x 3 planes = 24 KB), ds_read_b128;
//   - k = column index, the wavefront's own rows: straight from global by the lane that owns the row (B operand);
//   - k = ROW index (the operand is M^T): the same bytes fetched in the T-use cut (the LDS-DMA's per-lane source
//     address does the re-arrangement) and read with ds_read_b64_tr_b16 (hardware transpose).
//
// A workgroup = 128 rows (half a cloud) = EIGHT wavefronts, two per SIMD (round 4; rounds 2-3 ran 32-row tiles,
// v_mfma_f32_32x32x16_bf16, one wavefront per SIMD on ~400 registers: nobody hid its LDS latencies, DMA issue, fragment
// splits and memory phases).  With 16 rows a wavefront's accumulator sets are 64 registers, every kernel fits 256
// registers without scratch, and the second wavefront of a SIMD fills the first one's bubbles.  The price: every
// wavefront reads the whole slab, so a k-step moves 8 x 24 KB through LDS in the time of its 2 x 768 MFMA cycles per
// SIMD - LDS bandwidth (128 B/clk) and the matrix pipe are both at their limit (measured 1900-2200 cycles per step
// against 1536 of either).  The shared operand streams through LDS by LDS-DMA, three slots, one barrier per half-slab.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

#include "pzn_mfma.h"
#include "pzn_mfma16.h"

constexpr int L = 256, E = 256, DK = 64;
constexpr int QK_IMG = 2 * 3 * 16 * 1024;   // plane image of a [256][64] operand, bytes per cloud
constexpr int V_IMG = 8 * 3 * 16 * 1024;    // plane image of a [256][256] operand

// byte offsets inside one layer's weight-plane buffer: every matrix as a sequence of 24 KB slabs in consumption order,
// slab = [plane][8 row tiles][lane][16 B]
constexpr size_t W_QKV = 0;                          // rows n = (q | k), v[0:128], v[128:256] (3 groups), k = c: 8 x 3 slabs
constexpr size_t W_O = W_QKV + 24 * SLAB;            // rows o (2 halves), k = c: 8 x 2 slabs
constexpr size_t W_OT = W_O + 16 * SLAB;             // rows c, k = o
constexpr size_t W_QT = W_OT + 16 * SLAB;            // rows c, k = d: 2 x 2 slabs
constexpr size_t W_KVT = W_QT + 4 * SLAB;            // rows c, k = d (Wk) then c' (Wv): (2 + 8) x 2 slabs
constexpr size_t W_BYTES = W_KVT + 20 * SLAB;

// ---- weights -> slabs
struct PackJob {
  const float* src;
  long rs, cs;          // A[row][k] = src[row * rs + k * cs]
  int nrt, nks;         // rows = 16 nrt, K = 32 nks
  int groups, rt0, ks0; // row groups of 8 tiles in the whole image, first tile / k-step of this job inside it
  unsigned char* dst;
};
struct PackArgs {
  PackJob job[32];
  int njobs;
};
__global__ __launch_bounds__(256) void pack_rp_kernel(PackArgs a) {
  const PackJob& J = a.job[blockIdx.y];
  const int total = J.nks * J.nrt * 64;
  for (int ci = blockIdx.x * blockDim.x + threadIdx.x; ci < total; ci += gridDim.x * blockDim.x) {
    const int lane = ci & 63, rt = (ci >> 6) % J.nrt, ks = (ci >> 6) / J.nrt;
    const int c = lane & 15, g = lane >> 4;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3);
      v[j] = J.src[(long)(16 * rt + c) * J.rs + (long)k * J.cs];
    }
    bf16x8 b[3];
    split8(v, b);
    const int tg = J.rt0 + rt;
    unsigned char* slab = J.dst + (size_t)((J.ks0 + ks) * J.groups + (tg >> 3)) * SLAB;
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(slab + ((p * 8 + (tg & 7)) * 64 + lane) * 16) = b[p];
  }
}

#ifdef ATTN_STAMPS
__device__ long long g_stamps[4][2][64];
#define STAMPK(K, i)                                                                                   \
  do {                                                                                                 \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 77))                                      \
      g_stamps[K][blockIdx.x ? 1 : 0][i] = (long long)__builtin_amdgcn_s_memtime();                   \
  } while (0)
#else
#define STAMPK(K, i) do { } while (0)
#endif

// ---- DMA pieces (8 wavefronts: three 1 KB pieces per wavefront and slab, ONE in the single-plane mode)
// jmap: piece index of this wavefront's i-th piece (NPL = 1: only i = 0, the plane-0 piece)
template <int NPL>
struct Dma16 {
  unsigned char* lds;
  int wave, lane;
  __device__ __forceinline__ uint32_t slot_addr(int s) const {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (uint32_t)(s * SLAB);
  }
  __device__ __forceinline__ uint32_t lane_addr(int s) const { return slot_addr(s) + (uint32_t)lane * 16u; }
  __device__ __forceinline__ void go(const unsigned char* src, int slot, int j) const {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds + slot * SLAB + j * 1024), 16, 0, 0);
  }
  // contiguous 24 KB slab (weights): piece j = bytes [1024 j, ..)
  __device__ __forceinline__ void weights(const unsigned char* slab, int slot, int i) const {
    if (NPL == 1 && i > 0) return;
    const int j = i * 8 + wave;            // NPL = 1: j = wave < 8 = plane 0
    go(slab + j * 1024 + lane * 16, slot, j);
  }
  // plain half-slab (k-step ks, row half hf) of an activation image [ks][plane][16 rt]
  __device__ __forceinline__ void plain(const unsigned char* img, int ks, int hf, int slot, int i) const {
    if (NPL == 1 && i > 0) return;
    const int j = i * 8 + wave, plane = j >> 3, tile = j & 7;
    go(img + ((ks * 3 + plane) * 16 + 8 * hf + tile) * 1024 + lane * 16, slot, j);
  }
  // T-use half-slab (k-step kk of 32 rows, feature half fh) of an image with F = 256: [plane][ks_f local 0..3][row tile 2kk, 2kk+1]
  __device__ __forceinline__ void t256(const unsigned char* img, uint32_t tsrc, int kk, int fh, int slot, int i) const {
    if (NPL == 1 && i > 0) return;
    const int j = i * 8 + wave, plane = j >> 3, ksfl = (j & 7) >> 1, rtn = j & 1;
    go(img + (((4 * fh + ksfl) * 3 + plane) * 16 + 2 * kk + rtn) * 1024 + tsrc, slot, j);
  }
  // T-use slab of an image with F = 64: the two k-steps kk0, kk0 + 1: [k-step][plane][ks_f 0..1][row tile]
  __device__ __forceinline__ void t64(const unsigned char* img, uint32_t tsrc, int kk0, int slot, int i) const {
    if (NPL == 1 && i > 0) return;
    const int j = NPL == 1 ? (wave >> 2) * 12 + (wave & 3) : i * 8 + wave;
    const int kloc = j / 12, m = j % 12, plane = m >> 2, ksf = (m & 3) >> 1, rtn = m & 1;
    go(img + ((ksf * 3 + plane) * 16 + 2 * (kk0 + kloc) + rtn) * 1024 + tsrc, slot, j);
  }
};

// ================================================================================================================
// projection: q, k, v = x W^T + b for the wavefront's 16 points, written as Rp16 images
struct ProjProb {
  const float* x;
  const unsigned char* w;
  const float *bq, *bk, *bv;
  unsigned char *qrp, *krp, *vrp;
};
struct ProjArgs {
  ProjProb p[2];
  int nb;
};

template <int NPL>
__global__ __launch_bounds__(NT16, 2) void attn_proj_kernel(ProjArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SLAB + 8 * STG16_BYTES];
  constexpr int DPW = NPL == 3 ? 3 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
  const int lb = logical_block(blockIdx.x, gridDim.x);
  const ProjProb& P = a.p[lb / a.nb];
  const int cb = lb % a.nb, cloud = cb >> 1, rt = (cb & 1) * 8 + wave;
  const Dma16<NPL> dma{lds, wave, lane};
  const unsigned char* wsrc = P.w + W_QKV;
  const long row0 = (long)cloud * L + 16 * rt;
  float* stg = reinterpret_cast<float*>(lds + 3 * SLAB + wave * STG16_BYTES);
  constexpr int NS = 24;
  auto issue1 = [&](int c, int i) {
    if (c < NS) dma.weights(wsrc + (size_t)c * SLAB, c % 3, i);
  };
  STAMPK(0, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) issue1(0, i);
#pragma unroll
  for (int i = 0; i < 3; ++i) issue1(1, i);
  floatx4 X[16];
  load_rows16<16>(P.x, row0, E, X, stg, lane);
  STAMPK(0, 1);
  floatx4 acc[24];
  bias_tiles16<4>(acc, P.bq, g);
  bias_tiles16<4>(acc + 4, P.bk, g);
  bias_tiles16<16>(acc + 8, P.bv, g);
  bf16x8 b[2][3];
  make_b16<NPL>(X[0], X[1], b[0]);
#pragma unroll
  for (int c = 0; c < NS; ++c) {
    const int ks = c / 3, grp = c % 3;
    step_sync(c == 0 ? 0 : (c + 1 < NS ? DPW : 0));     // (c == 0: the staged load's own waits stand in between)
    BNext16<NPL> bn;
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
      if (grp == 0 && t >= 3 && t < 7 && ks < 7) bn.pair(X[2 * ks + 2], X[2 * ks + 3], t - 3);
    };
    kstep16<NPL>(acc + 8 * grp, dma.lane_addr(c % 3), b[ks & 1], fill);
    if (grp == 0 && ks < 7) bn.get(b[(ks + 1) & 1]);
  }
  STAMPK(0, 2);
  store_rp16<2, NPL>(P.qrp + (size_t)cloud * QK_IMG, rt, lane, acc);
  store_rp16<2, NPL>(P.krp + (size_t)cloud * QK_IMG, rt, lane, acc + 4);
  STAMPK(0, 3);
  store_rp16<8, NPL>(P.vrp + (size_t)cloud * V_IMG, rt, lane, acc + 8);
  STAMPK(0, 4);
}

// ================================================================================================================
// softmax over the keys of S^T (16 tiles x 4 registers x the 4 row groups of the lane's column): S <- P, returns ln sum exp + max
__device__ __forceinline__ float softmax16(floatx4* S) {
  float m = S[0][0];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, S[t][r]);
  m = col_max4(m);
  const float c = 0.125f * LOG2E;       // logits / sqrt(dk), dk = 64 (model5_b.py:70)
  const float mc = m * c;
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float e = __builtin_amdgcn_exp2f(S[t][r] * c - mc);
      S[t][r] = e;
      sum += e;
    }
  sum = col_sum4(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int t = 0; t < 16; ++t) S[t] *= inv;
  return m * 0.125f + __logf(sum);
}

// ================================================================================================================
// forward of one block for the wavefront's 16 points
struct FwdProb {
  const float* x;
  const unsigned char *qrp, *krp, *vrp;
  const unsigned char* w;
  const float* bo;
  float* r;
  float* t;
  uint32_t* mask;  // [B*L, 8]: row, lane group g, two words of (tile, register) gate bits
  float* map;
  float* lse;
};
struct FwdArgs {
  FwdProb p[2];
  int nb;
  int map_mode;      // bit 0: add to what map holds; bit 1: map holds column sums per 16-row strip, [B][L / 16][L]
  float map_scale;
};

template <int NPL>
__global__ __launch_bounds__(NT16, 2) void attn_fwd_kernel(FwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SLAB + 8 * STG16_BYTES];
  constexpr int DPW = NPL == 3 ? 3 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
  const int lb = logical_block(blockIdx.x, gridDim.x);
  const FwdProb& P = a.p[lb / a.nb];
  const int cb = lb % a.nb, cloud = cb >> 1, rt = (cb & 1) * 8 + wave;
  const long row0 = (long)cloud * L + 16 * rt;
  const long row = row0 + (lane & 15);
  const unsigned char* krp = P.krp + (size_t)cloud * QK_IMG;
  const unsigned char* vrp = P.vrp + (size_t)cloud * V_IMG;
  const unsigned char* wo = P.w + W_O;
  const Dma16<NPL> dma{lds, wave, lane};
  float* stg = reinterpret_cast<float*>(lds + 3 * SLAB + wave * STG16_BYTES);
  const uint32_t tsrc = tr16_src_lane_off(lane);
  // slab sequence: K (2 k-steps x 2 row halves) 0..3 | V^T (T use: 8 k-steps of 32 keys x 2 feature halves) 4..19 | Wo 20..35
  constexpr int NS = 36;
  auto issue1 = [&](int c, int i) {
    if (c < 4)
      dma.plain(krp, c >> 1, c & 1, c % 3, i);
    else if (c < 20)
      dma.t256(vrp, tsrc, (c - 4) >> 1, (c - 4) & 1, c % 3, i);
    else if (c < NS)
      dma.weights(wo + (size_t)(c - 20) * SLAB, c % 3, i);
  };
  STAMPK(1, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) issue1(0, i);
#pragma unroll
  for (int i = 0; i < 3; ++i) issue1(1, i);
  bf16x8 qf[2][3];
  own_frag16<NPL>(P.qrp + (size_t)cloud * QK_IMG, 0, rt, lane, qf[0]);
  own_frag16<NPL>(P.qrp + (size_t)cloud * QK_IMG, 1, rt, lane, qf[1]);
  floatx4 S[16];
  ZERO_TILES16(S, 16);
  // ---- S^T = K q^T
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    step_sync(DPW);      // (c == 0: the own-fragment loads are younger than slab 1's pieces: waiting them out is conservative)
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
    };
    kstep16<NPL>(S + 8 * (c & 1), dma.lane_addr(c % 3), qf[c >> 1], fill);
  }
  STAMPK(1, 1);
  const float lse = softmax16(S);
  if (g == 0) P.lse[row] = lse;
  STAMPK(1, 2);
  if (P.map) {
    if (a.map_mode & 2) {      // only the column sums of the wavefront's 16 rows: map is [B][L / 16][L]
      float* strip = P.map + ((size_t)cloud * (L / 16) + rt) * L;
      if (a.map_mode & 1)
        store_colsum16<1>(strip, S, stg, lane, a.map_scale);
      else
        store_colsum16<0>(strip, S, stg, lane, a.map_scale);
    } else if (a.map_mode & 1) {
      store_rows16<16, 1>(P.map, row0, L, S, stg, lane, a.map_scale);
    } else {
      store_rows16<16, 0>(P.map, row0, L, S, stg, lane, a.map_scale);
    }
  }
  STAMPK(1, 3);
  // ---- A^T = V^T P^T (8 k-steps over the keys, two feature halves each)
  floatx4 O[16];
  ZERO_TILES16(O, 16);
  bf16x8 b[2][3];
  make_b16<NPL>(S[0], S[1], b[0]);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int c = 4 + s, kk = s >> 1, fh = s & 1;
    step_sync(DPW);
    BNext16<NPL> bn;
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
      if (fh == 0 && t >= 3 && t < 7 && kk < 7) bn.pair(S[2 * kk + 2], S[2 * kk + 3], t - 3);
    };
    kstep16_tr<8, 8192, 0, NPL>(O + 8 * fh, dma.lane_addr(c % 3), b[kk & 1], fill);
    if (fh == 0 && kk < 7) bn.get(b[(kk + 1) & 1]);
  }
  STAMPK(1, 4);
  // ---- t^T = x^T - A^T
  {
    floatx4(&X)[16] = S;      // (P is dead)
    load_rows16<16>(P.x, row0, E, X, stg, lane);
#pragma unroll
    for (int t = 0; t < 16; ++t) O[t] = X[t] - O[t];
  }
  STAMPK(1, 5);
  // ---- z^T = Wo t^T
  step_sync(DPW);
  store_rows16<16>(P.t, row0, E, O, stg, lane);
  floatx4(&Z)[16] = S;
  bias_tiles16<16>(Z, P.bo, g);
  make_b16<NPL>(O[0], O[1], b[0]);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int c = 20 + s, ks = s >> 1, hf = s & 1;
    if (s > 0) step_sync(c + 1 < NS ? DPW : 0);
    BNext16<NPL> bn;
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
      if (hf == 0 && t >= 3 && t < 7 && ks < 7) bn.pair(O[2 * ks + 2], O[2 * ks + 3], t - 3);
    };
    kstep16<NPL>(Z + 8 * hf, dma.lane_addr(c % 3), b[ks & 1], fill);
    if (hf == 0 && ks < 7) bn.get(b[(ks + 1) & 1]);
  }
  STAMPK(1, 6);
  // ---- r = x + relu(z): the gate bits for the backward; x is added in row layout at the store
  {
    uint32_t bits[2] = {0u, 0u};
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on = Z[t][r] > 0.f;
        bits[t >> 3] |= (on ? 1u : 0u) << ((t & 7) * 4 + r);
        Z[t][r] = on ? Z[t][r] : 0.f;
      }
    *reinterpret_cast<uint2*>(P.mask + (row * 4 + g) * 2) = make_uint2(bits[0], bits[1]);
  }
  store_rows16<16>(P.r, row0, E, Z, stg, lane, 1.f, NoGate(), P.x, E);
  STAMPK(1, 7);
}

// ================================================================================================================
// backward, query side: the wavefront's 16 points as QUERIES.
//   dz = dr . gate;  dt^T = Wo^T dz^T;  da = -dt (image for the key-side pass);  dP^T = V da^T;  P^T recomputed;
//   delta = sum_key P dP;  dS^T = P^T (dP^T - delta) / 8;  dq^T = K^T dS^T;  u = dr + dt (partial dx: the key-side pass
//   adds dq Wq with its own terms).
// dr (+ dr2) is read once - the gate is applied where dz is consumed: in the split of the B fragments and in the store of
// dz, so dr is still there for u = dr + dt -, u is written once, in register order (tile image), and never more than
// three 64-register accumulator sets are live.
struct BwdQProb {
  const float* dr;
  const float* dr2;
  int ld_dr, ld_dr2;
  const uint32_t* mask;
  const unsigned char *qrp, *krp, *vrp;
  const unsigned char* w;
  float* dz;
  float* u;       // tile image
  float* dq;      // rows
  float* dqt;     // tile image
  unsigned char* darp;
  float* delta;
};
struct BwdQArgs {
  BwdQProb p[2];
  int nb;
};

template <int NPL>
__global__ __launch_bounds__(NT16, 2) void attn_bwd_q_kernel(BwdQArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SLAB + 8 * STG16_BYTES];
  constexpr int DPW = NPL == 3 ? 3 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
  const int lb = logical_block(blockIdx.x, gridDim.x);
  const BwdQProb& P = a.p[lb / a.nb];
  const int cb = lb % a.nb, cloud = cb >> 1, rt = (cb & 1) * 8 + wave;
  const long row0 = (long)cloud * L + 16 * rt;
  const long row = row0 + (lane & 15);
  const unsigned char* krp = P.krp + (size_t)cloud * QK_IMG;
  const unsigned char* vrp = P.vrp + (size_t)cloud * V_IMG;
  const unsigned char* wot = P.w + W_OT;
  const Dma16<NPL> dma{lds, wave, lane};
  float* stg = reinterpret_cast<float*>(lds + 3 * SLAB + wave * STG16_BYTES);
  const uint32_t tsrc = tr16_src_lane_off(lane);
  // slab sequence: Wo^T 0..15 | V (8 k-steps over c x 2 key halves) 16..31 | K (2 x 2) 32..35 | K^T (T use, 2 k-steps of 32 keys each) 36..39
  constexpr int NS = 40;
  auto issue1 = [&](int c, int i) {
    if (c < 16)
      dma.weights(wot + (size_t)c * SLAB, c % 3, i);
    else if (c < 32)
      dma.plain(vrp, (c - 16) >> 1, (c - 16) & 1, c % 3, i);
    else if (c < 36)
      dma.plain(krp, (c - 32) >> 1, (c - 32) & 1, c % 3, i);
    else if (c < NS)
      dma.t64(krp, tsrc, 2 * (c - 36), c % 3, i);
  };
  STAMPK(2, 0);
#pragma unroll
  for (int i = 0; i < 3; ++i) issue1(0, i);
#pragma unroll
  for (int i = 0; i < 3; ++i) issue1(1, i);
  floatx4 S[16];     // dr^T, then u^T = dr^T + dt^T, later the scores
  floatx4 DT[16];
  if (P.dr2)
    load_rows16_sum<16>(P.dr, P.ld_dr, P.dr2, P.ld_dr2, row0, S, DT, stg, lane);
  else
    load_rows16<16>(P.dr, row0, P.ld_dr, S, stg, lane);
  Gate16 gate;
  {
    const uint2 mb = *reinterpret_cast<const uint2*>(P.mask + (row * 4 + g) * 2);
    gate.w[0] = mb.x, gate.w[1] = mb.y;
  }
  STAMPK(2, 1);
  store_rows16<16, 0, Gate16>(P.dz, row0, E, S, stg, lane, 1.f, gate);
  STAMPK(2, 2);
  // ---- dt^T = Wo^T dz^T (8 k-steps over o x 2 halves of c); the gate is applied as the fragments are split
  ZERO_TILES16(DT, 16);
  bf16x8 b[2][3];
  {
    BNext16<NPL> b0;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) b0.pair_gated(S[0], S[1], jj, gate.w[0], 0);
    b0.get(b[0]);
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int c = s, ks = s >> 1, hf = s & 1;
    step_sync(s == 0 ? 0 : DPW);
    BNext16<NPL> bn;
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
      if (hf == 0 && t >= 3 && t < 7 && ks < 7) {
        const int t0 = 2 * ks + 2;
        bn.pair_gated(S[t0], S[t0 + 1], t - 3, gate.w[t0 >> 3], (t0 & 7) * 4);
      }
    };
    kstep16<NPL>(DT + 8 * hf, dma.lane_addr(c % 3), b[ks & 1], fill);
    if (hf == 0 && ks < 7) bn.get(b[(ks + 1) & 1]);
  }
  // ---- DP^T = V dt^T = -dP^T; u and the da image go to memory at the head of its first step
  STAMPK(2, 3);
  step_sync(DPW);
#pragma unroll
  for (int t = 0; t < 16; ++t) S[t] += DT[t];       // u = dr + dt
  store_tiles16<16>(P.u, (long)cloud * 16 + rt, lane, S);
  __builtin_amdgcn_sched_barrier(0);
  STAMPK(2, 5);
  store_rp16<8, NPL, true>(P.darp + (size_t)cloud * V_IMG, rt, lane, DT);
  __builtin_amdgcn_sched_barrier(0);
  STAMPK(2, 6);
  floatx4 DP[16];
  ZERO_TILES16(DP, 16);
  make_b16<NPL>(DT[0], DT[1], b[0]);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int c = 16 + s, ks = s >> 1, hf = s & 1;
    if (s > 0) step_sync(DPW);
    BNext16<NPL> bn;
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
      if (hf == 0 && t >= 3 && t < 7 && ks < 7) bn.pair(DT[2 * ks + 2], DT[2 * ks + 3], t - 3);
    };
    kstep16<NPL>(DP + 8 * hf, dma.lane_addr(c % 3), b[ks & 1], fill);
    if (hf == 0 && ks < 7) bn.get(b[(ks + 1) & 1]);
  }
  // ---- S^T = K q^T, P^T (u is in memory: its registers hold the scores now)
  __builtin_amdgcn_sched_barrier(0);
  STAMPK(2, 9);
  floatx4(&S2)[16] = S;
  ZERO_TILES16(S2, 16);
  bf16x8 qf[2][3];
  own_frag16<NPL>(P.qrp + (size_t)cloud * QK_IMG, 0, rt, lane, qf[0]);
  own_frag16<NPL>(P.qrp + (size_t)cloud * QK_IMG, 1, rt, lane, qf[1]);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = 32 + s;
    step_sync(s == 0 ? DPW + 2 * NPL : DPW);    // (s == 0: the own-fragment loads are younger than the slab waited for)
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
    };
    kstep16<NPL>(S2 + 8 * (s & 1), dma.lane_addr(c % 3), qf[s >> 1], fill);
  }
  STAMPK(2, 10);
  softmax16(S2);
  {  // delta = sum P dP;  dS = P (dP - delta) / 8  with dP = -DP
    float d = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) d -= S2[t][r] * DP[t][r];
    d = col_sum4(d);
    if (g == 0) P.delta[row] = d;
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) S2[t][r] = S2[t][r] * (-DP[t][r] - d) * 0.125f;
  }
  // ---- dq^T = K^T dS^T (8 k-steps of 32 keys, two per slab; rows = d: 4 tiles)
  __builtin_amdgcn_sched_barrier(0);
  STAMPK(2, 11);
  floatx4 DQ[4];
  ZERO_TILES16(DQ, 4);
  make_b16<NPL>(S2[0], S2[1], b[0]);
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int c = 36 + sl;
    step_sync(sl == 0 ? DPW + 1 : (sl == 3 ? 0 : DPW));   // (sl == 0: the store of delta is younger than the slab as well)
#pragma unroll
    for (int i = 0; i < 3; ++i) issue1(c + 2, i);
    const uint32_t la = dma.lane_addr(c % 3);
    static_for<0, 2>([&](auto iq) {
      constexpr int q2 = decltype(iq)::value;
      const int kk = 2 * sl + q2;
      BNext16<NPL> bn;
      auto fill = [&](int t) {
        if (kk < 7) bn.pair(S2[2 * kk + 2], S2[2 * kk + 3], t);
      };
      kstep16_tr<4, 4096, q2 * 12288, NPL>(DQ, la, b[kk & 1], fill);
      if (kk < 7) bn.get(b[(kk + 1) & 1]);
    });
  }
  STAMPK(2, 12);
  store_tiles16<4>(P.dqt, (long)cloud * 16 + rt, lane, DQ);
  store_rows16<4>(P.dq, row0, DK, DQ, stg, lane);
  STAMPK(2, 13);
}

// ================================================================================================================
// backward, key side: the wavefront's 16 points as KEYS against all 256 queries of the cloud (whole-cloud accumulators:
// every half-slab feeds 48 MFMAs per wavefront).
//   S = q k^T (key on the lane, query in the registers), P = exp(S/8 - lse_q), dP = da v^T, dS = P (dP - delta_q) / 8,
//   dk^T = q^T dS, dv^T = da^T P (in this order: dS dies before the dv accumulators are born);
//   then dx = u + dq Wq + dk Wk + dv Wv for the wavefront's points (u = dr + dt and dq from the query-side pass)
struct BwdKProb {
  const unsigned char *qrp, *krp, *vrp, *darp;
  const unsigned char* w;
  const float *lse, *delta;
  const float* u;     // tile image
  const float* dq;    // tile image
  float* dk;
  float* dv;
  float* dx;
};
struct BwdKArgs {
  BwdKProb p[2];
  int nb;
};

template <int NPL>
__global__ __launch_bounds__(NT16, 2) void attn_bwd_k_kernel(BwdKArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SLAB + 2048 + 8 * STG16_BYTES];   // ring | lse[256] | delta[256] | staging
  constexpr int DPW = NPL == 3 ? 3 : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
  const int lb = logical_block(blockIdx.x, gridDim.x);
  const BwdKProb& P = a.p[lb / a.nb];
  const int cb = lb % a.nb, cloud = cb >> 1, rt = (cb & 1) * 8 + wave;
  const unsigned char* qrp = P.qrp + (size_t)cloud * QK_IMG;
  const unsigned char* krp = P.krp + (size_t)cloud * QK_IMG;
  const unsigned char* vrp = P.vrp + (size_t)cloud * V_IMG;
  const unsigned char* darp = P.darp + (size_t)cloud * V_IMG;
  const unsigned char* wqkvt = P.w + W_QT;    // [Wq^T | Wk^T | Wv^T]: (2 + 2 + 8) k-steps x 2 halves = 24 slabs, contiguous
  const Dma16<NPL> dma{lds, wave, lane};
  const long row0 = (long)cloud * L + 16 * rt;
  float* stg = reinterpret_cast<float*>(lds + 3 * SLAB + 2048 + wave * STG16_BYTES);
  float* row_consts = reinterpret_cast<float*>(lds + 3 * SLAB);
  const uint32_t tsrc = tr16_src_lane_off(lane);
  STAMPK(3, 0);
  if (tid < 256) {
    row_consts[tid] = P.lse[(long)cloud * L + tid];
    row_consts[256 + tid] = P.delta[(long)cloud * L + tid];
  }
  // slab sequence: q (2 x 2) 0..3 | da (8 k-steps over c x 2 query halves) 4..19 | q^T (T use, 2 k-steps of 32 queries each) 20..23 |
  //                da^T (T use: 8 k-steps of 32 queries x 2 feature halves) 24..39 | weights 40..63
  constexpr int NS = 64;
  auto issue1 = [&](int c, int i) {
    if (c < 4)
      dma.plain(qrp, c >> 1, c & 1, c % 3, i);
    else if (c < 20)
      dma.plain(darp, (c - 4) >> 1, (c - 4) & 1, c % 3, i);
    else if (c < 24)
      dma.t64(qrp, tsrc, 2 * (c - 20), c % 3, i);
    else if (c < 40)
      dma.t256(darp, tsrc, (c - 24) >> 1, (c - 24) & 1, c % 3, i);
    else if (c < NS)
      dma.weights(wqkvt + (size_t)(c - 40) * SLAB, c % 3, i);
  };
#pragma unroll
  for (int i = 0; i < 3; ++i) issue1(0, i);
#pragma unroll
  for (int i = 0; i < 3; ++i) issue1(1, i);
  bf16x8 of[3][3];     // own fragments of global k-step gk (S loop: gk = ks, dP loop: gk = 2 + ks) in of[gk % 3]
  own_frag16<NPL>(krp, 0, rt, lane, of[0]);
  own_frag16<NPL>(krp, 1, rt, lane, of[1]);
  __syncthreads();   // row constants in LDS (drains the first two slabs once)
  STAMPK(3, 1);
  // ---- S = q k^T: rows = the cloud's 256 queries (16 tiles), B = this wavefront's key fragments
  floatx4 S[16];
  ZERO_TILES16(S, 16);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = s, ks = s >> 1, hf = s & 1;
    if (s > 0) step_sync(s == 2 ? DPW + NPL : DPW);
    if (s == 1) own_frag16<NPL>(vrp, 0, rt, lane, of[2]);      // (first value fragment, behind slab 2's pieces of step 0)
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
    };
    kstep16<NPL>(S + 8 * hf, dma.lane_addr(c % 3), of[ks], fill);
  }
  STAMPK(3, 2);
  {
    const float c = 0.125f * LOG2E;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float4 ls = *reinterpret_cast<const float4*>(row_consts + 16 * t + 4 * g);
      const float lv[4] = {ls.x, ls.y, ls.z, ls.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) S[t][r] = __builtin_amdgcn_exp2f(S[t][r] * c - lv[r] * LOG2E);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  STAMPK(3, 3);
  // ---- dP = da v^T (8 k-steps over c x 2 query halves); B = this wavefront's value fragments, one k-step ahead
  floatx4 DP[16];
  ZERO_TILES16(DP, 16);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int c = 4 + s, ks = s >> 1, hf = s & 1;
    step_sync(hf == 1 && ks < 7 ? DPW + NPL : DPW);     // (the next k-step's fragment loads were issued in the hf == 0 step)
    if (hf == 0 && ks < 7) own_frag16<NPL>(vrp, ks + 1, rt, lane, of[(ks + 3) % 3]);
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
    };
    kstep16<NPL>(DP + 8 * hf, dma.lane_addr(c % 3), of[(ks + 2) % 3], fill);
  }
  STAMPK(3, 4);
  // dS = P (dP - delta_q) / 8
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 de = *reinterpret_cast<const float4*>(row_consts + 256 + 16 * t + 4 * g);
    const float dv4[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) DP[t][r] = S[t][r] * (DP[t][r] - dv4[r]) * 0.125f;
  }
  // ---- dk^T = q^T dS (8 k-steps of 32 queries, two per slab; rows = d: 4 tiles)
  __builtin_amdgcn_sched_barrier(0);
  floatx4 DKt[4];
  ZERO_TILES16(DKt, 4);
  bf16x8 b[2][3];
  make_b16<NPL>(DP[0], DP[1], b[0]);
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int c = 20 + sl;
    step_sync(DPW);
#pragma unroll
    for (int i = 0; i < 3; ++i) issue1(c + 2, i);
    const uint32_t la = dma.lane_addr(c % 3);
    static_for<0, 2>([&](auto iq) {
      constexpr int q2 = decltype(iq)::value;
      const int kk = 2 * sl + q2;
      BNext16<NPL> bn;
      auto fill = [&](int t) {
        if (kk < 7) bn.pair(DP[2 * kk + 2], DP[2 * kk + 3], t);
      };
      kstep16_tr<4, 4096, q2 * 12288, NPL>(DKt, la, b[kk & 1], fill);
      if (kk < 7) bn.get(b[(kk + 1) & 1]);
    });
  }
  STAMPK(3, 5);
  // ---- dv^T = da^T P (8 k-steps of 32 queries x 2 feature halves); dS is dead: its registers are the accumulators
  __builtin_amdgcn_sched_barrier(0);
  floatx4(&DV)[16] = DP;
  ZERO_TILES16(DV, 16);
  make_b16<NPL>(S[0], S[1], b[0]);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int c = 24 + s, kk = s >> 1, fh = s & 1;
    step_sync(DPW);
    BNext16<NPL> bn;
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
      if (fh == 0 && t >= 3 && t < 7 && kk < 7) bn.pair(S[2 * kk + 2], S[2 * kk + 3], t - 3);
    };
    kstep16_tr<8, 8192, 0, NPL>(DV + 8 * fh, dma.lane_addr(c % 3), b[kk & 1], fill);
    if (fh == 0 && kk < 7) bn.get(b[(kk + 1) & 1]);
  }
  // ---- dx^T = u^T + Wq^T dq^T + Wk^T dk^T + Wv^T dv^T (2 + 2 + 8 k-steps x 2 halves); P is dead: its registers take u
  STAMPK(3, 6);
  __builtin_amdgcn_sched_barrier(0);
  floatx4(&DX)[16] = S;
  floatx4 DQ[4];
  step_sync(DPW);
  load_tiles16<4>(P.dq, (long)cloud * 16 + rt, lane, DQ);
  load_tiles16<16>(P.u, (long)cloud * 16 + rt, lane, DX);
  store_rows16<4>(P.dk, row0, DK, DKt, stg, lane);
  store_rows16<16>(P.dv, row0, E, DV, stg, lane);
  bf16x8 bt[2][3];
  make_b16<NPL>(DQ[0], DQ[1], bt[0]);
  STAMPK(3, 7);
#pragma unroll
  for (int s = 0; s < 24; ++s) {
    const int c = 40 + s, ks = s >> 1, hf = s & 1;
    if (s > 0) step_sync(s < 23 ? DPW : 0);
    BNext16<NPL> bn;
    auto fill = [&](int t) {
      if (t < 3) issue1(c + 2, t);
      if (hf == 0 && t >= 3 && t < 7 && ks < 11) {
        const int kn = ks + 1;
        if (kn < 2)
          bn.pair(DQ[2 * kn], DQ[2 * kn + 1], t - 3);
        else if (kn < 4)
          bn.pair(DKt[2 * (kn - 2)], DKt[2 * (kn - 2) + 1], t - 3);
        else
          bn.pair(DV[2 * (kn - 4)], DV[2 * (kn - 4) + 1], t - 3);
      }
    };
    kstep16<NPL>(DX + 8 * hf, dma.lane_addr(c % 3), bt[ks & 1], fill);
    if (hf == 0 && ks < 11) bn.get(bt[(ks + 1) & 1]);
  }
  STAMPK(3, 8);
  store_rows16<16>(P.dx, row0, E, DX, stg, lane);
  STAMPK(3, 9);
}


bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// ================================================================================================================
// C ABI.  All entry points take up to two independent problems (the two encoders of predict5, model5_b.py:700-707)
// so that one launch fills the chip: 2 B workgroups of eight wavefronts per problem, two wavefronts per SIMD.
#ifdef ATTN_STAMPS
extern "C" __attribute__((visibility("default"))) int pzn_attn_fused_read_stamps(long long* host, int clear) {
  if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return -1;
  if (clear) {
    static long long z[4][2][64];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
PZN_EXPORT size_t pzn_attn_fused_weight_bytes(void) { return W_BYTES; }

PZN_EXPORT size_t pzn_attn_fused_qk_image_bytes(int B) { return B > 0 ? (size_t)B * QK_IMG : 0; }
PZN_EXPORT size_t pzn_attn_fused_v_image_bytes(int B) { return B > 0 ? (size_t)B * V_IMG : 0; }

PZN_EXPORT int pzn_attn_fused_supported(int L_, int E_, int dk) { return L_ == L && E_ == E && dk == DK; }

// planes of the weights of n <= 4 blocks (Wq, Wk [dk, E]; Wv, Wo [E, E]) for all kernels, one launch
static void pack_jobs(PackArgs& a, int at, const float* Wq, const float* Wk, const float* Wv, const float* Wo, unsigned char* w) {
  PackJob* J = a.job + at;
  // W_QKV: rows n = q (4 tiles) | k (4) | v (16) = 3 groups of 8 tiles, k = c (8 k-steps)
  J[0] = PackJob{Wq, E, 1, 4, 8, 3, 0, 0, w + W_QKV};
  J[1] = PackJob{Wk, E, 1, 4, 8, 3, 4, 0, w + W_QKV};
  J[2] = PackJob{Wv, E, 1, 16, 8, 3, 8, 0, w + W_QKV};
  J[3] = PackJob{Wo, E, 1, 16, 8, 2, 0, 0, w + W_O};         // rows o, k = c
  J[4] = PackJob{Wo, 1, E, 16, 8, 2, 0, 0, w + W_OT};        // rows c, k = o:  A[c][o] = Wo[o][c]
  J[5] = PackJob{Wq, 1, E, 16, 2, 2, 0, 0, w + W_QT};        // rows c, k = d:  A[c][d] = Wq[d][c]
  J[6] = PackJob{Wk, 1, E, 16, 2, 2, 0, 0, w + W_KVT};       // rows c, k = d
  J[7] = PackJob{Wv, 1, E, 16, 8, 2, 0, 2, w + W_KVT};       // rows c, k = c' (k-steps 2..9)
}

PZN_EXPORT int pzn_attn_fused_prep_weights_n(int n, const float* const* Wq, const float* const* Wk, const float* const* Wv,
                                             const float* const* Wo, void* const* planes, pzn_stream_t stream) {
  PZN_CHECK_ARG(n >= 1 && n <= 4 && Wq && Wk && Wv && Wo && planes);
  PackArgs a;
  a.njobs = 8 * n;
  for (int i = 0; i < n; ++i) {
    PZN_CHECK_ARG(Wq[i] && Wk[i] && Wv[i] && Wo[i] && planes[i] && aligned16(planes[i]));
    pack_jobs(a, 8 * i, Wq[i], Wk[i], Wv[i], Wo[i], static_cast<unsigned char*>(planes[i]));
  }
  PZN_LAUNCH(pack_rp_kernel, dim3(12, a.njobs), dim3(256), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_attn_fused_prep_weights(const float* Wq, const float* Wk, const float* Wv, const float* Wo, void* planes,
                                           pzn_stream_t stream) {
  return pzn_attn_fused_prep_weights_n(1, &Wq, &Wk, &Wv, &Wo, &planes, stream);
}

// q, k, v images of `nprob` problems: x[i][B*L, E], weight planes w[i], biases; images: qrp, krp (QK size), vrp (V size)
PZN_EXPORT int pzn_attn_fused_proj(int nprob, const float* const* x, const void* const* w, const float* const* bq,
                                   const float* const* bk, const float* const* bv, int B, void* const* qrp,
                                   void* const* krp, void* const* vrp, pzn_stream_t stream) {
  PZN_CHECK_ARG(nprob >= 1 && nprob <= 2 && B > 0 && x && w && bq && bk && bv && qrp && krp && vrp);
  ProjArgs a;
  a.nb = 2 * B;
  for (int i = 0; i < nprob; ++i) {
    PZN_CHECK_ARG(x[i] && w[i] && bq[i] && bk[i] && bv[i] && qrp[i] && krp[i] && vrp[i]);
    PZN_CHECK_ARG(aligned16(x[i]) && aligned16(w[i]) && aligned16(bq[i]) && aligned16(bk[i]) && aligned16(bv[i]) &&
                  aligned16(qrp[i]) && aligned16(krp[i]) && aligned16(vrp[i]));
    a.p[i] = ProjProb{x[i], static_cast<const unsigned char*>(w[i]), bq[i], bk[i], bv[i],
                      static_cast<unsigned char*>(qrp[i]), static_cast<unsigned char*>(krp[i]),
                      static_cast<unsigned char*>(vrp[i])};
  }
  if (pzn_attn_precision_mode() == 1)
    PZN_LAUNCH(attn_proj_kernel<1>, dim3(a.nb * nprob), dim3(NT16), 0, pzn_hip_stream(stream), a);
  else
    PZN_LAUNCH(attn_proj_kernel<3>, dim3(a.nb * nprob), dim3(NT16), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}

// one block forward: r = x + relu(Wo (x - softmax(q k^T / 8) v) + bo); also t = x - attn v, the gate bits, ln-sum-exp per
// row and (map[i] != NULL) the running mean map: map = scale P (map_mode 0) or map += scale P (1); map_mode 2 / 3: the same
// for the column sums of P over each strip of 16 query rows, map[i] = [B][L / 16][L] (what a mean over the rows needs)
PZN_EXPORT int pzn_attn_fused_fwd(int nprob, const float* const* x, const void* const* qrp, const void* const* krp,
                                  const void* const* vrp, const void* const* w, const float* const* bo, int B,
                                  float* const* r, float* const* t, void* const* mask, float* const* map, float* const* lse,
                                  int map_mode, float map_scale, pzn_stream_t stream) {
  PZN_CHECK_ARG(nprob >= 1 && nprob <= 2 && B > 0 && x && qrp && krp && vrp && w && bo && r && t && mask && map && lse &&
                map_mode >= 0 && map_mode <= 3);
  FwdArgs a;
  a.nb = 2 * B;
  a.map_mode = map_mode;
  a.map_scale = map_scale;
  for (int i = 0; i < nprob; ++i) {
    PZN_CHECK_ARG(x[i] && qrp[i] && krp[i] && vrp[i] && w[i] && bo[i] && r[i] && t[i] && mask[i] && lse[i]);
    PZN_CHECK_ARG(aligned16(x[i]) && aligned16(qrp[i]) && aligned16(krp[i]) && aligned16(vrp[i]) && aligned16(w[i]) &&
                  aligned16(bo[i]) && aligned16(r[i]) && aligned16(t[i]) && aligned16(mask[i]) && aligned16(map[i]));
    a.p[i] = FwdProb{x[i], static_cast<const unsigned char*>(qrp[i]), static_cast<const unsigned char*>(krp[i]),
                     static_cast<const unsigned char*>(vrp[i]), static_cast<const unsigned char*>(w[i]), bo[i], r[i], t[i],
                     static_cast<uint32_t*>(mask[i]), map[i], lse[i]};
  }
  if (pzn_attn_precision_mode() == 1)
    PZN_LAUNCH(attn_fwd_kernel<1>, dim3(a.nb * nprob), dim3(NT16), 0, pzn_hip_stream(stream), a);
  else
    PZN_LAUNCH(attn_fwd_kernel<3>, dim3(a.nb * nprob), dim3(NT16), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}

// backward, query side (see attn_bwd_q_kernel): writes dz, dq (rows), delta, the image of da, and for the key-side pass
// u = dr + dt and dq again as tile images (same sizes as the row tensors, layout private to the two kernels)
PZN_EXPORT int pzn_attn_fused_bwd_q(int nprob, const float* const* dr, int ld_dr, const float* const* dr2, int ld_dr2,
                                    const void* const* mask, const void* const* qrp, const void* const* krp,
                                    const void* const* vrp, const void* const* w, int B, float* const* dz, float* const* u,
                                    float* const* dq, float* const* dqt, void* const* darp, float* const* delta,
                                    pzn_stream_t stream) {
  PZN_CHECK_ARG(nprob >= 1 && nprob <= 2 && B > 0 && dr && mask && qrp && krp && vrp && w && dz && u && dq && dqt && darp && delta);
  BwdQArgs a;
  a.nb = 2 * B;
  for (int i = 0; i < nprob; ++i) {
    PZN_CHECK_ARG(dr[i] && mask[i] && qrp[i] && krp[i] && vrp[i] && w[i] && dz[i] && u[i] && dq[i] && dqt[i] && darp[i] && delta[i]);
    PZN_CHECK_ARG(aligned16(dr[i]) && aligned16(mask[i]) && aligned16(qrp[i]) && aligned16(krp[i]) && aligned16(vrp[i]) &&
                  aligned16(w[i]) && aligned16(dz[i]) && aligned16(u[i]) && aligned16(dq[i]) && aligned16(dqt[i]) &&
                  aligned16(darp[i]));
    PZN_CHECK_ARG(ld_dr >= E && (ld_dr & 3) == 0 && (!dr2 || !dr2[i] || (aligned16(dr2[i]) && ld_dr2 >= E && (ld_dr2 & 3) == 0)));
    a.p[i] = BwdQProb{dr[i], dr2 ? dr2[i] : nullptr, ld_dr, ld_dr2, static_cast<const uint32_t*>(mask[i]),
                      static_cast<const unsigned char*>(qrp[i]), static_cast<const unsigned char*>(krp[i]),
                      static_cast<const unsigned char*>(vrp[i]), static_cast<const unsigned char*>(w[i]), dz[i], u[i], dq[i],
                      dqt[i], static_cast<unsigned char*>(darp[i]), delta[i]};
  }
  if (pzn_attn_precision_mode() == 1)
    PZN_LAUNCH(attn_bwd_q_kernel<1>, dim3(a.nb * nprob), dim3(NT16), 0, pzn_hip_stream(stream), a);
  else
    PZN_LAUNCH(attn_bwd_q_kernel<3>, dim3(a.nb * nprob), dim3(NT16), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}

// backward, key side (see attn_bwd_k_kernel): writes dk, dv and the block's input gradient dx = u + dq Wq + dk Wk + dv Wv
PZN_EXPORT int pzn_attn_fused_bwd_k(int nprob, const void* const* qrp, const void* const* krp, const void* const* vrp,
                                    const void* const* darp, const void* const* w, const float* const* lse,
                                    const float* const* delta, const float* const* u, const float* const* dq, int B,
                                    float* const* dk, float* const* dv, float* const* dx, pzn_stream_t stream) {
  PZN_CHECK_ARG(nprob >= 1 && nprob <= 2 && B > 0 && qrp && krp && vrp && darp && w && lse && delta && u && dq && dk && dv && dx);
  BwdKArgs a;
  a.nb = 2 * B;
  for (int i = 0; i < nprob; ++i) {
    PZN_CHECK_ARG(qrp[i] && krp[i] && vrp[i] && darp[i] && w[i] && lse[i] && delta[i] && u[i] && dq[i] && dk[i] && dv[i] && dx[i]);
    PZN_CHECK_ARG(aligned16(qrp[i]) && aligned16(krp[i]) && aligned16(vrp[i]) && aligned16(darp[i]) && aligned16(w[i]) &&
                  aligned16(lse[i]) && aligned16(delta[i]) && aligned16(u[i]) && aligned16(dq[i]) && aligned16(dk[i]) &&
                  aligned16(dv[i]) && aligned16(dx[i]));
    a.p[i] = BwdKProb{static_cast<const unsigned char*>(qrp[i]), static_cast<const unsigned char*>(krp[i]),
                      static_cast<const unsigned char*>(vrp[i]), static_cast<const unsigned char*>(darp[i]),
                      static_cast<const unsigned char*>(w[i]), lse[i], delta[i], u[i], dq[i], dk[i], dv[i], dx[i]};
  }
  if (pzn_attn_precision_mode() == 1)
    PZN_LAUNCH(attn_bwd_k_kernel<1>, dim3(a.nb * nprob), dim3(NT16), 0, pzn_hip_stream(stream), a);
  else
    PZN_LAUNCH(attn_bwd_k_kernel<3>, dim3(a.nb * nprob), dim3(NT16), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}
