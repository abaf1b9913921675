// attnfused.hip — layerAttention (model5_b.py:67-75, 83-101) as chained matrix-core kernels, gfx950 only.
//
// Shapes are the model's: L = 256 points per cloud, E = 256 channels, dk = 64 (model5_b.py:436-439: embed_dim 256,
// q/k of embed_dim / 4).  Everything is computed TRANSPOSED, with the point (query or key) on the MFMA lane:
//
//   v_mfma_f32_32x32x16_bf16: D[i][j] += sum_k A[i][k] B[k][j]; lane (r = l & 31, h = l >> 5) holds A[r][8h..8h+7],
//   B[8h..8h+7][r] and D[(reg & 3) + 8 (reg >> 2) + 4h][r].  A 32x32 result therefore has its COLUMN on the lane and
//   its rows in the 16 registers, which is exactly the B operand of a following product that sums over its rows.
//   With the shared operand (keys, values, weights) as A and the wavefront's own 32 points as the columns, every product
//   of the block takes the previous accumulator as its B operand straight from registers:
//
//     S^T = K q^T   ->  P^T = softmax over keys (registers + one half-lane exchange)
//     A^T = V^T P^T ->  t^T = x^T - A^T  ->  z^T = Wo t^T  ->  r^T = x^T + relu(z^T + bo)
//
//   No score matrix, no LDS transposes; a wavefront owns 32 points end to end and never talks to another one.
//
// Split precision: every fp32 operand is x = x1 + x2 + x3 (three bf16 "planes"); a product is the six MFMAs of
// magnitude >= 2^-16 (fp32-GEMM accuracy, see gemm.hip).  Shared operands are split ONCE by their producer and kept in
// memory as plane images in MFMA-fragment order; the per-wavefront operand is split in registers as it is consumed.
//
// Plane images (bf16, 16-byte chunks), ONE per operand (round 4: the separate transposed-read images are gone):
//   Rp  image of M[rows][K]: [k-step][plane][row tile][lane][8], the chunk of lane (r, h) holding
//        M[32 rt + r][16 ks + perm(h, j)], perm(h, j) = 8 (j >> 2) + 4h + (j & 3): the order in which an accumulator's
//        registers 8s..8s+7 come out as a B fragment.  Consumed three ways:
//        - k = column index, operand through LDS: plain slabs of one k-step, ds_read_b128;
//        - k = column index, the wavefront's own rows: straight from global by the lane that owns the row (B operand);
//        - k = ROW index (the operand is M^T): the same bytes fetched in the T-use cut of pzn_mfma.h (the LDS-DMA's
//          per-lane source address does the re-arrangement) and read with ds_read_b64_tr_b16 (hardware transpose).
//
// The shared operand streams through LDS in 24 KB slabs (36 KB in the projection kernel) by LDS-DMA, three slots,
// one barrier per slab.  One wavefront per SIMD (the chained accumulators need ~400 registers).
#include <stdio.h>

#include <type_traits>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

#include "pzn_mfma.h"

#ifdef ATTN_STAMPS
// diagnostic build only (tools/attn_stamps.py): cycle stamps of wavefront 0 of two workgroups per kernel
__device__ long long g_stamps[4][2][64];
#define STAMPK(K, i)                                                                                   \
  do {                                                                                                 \
    if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 77))                                      \
      g_stamps[K][blockIdx.x ? 1 : 0][i] = (long long)__builtin_amdgcn_s_memtime();                     \
  } while (0)
#else
#define STAMPK(K, i) do { } while (0)
#endif
#define STAMP(i) STAMPK(1, i)

constexpr int L = 256, E = 256, DK = 64;
constexpr int QK_IMG = 4 * 3 * 8 * 1024;  // Rp image of a [256][64] operand, bytes per cloud
constexpr int V_IMG = 16 * 3 * 8 * 1024;  // Rp image of a [256][256] operand

// ---- tile <-> memory through a per-wavefront LDS staging buffer --------------------------------------------------
// A wavefront's result tile has the point on the lane and the features in registers, so a direct store puts every
// lane in a different row: 32 partial cache lines per instruction (measured: ~300 cycles per store instruction, a third
// of the forward kernel).  Instead the tile goes through LDS: written in register layout (rows padded to 132 dwords:
// conflict-free both ways), read back row-contiguous, stored / loaded with 1 KB per instruction.  One pass moves 32 rows
// x up to 128 fp32 features (or 256 bf16); the buffer belongs to the wavefront, so only wave-scope syncs are needed.
constexpr int STG_LD = 132;                        // dwords per staged row
constexpr int STG_BYTES = 32 * STG_LD * 4;         // 16,896 per wavefront

// registers -> staging: tiles ft0 .. ft0 + NFT - 1 (NFT <= 4) of x at columns 32 (ft - ft0); gate != NULL: element i of
// tile ft is replaced by zero unless bit (ft & 1) * 16 + i of gate[ft >> 1] is set (ReLU mask of the forward pass)
struct Gate4 {
  uint32_t w[4];
};
template <int NFT, class G = NoGate>
__device__ __forceinline__ void stage_put(float* stg, const floatx16* x, int ft0, int lane, const G& gate = G()) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = x[ft0 + ft][4 * g + e];
        if constexpr (std::is_same<G, Gate4>::value) {
          if (!((gate.w[(ft0 + ft) >> 1] >> (((ft0 + ft) & 1) * 16 + 4 * g + e)) & 1u)) v[e] = 0.f;
        }
      }
      *reinterpret_cast<float4*>(stg + r * STG_LD + 32 * ft + 8 * g + 4 * h) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
template <int NFT, bool ADD = false>
__device__ __forceinline__ void stage_get(const float* stg, floatx16* x, int ft0, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(stg + r * STG_LD + 32 * ft + 8 * g + 4 * h);
      if (ADD) {
        x[ft0 + ft][4 * g] += v.x, x[ft0 + ft][4 * g + 1] += v.y, x[ft0 + ft][4 * g + 2] += v.z, x[ft0 + ft][4 * g + 3] += v.w;
      } else {
        x[ft0 + ft][4 * g] = v.x, x[ft0 + ft][4 * g + 1] = v.y, x[ft0 + ft][4 * g + 2] = v.z, x[ft0 + ft][4 * g + 3] = v.w;
      }
    }
}

// fp32 rows [32 rows of this wavefront][ld]: FT feature tiles starting at column 0; row0 = the wavefront's first row.
// MODE 0: store; 1: out = old + scale * tile (read-modify-write; rows are wave-private)
template <int FT, int MODE = 0, class G = NoGate>
__device__ __forceinline__ void store_rows(float* base, long row0, int ld, const floatx16* x, float* stg, int lane,
                                           float scale = 1.f, const G& gate = G()) {
  constexpr int PASS = FT >= 4 ? 4 : FT;            // tiles per pass
  constexpr int LPR = PASS * 8;                     // lanes per row (16 bytes each)
  constexpr int RPI = 64 / LPR;                     // rows per instruction
#pragma unroll
  for (int ft0 = 0; ft0 < FT; ft0 += PASS) {
    stage_put<PASS, G>(stg, x, ft0, lane, gate);
    pzn::wave_lds_sync();
    float* g0 = base + row0 * ld + 32 * ft0 + 4 * (lane % LPR);
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int rr = i * RPI + lane / LPR;
      float4 v = *reinterpret_cast<const float4*>(stg + rr * STG_LD + 4 * (lane % LPR));
      float4* dst = reinterpret_cast<float4*>(g0 + (long)rr * ld);
      if (MODE == 1) {
        const float4 o = *dst;
        v = make_float4(o.x + scale * v.x, o.y + scale * v.y, o.z + scale * v.z, o.w + scale * v.w);
      } else if (scale != 1.f) {
        v = make_float4(scale * v.x, scale * v.y, scale * v.z, scale * v.w);
      }
      *dst = v;
    }
    pzn::wave_lds_sync();
  }
}
// ADD: x += the rows (a second addend of the same tile)
template <int FT, bool ADD = false>
__device__ __forceinline__ void load_rows(const float* base, long row0, int ld, floatx16* x, float* stg, int lane) {
  constexpr int PASS = FT >= 4 ? 4 : FT;
  constexpr int LPR = PASS * 8, RPI = 64 / LPR;
#pragma unroll
  for (int ft0 = 0; ft0 < FT; ft0 += PASS) {
    const float* g0 = base + row0 * ld + 32 * ft0 + 4 * (lane % LPR);
    float4 v[32 / RPI];
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) v[i] = *reinterpret_cast<const float4*>(g0 + (long)(i * RPI + lane / LPR) * ld);
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i)
      *reinterpret_cast<float4*>(stg + (i * RPI + lane / LPR) * STG_LD + 4 * (lane % LPR)) = v[i];
    pzn::wave_lds_sync();
    stage_get<PASS, ADD>(stg, x, ft0, lane);
    pzn::wave_lds_sync();
  }
}

// fp32 tensors that only travel between these kernels (u, dq: query-side -> key-side backward) skip the staging: the
// "tile image" of M[rows][32 FT] keeps a tile in register order, [row tile][feature tile][register group g][lane][4],
// so a lane's four registers 4g..4g+3 are one 16-byte piece and every instruction moves 1 KB contiguously.
template <int FT>
__device__ __forceinline__ void store_tiles(float* img, long tile0, int lane, const floatx16* x) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(img + (((tile0 * FT + ft) * 4 + g) * 64 + lane) * 4) =
          make_float4(x[ft][4 * g], x[ft][4 * g + 1], x[ft][4 * g + 2], x[ft][4 * g + 3]);
}
template <int FT>
__device__ __forceinline__ void load_tiles(const float* img, long tile0, int lane, floatx16* x) {
  float4 v[FT * 4];
#pragma unroll
  for (int i = 0; i < FT * 4; ++i) v[i] = *reinterpret_cast<const float4*>(img + ((tile0 * FT * 4 + i) * 64 + lane) * 4);
#pragma unroll
  for (int i = 0; i < FT * 4; ++i) {
    x[i >> 2][4 * (i & 3)] = v[i].x, x[i >> 2][4 * (i & 3) + 1] = v[i].y;
    x[i >> 2][4 * (i & 3) + 2] = v[i].z, x[i >> 2][4 * (i & 3) + 3] = v[i].w;
  }
}

// Rp image of M[256 rows][16 KS features]: tile ft of the accumulators = k-steps 2 ft, 2 ft + 1; the lane's registers
// 8s..8s+7 are exactly its own chunk.  img = this cloud's image, rt = row tile of the wavefront.
template <int FT, int NPL, bool NEG = false>
__device__ __forceinline__ void store_rp(unsigned char* img, int rt, int lane, const floatx16* x) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 b[3];
      make_bn<NPL, NEG>(x[ft], s, b);
      const int ks = 2 * ft + s;
#pragma unroll
      for (int p = 0; p < NPL; ++p) *reinterpret_cast<bf16x8*>(img + (((ks * 3 + p) * 8 + rt) * 64 + lane) * 16) = b[p];
      __builtin_amdgcn_sched_barrier(0);   // (keeps the scheduler from splitting all 16 fragments before the first store)
    }
}

// ================================================================================================================
// weight planes: Rp image of a logical matrix A[row][k] = src[row * rs + k * cs], rows = 32 nrt, K = 16 nks,
// written at row tile offset rt0 of an image with RT row tiles and k-step offset ks0
struct PackJob {
  const float* src;
  long rs, cs;
  int nrt, nks, RT, rt0, ks0;
  unsigned char* dst;
};
struct PackArgs {
  PackJob job[32];
  int njobs;
};

__global__ __launch_bounds__(256) void pack_rp_kernel(PackArgs a) {
  const PackJob& J = a.job[blockIdx.y];
  const int total = J.nks * J.nrt * 64;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    const int lane = c & 63, rt = (c >> 6) % J.nrt, ks = (c >> 6) / J.nrt;
    const int r = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
      v[j] = J.src[(long)(32 * rt + r) * J.rs + (long)k * J.cs];
    }
    bf16x8 b[3];
    split8(v, b);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<bf16x8*>(J.dst + ((((long)(J.ks0 + ks) * 3 + p) * J.RT + J.rt0 + rt) * 64 + lane) * 16) = b[p];
  }
}

// byte offsets inside one layer's weight-plane buffer
constexpr size_t W_QKV = 0;                                   // rows n (q | k | v = 384), k = c: 16 x 36864
constexpr size_t W_O = W_QKV + 16 * 36864;                    // rows o, k = c: 16 x 24576
constexpr size_t W_OT = W_O + 16 * SLAB;                      // rows c, k = o
constexpr size_t W_QT = W_OT + 16 * SLAB;                     // rows c, k = d: 4 x 24576
constexpr size_t W_KVT = W_QT + 4 * SLAB;                     // rows c, k = d (Wk) then c' (Wv): 20 x 24576
constexpr size_t W_BYTES = W_KVT + 20 * SLAB;

// ================================================================================================================
// projection: q, k, v = x W^T + b for the wavefront's 32 points, written as plane images (Rp and T each)
struct ProjProb {
  const float* x;                 // [B*L, E]
  const unsigned char* w;         // the layer's weight planes
  const float *bq, *bk, *bv;
  unsigned char *qrp, *krp, *vrp;
};
struct ProjArgs {
  ProjProb p[2];
  int nb;   // workgroups per problem = 2 B
};

// acc tile ft <- bias[32 ft + 8 g + 4 h + e] in register 4 g + e: the row of a transposed result is the output feature,
// so a bias is the INITIAL accumulator (all loads are independent and issued together)
template <int FT>
__device__ __forceinline__ void bias_tiles(floatx16* acc, const float* bias, int h) {
  float4 v[FT * 4];
#pragma unroll
  for (int i = 0; i < FT * 4; ++i) v[i] = *reinterpret_cast<const float4*>(bias + 8 * i + 4 * h);
#pragma unroll
  for (int i = 0; i < FT * 4; ++i) {
    acc[i >> 2][4 * (i & 3)] = v[i].x, acc[i >> 2][4 * (i & 3) + 1] = v[i].y;
    acc[i >> 2][4 * (i & 3) + 2] = v[i].z, acc[i >> 2][4 * (i & 3) + 3] = v[i].w;
  }
}

#define ZERO_TILES(A, N)            \
  _Pragma("unroll") for (int i_ = 0; i_ < (N); ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) A[i_][j_] = 0.f

template <int NPL>
__global__ __launch_bounds__(NT, 1) void attn_proj_kernel(ProjArgs a) {
  constexpr int WSLAB = 36864;
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * WSLAB];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const int lb = logical_block(blockIdx.x, gridDim.x);
  const ProjProb& P = a.p[lb / a.nb];
  const int cb = lb % a.nb, cloud = cb >> 1, rt = (cb & 1) * 4 + wave;
  const Ring ring{lds, wave, lane, WSLAB};
  const unsigned char* wsrc = P.w + W_QKV;
  const long row0 = (long)cloud * L + 32 * rt;
  float* stg = reinterpret_cast<float*>(lds + wave * STG_BYTES);   // staging aliases the ring: used before / after it

  STAMPK(0, 0);
  floatx16 X[8];
  load_rows<8>(P.x, row0, E, X, stg, lane);
  __syncthreads();
  STAMPK(0, 1);
#pragma unroll
  for (int i = 0; i < 9; ++i) ring.issue1<NPL, 12>(wsrc, 0, i);
#pragma unroll
  for (int i = 0; i < 9; ++i) ring.issue1<NPL, 12>(wsrc + WSLAB, 1, i);
  floatx16 acc[12];
  bias_tiles<2>(acc, P.bq, h);
  bias_tiles<2>(acc + 2, P.bk, h);
  bias_tiles<8>(acc + 4, P.bv, h);
  bf16x8 b[2][3];
  make_bn<NPL>(X[0], 0, b[0]);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    step_sync(ks < 15 ? (NPL == 3 ? 9 : 3) : 0);
    BNextN<NPL> bn;
    auto fill = [&](int t) {
      if (t < 5) {
        if (ks + 2 < 16) {
          ring.issue1<NPL, 12>(wsrc + (ks + 2) * WSLAB, (ks + 2) % 3, 2 * t);
          if (2 * t + 1 < 9) ring.issue1<NPL, 12>(wsrc + (ks + 2) * WSLAB, (ks + 2) % 3, 2 * t + 1);
        }
      } else if (t < 9 && ks < 15) {
        bn.pair(X[(ks + 1) >> 1], (ks + 1) & 1, t - 5);
      }
    };
    kstep_rp_n<12, NPL>(acc, ring.lane_addr(ks % 3), b[ks & 1], fill);
    if (ks < 15) bn.get(b[(ks + 1) & 1]);
  }
  const floatx16 *q = acc, *k = acc + 2, *v = acc + 4;
  STAMPK(0, 2);
  store_rp<2, NPL>(P.qrp + (size_t)cloud * QK_IMG, rt, lane, q);
  store_rp<2, NPL>(P.krp + (size_t)cloud * QK_IMG, rt, lane, k);
  STAMPK(0, 3);
  store_rp<8, NPL>(P.vrp + (size_t)cloud * V_IMG, rt, lane, v);
  STAMPK(0, 4);
}

// ================================================================================================================
// softmax over the keys of S^T (8 tiles x 16 registers + the other half-lane): S <- P, returns ln sum exp + max
__device__ __forceinline__ float softmax_regs(floatx16 (&S)[8]) {
  float m = S[0][0];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, S[t][i]);
  m = fmaxf(m, xor32(m));
  const float c = 0.125f * LOG2E;       // logits / sqrt(dk), dk = 64 (model5_b.py:70)
  const float mc = m * c;
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float e = __builtin_amdgcn_exp2f(S[t][i] * c - mc);
      S[t][i] = e;
      sum += e;
    }
  sum += xor32(sum);
  const float inv = 1.f / sum;
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) S[t][i] *= inv;
  return m * 0.125f + __logf(sum);
}

// the wavefront's query fragments (B operand of S^T = K q^T): 4 k-steps x 3 planes, straight from the Rp image
template <int NPL>
__device__ __forceinline__ void load_own_frags(const unsigned char* img, int rt, int lane, bf16x8 (&f)[4][3]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int p = 0; p < NPL; ++p) f[ks][p] = *reinterpret_cast<const bf16x8*>(img + (((ks * 3 + p) * 8 + rt) * 64 + lane) * 16);
}

// ================================================================================================================
// forward of one block for the wavefront's 32 points
struct FwdProb {
  const float* x;                          // [B*L, E] block input
  const unsigned char *qrp, *krp, *vrp;    // images of this layer's q, k, v
  const unsigned char* w;                  // weight planes
  const float* bo;
  float* r;        // [B*L, E] block output
  float* t;        // [B*L, E] x - attn v (the out projection's input: its weight gradient needs it)
  uint32_t* mask;  // [B*L, 8] bits of (Wo t + bo > 0)
  float* map;      // [B, L, L] mean attention map (may be NULL)
  float* lse;      // [B*L]
};
struct FwdArgs {
  FwdProb p[2];
  int nb;
  int map_accumulate;   // 0: map = scale * P, 1: map += scale * P
  float map_scale;
};

template <int NPL>
__global__ __launch_bounds__(NT, 1) void attn_fwd_kernel(FwdArgs a) {
  constexpr int DPW = NPL == 3 ? 6 : 2;   // DMA pieces per wavefront and slab (NPL = 1: plane 0 only)
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SLAB + 4 * STG_BYTES];   // ring | staging per wavefront
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const int lb = logical_block(blockIdx.x, gridDim.x);
  const FwdProb& P = a.p[lb / a.nb];
  const int cb = lb % a.nb, cloud = cb >> 1, rt = (cb & 1) * 4 + wave;
  const long row = (long)cloud * L + 32 * rt + (lane & 31);
  const unsigned char* krp = P.krp + (size_t)cloud * QK_IMG;
  const unsigned char* vrp = P.vrp + (size_t)cloud * V_IMG;
  const unsigned char* wo = P.w + W_O;
  const Ring ring{lds, wave, lane, SLAB};
  const long row0 = (long)cloud * L + 32 * rt;
  float* stg = reinterpret_cast<float*>(lds + 3 * SLAB + wave * STG_BYTES);
  const uint32_t tsrc = tr_src_lane_off(lane);
  // slab sequence: K 0..3 | V^T (T use of the v image, k-step = 16 keys) 0..15 | Wo 0..15
  constexpr int NS = 36;
  auto issue1 = [&](int c, int i) {   // piece i of this wavefront of slab c -> slot c % 3
    if (c < 4)
      ring.issue1<NPL>(krp + c * SLAB, c % 3, i);
    else if (c < 20)
      ring.issue1_t<256, NPL>(vrp, tsrc, c - 4, c % 3, i);
    else if (c < NS)
      ring.issue1<NPL>(wo + (c - 20) * SLAB, c % 3, i);
  };

  STAMP(0);
  bf16x8 qf[4][3];
  load_own_frags<NPL>(P.qrp + (size_t)cloud * QK_IMG, rt, lane, qf);
#pragma unroll
  for (int i = 0; i < 6; ++i) issue1(0, i);
#pragma unroll
  for (int i = 0; i < 6; ++i) issue1(1, i);
  floatx16 S[8];
  ZERO_TILES(S, 8);
  // ---- S^T = K q^T (4 k-steps over d)
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    step_sync(DPW);
    auto fill = [&](int t) {
      if (t < 3) {
        issue1(c + 2, 2 * t);
        issue1(c + 2, 2 * t + 1);
      }
    };
    kstep_rp_n<8, NPL>(S, ring.lane_addr(c % 3), qf[c], fill);
  }
  STAMP(1);
  const float lse = softmax_regs(S);
  if (h == 0) P.lse[row] = lse;
  STAMP(2);
  if (P.map) {   // mean of the four blocks' maps (model5_b.py:468-469): this wavefront owns its 32 rows
    if (a.map_accumulate)
      store_rows<8, 1>(P.map, row0, L, S, stg, lane, a.map_scale);
    else
      store_rows<8, 0>(P.map, row0, L, S, stg, lane, a.map_scale);
  }
  STAMP(3);
  // ---- A^T = V^T P^T (16 k-steps over the keys)
  floatx16 O[8];
  ZERO_TILES(O, 8);
  bf16x8 b[2][3];
  make_bn<NPL>(S[0], 0, b[0]);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int c = 4 + ks;
    if (ks >= 4 && ks < 8) STAMP(16 + 2 * (ks - 4));
    step_sync(DPW);
    if (ks >= 4 && ks < 8) STAMP(17 + 2 * (ks - 4));
    BNextN<NPL> bn;
    auto fill = [&](int t) {
      if (t < 3) {
        issue1(c + 2, 2 * t);
        issue1(c + 2, 2 * t + 1);
      } else if (t < 7 && ks < 15) {
        bn.pair(S[(ks + 1) >> 1], (ks + 1) & 1, t - 3);
      }
    };
    kstep_tr<8, 256, 0, NPL>(O, tr_lane_addr(ring.slot_addr(c % 3), lane), b[ks & 1], fill);
    if (ks < 15) bn.get(b[(ks + 1) & 1]);
  }
  STAMP(4);
  // ---- t^T = x^T - A^T (x stays in registers for the residual at the end)
  floatx16 X[8];
  load_rows<8>(P.x, row0, E, X, stg, lane);
#pragma unroll
  for (int ft = 0; ft < 8; ++ft)
#pragma unroll
    for (int i = 0; i < 16; ++i) O[ft][i] = X[ft][i] - O[ft][i];
  STAMP(5);
  // ---- z^T = Wo t^T (16 k-steps over c)
  step_sync(DPW);
  store_rows<8>(P.t, row0, E, O, stg, lane);   // (behind the barrier, ahead of this step's DMA: see Ring)
  floatx16 Z[8];
  bias_tiles<8>(Z, P.bo, h);
  make_bn<NPL>(O[0], 0, b[0]);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int c = 20 + ks;
    if (ks >= 4 && ks < 8) STAMP(8 + 2 * (ks - 4));
    if (ks > 0) step_sync(c + 1 < NS ? DPW : 0);
    if (ks >= 4 && ks < 8) STAMP(9 + 2 * (ks - 4));
    BNextN<NPL> bn;
    auto fill = [&](int t) {
      if (t < 3) {
        issue1(c + 2, 2 * t);
        issue1(c + 2, 2 * t + 1);
      } else if (t < 7 && ks < 15) {
        bn.pair(O[(ks + 1) >> 1], (ks + 1) & 1, t - 3);
      }
    };
    kstep_rp_n<8, NPL>(Z, ring.lane_addr(c % 3), b[ks & 1], fill);
    if (ks < 15) bn.get(b[(ks + 1) & 1]);
  }
  STAMP(6);
  // ---- r^T = x^T + relu(z^T + bo); gate bits for the backward
  {
    uint32_t bits[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ft = 0; ft < 8; ++ft)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float z = Z[ft][i];
        const bool on = z > 0.f;
        bits[ft >> 1] |= (on ? 1u : 0u) << ((ft & 1) * 16 + i);
        Z[ft][i] = X[ft][i] + (on ? z : 0.f);
      }
    *reinterpret_cast<uint4*>(P.mask + (row * 2 + h) * 4) = make_uint4(bits[0], bits[1], bits[2], bits[3]);
  }
  store_rows<8>(P.r, row0, E, Z, stg, lane);
  STAMP(7);
}

// ================================================================================================================
// backward, query side: the wavefront's 32 points as QUERIES.
//   dz = dr . gate;  dt^T = Wo^T dz^T;  da = -dt (image for the key-side pass);  dP^T = V da^T;  P^T recomputed;
//   delta = sum_key P dP;  dS^T = P^T (dP^T - delta) / 8;  dq^T = K^T dS^T;  u = dr + dt (partial dx: the key-side pass
//   adds dq Wq with its own terms).
// Round 4: dr is read once (the gate is applied where dz is consumed: in the split of the B fragments and in the store
// of dz, so dr is still there for u = dr + dt), u is written once, and never more than two accumulator sets are live.
struct BwdQProb {
  const float* dr;       // gradient of the block output: rows of ld_dr floats (a column slice of a wider matrix is fine)
  const float* dr2;      // optional second addend of that gradient (NULL: none), rows of ld_dr2 floats
  int ld_dr, ld_dr2;
  const uint32_t* mask;
  const unsigned char *qrp, *krp, *vrp;
  const unsigned char* w;
  float* dz;             // [B*L, E]
  float* u;              // [B*L, E] as a tile image (store_tiles): read by the key-side pass only
  float* dq;             // [B*L, DK] rows (weight gradients)
  float* dqt;            // [B*L, DK] as a tile image (key-side pass)
  unsigned char* darp;   // image of da
  float* delta;          // [B*L]
};
struct BwdQArgs {
  BwdQProb p[2];
  int nb;
};

template <int NPL>
__global__ __launch_bounds__(NT, 1) void attn_bwd_q_kernel(BwdQArgs a) {
  constexpr int DPW = NPL == 3 ? 6 : 2;   // DMA pieces per wavefront and slab (NPL = 1: plane 0 only)
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SLAB + 4 * STG_BYTES];   // ring | staging per wavefront
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lb = logical_block(blockIdx.x, gridDim.x);
  const BwdQProb& P = a.p[lb / a.nb];
  const int cb = lb % a.nb, cloud = cb >> 1, rt = (cb & 1) * 4 + wave;
  const unsigned char* krp = P.krp + (size_t)cloud * QK_IMG;
  const unsigned char* vrp = P.vrp + (size_t)cloud * V_IMG;
  const unsigned char* wot = P.w + W_OT;
  const long row0 = (long)cloud * L + 32 * rt;
  float* stg = reinterpret_cast<float*>(lds + 3 * SLAB + wave * STG_BYTES);
  constexpr int NS = 40;
  STAMPK(2, 0);
  // ---- prologue: dr (+ dr2) -> registers, gate bits, dz -> memory.  What derives from the lane id here does not outlive
  // the prologue: the loops below take the lane id from the hardware again (fresh_lane) instead of keeping it, and its
  // derived addresses, alive through the register-hungry transposes (the allocator spilled them otherwise).
  floatx16 S[8];     // dr^T, then u^T = dr^T + dt^T, later the scores
  Gate4 gate;
  {
    const int lane0 = tid & 63;
    const Ring ring0{lds, wave, lane0, SLAB};
#pragma unroll
    for (int i = 0; i < 6; ++i) ring0.issue1<NPL>(wot, 0, i);
#pragma unroll
    for (int i = 0; i < 6; ++i) ring0.issue1<NPL>(wot + SLAB, 1, i);
    load_rows<8>(P.dr, row0, P.ld_dr, S, stg, lane0);
    if (P.dr2) load_rows<8, true>(P.dr2, row0, P.ld_dr2, S, stg, lane0);
    const long row = row0 + (lane0 & 31);
    const uint4 mb = *reinterpret_cast<const uint4*>(P.mask + (row * 2 + (lane0 >> 5)) * 4);
    gate.w[0] = mb.x, gate.w[1] = mb.y, gate.w[2] = mb.z, gate.w[3] = mb.w;
    STAMPK(2, 1);
    store_rows<8, 0, Gate4>(P.dz, row0, E, S, stg, lane0, 1.f, gate);
    STAMPK(2, 2);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int lane = fresh_lane();
  const Ring ring{lds, wave, lane, SLAB};
  // slab sequence: Wo^T 0..15 | V 0..15 | K 0..3 | K^T (T use of the k image, 4 k-steps of 16 keys each) 0..3
  auto issue1 = [&](int c, int i) {
    if (c < 16)
      ring.issue1<NPL>(wot + c * SLAB, c % 3, i);
    else if (c < 32)
      ring.issue1<NPL>(vrp + (c - 16) * SLAB, c % 3, i);
    else if (c < 36)
      ring.issue1<NPL>(krp + (c - 32) * SLAB, c % 3, i);
    else if (c < NS)   // (the lane's source offset is derived on the spot: 4 of 40 slabs, and one register less across the loops)
      ring.issue1_t<64, NPL>(krp, tr_src_lane_off(fresh_lane()), 4 * (c - 36), c % 3, i);
  };
  auto fill_dma = [&](int c, int t) {
    if (t < 3) {
      issue1(c + 2, 2 * t);
      issue1(c + 2, 2 * t + 1);
    }
  };

  // ---- dt^T = Wo^T dz^T (16 k-steps over o); the gate is applied as the fragments are split
  floatx16 DT[8];
  ZERO_TILES(DT, 8);
  bf16x8 b[2][3];
  Stash stash;
  {
    BNextN<NPL> b0;
#pragma unroll
    for (int j = 0; j < 4; ++j) b0.pair_gated(S[0], 0, j, gate.w[0], 0);
    b0.get(b[0]);
  }
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int c = ks;
    step_sync(ks == 0 ? 0 : DPW);   // (first step: the stores of dz stand between the two slabs and this wait)
    BNextN<NPL> bn;
    auto fill = [&](int t) {
      fill_dma(c, t);
      if (t >= 3 && t < 7 && ks < 15) {
        const int kn = ks + 1, ft = kn >> 1;
        bn.pair_gated(S[ft], kn & 1, t - 3, gate.w[ft >> 1], (ft & 1) * 16);
      }
    };
    kstep_rp_d<8, NPL>(DT, ring.lane_addr(c % 3), b[ks & 1], b[(ks + 1) & 1], stash, ks == 0, ks == 15, fill);
    if (ks < 15) bn.get(b[(ks + 1) & 1]);
  }
  // ---- DP^T = V dt^T = -dP^T (16 k-steps over c); the da image and u go to memory at the head of its first step
  STAMPK(2, 3);
  step_sync(DPW);
  STAMPK(2, 4);
#pragma unroll
  for (int ft = 0; ft < 8; ++ft) S[ft] += DT[ft];       // u = dr + dt
  store_tiles<8>(P.u, (long)cloud * 8 + rt, lane, S);
  __builtin_amdgcn_sched_barrier(0);   // (one accumulator set leaves before the next piece of work needs registers)
  STAMPK(2, 5);
  store_rp<8, NPL, true>(P.darp + (size_t)cloud * V_IMG, rt, lane, DT);
  __builtin_amdgcn_sched_barrier(0);
  STAMPK(2, 6);
  floatx16 DP[8];
  ZERO_TILES(DP, 8);
  make_bn<NPL>(DT[0], 0, b[0]);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int c = 16 + ks;
    if (ks > 0) step_sync(DPW);
    BNextN<NPL> bn;
    auto fill = [&](int t) {
      fill_dma(c, t);
      if (t >= 3 && t < 7 && ks < 15) bn.pair(DT[(ks + 1) >> 1], (ks + 1) & 1, t - 3);
    };
    kstep_rp_d<8, NPL>(DP, ring.lane_addr(c % 3), b[ks & 1], b[(ks + 1) & 1], stash, ks == 0, ks == 15, fill);
    if (ks < 15) bn.get(b[(ks + 1) & 1]);
  }
  // ---- S^T = K q^T, P^T (u is in memory: its registers hold the scores now)
  __builtin_amdgcn_sched_barrier(0);
  STAMPK(2, 9);
  // (what the tail derives from the lane id is derived where it is used, from a fresh read of the lane id, instead of
  //  occupying registers through the loops)
  floatx16(&S2)[8] = S;
  ZERO_TILES(S2, 8);
  bf16x8 qf[2][3];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c = 32 + ks;
#pragma unroll
    for (int p = 0; p < NPL; ++p)
      qf[ks & 1][p] = *reinterpret_cast<const bf16x8*>(P.qrp + (size_t)cloud * QK_IMG + (((ks * 3 + p) * 8 + rt) * 64 + fresh_lane()) * 16);
    step_sync(DPW + NPL);    // (the three fragment loads above are younger than the slab waited for)
    auto fill = [&](int t) { fill_dma(c, t); };
    kstep_rp_d<8, NPL>(S2, ring.lane_addr(c % 3), qf[ks & 1], qf[(ks + 1) & 1], stash, ks == 0, ks == 3, fill);
  }
  STAMPK(2, 10);
  softmax_regs(S2);
  {  // delta = sum P dP;  dS = P (dP - delta) / 8  with dP = -DP
    float d = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) d -= S2[t][i] * DP[t][i];
    d += xor32(d);
    const int l2 = fresh_lane();
    if (l2 < 32) P.delta[row0 + l2] = d;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) S2[t][i] = S2[t][i] * (-DP[t][i] - d) * 0.125f;
  }
  // ---- dq^T = K^T dS^T (16 k-steps over the keys, 4 per slab; rows = d)
  __builtin_amdgcn_sched_barrier(0);
  STAMPK(2, 11);
  floatx16 DQ[2];
  ZERO_TILES(DQ, 2);
  make_bn<NPL>(S2[0], 0, b[0]);
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int c = 36 + sl;
    step_sync(sl == 0 ? DPW + 1 : sl == 3 ? 0 : DPW);   // (sl == 0: the store of delta is younger than the slab as well)
#pragma unroll
    for (int i = 0; i < 6; ++i) issue1(c + 2, i);
    const uint32_t ta = tr_lane_addr(ring.slot_addr(c % 3), fresh_lane());
    static_for<0, 4>([&](auto iq) {
      constexpr int q4 = decltype(iq)::value;
      const int ks = 4 * sl + q4;
      BNextN<NPL> bn;
      auto fill = [&](int t) {
        if (ks < 15) {
          bn.pair(S2[(ks + 1) >> 1], (ks + 1) & 1, 2 * t);
          bn.pair(S2[(ks + 1) >> 1], (ks + 1) & 1, 2 * t + 1);
        }
      };
      kstep_tr<2, 64, q4 * 6144, NPL>(DQ, ta, b[ks & 1], fill);
      if (ks < 15) bn.get(b[(ks + 1) & 1]);
    });
  }
  STAMPK(2, 12);
  {
    const int l2 = fresh_lane();
    store_tiles<2>(P.dqt, (long)cloud * 8 + rt, l2, DQ);
    store_rows<2>(P.dq, row0, DK, DQ, stg, l2);
  }
  STAMPK(2, 13);
}

// ================================================================================================================
// backward, key side: the wavefront's 32 points as KEYS against all 256 queries of the cloud (round 4: whole-cloud
// accumulators as in the forward kernel instead of query-tile pairs: every slab feeds 48 MFMAs per wavefront).
//   S = q k^T (key on the lane, query in the registers), P = exp(S/8 - lse_q), dP = da v^T, dS = P (dP - delta_q) / 8,
//   dk^T = q^T dS, dv^T = da^T P (in this order: dS dies before the dv accumulators are born);
//   then dx = u + dq Wq + dk Wk + dv Wv for the wavefront's points (u = dr + dt and dq from the query-side pass)
struct BwdKProb {
  const unsigned char *qrp, *krp, *vrp, *darp;
  const unsigned char* w;
  const float *lse, *delta;
  const float* u;     // [B*L, E] tile image
  const float* dq;    // [B*L, DK] tile image
  float* dk;          // [B*L, DK]
  float* dv;          // [B*L, E]
  float* dx;          // [B*L, E]
};
struct BwdKArgs {
  BwdKProb p[2];
  int nb;
};

template <int NPL>
__global__ __launch_bounds__(NT, 1) void attn_bwd_k_kernel(BwdKArgs a) {
  constexpr int DPW = NPL == 3 ? 6 : 2;   // DMA pieces per wavefront and slab (NPL = 1: plane 0 only)
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SLAB + 2048 + 4 * STG_BYTES];   // ring | lse[256] | delta[256] | staging
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  const int lb = logical_block(blockIdx.x, gridDim.x);
  const BwdKProb& P = a.p[lb / a.nb];
  const int cb = lb % a.nb, cloud = cb >> 1, rt = (cb & 1) * 4 + wave;
  const unsigned char* qrp = P.qrp + (size_t)cloud * QK_IMG;
  const unsigned char* krp = P.krp + (size_t)cloud * QK_IMG;
  const unsigned char* vrp = P.vrp + (size_t)cloud * V_IMG;
  const unsigned char* darp = P.darp + (size_t)cloud * V_IMG;
  const unsigned char* wqkvt = P.w + W_QT;    // [Wq^T | Wk^T | Wv^T]: 4 + 4 + 16 k-step slabs, contiguous
  const Ring ring{lds, wave, lane, SLAB};
  const long row0 = (long)cloud * L + 32 * rt;
  float* stg = reinterpret_cast<float*>(lds + 3 * SLAB + 2048 + wave * STG_BYTES);
  float* row_consts = reinterpret_cast<float*>(lds + 3 * SLAB);
  const uint32_t tsrc = tr_src_lane_off(lane);
  STAMPK(3, 0);
  row_consts[tid] = P.lse[(long)cloud * L + tid];
  row_consts[256 + tid] = P.delta[(long)cloud * L + tid];
  // slab sequence: q 0..3 (k = d) | da 0..15 (k = c) | q^T (T use, 4 k-steps of 16 queries each) 0..3 |
  //                da^T (T use, k-step = 16 queries) 0..15 | [Wq^T | Wk^T | Wv^T] 0..23
  constexpr int NS = 64;
  auto issue1 = [&](int c, int i) {
    if (c < 4)
      ring.issue1<NPL>(qrp + c * SLAB, c % 3, i);
    else if (c < 20)
      ring.issue1<NPL>(darp + (c - 4) * SLAB, c % 3, i);
    else if (c < 24)
      ring.issue1_t<64, NPL>(qrp, tsrc, 4 * (c - 20), c % 3, i);
    else if (c < 40)
      ring.issue1_t<256, NPL>(darp, tsrc, c - 24, c % 3, i);
    else if (c < NS)
      ring.issue1<NPL>(wqkvt + (c - 40) * SLAB, c % 3, i);
  };
  auto fill_dma = [&](int c, int t) {
    if (t < 3) {
      issue1(c + 2, 2 * t);
      issue1(c + 2, 2 * t + 1);
    }
  };
  // the wavefront's own rows of an Rp image as the B operand of k-step ks: three chunks straight from global
  auto own_frag = [&](const unsigned char* img, int ks, bf16x8 (&f)[3]) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) f[p] = *reinterpret_cast<const bf16x8*>(img + (((ks * 3 + p) * 8 + rt) * 64 + lane) * 16);
  };
#pragma unroll
  for (int i = 0; i < 6; ++i) issue1(0, i);
#pragma unroll
  for (int i = 0; i < 6; ++i) issue1(1, i);
  bf16x8 of[3][3];   // own fragments of global step g (S loop: g = ks, dP loop: g = 4 + ks) in of[g % 3]: this step's, the
                     // previous one's (the deferred tile still multiplies by it) and the next one's (in flight)
  Stash stash;
  own_frag(krp, 0, of[0]);
  __syncthreads();   // row constants in LDS (drains the first two slabs once)
  STAMPK(3, 1);

  // ---- S = q k^T (4 k-steps over d): rows = the cloud's 256 queries, B = this wavefront's key fragments
  floatx16 S[8];
  ZERO_TILES(S, 8);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    const int c = ks;
    if (ks > 0) step_sync(DPW + NPL);   // (the fragment loads of the next step are younger than the slab waited for)
    if (ks < 3)
      own_frag(krp, ks + 1, of[(ks + 1) % 3]);
    else
      own_frag(vrp, 0, of[(ks + 1) % 3]);
    auto fill = [&](int t) { fill_dma(c, t); };
    kstep_rp_d<8, NPL>(S, ring.lane_addr(c % 3), of[ks % 3], of[(ks + 2) % 3], stash, ks == 0, ks == 3, fill);
  }
  STAMPK(3, 2);
  // P = exp(S / 8 - lse_q): the query is the register's row
  {
    const float c = 0.125f * LOG2E;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 ls = *reinterpret_cast<const float4*>(row_consts + 32 * t + 8 * g + 4 * h);
        const float lv[4] = {ls.x, ls.y, ls.z, ls.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) S[t][4 * g + e] = __builtin_amdgcn_exp2f(S[t][4 * g + e] * c - lv[e] * LOG2E);
      }
  }
  __builtin_amdgcn_sched_barrier(0);
  STAMPK(3, 3);
  // ---- dP = da v^T (16 k-steps over c); B = this wavefront's value fragments, one step ahead
  floatx16 DP[8];
  ZERO_TILES(DP, 8);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int c = 4 + ks;
    step_sync(DPW + NPL);
    if (ks < 15) own_frag(vrp, ks + 1, of[(c + 1) % 3]);
    auto fill = [&](int t) { fill_dma(c, t); };
    kstep_rp_d<8, NPL>(DP, ring.lane_addr(c % 3), of[c % 3], of[(c + 2) % 3], stash, ks == 0, ks == 15, fill);
  }
  STAMPK(3, 4);
  // dS = P (dP - delta_q) / 8
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 de = *reinterpret_cast<const float4*>(row_consts + 256 + 32 * t + 8 * g + 4 * h);
      const float dv4[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) DP[t][4 * g + e] = S[t][4 * g + e] * (DP[t][4 * g + e] - dv4[e]) * 0.125f;
    }
  // ---- dk^T = q^T dS (16 k-steps of 16 queries, four per slab; rows = d)
  __builtin_amdgcn_sched_barrier(0);
  floatx16 DKt[2];
  ZERO_TILES(DKt, 2);
  bf16x8 b[2][3];
  make_bn<NPL>(DP[0], 0, b[0]);
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    const int c = 20 + sl;
    step_sync(DPW);
#pragma unroll
    for (int i = 0; i < 6; ++i) issue1(c + 2, i);
    const uint32_t ta = tr_lane_addr(ring.slot_addr(c % 3), lane);
    static_for<0, 4>([&](auto iq) {
      constexpr int q4 = decltype(iq)::value;
      const int ks = 4 * sl + q4;
      BNextN<NPL> bn;
      auto fill = [&](int t) {
        if (ks < 15) {
          bn.pair(DP[(ks + 1) >> 1], (ks + 1) & 1, 2 * t);
          bn.pair(DP[(ks + 1) >> 1], (ks + 1) & 1, 2 * t + 1);
        }
      };
      kstep_tr<2, 64, q4 * 6144, NPL>(DKt, ta, b[ks & 1], fill);
      if (ks < 15) bn.get(b[(ks + 1) & 1]);
    });
  }
  STAMPK(3, 5);
  // ---- dv^T = da^T P (16 k-steps of 16 queries, rows = c); dS is dead: its registers are the accumulators
  __builtin_amdgcn_sched_barrier(0);
  floatx16(&DV)[8] = DP;
  ZERO_TILES(DV, 8);
  make_bn<NPL>(S[0], 0, b[0]);
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
    const int c = 24 + ks;
    step_sync(DPW);
    BNextN<NPL> bn;
    auto fill = [&](int t) {
      fill_dma(c, t);
      if (t >= 3 && t < 7 && ks < 15) bn.pair(S[(ks + 1) >> 1], (ks + 1) & 1, t - 3);
    };
    kstep_tr<8, 256, 0, NPL>(DV, tr_lane_addr(ring.slot_addr(c % 3), lane), b[ks & 1], fill);
    if (ks < 15) bn.get(b[(ks + 1) & 1]);
  }
  // ---- dx^T = u^T + Wq^T dq^T + Wk^T dk^T + Wv^T dv^T (4 + 4 + 16 k-steps); P is dead: its registers take u
  STAMPK(3, 6);
  __builtin_amdgcn_sched_barrier(0);
  floatx16(&DX)[8] = S;
  floatx16 DQ[2];
  step_sync(DPW);
  load_tiles<2>(P.dq, (long)cloud * 8 + rt, lane, DQ);    // (in flight while dk and dv leave through the staging buffer)
  load_tiles<8>(P.u, (long)cloud * 8 + rt, lane, DX);
  store_rows<2>(P.dk, row0, DK, DKt, stg, lane);
  store_rows<8>(P.dv, row0, E, DV, stg, lane);
  bf16x8 bt[2][3];
  make_bn<NPL>(DQ[0], 0, bt[0]);
  STAMPK(3, 7);
#pragma unroll
  for (int ks = 0; ks < 24; ++ks) {
    const int c = 40 + ks;
    if (ks > 0) step_sync(ks < 23 ? DPW : 0);
    BNextN<NPL> bn;
    auto fill = [&](int t) {
      if (c + 2 < NS) fill_dma(c, t);
      if (t >= 3 && t < 7 && ks < 23) {
        const int kn = ks + 1;
        if (kn < 4)
          bn.pair(DQ[kn >> 1], kn & 1, t - 3);
        else if (kn < 8)
          bn.pair(DKt[(kn - 4) >> 1], kn & 1, t - 3);
        else
          bn.pair(DV[(kn - 8) >> 1], kn & 1, t - 3);
      }
    };
    kstep_rp_n<8, NPL>(DX, ring.lane_addr(c % 3), bt[ks & 1], fill);
    if (ks < 23) bn.get(bt[(ks + 1) & 1]);
  }
  STAMPK(3, 8);
  store_rows<8>(P.dx, row0, E, DX, stg, lane);
  STAMPK(3, 9);
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// ================================================================================================================
// C ABI.  All entry points take up to two independent problems (the two encoders of predict5, model5_b.py:700-707)
// so that one launch fills the chip: 2 B workgroups of four wavefronts per problem, one wavefront per SIMD.
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

// tile shape of the chained kernels: 32 (four wavefronts per workgroup, v_mfma_f32_32x32x16_bf16) or 16 (eight wavefronts,
// v_mfma_f32_16x16x32_bf16: attn16.hip).  PZN_ATTN_ROWS selects; read once.  The images, the gate words and the tile
// images differ between the two, so the mode is a property of the process, not of a call.
int pzn_attn_rows_mode() {
  static const int mode = [] {
    const char* e = getenv("PZN_ATTN_ROWS");
    return e && atoi(e) == 32 ? 32 : 16;
  }();
  return mode;
}
PZN_EXPORT int pzn_attn_fused_rows(void) { return pzn_attn_rows_mode(); }
PZN_EXPORT size_t pzn_attn_fused_qk_image_bytes(int B) { return B > 0 ? (size_t)B * QK_IMG : 0; }
PZN_EXPORT size_t pzn_attn_fused_v_image_bytes(int B) { return B > 0 ? (size_t)B * V_IMG : 0; }

PZN_EXPORT int pzn_attn_fused_supported(int L_, int E_, int dk) { return L_ == L && E_ == E && dk == DK; }

// planes of the weights of n <= 4 blocks (Wq, Wk [dk, E]; Wv, Wo [E, E]) for all kernels, one launch
static void pack_jobs(PackArgs& a, int at, const float* Wq, const float* Wk, const float* Wv, const float* Wo, unsigned char* w) {
  // W_QKV: rows n = q | k | v, k = c
  a.job[at + 0] = PackJob{Wq, E, 1, 2, 16, 12, 0, 0, w + W_QKV};
  a.job[at + 1] = PackJob{Wk, E, 1, 2, 16, 12, 2, 0, w + W_QKV};
  a.job[at + 2] = PackJob{Wv, E, 1, 8, 16, 12, 4, 0, w + W_QKV};
  a.job[at + 3] = PackJob{Wo, E, 1, 8, 16, 8, 0, 0, w + W_O};        // rows o, k = c
  a.job[at + 4] = PackJob{Wo, 1, E, 8, 16, 8, 0, 0, w + W_OT};       // rows c, k = o:  A[c][o] = Wo[o][c]
  a.job[at + 5] = PackJob{Wq, 1, E, 8, 4, 8, 0, 0, w + W_QT};        // rows c, k = d:  A[c][d] = Wq[d][c]
  a.job[at + 6] = PackJob{Wk, 1, E, 8, 4, 8, 0, 0, w + W_KVT};       // rows c, k = d
  a.job[at + 7] = PackJob{Wv, 1, E, 8, 16, 8, 0, 4, w + W_KVT};      // rows c, k = c' (k-steps 4..19)
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
  if (pzn_attn_rows_mode() == 16) return pzn_attn16_prep_weights(n, Wq, Wk, Wv, Wo, planes, pzn_hip_stream(stream));
  hipLaunchKernelGGL(pack_rp_kernel, dim3(12, a.njobs), dim3(256), 0, pzn_hip_stream(stream), a);
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
  if (pzn_attn_rows_mode() == 16) return pzn_attn16_proj(nprob, x, w, bq, bk, bv, B, qrp, krp, vrp, pzn_hip_stream(stream));
  if (pzn_attn_precision_mode() == 1)
    hipLaunchKernelGGL(attn_proj_kernel<1>, dim3(a.nb * nprob), dim3(NT), 0, pzn_hip_stream(stream), a);
  else
    hipLaunchKernelGGL(attn_proj_kernel<3>, dim3(a.nb * nprob), dim3(NT), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}

// one block forward: r = x + relu(Wo (x - softmax(q k^T / 8) v) + bo); also t = x - attn v, the gate bits, ln-sum-exp per
// row and (map[i] != NULL) the running mean map: map = scale P (accumulate = 0) or map += scale P
PZN_EXPORT int pzn_attn_fused_fwd(int nprob, const float* const* x, const void* const* qrp, const void* const* krp,
                                  const void* const* vrp, const void* const* w, const float* const* bo, int B,
                                  float* const* r, float* const* t, void* const* mask, float* const* map, float* const* lse,
                                  int map_accumulate, float map_scale, pzn_stream_t stream) {
  PZN_CHECK_ARG(nprob >= 1 && nprob <= 2 && B > 0 && x && qrp && krp && vrp && w && bo && r && t && mask && map && lse);
  FwdArgs a;
  a.nb = 2 * B;
  a.map_accumulate = map_accumulate;
  a.map_scale = map_scale;
  for (int i = 0; i < nprob; ++i) {
    PZN_CHECK_ARG(x[i] && qrp[i] && krp[i] && vrp[i] && w[i] && bo[i] && r[i] && t[i] && mask[i] && lse[i]);
    PZN_CHECK_ARG(aligned16(x[i]) && aligned16(qrp[i]) && aligned16(krp[i]) && aligned16(vrp[i]) && aligned16(w[i]) &&
                  aligned16(bo[i]) && aligned16(r[i]) && aligned16(t[i]) && aligned16(mask[i]) && aligned16(map[i]));
    a.p[i] = FwdProb{x[i], static_cast<const unsigned char*>(qrp[i]), static_cast<const unsigned char*>(krp[i]),
                     static_cast<const unsigned char*>(vrp[i]), static_cast<const unsigned char*>(w[i]), bo[i], r[i], t[i],
                     static_cast<uint32_t*>(mask[i]), map[i], lse[i]};
  }
  if (pzn_attn_rows_mode() == 16)
    return pzn_attn16_fwd(nprob, x, qrp, krp, vrp, w, bo, B, r, t, mask, map, lse, map_accumulate, map_scale, pzn_hip_stream(stream));
  if (pzn_attn_precision_mode() == 1)
    hipLaunchKernelGGL(attn_fwd_kernel<1>, dim3(a.nb * nprob), dim3(NT), 0, pzn_hip_stream(stream), a);
  else
    hipLaunchKernelGGL(attn_fwd_kernel<3>, dim3(a.nb * nprob), dim3(NT), 0, pzn_hip_stream(stream), a);
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
  if (pzn_attn_rows_mode() == 16)
    return pzn_attn16_bwd_q(nprob, dr, ld_dr, dr2, ld_dr2, mask, qrp, krp, vrp, w, B, dz, u, dq, dqt, darp, delta, pzn_hip_stream(stream));
  if (pzn_attn_precision_mode() == 1)
    hipLaunchKernelGGL(attn_bwd_q_kernel<1>, dim3(a.nb * nprob), dim3(NT), 0, pzn_hip_stream(stream), a);
  else
    hipLaunchKernelGGL(attn_bwd_q_kernel<3>, dim3(a.nb * nprob), dim3(NT), 0, pzn_hip_stream(stream), a);
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
  if (pzn_attn_rows_mode() == 16)
    return pzn_attn16_bwd_k(nprob, qrp, krp, vrp, darp, w, lse, delta, u, dq, B, dk, dv, dx, pzn_hip_stream(stream));
  if (pzn_attn_precision_mode() == 1)
    hipLaunchKernelGGL(attn_bwd_k_kernel<1>, dim3(a.nb * nprob), dim3(NT), 0, pzn_hip_stream(stream), a);
  else
    hipLaunchKernelGGL(attn_bwd_k_kernel<3>, dim3(a.nb * nprob), dim3(NT), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}
