// gemm.hip — exact-fp32 dense engine on the matrix cores (v_mfma_f32_32x32x2_f32).
//
// Everything dense in model5_b.py funnels through ONE tile engine:
//   * nn.Linear (+bias, +ReLU) forward              (model5_b.py:417-422, 559-599)   "NT"
//   * shared MLP second layer with max-over-K=32    (model5_b.py:452-454, 459-461)   "NT" + max-pool epilogue
//   * input gradients  dX = dY W                    "NN"
//   * weight gradients dW = dY^T X, db = sum dY     "TN", split over the row range, fp32 atomics
//   * attention products q k^T, attn v, and their backward (model5_b.py:67-75), batched.
//
// gfx950 has an f32-input MFMA whose result is bit-for-bit an fp32 fma chain (no TF32-like
// truncation), at 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak.  That keeps the 1e-4 parity bar
// reachable with plain fp32 semantics.
//
// Tile engine (256 threads = 4 wavefronts):
//   block tile BM x BN (128x128 or 128x64), K-step 16, double-buffered LDS.
//   LDS images are K-MAJOR for both operands (As[k][m], Bs[k][n]) so a fragment read for
//   v_mfma_f32_32x32x2 (lane l: A[m = l&31][k = l>>5], B[k = l>>5][n = l&31]) is 32
//   consecutive dwords per half-wave: conflict-free ds_read_b32.  Operands that are
//   k-contiguous in HBM are transposed on the way in (float4 global load, 4 ds_write_b32);
//   operands that are m/n-contiguous go in with ds_write_b128.
//   Each wave owns a (BM/WM) x (BN/WN) sub-tile = TM x TN MFMA tiles of 32x32, accumulators
//   in registers.  The 32 rows of one MFMA tile are exactly one (centroid, K=32 neighbours)
//   group, so max-over-K is a register epilogue: 15 v_max per lane + one cross-half exchange.
//   One barrier per K-step: next tile's global loads are issued before the MFMAs of the
//   current one and written to the other LDS buffer after them.
//
// "Generators" synthesise the A operand on the fly so that sparse / masked gradients are
// never materialised:  dY * (Y > 0)  (ReLU backward)  and the max-pool scatter
// dy[g*32 + k, c] = (argmax[g,c] == k && out[g,c] > 0) ? dOut[g,c] : 0.
#include <stdlib.h>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---- split precision ("bf16x3") -------------------------------------------------------------
// x = x1 + x2 + x3 exactly, each xi a bf16 (8 significant bits, fp32's exponent range): x1 = bf16(x),
// x2 = bf16(x - x1), x3 = bf16(x - x1 - x2).  a*b is then summed from the six products whose magnitude is
// >= 2^-16 of the leading one: (1,1) (1,2) (2,1) (1,3) (2,2) (3,1); the three dropped ones are <= 2^-24
// relative, i.e. below fp32 rounding.  Each product of two bf16 is exact in fp32 and the MFMA accumulates
// in fp32, so the result has fp32-GEMM accuracy — at 6 v_mfma_f32_32x32x16_bf16 (32 cycles for 16 k) against
// 8 v_mfma_f32_32x32x2_f32 (64 cycles for 2 k): 2.67x the matrix-pipe throughput.
__device__ __forceinline__ uint32_t f2bf(float x) {
  __bf16 b = (__bf16)x;  // v_cvt_pk_bf16_f32, round to nearest even
  return (uint32_t)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ void split3(float x, uint32_t& a, uint32_t& b, uint32_t& c) {
  a = f2bf(x);
  float r = x - bf2f(a);
  b = f2bf(r);
  float r2 = r - bf2f(b);
  c = f2bf(r2);
}

constexpr int BK = 16;
constexpr int GT = 256;  // threads per block
constexpr int PAD = 4;

enum { GEN_NONE = 0, GEN_RELU = 1, GEN_MAXPOOL = 2 };
enum { EPI_STORE = 0, EPI_MAXPOOL = 1, EPI_ATOMIC = 2 };

struct GemmArgs {
  int plain_bf16;  // EPI_STORE launches only: operands rounded to bf16 once, ONE MFMA per product (opt-in attention mode)
  // logical problem: C[M,N] = sum_k A(m,k) B(k,n)
  int M, N, K;
  const float* A;  // A_KC: A[m*lda + k]   else: A[k*lda + m]
  int lda;
  const float* B;  // B_KC: B[n*ldb + k]   else: B[k*ldb + n]
  int ldb;
  float* C;
  int ldc;
  long sA, sB, sC;  // batch strides (elements); batch index = blockIdx.z when splits == 1
  int splits;       // > 1: blockIdx.z splits the K range, epilogue accumulates atomically
  int k_chunk;      // K elements per split (multiple of BK)
  // A generator (the "dY" matrix has rows r, columns c;  A_KC: (r,c) = (m,k)   else (r,c) = (k,m))
  int gen;
  const float* genY;        // GEN_RELU: same layout as A
  const int32_t* genArg;    // GEN_MAXPOOL: argmax[r/32][c], ld = lda
  const float* genOut;      // GEN_MAXPOOL: out[r/32][c]; A points at dOut[r/32][c]
  // epilogue
  const float* bias;        // per column n (or NULL)
  int relu;
  float alpha;
  const float* maskH;       // multiply C by (maskH[m*ldc + n] > 0)
  const float* addend;      // EPI_STORE: C = alpha * (A B) + addend (same layout and batch stride as C)
  int32_t* argmax;          // EPI_MAXPOOL: C is out[M/32][N], argmax[M/32][N]
  float* bias_grad;         // "TN" (A k-major) only: bias_grad[m] += sum_k A(m,k), taken from the A tiles as
                            // they stream through the loader of the blockIdx.x == 0 column of workgroups
  const float* side;        // same mechanism with weights: side_out[m*ld_side_out + c] += sum_k A(m,k) side[k*ld_side + c],
  int ld_side;              // c = 0..2  (the three xyz columns of a grouped row: a 3-wide GEMM for free)
  float* side_out;
  int ld_side_out;
};

// ---- operand loaders ---------------------------------------------------------------------
// Each fetch() pulls this thread's share of a BR x BK tile (BR = BM or BN) into registers,
// store() puts it into the k-major LDS image.

template <int BR, bool KC>
struct Loader {
  static constexpr int NV = BR * BK / 4 / GT;  // float4 per thread
  float4 v[NV];

  // (row index in the R dimension, k index) -> value, fully guarded scalar path
  __device__ __forceinline__ static float fetch1(const GemmArgs& p, bool isA, const float* base, int ld, int R,
                                                 int rr, int Kend, int kk) {
    if (rr >= R || kk >= Kend) return 0.f;
    if (isA && p.gen == GEN_MAXPOOL) {
      // dY rows r, cols c:  A_KC: r = rr (m), c = kk   else r = kk, c = rr
      long r = KC ? rr : kk;
      int c = KC ? kk : rr;
      long g = r >> 5;
      long off = g * ld + c;
      float o = p.genOut[off];
      return (p.genArg[off] == (int)(r & 31) && o > 0.f) ? base[off] : 0.f;
    }
    long off = KC ? (long)rr * ld + kk : (long)kk * ld + rr;
    float x = base[off];
    if (isA && p.gen == GEN_RELU) x = p.genY[off] > 0.f ? x : 0.f;
    return x;
  }

  __device__ __forceinline__ void fetch(const GemmArgs& p, bool isA, const float* base, int ld, int R, int r0,
                                        int k0, int Kend, bool vec_ok, int tid) {
    // Interior tiles (all but the last row/K tile of a problem): one workgroup-uniform branch, then
    // straight-line 16-byte loads — no per-element bounds tests, no exec-mask divergence.
    if (vec_ok && r0 + BR <= R && k0 + BK <= Kend) {
      const int gen = isA ? p.gen : GEN_NONE;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int f = tid + i * GT;
        const int rr = KC ? r0 + f / (BK / 4) : r0 + (f % (BR / 4)) * 4;
        const int kk = KC ? k0 + (f % (BK / 4)) * 4 : k0 + f / (BR / 4);
        if (gen == GEN_MAXPOOL) {
          const long r = KC ? rr : kk;
          const long off = (r >> 5) * ld + (KC ? kk : rr);
          const float4 d = *reinterpret_cast<const float4*>(base + off);
          const float4 o = *reinterpret_cast<const float4*>(p.genOut + off);
          const int4 a = *reinterpret_cast<const int4*>(p.genArg + off);
          const int rl = (int)(r & 31);
          v[i] = make_float4((a.x == rl && o.x > 0.f) ? d.x : 0.f, (a.y == rl && o.y > 0.f) ? d.y : 0.f,
                             (a.z == rl && o.z > 0.f) ? d.z : 0.f, (a.w == rl && o.w > 0.f) ? d.w : 0.f);
        } else {
          const long off = KC ? (long)rr * ld + kk : (long)kk * ld + rr;
          float4 d = *reinterpret_cast<const float4*>(base + off);
          if (gen == GEN_RELU) {
            const float4 y = *reinterpret_cast<const float4*>(p.genY + off);
            d.x = y.x > 0.f ? d.x : 0.f;
            d.y = y.y > 0.f ? d.y : 0.f;
            d.z = y.z > 0.f ? d.z : 0.f;
            d.w = y.w > 0.f ? d.w : 0.f;
          }
          v[i] = d;
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = tid + i * GT;
      int rr, kk;
      if (KC) {  // float4 along k
        rr = r0 + f / (BK / 4);
        kk = k0 + (f % (BK / 4)) * 4;
      } else {  // float4 along the row dimension
        kk = k0 + f / (BR / 4);
        rr = r0 + (f % (BR / 4)) * 4;
      }
      bool full = KC ? (rr < R && kk + 3 < Kend) : (kk < Kend && rr + 3 < R);
      if (vec_ok && full) {
        if (isA && p.gen == GEN_MAXPOOL) {
          long r = KC ? rr : kk;
          int c = KC ? kk : rr;
          long off = (r >> 5) * ld + c;
          float4 d = *reinterpret_cast<const float4*>(base + off);
          float4 o = *reinterpret_cast<const float4*>(p.genOut + off);
          int4 a = *reinterpret_cast<const int4*>(p.genArg + off);
          int rl = (int)(r & 31);
          if (KC) {  // the 4 elements share the row r
            v[i] = make_float4((a.x == rl && o.x > 0.f) ? d.x : 0.f, (a.y == rl && o.y > 0.f) ? d.y : 0.f,
                               (a.z == rl && o.z > 0.f) ? d.z : 0.f, (a.w == rl && o.w > 0.f) ? d.w : 0.f);
          } else {  // also the same row r (k index), 4 consecutive columns
            v[i] = make_float4((a.x == rl && o.x > 0.f) ? d.x : 0.f, (a.y == rl && o.y > 0.f) ? d.y : 0.f,
                               (a.z == rl && o.z > 0.f) ? d.z : 0.f, (a.w == rl && o.w > 0.f) ? d.w : 0.f);
          }
        } else {
          long off = KC ? (long)rr * ld + kk : (long)kk * ld + rr;
          float4 d = *reinterpret_cast<const float4*>(base + off);
          if (isA && p.gen == GEN_RELU) {
            float4 y = *reinterpret_cast<const float4*>(p.genY + off);
            d.x = y.x > 0.f ? d.x : 0.f;
            d.y = y.y > 0.f ? d.y : 0.f;
            d.z = y.z > 0.f ? d.z : 0.f;
            d.w = y.w > 0.f ? d.w : 0.f;
          }
          v[i] = d;
        }
      } else {
        float t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          t[c] = KC ? fetch1(p, isA, base, ld, R, rr, Kend, kk + c) : fetch1(p, isA, base, ld, R, rr + c, Kend, kk);
        v[i] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  }

  // weighted sums for the three side columns (k-major loaders only)
  __device__ __forceinline__ void accumulate_side(float4* acc3, const float* side, int ld, int k0, int Kend,
                                                  int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int kk = k0 + (tid + i * GT) / (BR / 4);
      if (kk < Kend) {
        const float* sp = side + (long)kk * ld;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float w = sp[c];
          acc3[c].x += v[i].x * w;
          acc3[c].y += v[i].y * w;
          acc3[c].z += v[i].z * w;
          acc3[c].w += v[i].w * w;
        }
      }
    }
  }

  // sum of this thread's float4s (k-major loaders only: every float4 of a thread covers the same 4 rows)
  __device__ __forceinline__ void accumulate(float4& acc) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      acc.x += v[i].x;
      acc.y += v[i].y;
      acc.z += v[i].z;
      acc.w += v[i].w;
    }
  }

  // bf16x3 image: three planes of BR x 16 bf16, each in MFMA-fragment order: the 16-byte chunk of
  // (row r, k-half h) sits at chunk index (r/32 * 2 + h) * 32 + r % 32, so a wave's fragment read for one
  // 32-row tile is 64 consecutive chunks (lane = h*32 + r%32): conflict-free ds_read_b128, no padding.
  // ONE: plain bf16 operands (the opt-in attention mode): only the first plane is produced
  template <bool ONE = false>
  __device__ __forceinline__ void store_x3(unsigned char* img, int tid) const {
    constexpr int PLANE = KC ? BR * 32 : 16 * (BR + 32) * 2;  // bytes
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = tid + i * GT;
      float t[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
      uint32_t a[4], b[4], c[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ONE)
          a[e] = f2bf(t[e]), b[e] = c[e] = 0;
        else
          split3(t[e], a[e], b[e], c[e]);
      }
      if (KC) {  // 4 consecutive k of one row: one 8-byte store per plane
        int r = f / (BK / 4), kq = (f % (BK / 4)) * 4;
        int off = (((r >> 5) * 2 + (kq >> 3)) * 32 + (r & 31)) * 16 + (kq & 7) * 2;
        *reinterpret_cast<uint2*>(img + off) = make_uint2(a[0] | (a[1] << 16), a[2] | (a[3] << 16));
        if (ONE) continue;
        *reinterpret_cast<uint2*>(img + PLANE + off) = make_uint2(b[0] | (b[1] << 16), b[2] | (b[3] << 16));
        *reinterpret_cast<uint2*>(img + 2 * PLANE + off) = make_uint2(c[0] | (c[1] << 16), c[2] | (c[3] << 16));
      } else {  // 4 consecutive rows at one k: k-major plane [16][BR+32] bf16, one 8-byte store per plane;
                // the fragment is produced at read time by ds_read_b64_tr_b16 (hardware transpose)
        int k = f / (BR / 4), r0 = (f % (BR / 4)) * 4;
        int off = (k * (BR + 32) + r0) * 2;
        *reinterpret_cast<uint2*>(img + off) = make_uint2(a[0] | (a[1] << 16), a[2] | (a[3] << 16));
        if (ONE) continue;
        *reinterpret_cast<uint2*>(img + PLANE + off) = make_uint2(b[0] | (b[1] << 16), b[2] | (b[3] << 16));
        *reinterpret_cast<uint2*>(img + 2 * PLANE + off) = make_uint2(c[0] | (c[1] << 16), c[2] | (c[3] << 16));
      }
    }
  }

  // One bf16 MFMA fragment (8 consecutive k of row tile_row + lane%32, k-half lane/32) from plane `pl`.
  __device__ __forceinline__ static bf16x8 read_frag(const unsigned char* pl, int tile_row, int lane) {
    if (KC) return *reinterpret_cast<const bf16x8*>(pl + ((tile_row >> 5) * 64 + lane) * 16);
    // k-major image: each 16-lane group transposes a 4(k) x 16(rows) block per read; lane 4q+p of the group
    // supplies the address of k-row q, rows 4p..4p+3 and receives the 4 k-values of row `i`.
    const int g = lane >> 4, i = lane & 15;
    const int krow = 8 * (g >> 1) + (i >> 2);
    const int col = tile_row + 16 * (g & 1) + 4 * (i & 3);
    const unsigned char* a0 = pl + (krow * (BR + 32) + col) * 2;
    typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(a0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(a0 + 4 * (BR + 32) * 2));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  }

  __device__ __forceinline__ void store(float (*S)[BR + PAD], int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int f = tid + i * GT;
      if (KC) {
        int r = f / (BK / 4), k = (f % (BK / 4)) * 4;
        S[k + 0][r] = v[i].x;
        S[k + 1][r] = v[i].y;
        S[k + 2][r] = v[i].z;
        S[k + 3][r] = v[i].w;
      } else {
        int k = f / (BR / 4), r = (f % (BR / 4)) * 4;
        *reinterpret_cast<float4*>(&S[k][r]) = v[i];
      }
    }
  }
};

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int EPI, bool X3, bool ONE = false>
__global__ __launch_bounds__(GT) void gemm_kernel(GemmArgs p) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(WM * WN == GT / PZN_WAVE, "4 waves");
  // ONE LDS array: operand images during the K loop, per-wave 32 x 64 staging tiles in the epilogue.
  // fp32 images: k-major [2][BK][rows+PAD] floats; bf16x3 images: [2][3 planes][rows*16] bf16 = rows*24 floats/buffer.
  constexpr int A_PLANE = A_KC ? BM * 32 : 16 * (BM + 32) * 2, B_PLANE = B_KC ? BN * 32 : 16 * (BN + 32) * 2;  // bytes
  constexpr int A_ELEMS = X3 ? 2 * 3 * A_PLANE / 4 : 2 * BK * (BM + PAD);
  constexpr int B_ELEMS = X3 ? 2 * 3 * B_PLANE / 4 : 2 * BK * (BN + PAD);
  constexpr int STG_LD = 64 + PAD, STG_ELEMS = (GT / PZN_WAVE) * 32 * STG_LD;
  constexpr int SMEM_ELEMS = A_ELEMS + B_ELEMS > STG_ELEMS ? A_ELEMS + B_ELEMS : STG_ELEMS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_ELEMS];
  float (*As)[BK][BM + PAD] = reinterpret_cast<float (*)[BK][BM + PAD]>(smem);
  float (*Bs)[BK][BN + PAD] = reinterpret_cast<float (*)[BK][BN + PAD]>(smem + A_ELEMS);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  const float* A = p.A;
  const float* B = p.B;
  float* C = p.C;
  int kbeg = 0, kend = p.K;
  if (p.splits > 1) {
    kbeg = blockIdx.z * p.k_chunk;
    kend = min(p.K, kbeg + p.k_chunk);
    if (kbeg >= kend) return;
  } else {
    A += (long)blockIdx.z * p.sA;
    B += (long)blockIdx.z * p.sB;
    C += (long)blockIdx.z * p.sC;
    if (p.addend) p.addend += (long)blockIdx.z * p.sC;
  }
  const bool a_vec = (p.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
                     (p.gen != GEN_RELU || (reinterpret_cast<uintptr_t>(p.genY) & 15) == 0) &&
                     (p.gen != GEN_MAXPOOL || ((reinterpret_cast<uintptr_t>(p.genOut) & 15) == 0 &&
                                               (reinterpret_cast<uintptr_t>(p.genArg) & 15) == 0));
  const bool b_vec = (p.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Loader<BM, A_KC> la;
  Loader<BN, B_KC> lb;
  // bias gradient: this thread's A float4s always cover the same 4 output rows (m-quad tid % (BM/4))
  const bool do_bias = !A_KC && p.bias_grad != nullptr && blockIdx.x == 0;
  const bool do_side = !A_KC && p.side != nullptr && blockIdx.x == 0;
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ssum[3] = {bsum, bsum, bsum};
  la.fetch(p, true, A, p.lda, p.M, m0, kbeg, kend, a_vec, tid);
  lb.fetch(p, false, B, p.ldb, p.N, n0, kbeg, kend, b_vec, tid);
  if (do_bias) la.accumulate(bsum);
  if (do_side) la.accumulate_side(ssum, p.side, p.ld_side, kbeg, kend, tid);
  unsigned char* imgA = reinterpret_cast<unsigned char*>(smem);               // bf16x3: [2][3][A_PLANE]
  unsigned char* imgB = reinterpret_cast<unsigned char*>(smem + A_ELEMS);     //         [2][3][B_PLANE]
  if (X3) {
    la.template store_x3<ONE>(imgA, tid);
    lb.template store_x3<ONE>(imgB, tid);
  } else {
    la.store(As[0], tid);
    lb.store(Bs[0], tid);
  }
  __syncthreads();

  const int nkt = (kend - kbeg + BK - 1) / BK;
  const int half = lane >> 5, l31 = lane & 31;
  if (X3) {
    // Software pipeline, two tiles deep:  LDS[cur] = tile kt (converted), registers = tile kt+1 (raw fp32,
    // loads issued one iteration ago), and tile kt+2's loads are issued at the end of the iteration.
    // The fp32 -> 3 x bf16 split of tile kt+1 (~110 VALU + 6 ds_write per thread) is interleaved with
    // the 24 MFMAs of tile kt (sched_group_barrier): it runs in the shadow of the matrix pipe.
    // register sets: (la, lb) and (la2, lb2) alternate between "landed tile kt+1, being converted" and
    // "tile kt+2, loads in flight": the loads get a whole K-step of MFMA time to land.
    Loader<BM, A_KC> la2;
    Loader<BN, B_KC> lb2;
    if (nkt > 1) {
      la.fetch(p, true, A, p.lda, p.M, m0, kbeg + BK, kend, a_vec, tid);
      lb.fetch(p, false, B, p.ldb, p.N, n0, kbeg + BK, kend, b_vec, tid);
    }
    auto mma_tile = [&](int cur, bool convert_next, Loader<BM, A_KC>& ca_regs, Loader<BN, B_KC>& cb_regs) {
      const unsigned char* ca = imgA + cur * (3 * A_PLANE);
      const unsigned char* cb = imgB + cur * (3 * B_PLANE);
      constexpr int NPL = ONE ? 1 : 3;
      bf16x8 af[TM][NPL], bf[TN][NPL];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < NPL; ++q) af[i][q] = Loader<BM, A_KC>::read_frag(ca + q * A_PLANE, wm * (BM / WM) + i * 32, lane);
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < NPL; ++q) bf[j][q] = Loader<BN, B_KC>::read_frag(cb + q * B_PLANE, wn * (BN / WN) + j * 32, lane);
      if (convert_next) {  // compile-time constant at every call site: no branch inside the scheduled region
        ca_regs.template store_x3<ONE>(imgA + (cur ^ 1) * (3 * A_PLANE), tid);
        cb_regs.template store_x3<ONE>(imgB + (cur ^ 1) * (3 * B_PLANE), tid);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          floatx16 c = acc[i][j];
          if constexpr (ONE) {  // plain bf16 operands, fp32 accumulation: one product
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
          } else {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], c, 0, 0, 0);  // small terms first
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
          }
          acc[i][j] = c;
        }
      if (convert_next && !ONE) {
#pragma unroll
        for (int g = 0; g < TM * TN * 6; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);  // five VALU
          if ((g & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // one DS write
        }
      }
    };
    auto step = [&](int kt, Loader<BM, A_KC>& xa, Loader<BN, B_KC>& xb, Loader<BM, A_KC>& ya, Loader<BN, B_KC>& yb) {
      if (kt + 2 < nkt) {  // tile kt+2 -> the free register set, a full K-step ahead of its use
        ya.fetch(p, true, A, p.lda, p.M, m0, kbeg + (kt + 2) * BK, kend, a_vec, tid);
        yb.fetch(p, false, B, p.ldb, p.N, n0, kbeg + (kt + 2) * BK, kend, b_vec, tid);
      }
      if (do_bias) xa.accumulate(bsum);
      if (do_side) xa.accumulate_side(ssum, p.side, p.ld_side, kbeg + (kt + 1) * BK, kend, tid);
      mma_tile(kt & 1, true, xa, xb);
      __syncthreads();
    };
    int kt = 0;
    for (; kt + 2 < nkt; kt += 2) {
      step(kt, la, lb, la2, lb2);
      step(kt + 1, la2, lb2, la, lb);
    }
    if (kt + 1 < nkt) {
      step(kt, la, lb, la2, lb2);
      ++kt;
    }
    mma_tile((nkt - 1) & 1, false, la, lb);
    __syncthreads();
  } else {
    for (int kt = 0; kt < nkt; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nkt) {
        la.fetch(p, true, A, p.lda, p.M, m0, kbeg + (kt + 1) * BK, kend, a_vec, tid);
        lb.fetch(p, false, B, p.ldb, p.N, n0, kbeg + (kt + 1) * BK, kend, b_vec, tid);
      }
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[cur][kk + half][wm * (BM / WM) + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[cur][kk + half][wn * (BN / WN) + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (kt + 1 < nkt) {
        if (do_bias) la.accumulate(bsum);
        if (do_side) la.accumulate_side(ssum, p.side, p.ld_side, kbeg + (kt + 1) * BK, kend, tid);
        la.store(As[cur ^ 1], tid);
        lb.store(Bs[cur ^ 1], tid);
      }
      __syncthreads();
    }
  }
  if (!A_KC && blockIdx.x == 0 && (p.bias_grad != nullptr || p.side != nullptr)) {  // (uniform over the workgroup)
    // 8 threads (lane, lane+32 in each of the 4 waves) hold partial sums of the same 4 rows:
    // fold them on chip so that the contended atomics are 128 per workgroup, not 1024.
    float4* red = reinterpret_cast<float4*>(smem);  // the K loop is over: LDS is free (last barrier passed)
    auto fold = [&](float4 v, float* out, int ld, int col) {
      v.x += __shfl_xor(v.x, 32, PZN_WAVE);
      v.y += __shfl_xor(v.y, 32, PZN_WAVE);
      v.z += __shfl_xor(v.z, 32, PZN_WAVE);
      v.w += __shfl_xor(v.w, 32, PZN_WAVE);
      if (half == 0) red[wave * 32 + l31] = v;
      __syncthreads();
      if (wave == 0 && half == 0) {
        float4 a = red[l31], b = red[32 + l31], c = red[64 + l31], d = red[96 + l31];
        int r = m0 + l31 * 4;
        if (r + 0 < p.M) atomicAdd(out + (long)(r + 0) * ld + col, (a.x + b.x) + (c.x + d.x));
        if (r + 1 < p.M) atomicAdd(out + (long)(r + 1) * ld + col, (a.y + b.y) + (c.y + d.y));
        if (r + 2 < p.M) atomicAdd(out + (long)(r + 2) * ld + col, (a.z + b.z) + (c.z + d.z));
        if (r + 3 < p.M) atomicAdd(out + (long)(r + 3) * ld + col, (a.w + b.w) + (c.w + d.w));
      }
      __syncthreads();
    };
    if (p.bias_grad != nullptr) fold(bsum, p.bias_grad, 1, 0);
    if (p.side != nullptr) {
      fold(ssum[0], p.side_out, p.ld_side_out, 0);
      fold(ssum[1], p.side_out, p.ld_side_out, 1);
      fold(ssum[2], p.side_out, p.ld_side_out, 2);
    }
  }

  // ---- epilogue ----  C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if (EPI == EPI_STORE) {
    // Stage each 32 x 64 strip of the wave's sub-tile through LDS and write it as whole 256-B row
    // segments with 16-B stores (a register-direct store is one dword per lane: 4x the store
    // instructions, 128-B segments, and a dword-gather for the ReLU mask).
    static_assert(TN == 2, "a wave's sub-tile is 64 columns wide");
    float* stg = smem + wave * 32 * STG_LD;
    const int cbase = n0 + wn * (BN / WN);
    const bool c_vec = (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
                       (!p.maskH || (reinterpret_cast<uintptr_t>(p.maskH) & 15) == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = m0 + wm * (BM / WM) + i * 32;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = cbase + j * 32 + l31;
        const float bv = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = acc[i][j][r] * p.alpha + bv;
          if (p.relu) v = v > 0.f ? v : 0.f;
          stg[rl * STG_LD + j * 32 + l31] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ps = 0; ps < 8; ++ps) {
        const int rl = ps * 4 + (lane >> 4), c4 = (lane & 15) * 4;
        const int row = rbase + rl, col = cbase + c4;
        if (row >= p.M || col >= p.N) continue;
        float4 v = *reinterpret_cast<const float4*>(stg + rl * STG_LD + c4);
        const long off = (long)row * p.ldc + col;
        if (c_vec && col + 3 < p.N && p.addend && (reinterpret_cast<uintptr_t>(p.addend) & 15) == 0) {
          const float4 a = *reinterpret_cast<const float4*>(p.addend + off);
          v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
        } else if (p.addend) {
          if (col < p.N) v.x += p.addend[off];
          if (col + 1 < p.N) v.y += p.addend[off + 1];
          if (col + 2 < p.N) v.z += p.addend[off + 2];
          if (col + 3 < p.N) v.w += p.addend[off + 3];
        }
        if (c_vec && col + 3 < p.N) {
          if (p.maskH) {
            float4 h = *reinterpret_cast<const float4*>(p.maskH + off);
            v.x = h.x > 0.f ? v.x : 0.f;
            v.y = h.y > 0.f ? v.y : 0.f;
            v.z = h.z > 0.f ? v.z : 0.f;
            v.w = h.w > 0.f ? v.w : 0.f;
          }
          *reinterpret_cast<float4*>(C + off) = v;
        } else {
          float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (col + c < p.N) {
              float x = t[c];
              if (p.maskH) x = p.maskH[off + c] > 0.f ? x : 0.f;
              C[off + c] = x;
            }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 32 + l31;
      const int rbase = m0 + wm * (BM / WM) + i * 32;
      const bool col_ok = col < p.N;
      const float bv = (p.bias && col_ok) ? p.bias[col] : 0.f;
      if (EPI == EPI_MAXPOOL) {
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int rl = (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = acc[i][j][r] * p.alpha + bv;
          v = v > 0.f ? v : 0.f;  // ReLU before the max (model5_b.py:453-454)
          bool gt = v > best;
          best = gt ? v : best;
          bi = gt ? rl : bi;
        }
        float ob = __shfl_xor(best, 32, PZN_WAVE);
        int oi = __shfl_xor(bi, 32, PZN_WAVE);
        bool take = ob > best || (ob == best && oi < bi);
        best = take ? ob : best;
        bi = take ? oi : bi;
        if (half == 0 && col_ok && rbase < p.M) {
          long g = rbase >> 5;
          C[g * p.ldc + col] = best;
          p.argmax[g * p.ldc + col] = bi;
        }
      } else {  // EPI_ATOMIC
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = rbase + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < p.M && col_ok) atomicAdd(C + (long)row * p.ldc + col, acc[i][j][r] * p.alpha);
        }
      }
    }
  }
}

// Which matrix-core path a launch takes: bf16x3 split precision (default, "auto" == "x3" today) or the
// exact-fp32 MFMA (PZN_GEMM_PRECISION=f32, pzn_gemm_set_precision(0)) — same results to fp32 rounding.
int g_precision = -1;  // -1: not decided yet (environment, else auto)
int gemm_precision() {
  if (g_precision < 0) {
    const char* e = getenv("PZN_GEMM_PRECISION");
    g_precision = (e && (e[0] == 'f' || e[0] == 'F')) ? 0 : (e && (e[0] == 'x' || e[0] == 'X')) ? 1 : 2;
  }
  return g_precision;
}

template <int BM, int BN, int WM, int WN, bool A_KC, bool B_KC, int EPI>
void launch_cfg(const GemmArgs& p, int batch, hipStream_t st) {
  dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, p.splits > 1 ? p.splits : batch);
  const int mode = gemm_precision();
  const bool x3 = mode != 0;  // auto == bf16x3 on every product (the split-K grid is sized for it, choose_splits)
  if constexpr (EPI == EPI_STORE) {
    if (x3 && p.plain_bf16) {
      PZN_LAUNCH((gemm_kernel<BM, BN, WM, WN, A_KC, B_KC, EPI, true, true>), grid, dim3(GT), 0, st, p);
      return;
    }
  }
  if (x3)
    PZN_LAUNCH((gemm_kernel<BM, BN, WM, WN, A_KC, B_KC, EPI, true>), grid, dim3(GT), 0, st, p);
  else
    PZN_LAUNCH((gemm_kernel<BM, BN, WM, WN, A_KC, B_KC, EPI, false>), grid, dim3(GT), 0, st, p);
}

template <bool A_KC, bool B_KC, int EPI>
void launch(const GemmArgs& p, int batch, hipStream_t st) {
  if constexpr (EPI == EPI_STORE) {
    // batched small products (the attention maps: 64 x (256 x 256 x 64..256)): 128 x 128 tiles give one 4-wave
    // workgroup per CU, i.e. nothing to hide a load behind; 64 x 128 tiles double the resident wavefronts
    constexpr bool small_tiles = true;
    if (small_tiles && batch > 1 && p.N > 64 && p.M <= 512 && p.splits <= 1) {
      launch_cfg<64, 128, 2, 2, A_KC, B_KC, EPI>(p, batch, st);
      return;
    }
  }
  if (p.N > 64)
    launch_cfg<128, 128, 2, 2, A_KC, B_KC, EPI>(p, batch, st);
  else
    launch_cfg<128, 64, 4, 1, A_KC, B_KC, EPI>(p, batch, st);
}

GemmArgs base_args(int M, int N, int K) {
  GemmArgs p = {};
  p.M = M;
  p.N = N;
  p.K = K;
  p.splits = 1;
  p.alpha = 1.f;
  return p;
}

// split the reduction range of a weight-gradient GEMM so that the grid fills the chip
void choose_splits(GemmArgs& p) {
  long tiles = (long)((p.N + 127) / 128) * ((p.M + 127) / 128);
  // one full wave of resident workgroups: the bf16x3 TN kernel holds 61 KB of LDS (2 per CU), the fp32 one 3 per CU;
  // a grid of 1.5 waves costs as much as 2 (measured: 0.86 -> 0.69 ms on a 256x256x524288 weight gradient)
  constexpr long forced = 0;
  const long target = forced ? forced : (gemm_precision() == 0 ? 768L : 512L);
  long want = target / (tiles > 0 ? tiles : 1);
  long ksteps = (p.K + BK - 1) / BK;
  long splits = want < 1 ? 1 : want;
  if (splits > ksteps / 16) splits = ksteps / 16;  // >= 16 K-steps (256 rows) per split: the atomic epilogue
  if (splits < 1) splits = 1;                      // and the zero-fill must stay small next to the MFMA work
  if (splits > 65535) splits = 65535;
  long per = (ksteps + splits - 1) / splits;
  p.k_chunk = (int)(per * BK);
  p.splits = (int)((p.K + p.k_chunk - 1) / p.k_chunk);
}

// Few-row layers (the pose head: 64 rows through 2048->1024->512->512->256): one 128-row tile per 128 output
// columns leaves the grid at 2-16 workgroups walking the whole K range, so the launch is latency-bound
// (176 us for 64x2048x1024).  Split K across the chip instead and finish bias / ReLU in a second tiny launch.
bool few_rows(int M, int K) { return M <= 128 && K >= 256; }

void few_rows_splits(GemmArgs& p) {
  const long ksteps = (p.K + BK - 1) / BK;
  long per = 8;  // 128 K elements per workgroup
  p.k_chunk = (int)(per * BK);
  p.splits = (int)((ksteps + per - 1) / per);
  if (p.splits < 2) p.splits = 2, p.k_chunk = (int)(((ksteps + 1) / 2) * BK);
}

__global__ void bias_act_kernel(float* __restrict__ y, const float* __restrict__ bias, long total, int N, int relu) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float v = y[i] + (bias ? bias[i % N] : 0.f);
  y[i] = relu ? fmaxf(v, 0.f) : v;
}

}  // namespace

// ------------------------------------------------------------------------------- C ABI ----

PZN_EXPORT int pzn_linear_fwd_f32(const float* x, const float* W, const float* bias, int M, int Kin, int Nout, int relu,
                                  float* y, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && W && y && M > 0 && Kin > 0 && Nout > 0);
  if (gemm_precision() != 0 && pzn_ws_gemm_supported(M, Nout, Kin, x, Kin, nullptr, false)) {  // skinny layer: wsgemm.hip
    const int rc = pzn_ws_gemm(x, Kin, W, Kin, 0, y, Nout, M, Nout, Kin, bias, relu, nullptr, nullptr, nullptr, nullptr,
                               0, 0, pzn_hip_stream(stream));
    if (rc != PZN_EUNSUPPORTED) return rc;  // e.g. an unaligned bias / output: the general engine takes it
  }
  GemmArgs p = base_args(M, Nout, Kin);
  p.A = x, p.lda = Kin, p.B = W, p.ldb = Kin, p.C = y, p.ldc = Nout, p.bias = bias, p.relu = relu;
  if (few_rows(M, Kin)) {
    hipStream_t st = pzn_hip_stream(stream);
    if (pzn_zero_async(y, (size_t)M * Nout, st) != PZN_OK) return PZN_ELAUNCH;
    few_rows_splits(p);
    launch<true, true, EPI_ATOMIC>(p, 1, st);
    if (bias || relu) {
      const long total = (long)M * Nout;
      PZN_LAUNCH(bias_act_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, bias, total, Nout,
                         relu);
    }
    PZN_RETURN_LAUNCH_STATUS();
  }
  launch<true, true, EPI_STORE>(p, 1, pzn_hip_stream(stream));
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_linear_maxpool_fwd_f32(const float* x, const float* W, const float* bias, int R, int Kin, int Nout,
                                          float* out, int32_t* argmax, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && W && out && argmax && R > 0 && Kin > 0 && Nout > 0);
  if (gemm_precision() != 0 && pzn_ws_gemm_supported(R * 32, Nout, Kin, x, Kin, nullptr, true))
    return pzn_ws_gemm(x, Kin, W, Kin, 0, out, Nout, R * 32, Nout, Kin, bias, 1, nullptr, nullptr, argmax, nullptr, 0, 0,
                       pzn_hip_stream(stream));
  GemmArgs p = base_args(R * 32, Nout, Kin);
  p.A = x, p.lda = Kin, p.B = W, p.ldb = Kin, p.C = out, p.ldc = Nout, p.bias = bias, p.relu = 1, p.argmax = argmax;
  launch<true, true, EPI_MAXPOOL>(p, 1, pzn_hip_stream(stream));
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_linear_dgrad_f32(const float* dy, const float* y_relu, const float* W, int M, int Kin, int Nout,
                                    const float* x_relu, float* dx, pzn_stream_t stream) {
  PZN_CHECK_ARG(dy && W && dx && M > 0 && Kin > 0 && Nout > 0);
  if (gemm_precision() != 0 && pzn_ws_gemm_supported(M, Kin, Nout, dy, Nout, y_relu, false)) {
    const int rc = pzn_ws_gemm(dy, Nout, W, Kin, 1, dx, Kin, M, Kin, Nout, nullptr, 0, y_relu, x_relu, nullptr, nullptr,
                               0, 0, pzn_hip_stream(stream));
    if (rc != PZN_EUNSUPPORTED) return rc;
  }
  GemmArgs p = base_args(M, Kin, Nout);
  p.A = dy, p.lda = Nout, p.B = W, p.ldb = Kin, p.C = dx, p.ldc = Kin;
  if (y_relu) p.gen = GEN_RELU, p.genY = y_relu;
  p.maskH = x_relu;
  if (!x_relu && few_rows(M, Nout)) {
    hipStream_t st = pzn_hip_stream(stream);
    if (pzn_zero_async(dx, (size_t)M * Kin, st) != PZN_OK) return PZN_ELAUNCH;
    few_rows_splits(p);
    launch<true, false, EPI_ATOMIC>(p, 1, st);
    PZN_RETURN_LAUNCH_STATUS();
  }
  launch<true, false, EPI_STORE>(p, 1, pzn_hip_stream(stream));
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_linear_maxpool_dgrad_f32(const float* dout, const int32_t* argmax, const float* out, const float* W,
                                            int R, int Kin, int Nout, const float* x_relu, float* dx,
                                            pzn_stream_t stream) {
  PZN_CHECK_ARG(dout && argmax && out && W && dx && R > 0 && Kin > 0 && Nout > 0);
  // (the encoder's levels never take this: their rows' gradient is summed per point where it is computed, csrc/sapool.hip)
  GemmArgs p = base_args(R * 32, Kin, Nout);
  p.A = dout, p.lda = Nout, p.B = W, p.ldb = Kin, p.C = dx, p.ldc = Kin;
  p.gen = GEN_MAXPOOL, p.genArg = argmax, p.genOut = out;
  p.maskH = x_relu;
  launch<true, false, EPI_STORE>(p, 1, pzn_hip_stream(stream));
  PZN_RETURN_LAUNCH_STATUS();
}

// dW[Nout,Kin] = dY^T X, db[Nout] = column sums of dY (row sums of the streamed A tiles): both overwritten.
static int wgrad_common(GemmArgs p, int Kin, int Nout, const float* x, float* dW, float* db, int accumulate,
                        hipStream_t st, int ldw = 0) {
  // logical C[Nout, Kin] = sum_r dY[r][n] * X[r][k];  db[n] = sum_r dY[r][n] from the A tiles.
  // accumulate != 0: add into dW / db as they are (e.g. straight into the flat gradient bucket, which the
  // step zeroes once) — no zero-fill launches here and no separate "grad += dW" pass afterwards.
  p.B = x, p.ldb = Kin, p.C = dW, p.ldc = ldw ? ldw : Kin;
  if (db) {
    p.bias_grad = db;
    if (!accumulate && pzn_zero_async(db, (size_t)Nout, st) != PZN_OK) return PZN_ELAUNCH;
  }
  if (!accumulate && pzn_zero_async(dW, (size_t)Nout * Kin, st) != PZN_OK) return PZN_ELAUNCH;
  choose_splits(p);
  if (p.splits == 1) p.splits = 2, p.k_chunk = ((p.K + 2 * BK - 1) / (2 * BK)) * BK;  // keep the atomic epilogue path
  launch<false, false, EPI_ATOMIC>(p, 1, st);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_linear_wgrad_f32(const float* dy, const float* y_relu, const float* x, int M, int Kin, int Nout,
                                    float* dW, float* db, int accumulate, pzn_stream_t stream) {
  PZN_CHECK_ARG(dy && x && dW && M > 0 && Kin > 0 && Nout > 0);
  if (gemm_precision() != 0 && pzn_df_wgrad_supported(M, Nout, Kin)) {  // small weight matrix: dfgemm.hip
    hipStream_t st = pzn_hip_stream(stream);
    if (!accumulate) {
      if (pzn_zero_async(dW, (size_t)Nout * Kin, st) != PZN_OK) return PZN_ELAUNCH;
      if (db && pzn_zero_async(db, (size_t)Nout, st) != PZN_OK) return PZN_ELAUNCH;
    }
    return pzn_df_wgrad(dy, Nout, y_relu, x, Kin, M, Nout, Kin, dW, Kin, db, -1, st);
  }
  GemmArgs p = base_args(Nout, Kin, M);
  p.A = dy, p.lda = Nout;
  if (y_relu) p.gen = GEN_RELU, p.genY = y_relu;
  return wgrad_common(p, Kin, Nout, x, dW, db, accumulate, pzn_hip_stream(stream));
}

// ---- the same three products on a COLUMN SLICE of a wider weight matrix: W points at column k0 of W_full[Nout, ldw]
// (Kin columns wide).  What cat(x_1 .. x_n) W_full^T needs when the concatenation is never built (model5_b.py:466-474:
// y = sum_i x_i W_i^T + b), and what adds a weight gradient straight into a slice of a parameter (the feature block
// of the first set-abstraction layer).  fwd: accumulate != 0 adds to y (bias then ignored).  dgrad: dx = dy W (+ addend).
// wgrad: ADDS into dW (the slice's other columns are not touched) and into db when non-NULL.
PZN_EXPORT int pzn_linear_slice_fwd_f32(const float* x, const float* W, int ldw, const float* bias, int M, int Kin, int Nout,
                                        int accumulate, float* y, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && W && y && M > 0 && Kin > 0 && Nout > 0 && ldw >= Kin);
  hipStream_t st = pzn_hip_stream(stream);
  if (gemm_precision() != 0 && pzn_ws_gemm_supported(M, Nout, Kin, x, Kin, nullptr, false)) {
    const int rc = pzn_ws_gemm_ex(x, Kin, W, ldw, 0, y, Nout, M, Nout, Kin, accumulate ? nullptr : bias, 0, nullptr, nullptr,
                                  nullptr, nullptr, 0, 0, nullptr, nullptr, accumulate, st);
    if (rc != PZN_EUNSUPPORTED) return rc;
  }
  GemmArgs p = base_args(M, Nout, Kin);
  p.A = x, p.lda = Kin, p.B = W, p.ldb = ldw, p.C = y, p.ldc = Nout, p.bias = accumulate ? nullptr : bias;
  if (accumulate) p.addend = y;      // EPI_STORE: C = A B + addend, element for element
  launch<true, true, EPI_STORE>(p, 1, st);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_linear_slice_dgrad_f32(const float* dy, const float* W, int ldw, int M, int Kin, int Nout,
                                          const float* addend, float* dx, pzn_stream_t stream) {
  PZN_CHECK_ARG(dy && W && dx && M > 0 && Kin > 0 && Nout > 0 && ldw >= Kin);
  GemmArgs p = base_args(M, Kin, Nout);
  p.A = dy, p.lda = Nout, p.B = W, p.ldb = ldw, p.C = dx, p.ldc = Kin, p.addend = addend;
  launch<true, false, EPI_STORE>(p, 1, pzn_hip_stream(stream));
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_linear_slice_wgrad_f32(const float* dy, const float* x, int M, int Kin, int Nout, float* dW, int ldw,
                                          float* db, pzn_stream_t stream) {
  PZN_CHECK_ARG(dy && x && dW && M > 0 && Kin > 0 && Nout > 0 && ldw >= Kin);
  hipStream_t st = pzn_hip_stream(stream);
  if (gemm_precision() != 0 && pzn_df_wgrad_supported(M, Nout, Kin))
    return pzn_df_wgrad(dy, Nout, nullptr, x, Kin, M, Nout, Kin, dW, ldw, db, -1, st);
  GemmArgs p = base_args(Nout, Kin, M);
  p.A = dy, p.lda = Nout;
  return wgrad_common(p, Kin, Nout, x, dW, db, 1, st, ldw);
}

PZN_EXPORT int pzn_linear_maxpool_wgrad_f32(const float* dout, const int32_t* argmax, const float* out, const float* x,
                                            int R, int Kin, int Nout, float* dW, float* db, int accumulate,
                                            pzn_stream_t stream) {
  PZN_CHECK_ARG(dout && argmax && out && x && dW && R > 0 && Kin > 0 && Nout > 0);
  if (pzn_pool_wgrad_supported(Kin, Nout, x)) {      // one non-zero per (group, channel): the sparse pass of poolbwd.hip
    hipStream_t st = pzn_hip_stream(stream);
    if (!accumulate) {
      if (pzn_zero_async(dW, (size_t)Nout * Kin, st) != PZN_OK) return PZN_ELAUNCH;
      if (db && pzn_zero_async(db, (size_t)Nout, st) != PZN_OK) return PZN_ELAUNCH;
    }
    return pzn_pool_wgrad_sparse(dout, argmax, out, x, dW, db, R, Kin, Nout, st);
  }
  GemmArgs p = base_args(Nout, Kin, R * 32);
  p.A = dout, p.lda = Nout, p.gen = GEN_MAXPOOL, p.genArg = argmax, p.genOut = out;
  return wgrad_common(p, Kin, Nout, x, dW, db, accumulate, pzn_hip_stream(stream));
}

// Batched C[b] = alpha * op(A[b]) op(B[b]);  mode 0 = "NT": A[M,K] B[N,K];  1 = "NN": A[M,K] B[K,N];
// 2 = "TN": A[K,M] B[K,N].  Row-major, dense (leading dimension = row length).
namespace {
// Attention contraction precision (model5_b.py:67-75 QK^T, attn V and their backward products): 0 = follow the matrix-core
// path of pzn_gemm_set_precision (default: fp32 results), 1 = operands rounded to bf16 once, ONE bf16 MFMA per product,
// fp32 accumulation and fp32 softmax (the "bf16 attn with MFMA" configuration of BASELINE configs[4]).
int g_attn_precision = -1;
int attn_precision() {
  if (g_attn_precision < 0) {
    const char* e = getenv("PZN_ATTN_PRECISION");
    g_attn_precision = (e && (e[0] == 'b' || e[0] == 'B')) ? 1 : 0;
  }
  return g_attn_precision;
}

int bgemm_impl(int mode, const float* A, const float* B, float* C, int batch, int M, int N, int K, float alpha, int plain,
               hipStream_t st) {
  GemmArgs p = base_args(M, N, K);
  p.A = A, p.B = B, p.C = C, p.ldc = N, p.alpha = alpha, p.plain_bf16 = plain;
  p.sA = (long)M * K, p.sB = (long)N * K, p.sC = (long)M * N;
  if (mode == 0) {
    p.lda = K, p.ldb = K;
    launch<true, true, EPI_STORE>(p, batch, st);
  } else if (mode == 1) {
    p.lda = K, p.ldb = N;
    launch<true, false, EPI_STORE>(p, batch, st);
  } else {
    p.lda = M, p.ldb = N;
    launch<false, false, EPI_STORE>(p, batch, st);
  }
  PZN_RETURN_LAUNCH_STATUS();
}

// the batched products of the attention entry points
int attn_bgemm(int mode, const float* A, const float* B, float* C, int batch, int M, int N, int K, float alpha,
               pzn_stream_t stream) {
  return bgemm_impl(mode, A, B, C, batch, M, N, K, alpha, attn_precision() == 1, pzn_hip_stream(stream));
}
}  // namespace

PZN_EXPORT int pzn_attn_set_precision(int mode) {
  PZN_CHECK_ARG(mode == 0 || mode == 1);
  g_attn_precision = mode;
  return PZN_OK;
}
PZN_EXPORT int pzn_attn_get_precision(void) { return attn_precision(); }
int pzn_attn_precision_mode() { return attn_precision(); }

// ------------------------------------------------------------------ softmax rows (attention) --
namespace {

// One wavefront per row.  fwd: io[r,:] = softmax(io[r,:] / div)   (model5_b.py:70,73)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(float* __restrict__ io, long rows, int cols, float div) {
  const int lane = threadIdx.x & 63;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float* x = io + r * cols;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64) m = fmaxf(m, x[c] / div);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) {
    float e = expf(x[c] / div - m);
    x[c] = e;
    s += e;
  }
  s = pzn::wave_sum_f32(s);
  for (int c = lane; c < cols; c += 64) x[c] = x[c] / s;
}

// Rows of up to 64*PER columns stay in registers: one read and one write of the row (same arithmetic, same order).
template <int PER>
__global__ __launch_bounds__(256) void softmax_fwd_reg_kernel(float* __restrict__ io, long rows, int cols, float div) {
  const int lane = threadIdx.x & 63;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float* x = io + r * cols;
  float v[PER];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < cols ? x[c] / div : -INFINITY;
    m = fmaxf(m, v[i]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < cols ? expf(v[i] - m) : 0.f;
    s += v[i];
  }
  s = pzn::wave_sum_f32(s);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    if (c < cols) x[c] = v[i] / s;
  }
}

template <int PER>
__global__ __launch_bounds__(256) void softmax_bwd_reg_kernel(const float* __restrict__ attn, float* __restrict__ io,
                                                              const float* __restrict__ extra, long rows, int cols,
                                                              float div) {
  const int lane = threadIdx.x & 63;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* a = attn + r * cols;
  float* d = io + r * cols;
  const float* e = extra ? extra + r * cols : nullptr;
  float g[PER], av[PER];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    const bool ok = c < cols;
    av[i] = ok ? a[c] : 0.f;
    g[i] = ok ? d[c] + (e ? e[c] : 0.f) : 0.f;
    s += g[i] * av[i];
  }
  s = pzn::wave_sum_f32(s);
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = lane + 64 * i;
    if (c < cols) d[c] = av[i] * (g[i] - s) / div;
  }
}

void launch_softmax_fwd(float* io, long rows, int cols, float div, hipStream_t st) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (cols <= 256)
    PZN_LAUNCH(softmax_fwd_reg_kernel<4>, grid, block, 0, st, io, rows, cols, div);
  else if (cols <= 512)
    PZN_LAUNCH(softmax_fwd_reg_kernel<8>, grid, block, 0, st, io, rows, cols, div);
  else
    PZN_LAUNCH(softmax_fwd_kernel, grid, block, 0, st, io, rows, cols, div);
}

// bwd: io[r,:] (= dAttn, optionally + extra) -> dLogits = attn * (dAttn - sum(dAttn * attn)) / div
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ attn, float* __restrict__ io,
                                                          const float* __restrict__ extra, long rows, int cols,
                                                          float div) {
  const int lane = threadIdx.x & 63;
  long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* a = attn + r * cols;
  float* d = io + r * cols;
  const float* e = extra ? extra + r * cols : nullptr;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) {
    float g = d[c] + (e ? e[c] : 0.f);
    d[c] = g;
    s += g * a[c];
  }
  s = pzn::wave_sum_f32(s);
  for (int c = lane; c < cols; c += 64) d[c] = a[c] * (d[c] - s) / div;
}

void launch_softmax_bwd(const float* attn, float* io, const float* extra, long rows, int cols, float div,
                        hipStream_t st) {
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  if (cols <= 256)
    PZN_LAUNCH(softmax_bwd_reg_kernel<4>, grid, block, 0, st, attn, io, extra, rows, cols, div);
  else if (cols <= 512)
    PZN_LAUNCH(softmax_bwd_reg_kernel<8>, grid, block, 0, st, attn, io, extra, rows, cols, div);
  else
    PZN_LAUNCH(softmax_bwd_kernel, grid, block, 0, st, attn, io, extra, rows, cols, div);
}

}  // namespace

// scaled_dot_production (model5_b.py:67-75): attn[B,L,L] = softmax(q k^T / sqrt(dk)), out[B,L,dv] = attn v
PZN_EXPORT int pzn_attn_fwd_f32(const float* q, const float* k, const float* v, int B, int L, int dk, int dv,
                                float* attn, float* out, pzn_stream_t stream) {
  PZN_CHECK_ARG(q && k && v && attn && out && B > 0 && L > 0 && dk > 0 && dv > 0);
  int rc = attn_bgemm(0, q, k, attn, B, L, L, dk, 1.f, stream);
  if (rc != PZN_OK) return rc;
  long rows = (long)B * L;
  launch_softmax_fwd(attn, rows, L, sqrtf((float)dk), pzn_hip_stream(stream));
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  return attn_bgemm(1, attn, v, out, B, L, dv, L, 1.f, stream);
}

PZN_EXPORT size_t pzn_attn_bwd_workspace_bytes(int B, int L, int dk, int dv) {
  (void)dk;
  (void)dv;
  return B > 0 && L > 0 ? sizeof(float) * (size_t)B * L * L : 0;
}

// dq, dk_out, dv_out from d_out[B,L,dv] and (optional) d_attn[B,L,L]
PZN_EXPORT int pzn_attn_bwd_f32(const float* q, const float* k, const float* v, const float* attn, const float* d_out,
                                const float* d_attn, int B, int L, int dk, int dv, float* dq, float* dk_out,
                                float* dv_out, void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(q && k && v && attn && d_out && dq && dk_out && dv_out && workspace && B > 0 && L > 0 && dk > 0 &&
                dv > 0);
  float* ds = static_cast<float*>(workspace);
  int rc = attn_bgemm(2, attn, d_out, dv_out, B, L, dv, L, 1.f, stream);  // dV = attn^T dO
  if (rc != PZN_OK) return rc;
  rc = attn_bgemm(0, d_out, v, ds, B, L, L, dv, 1.f, stream);  // dAttn = dO V^T
  if (rc != PZN_OK) return rc;
  long rows = (long)B * L;
  launch_softmax_bwd(attn, ds, d_attn, rows, L, sqrtf((float)dk), pzn_hip_stream(stream));
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  rc = attn_bgemm(1, ds, k, dq, B, L, dk, L, 1.f, stream);  // dQ = dS K
  if (rc != PZN_OK) return rc;
  return attn_bgemm(2, ds, q, dk_out, B, L, dk, L, 1.f, stream);  // dK = dS^T Q
}

// ---- layerAttention as ONE unit (model5_b.py:83-101): q,k,v = Linear(x); (a, attn) = scaled_dot_production(q,k,v);
//      r = x - a;  out = x + relu(Linear_o(r)).  The two elementwise lines ride in GEMM epilogues (r: addend of the
//      attn*v product with alpha = -1; out: residual output of the weight-stationary kernel), and in the backward the
//      five contributions to dx (residual, Linear_o path, q, k, v) are summed by accumulate / residual epilogues
//      instead of one tensor add each.  Needs shapes the weight-stationary kernel takes (else PZN_EUNSUPPORTED and
//      the caller composes the block from the single entry points).
static bool attn_block_ok(int M, int E, int dk, const float* x) {
  return gemm_precision() != 0 && pzn_ws_gemm_supported(M, E, E, x, E, nullptr, false) &&
         pzn_ws_gemm_supported(M, E, dk, x, dk, nullptr, false) && pzn_ws_gemm_supported(M, dk, E, x, E, nullptr, false);
}

PZN_EXPORT int pzn_attn_block_fwd_f32(const float* x, const float* Wq, const float* bq, const float* Wk, const float* bk,
                                      const float* Wv, const float* bv, const float* Wo, const float* bo, int B, int L,
                                      int E, int dk, float* q, float* k, float* v, float* attn, float* r, float* yo,
                                      float* out, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && Wq && Wk && Wv && Wo && q && k && v && attn && r && yo && out && B > 0 && L > 0 && E > 0 && dk > 0);
  const int M = B * L;
  if (!attn_block_ok(M, E, dk, x)) return PZN_EUNSUPPORTED;
  hipStream_t st = pzn_hip_stream(stream);
  int rc = PZN_EUNSUPPORTED;
  if (bq && bk && bv) {  // the three projections share x: one launch over the concatenated column slices
    const float* const Ws[3] = {Wq, Wk, Wv};
    const float* const bs[3] = {bq, bk, bv};
    float* const Cs[3] = {q, k, v};
    const int Ns[3] = {dk, dk, E};
    rc = pzn_ws_gemm3(x, E, Ws, bs, Cs, Ns, M, E, st);
  }
  if (rc == PZN_EUNSUPPORTED) {
    rc = pzn_linear_fwd_f32(x, Wq, bq, M, E, dk, 0, q, stream);
    if (rc == PZN_OK) rc = pzn_linear_fwd_f32(x, Wk, bk, M, E, dk, 0, k, stream);
    if (rc == PZN_OK) rc = pzn_linear_fwd_f32(x, Wv, bv, M, E, E, 0, v, stream);
  }
  if (rc == PZN_OK) rc = attn_bgemm(0, q, k, attn, B, L, L, dk, 1.f, stream);
  if (rc != PZN_OK) return rc;
  const long rows = (long)B * L;
  launch_softmax_fwd(attn, rows, L, sqrtf((float)dk), st);
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  {  // r = x - attn v
    GemmArgs p = base_args(L, E, L);
    p.A = attn, p.lda = L, p.B = v, p.ldb = E, p.C = r, p.ldc = E, p.alpha = -1.f, p.addend = x;
    p.sA = (long)L * L, p.sB = (long)L * E, p.sC = (long)L * E;
    p.plain_bf16 = attn_precision() == 1;
    launch<true, false, EPI_STORE>(p, B, st);
    if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  }
  // yo = relu(r Wo^T + bo) (kept for the backward's gate), out = x + yo
  return pzn_ws_gemm_ex(r, E, Wo, E, 0, yo, E, M, E, E, bo, 1, nullptr, nullptr, nullptr, nullptr, 0, 0, x, out, 0, st);
}

// workspace: dd[M,E] | ds[B,L,L] | dq[M,dk] | dk[M,dk] | dv[M,E]
PZN_EXPORT size_t pzn_attn_block_bwd_workspace_bytes(int B, int L, int E, int dk) {
  if (B <= 0 || L <= 0 || E <= 0 || dk <= 0) return 0;
  const size_t M = (size_t)B * L;
  return sizeof(float) * (2 * M * E + (size_t)B * L * L + 2 * M * dk);
}

PZN_EXPORT int pzn_attn_block_bwd_f32(const float* x, const float* Wq, const float* Wk, const float* Wv, const float* Wo,
                                      const float* q, const float* k, const float* v, const float* attn, const float* r,
                                      const float* yo, const float* dout, const float* dattn, int B, int L, int E, int dk,
                                      void* workspace, float* dx, float* dWq, float* dbq, float* dWk, float* dbk,
                                      float* dWv, float* dbv, float* dWo, float* dbo, int accumulate,
                                      pzn_stream_t stream) {
  PZN_CHECK_ARG(x && Wq && Wk && Wv && Wo && q && k && v && attn && r && yo && dout && workspace && dx && dWq && dbq &&
                dWk && dbk && dWv && dbv && dWo && dbo && B > 0 && L > 0 && E > 0 && dk > 0);
  PZN_CHECK_ARG((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (E & 3) == 0 && (dk & 3) == 0 && (L & 3) == 0);
  const int M = B * L;
  if (!attn_block_ok(M, E, dk, x)) return PZN_EUNSUPPORTED;
  hipStream_t st = pzn_hip_stream(stream);
  float* dd = static_cast<float*>(workspace);
  float* ds = dd + (size_t)M * E;
  float* dq = ds + (size_t)B * L * L;
  float* dkk = dq + (size_t)M * dk;
  float* dvv = dkk + (size_t)M * dk;
  // Linear_o: dd = (dout . [yo > 0]) Wo;  dWo, dbo
  int rc = pzn_linear_dgrad_f32(dout, yo, Wo, M, E, E, nullptr, dd, stream);
  if (rc == PZN_OK) rc = pzn_linear_wgrad_f32(dout, yo, r, M, E, E, dWo, dbo, accumulate, stream);
  // attention with d(values) = -dd  (r = x - a)
  if (rc == PZN_OK) rc = attn_bgemm(2, attn, dd, dvv, B, L, E, L, -1.f, stream);   // dV = attn^T dO
  if (rc == PZN_OK) rc = attn_bgemm(0, dd, v, ds, B, L, L, E, -1.f, stream);        // dAttn = dO V^T
  if (rc != PZN_OK) return rc;
  const long rows = (long)B * L;
  launch_softmax_bwd(attn, ds, dattn, rows, L, sqrtf((float)dk), st);
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  rc = attn_bgemm(1, ds, k, dq, B, L, dk, L, 1.f, stream);                           // dQ = dS K
  if (rc == PZN_OK) rc = attn_bgemm(2, ds, q, dkk, B, L, dk, L, 1.f, stream);        // dK = dS^T Q
  // dx = dv Wv + dd;  += dk Wk;  += dq Wq + dout   (no tensor adds)
  if (rc == PZN_OK)
    rc = pzn_ws_gemm_ex(dvv, E, Wv, E, 1, dx, E, M, E, E, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, dd, nullptr,
                        0, st);
  if (rc == PZN_OK)
    rc = pzn_ws_gemm_ex(dkk, dk, Wk, E, 1, dx, E, M, E, dk, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr,
                        nullptr, 1, st);
  if (rc == PZN_OK)
    rc = pzn_ws_gemm_ex(dq, dk, Wq, E, 1, dx, E, M, E, dk, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, dout,
                        nullptr, 1, st);
  if (rc != PZN_OK) return rc;
  // weight gradients of the three projections: they share x, so one launch over the concatenated N axis
  {
    const float* const dys[3] = {dq, dkk, dvv};
    const int ns[3] = {dk, dk, E};
    float* const dWs[3] = {dWq, dWk, dWv};
    float* const dbs[3] = {dbq, dbk, dbv};
    if (!accumulate) {
      for (int i = 0; i < 3; ++i) {
        if (pzn_zero_async(dWs[i], (size_t)ns[i] * E, st) != PZN_OK) return PZN_ELAUNCH;
        if (pzn_zero_async(dbs[i], (size_t)ns[i], st) != PZN_OK) return PZN_ELAUNCH;
      }
    }
    rc = pzn_df_wgrad3(dys, ns, dWs, dbs, x, E, M, E, st);
    if (rc != PZN_EUNSUPPORTED) return rc;
  }
  rc = pzn_linear_wgrad_f32(dq, nullptr, x, M, E, dk, dWq, dbq, 1, stream);   // (already zeroed when not accumulating)
  if (rc == PZN_OK) rc = pzn_linear_wgrad_f32(dkk, nullptr, x, M, E, dk, dWk, dbk, 1, stream);
  if (rc == PZN_OK) rc = pzn_linear_wgrad_f32(dvv, nullptr, x, M, E, E, dWv, dbv, 1, stream);
  return rc;
}

// Weight gradients of one layerAttention block from what the chained kernels (attnfused.hip) leave in memory:
//   dWo += dz^T t, dbo += sum dz;   dWq/k/v += dq/dk/dv^T x, dbq/k/v += column sums   (overwritten when !accumulate)
PZN_EXPORT int pzn_attn_fused_wgrads(const float* dz, const float* t, const float* dq, const float* dkk, const float* dvv,
                                     const float* x, int M, int E, int dk, float* dWq, float* dbq, float* dWk, float* dbk,
                                     float* dWv, float* dbv, float* dWo, float* dbo, int accumulate, pzn_stream_t stream) {
  return pzn_attn_fused_wgrads_ws(dz, t, dq, dkk, dvv, x, M, E, dk, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo, accumulate, nullptr, 0,
                                  stream);
}

int pzn_attn_fused_wgrads_ws(const float* dz, const float* t, const float* dq, const float* dkk, const float* dvv, const float* x,
                             int M, int E, int dk, float* dWq, float* dbq, float* dWk, float* dbk, float* dWv, float* dbv,
                             float* dWo, float* dbo, int accumulate, void* ws, size_t ws_bytes, pzn_stream_t stream) {
  PZN_CHECK_ARG(dz && t && dq && dkk && dvv && x && dWq && dbq && dWk && dbk && dWv && dbv && dWo && dbo && M > 0 && E > 0 &&
                dk > 0);
  hipStream_t st = pzn_hip_stream(stream);
  if (E == 256 && dk == 64 && (M & 63) == 0 && gemm_precision() != 0)      // the encoder's shape: one launch (csrc/attnwgrad.hip)
    return pzn_attn_wgrad_tiled(dz, t, dq, dkk, dvv, x, M, dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo, accumulate, ws, ws_bytes, st);
  int rc = pzn_linear_wgrad_f32(dz, nullptr, t, M, E, E, dWo, dbo, accumulate, stream);
  if (rc != PZN_OK) return rc;
  const float* const dys[3] = {dq, dkk, dvv};
  const int ns[3] = {dk, dk, E};
  float* const dWs[3] = {dWq, dWk, dWv};
  float* const dbs[3] = {dbq, dbk, dbv};
  if (!accumulate) {
    for (int i = 0; i < 3; ++i) {
      if (pzn_zero_async(dWs[i], (size_t)ns[i] * E, st) != PZN_OK) return PZN_ELAUNCH;
      if (pzn_zero_async(dbs[i], (size_t)ns[i], st) != PZN_OK) return PZN_ELAUNCH;
    }
  }
  rc = gemm_precision() != 0 ? pzn_df_wgrad3(dys, ns, dWs, dbs, x, E, M, E, st) : PZN_EUNSUPPORTED;
  if (rc != PZN_EUNSUPPORTED) return rc;
  rc = pzn_linear_wgrad_f32(dq, nullptr, x, M, E, dk, dWq, dbq, 1, stream);
  if (rc == PZN_OK) rc = pzn_linear_wgrad_f32(dkk, nullptr, x, M, E, dk, dWk, dbk, 1, stream);
  if (rc == PZN_OK) rc = pzn_linear_wgrad_f32(dvv, nullptr, x, M, E, E, dWv, dbv, 1, stream);
  return rc;
}

// Shared MLP + max over the K = 32 neighbours (model5_b.py:452-454 / 459-461):
//   h = relu(x W1^T + b1) [R*32, C1] (kept for the backward);  out[R,C2] = max_k relu(h W2^T + b2), argmax
PZN_EXPORT int pzn_sharedmlp_max_fwd_f32(const float* x, const float* W1, const float* b1, const float* W2,
                                         const float* b2, int R, int C0, int C1, int C2, float* h, float* out,
                                         int32_t* argmax, pzn_stream_t stream) {
  PZN_CHECK_ARG(h);
  int rc = pzn_linear_fwd_f32(x, W1, b1, R * 32, C0, C1, 1, h, stream);
  if (rc != PZN_OK) return rc;
  return pzn_linear_maxpool_fwd_f32(h, W2, b2, R, C1, C2, out, argmax, stream);
}

// Second-layer backward of the pooled shared MLP (rows h in memory: the grouped-row composition): dh (ReLU-masked by h) from
// the generated-operand GEMM, dW2 / db2 from the sparse pass of poolbwd.hip where the shape allows (else the GEMM form).
static int pool_layer_bwd(const float* dout, const int32_t* argmax, const float* out, const float* W2, const float* h,
                          int R, int C1, int C2, float* dh_ws, float* dW2, float* db2, int accumulate,
                          pzn_stream_t stream) {
  int rc = pzn_linear_maxpool_dgrad_f32(dout, argmax, out, W2, R, C1, C2, h, dh_ws, stream);
  if (rc != PZN_OK) return rc;
  return pzn_linear_maxpool_wgrad_f32(dout, argmax, out, h, R, C1, C2, dW2, db2, accumulate, stream);
}

// Set-abstraction level with the first layer per point AND never in memory (model5_b.py:449-454 / :456-461): the
// activation rows relu(Pp[idx] + Q[g]) are generated inside the matrix-core kernel's operand loader (wsgemm.hip, GATH).
//   Pp[B*N, C1] = feat W1[:,3:]^T + W1[:,0:3] xyz,  Q[B*S, C1] = b1 - W1[:,0:3] new_xyz   (pzn_sa_prep_f32)
//   out[B*S, C2] = max_k relu(relu(Pp[idx[., k]] + Q) W2^T + b2),  argmax[B*S, C2]
PZN_EXPORT int pzn_sa_level_fwd_f32(const float* Pp, const float* Q, const int64_t* idx, const float* W2, const float* b2,
                                    int B, int N, int S, int C1, int C2, float* out, int32_t* argmax,
                                    pzn_stream_t stream) {
  PZN_CHECK_ARG(Pp && Q && idx && W2 && out && argmax && B > 0 && N > 0 && S > 0 && C1 > 0 && C2 > 0);
  PZN_CHECK_ARG((long)B * S * 32 < 2147483647L);
  if (gemm_precision() == 0) return PZN_EUNSUPPORTED;      // (the exact-fp32 engine has no generated-operand loader)
  return pzn_ws_gemm_gather_maxpool(Pp, Q, idx, W2, b2, B * S, N, S, C1, C2, out, argmax, pzn_hip_stream(stream));
}

// The same level with a caller-owned workspace (pzn_sa_level_fwd_workspace_bytes; 0 = this shape has no streamed
// form): the streamed-weights kernel of salevel.hip generates every row ONCE per group (the weight-stationary kernel
// regenerates it once per 64-column slice of W2) and streams the split weight planes through LDS instead.
PZN_EXPORT size_t pzn_sa_level_fwd_workspace_bytes(int C1, int C2) { return pzn_sa_level_stream_workspace_bytes(C1, C2); }

PZN_EXPORT int pzn_sa_level_fwd_ws_f32(const float* Pp, const float* Q, const int64_t* idx, const float* W2,
                                       const float* b2, int B, int N, int S, int C1, int C2, float* out,
                                       int32_t* argmax, void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(Pp && Q && idx && W2 && out && argmax && B > 0 && N > 0 && S > 0 && C1 > 0 && C2 > 0);
  PZN_CHECK_ARG((long)B * S * 32 < 2147483647L);
  if (gemm_precision() == 0) return PZN_EUNSUPPORTED;
  if (workspace && b2) {
    int rc = pzn_sa_level_stream(Pp, Q, idx, W2, b2, B * S, N, S, C1, C2, out, argmax, workspace, pzn_hip_stream(stream));
    if (rc != PZN_EUNSUPPORTED) return rc;
  }
  return pzn_ws_gemm_gather_maxpool(Pp, Q, idx, W2, b2, B * S, N, S, C1, C2, out, argmax, pzn_hip_stream(stream));
}

// The two halves of pzn_sa_level_fwd_ws_f32 as separate entry points (as pzn_attn_fused_prep_weights is for the attention
// blocks): the split of W2 into the streamed kernel's plane image, and the level on a workspace that already holds it.
// PZN_EUNSUPPORTED where the shape has no streamed form (use pzn_sa_level_fwd_ws_f32 then).
PZN_EXPORT int pzn_sa_level_prep_weights_f32(const float* W2, int C1, int C2, void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(W2 && C1 > 0 && C2 > 0);
  if (gemm_precision() == 0) return PZN_EUNSUPPORTED;
  return pzn_sa_level_stream_pack(W2, C1, C2, workspace, pzn_hip_stream(stream));
}

PZN_EXPORT int pzn_sa_level_fwd_packed_f32(const float* Pp, const float* Q, const int64_t* idx, const float* b2, int B, int N,
                                           int S, int C1, int C2, float* out, int32_t* argmax, const void* workspace,
                                           pzn_stream_t stream) {
  PZN_CHECK_ARG(Pp && Q && idx && b2 && out && argmax && workspace && B > 0 && N > 0 && S > 0 && C1 > 0 && C2 > 0);
  PZN_CHECK_ARG((long)B * S * 32 < 2147483647L);
  if (gemm_precision() == 0) return PZN_EUNSUPPORTED;
  return pzn_sa_level_stream(Pp, Q, idx, nullptr, b2, B * S, N, S, C1, C2, out, argmax, const_cast<void*>(workspace),
                             pzn_hip_stream(stream), 1);
}

// Backward: dh_ws is [R*32, C1] scratch; dx may be NULL.  dW*, db* are overwritten.
PZN_EXPORT int pzn_sharedmlp_max_bwd_f32(const float* x, const float* W1, const float* W2, const float* h,
                                         const float* out, const int32_t* argmax, const float* dout, int R, int C0,
                                         int C1, int C2, float* dh_ws, float* dx, float* dW1, float* db1, float* dW2,
                                         float* db2, int accumulate, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && W1 && W2 && h && out && argmax && dout && dh_ws && dW1 && db1 && dW2 && db2);
  int rc = pool_layer_bwd(dout, argmax, out, W2, h, R, C1, C2, dh_ws, dW2, db2, accumulate, stream);
  if (rc != PZN_OK) return rc;
  rc = pzn_linear_wgrad_f32(dh_ws, nullptr, x, R * 32, C0, C1, dW1, db1, accumulate, stream);
  if (rc != PZN_OK) return rc;
  if (dx) rc = pzn_linear_dgrad_f32(dh_ws, nullptr, W1, R * 32, C0, C1, nullptr, dx, stream);
  return rc;
}

// 0 = exact-fp32 MFMA everywhere, 1 = bf16x3 split precision everywhere, 2 = auto (default; see gemm.hip).
// Process-wide setting read at launch time; also settable with PZN_GEMM_PRECISION=f32|x3|auto.
PZN_EXPORT int pzn_gemm_set_precision(int mode) {
  PZN_CHECK_ARG(mode >= 0 && mode <= 2);
  g_precision = mode;
  return PZN_OK;
}
PZN_EXPORT int pzn_gemm_get_precision(void) { return gemm_precision(); }
