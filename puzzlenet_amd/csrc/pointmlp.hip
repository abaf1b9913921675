// pointmlp.hip — the per-point MLP chains of the boundary heads (model5_b.py:571-592, 738-754) as ONE launch each way,
// gfx950 only.
//
// A chain is three Linear layers on [M = B N, 64] rows with ReLU after the first two:
//     MLPLocalPreFpc / MLPLocalPreRpc   64 -> 64 -> 64 -> 64
//     MLPFpcb / MLPRpcb                 (64 global | 64 local) -> 64 -> 32 -> 2, the global half of the first layer folded
//                                       into a per-cloud bias (ops._CatGlobalLinearRelu)
// Layer by layer these are 131 072-row launches that each read and write their operands at a third of the HBM rate
// (25 us against a 13 us floor).  Here a wavefront keeps its 32 rows in registers through the whole
// chain — transposed, the point on the MFMA lane as in attnfused.hip, so that an accumulator tile is the B operand of the
// next layer without leaving registers — and the rows are read once and every activation written once.  Split precision
// (bf16x3, six MFMAs per product, fp32-GEMM accuracy) as everywhere; the weights (<= 3 x 24 KB of planes) are split by
// every workgroup into LDS when it starts.
//
// Backward (point_mlp3_bwd_kernel): the same walk in reverse for the input gradients, and the three weight gradients in
// the same pass: dW[o][i] = sum_p g[p][o] x[p][i] has the POINT as the k index, so both operands are read back from the
// wavefront's row-major staging tile in LDS (8 strided dwords per fragment) and split there; a wavefront accumulates its
// tiles' contributions in registers (one wavefront per SIMD: 12 accumulator tiles + the working set), the workgroup's
// four wavefronts meet in LDS in a fixed order, and a second launch adds the workgroups' partial sums, again in a fixed
// order: the result does not depend on timing.
#include <stdlib.h>

#include <type_traits>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

#include "pzn_mfma.h"

#define ZERO16(T) _Pragma("unroll") for (int z_ = 0; z_ < 16; ++z_) (T)[z_] = 0.f

constexpr int PM_C = 64;                 // channels of the chain's input and of its first hidden layer
constexpr int PM_LD = 68;                // dwords per staged row (68 = 4 mod 32: 16-byte accesses of 8 lanes tile the banks)
constexpr int PM_STG = 32 * PM_LD * 4;   // bytes of one staging tile (32 rows x up to 64 features)

// D layout of a 32x32 accumulator tile: lane (r = l & 31, h = l >> 5) register i holds [feature 32 ft + (i & 3) + 8 (i >> 2) + 4 h][point r]
template <int NFT>
__device__ __forceinline__ void pm_put(float* stg, const floatx16* x, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(stg + r * PM_LD + 32 * ft + 8 * g + 4 * h) =
          make_float4(x[ft][4 * g], x[ft][4 * g + 1], x[ft][4 * g + 2], x[ft][4 * g + 3]);
}
template <int NFT>
__device__ __forceinline__ void pm_get(const float* stg, floatx16* x, int lane) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(stg + r * PM_LD + 32 * ft + 8 * g + 4 * h);
      x[ft][4 * g] = v.x, x[ft][4 * g + 1] = v.y, x[ft][4 * g + 2] = v.z, x[ft][4 * g + 3] = v.w;
    }
}

// 32 dense rows of F floats (one contiguous block of 128 F bytes) <-> registers: F / 8 16-byte pieces per lane
// (a plain vector type: an array of HIP's float4 — a struct around a union — stays in scratch memory across the loop)
typedef float pm_f4 __attribute__((ext_vector_type(4)));
template <int F>
__device__ __forceinline__ void pm_rows_load(const float* g, pm_f4 (&v)[F / 8], int lane) {
  const pm_f4* g4 = reinterpret_cast<const pm_f4*>(g);
#pragma unroll
  for (int i = 0; i < F / 8; ++i) v[i] = g4[i * 64 + lane];
}
template <int F>
__device__ __forceinline__ void pm_rows_to_stage(float* stg, const pm_f4 (&v)[F / 8], int lane) {
#pragma unroll
  for (int i = 0; i < F / 8; ++i) {
    const int q = i * 64 + lane, rr = q / (F / 4), c4 = q % (F / 4);
    *reinterpret_cast<pm_f4*>(stg + rr * PM_LD + 4 * c4) = v[i];
  }
}
template <int F>
__device__ __forceinline__ void pm_stage_to_rows(const float* stg, float* g, int lane) {
  pm_f4* g4 = reinterpret_cast<pm_f4*>(g);
  pm_f4 v[F / 8];
#pragma unroll
  for (int i = 0; i < F / 8; ++i) {
    const int q = i * 64 + lane, rr = q / (F / 4), c4 = q % (F / 4);
    v[i] = *reinterpret_cast<const pm_f4*>(stg + rr * PM_LD + 4 * c4);
  }
#pragma unroll
  for (int i = 0; i < F / 8; ++i) g4[i * 64 + lane] = v[i];
}
// accumulator tiles -> dense rows of 32 NFT floats (through the wavefront's staging tile)
template <int NFT>
__device__ __forceinline__ void pm_store_tiles(float* g, const floatx16* x, float* stg, int lane) {
  pm_put<NFT>(stg, x, lane);
  pzn::wave_lds_sync();
  pm_stage_to_rows<32 * NFT>(stg, g, lane);
  pzn::wave_lds_sync();
}

// Plane image of a weight matrix in LDS: [k-step][plane][row tile][lane][8 bf16]; the chunk of lane (r, h) holds
// A[32 rt + r][16 ks + perm(h, j)], perm(h, j) = 8 (j >> 2) + 4 h + (j & 3) (the order in which registers 8 s .. 8 s + 7 of
// an accumulator tile come out as a B fragment, see attnfused.hip).  A[row][k] = W[row * ldw + k], or W[k * ldw + row]
// (TRANS: the input-gradient product), zero outside rows x K.
template <int RT, int KS, bool TRANS>
__device__ __forceinline__ void pm_build_image(unsigned char* img, const float* __restrict__ W, int ldw, int rows, int K,
                                               int tid, int nthreads) {
  for (int c = tid; c < KS * RT * 64; c += nthreads) {
    const int l = c & 63, rt = (c >> 6) % RT, ks = (c >> 6) / RT;
    const int row = 32 * rt + (l & 31), h = l >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
      v[j] = (row < rows && k < K) ? (TRANS ? W[(long)k * ldw + row] : W[(long)row * ldw + k]) : 0.f;
    }
    bf16x8 b[3];
    split8(v, b);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(img + ((ks * 3 + p) * RT + rt) * 1024 + l * 16) = b[p];
  }
}

// acc[rt] += A_rt B over KS k-steps; B = the tiles `in` (k-step ks = registers 8 (ks & 1) .. of tile ks >> 1)
template <int RT, int KS>
__device__ __forceinline__ void pm_layer(floatx16 (&acc)[RT], const floatx16* in, const unsigned char* img, int lane) {
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bf16x8 b[3];
    make_b(in[ks >> 1], ks & 1, b);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      bf16x8 a[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(img + ((ks * 3 + p) * RT + rt) * 1024 + lane * 16);
      acc[rt] = mma6(a, b, acc[rt]);
    }
  }
}

// accumulator tiles <- bias (the row of a transposed result is the output feature); n = valid features
template <int RT>
__device__ __forceinline__ void pm_bias(floatx16 (&acc)[RT], const float* __restrict__ bias, int n, int h) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f0 = 32 * rt + 8 * g + 4 * h;
      if (f0 + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(bias + f0);
        acc[rt][4 * g] = v.x, acc[rt][4 * g + 1] = v.y, acc[rt][4 * g + 2] = v.z, acc[rt][4 * g + 3] = v.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[rt][4 * g + e] = f0 + e < n ? bias[f0 + e] : 0.f;
      }
    }
}

template <int RT>
__device__ __forceinline__ void pm_relu(floatx16 (&x)[RT]) {
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int i = 0; i < 16; ++i) x[rt][i] = fmaxf(x[rt][i], 0.f);
}

struct PmFwdArgs {
  const float* x;          // [M, 64]
  const float* w1;         // [64, ldw1] (a column slice of a wider matrix is fine)
  const float* b1;         // [64], or [clouds, 64] with b1_stride = 64
  const float *w2, *b2;    // [C2, 64], [C2]
  const float *w3, *b3;    // [C3, C2], [C3]
  float *h1, *h2, *y;      // [M, 64], [M, C2], [M, C3]
  int ldw1, b1_stride, rows_per_cloud, ntiles;
};

constexpr int PM_FWD_WAVES = 8;

template <int C2, int C3>
constexpr int pm_fwd_lds() {
  return (4 * 3 * 2 + 4 * 3 * (C2 / 32) + (C2 / 16) * 3 * (C3 >= 32 ? C3 / 32 : 1)) * 1024 + PM_FWD_WAVES * PM_STG;
}

template <int C2, int C3>
__global__ __launch_bounds__(PM_FWD_WAVES * 64) void point_mlp3_fwd_kernel(PmFwdArgs a) {
  constexpr int T2 = C2 / 32, T3 = C3 >= 32 ? C3 / 32 : 1, K3 = C2 / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* img1 = lds;
  unsigned char* img2 = img1 + 4 * 3 * 2 * 1024;
  unsigned char* img3 = img2 + 4 * 3 * T2 * 1024;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5;
  float* stg = reinterpret_cast<float*>(img3 + K3 * 3 * T3 * 1024 + wave * PM_STG);
  const int stride = gridDim.x * PM_FWD_WAVES;
  int tile = blockIdx.x * PM_FWD_WAVES + wave;
  pm_f4 pre[8];
  pm_rows_load<64>(a.x + (long)min(tile, a.ntiles - 1) * 32 * PM_C, pre, lane);
  pm_build_image<2, 4, false>(img1, a.w1, a.ldw1, PM_C, PM_C, tid, PM_FWD_WAVES * 64);
  pm_build_image<T2, 4, false>(img2, a.w2, PM_C, C2, PM_C, tid, PM_FWD_WAVES * 64);
  pm_build_image<T3, K3, false>(img3, a.w3, C2, C3, C2, tid, PM_FWD_WAVES * 64);
  __syncthreads();
  while (tile < a.ntiles) {
    const long row0 = (long)tile * 32;
    floatx16 X[2];
    pm_rows_to_stage<64>(stg, pre, lane);
    pzn::wave_lds_sync();
    pm_get<2>(stg, X, lane);
    pzn::wave_lds_sync();
    const int next = tile + stride;
    pm_rows_load<64>(a.x + (long)min(next, a.ntiles - 1) * 32 * PM_C, pre, lane);   // (the last tile again past the end)
    // layer 1
    floatx16 H1[2];
    pm_bias<2>(H1, a.b1 + (a.b1_stride ? (row0 / a.rows_per_cloud) * a.b1_stride : 0), PM_C, h);
    pm_layer<2, 4>(H1, X, img1, lane);
    pm_relu<2>(H1);
    pm_store_tiles<2>(a.h1 + row0 * PM_C, H1, stg, lane);
    // layer 2
    floatx16 H2[T2];
    pm_bias<T2>(H2, a.b2, C2, h);
    pm_layer<T2, 4>(H2, H1, img2, lane);
    pm_relu<T2>(H2);
    pm_store_tiles<T2>(a.h2 + row0 * C2, H2, stg, lane);
    // layer 3 (no ReLU)
    floatx16 Y[T3];
    pm_bias<T3>(Y, a.b3, C3, h);
    pm_layer<T3, K3>(Y, H2, img3, lane);
    if constexpr (C3 >= 32) {
      pm_store_tiles<T3>(a.y + row0 * C3, Y, stg, lane);
    } else {   // C3 = 2: features 0, 1 are registers 0, 1 of the lanes with h = 0
      static_assert(C3 == 2, "narrow output");
      if (h == 0) *reinterpret_cast<float2*>(a.y + (row0 + lane) * 2) = make_float2(Y[0][0], Y[0][1]);
    }
    tile = next;
  }
}

// ================================================================================================================
// backward
// Row-major staging tiles WITHOUT padding, filled by LDS-DMA (no register staging, the load of the next operand flies
// while the current one is multiplied): a tile is 32 rows of F floats, the 16-byte chunk c of row p at chunk position
// c ^ (p & 7) (the DMA writes lane l's 16 bytes at 16 l, the lane is free to FETCH any chunk of the 1 KB group: it
// fetches the chunk that belongs at its position).  Conflict-free for the three access shapes: a lane per row (16
// bytes of chunk c), a row per 16 / 8 lanes (row-contiguous), a lane per feature (one dword of row p).
template <int F>
__device__ __forceinline__ int pm_sw(int p, int f) {      // dword index of element (row p, feature f)
  return p * F + 4 * ((f >> 2) ^ (p & 7)) + (f & 3);
}
template <int F>
__device__ __forceinline__ void pm_dma_tile(const float* __restrict__ g, float* stg, int lane) {
  constexpr int RPI = 1024 / (4 * F);      // rows per DMA instruction (1 KB)
  constexpr int CPR = F / 4;               // chunks per row
#pragma unroll
  for (int i = 0; i < 32 / RPI; ++i) {
    const int row = i * RPI + lane / CPR, pos = lane % CPR;
    const float* src = g + row * F + 4 * (pos ^ (row & 7));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(stg + i * 256), 16, 0, 0);
  }
}
__device__ __forceinline__ void pm_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The lane's share of those addresses, computed once per kernel (left in the loop, the swizzle arithmetic was a fifth of the
// backward pass's instructions); everything else is an immediate offset.  All in dwords.
template <int F>
struct PmOff {
  int fo[8];   // fragment of a sum over points: point 8 h + j, feature r          (+ 16 s F + 32 ot)
  int go[4];   // D layout <-> tile: row r, chunk 2 g + h                           (+ 32 ft)
  int co[8];   // column sums, lane = feature: point j                              (+ 8 a F)
  __device__ __forceinline__ PmOff(int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      fo[j] = (8 * h + j) * F + 4 * ((r >> 2) ^ j) + (r & 3);
      co[j] = j * F + 4 * (((lane >> 2) & (F / 4 - 1)) ^ j) + (lane & 3);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) go[g] = r * F + 4 * ((2 * g + h) ^ (r & 7));
  }
};

template <int NFT, int F>
__device__ __forceinline__ void pm_put_sw(float* stg, const floatx16* x, const PmOff<F>& o) {
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<pm_f4*>(stg + o.go[g] + 32 * ft) =
          pm_f4{x[ft][4 * g], x[ft][4 * g + 1], x[ft][4 * g + 2], x[ft][4 * g + 3]};
}
template <int NFT, int F>
__device__ __forceinline__ void pm_get_sw(const float* stg, floatx16* x, const PmOff<F>& o) {
#pragma unroll
  for (int ft = 0; ft < NFT; ++ft)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const pm_f4 v = *reinterpret_cast<const pm_f4*>(stg + o.go[g] + 32 * ft);
      x[ft][4 * g] = v[0], x[ft][4 * g + 1] = v[1], x[ft][4 * g + 2] = v[2], x[ft][4 * g + 3] = v[3];
    }
}
// staging tile -> 32 dense rows of F floats
template <int F>
__device__ __forceinline__ void pm_sw_to_rows(const float* stg, float* g, int lane) {
  constexpr int CPR = F / 4;
  pm_f4* g4 = reinterpret_cast<pm_f4*>(g);
  pm_f4 v[F / 8];
#pragma unroll
  for (int i = 0; i < F / 8; ++i) {
    const int q = i * 64 + lane, rr = q / CPR, c4 = q % CPR;
    v[i] = *reinterpret_cast<const pm_f4*>(stg + rr * F + 4 * (c4 ^ (rr & 7)));
  }
#pragma unroll
  for (int i = 0; i < F / 8; ++i) g4[i * 64 + lane] = v[i];
}
// fragment of a product that sums over the POINTS (k-step s = points 16 s .. 16 s + 15): lane (r, h) takes feature
// 32 ot + r of points 16 s + 8 h + j, j = 0..7, from a row-major tile; three planes
template <int F>
__device__ __forceinline__ void pm_point_frag(const float* stg, int ot, int s, const PmOff<F>& o, bf16x8 (&b)[3]) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = stg[o.fo[j] + 16 * s * F + 32 * ot];
  split8(v, b);
}
// column sums of a tile: lane = feature (< F); four chains
template <int F>
__device__ __forceinline__ float pm_colsum(const float* stg, int lane, const PmOff<F>& o) {
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (F == 64 || lane < F) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j & 3] += stg[o.co[j] + 8 * a * F];
  }
  return (s[0] + s[1]) + (s[2] + s[3]);
}

struct PmBwdArgs {
  const float* dy;                    // [M, C3]
  const float *x, *h1, *h2;           // [M, 64], [M, 64], [M, C2]
  const float *w1, *w2, *w3;          // [64, ldw1], [C2, 64], [C3, C2]
  float* dx;                          // [M, 64]
  float* part_w;                      // [workgroups][NACC tiles][16][64]: accumulator tiles as they stand
  float* part_b;                      // [workgroups * 4 wavefronts][3][64]: column sums of the three gated gradients
  int ldw1, ntiles, tiles_per_wave;
};

template <int C2, int C3>
struct PmShape {
  static constexpr int T2 = C2 / 32, T3 = C3 >= 32 ? C3 / 32 : 1;
  static constexpr int K3 = C3 >= 16 ? C3 / 16 : 1;             // k-steps of the input-gradient product of layer 3 (over o)
  static constexpr int NACC = 2 * 2 + T2 * 2 + T3 * T2;         // accumulator tiles of dW1, dW2, dW3
  static constexpr int IMG1 = 4 * 3 * 2 * 1024;                 // W1^T: rows i (64), k = o (64)
  static constexpr int IMG2 = (C2 / 16) * 3 * 2 * 1024;         // W2^T: rows i (64), k = o (C2)
  static constexpr int IMG3 = K3 * 3 * T2 * 1024;               // W3^T: rows i (C2), k = o (C3, padded to 16)
  static constexpr int STG = 2 * 32 * 64 * 4;                   // two staging tiles per wavefront
  static constexpr int RED = NACC * 16 * 64 * 4;
  static constexpr int LDS = (IMG1 + IMG2 + IMG3 > RED ? IMG1 + IMG2 + IMG3 : RED) + 4 * STG;
};

template <int C2, int C3>
__global__ __launch_bounds__(256, 1) void point_mlp3_bwd_kernel(PmBwdArgs a) {
  using S = PmShape<C2, C3>;
  constexpr int T2 = S::T2, T3 = S::T3;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* img1 = lds;
  unsigned char* img2 = img1 + S::IMG1;
  unsigned char* img3 = img2 + S::IMG2;
  constexpr int IMGS = S::IMG1 + S::IMG2 + S::IMG3 > S::RED ? S::IMG1 + S::IMG2 + S::IMG3 : S::RED;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, r = lane & 31;
  float* P = reinterpret_cast<float*>(lds + IMGS + wave * S::STG);     // gradient tile (dy, G2, G1, dx)
  float* Q = P + 32 * 64;                                              // activation tile (h2, h1, x)
  const int wg = blockIdx.x * 4 + wave;                                // this wavefront's slot
  const int t0 = wg * a.tiles_per_wave, t1 = min(a.ntiles, t0 + a.tiles_per_wave);

  floatx16 dW1[2][2], dW2[T2][2], dW3[T3][T2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) ZERO16(dW1[i][j]);
#pragma unroll
  for (int i = 0; i < T2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) ZERO16(dW2[i][j]);
#pragma unroll
  for (int i = 0; i < T3; ++i)
#pragma unroll
    for (int j = 0; j < T2; ++j) ZERO16(dW3[i][j]);
  float db1 = 0.f, db2 = 0.f, db3 = 0.f;      // lane = feature
  const PmOff<64> o1(lane);                    // tiles of 64 features (x, h1, G1, dx)
  const PmOff<C2> o2(lane);                    // h2, G2
  const PmOff<(C3 >= 32 ? C3 : 32)> o3(lane);  // dy (the two-column form is read directly)

  if (t0 < t1) {
    if constexpr (C3 >= 32) pm_dma_tile<C3>(a.dy + (long)t0 * 32 * C3, P, lane);
    pm_dma_tile<C2>(a.h2 + (long)t0 * 32 * C2, Q, lane);
  }
  pm_build_image<2, 4, true>(img1, a.w1, a.ldw1, PM_C, PM_C, tid, 256);
  pm_build_image<2, C2 / 16, true>(img2, a.w2, PM_C, PM_C, C2, tid, 256);
  pm_build_image<T2, S::K3, true>(img3, a.w3, C2, C2, C3, tid, 256);
  __syncthreads();

  for (int tile = t0; tile < t1; ++tile) {
    const long row0 = (long)tile * 32;
    // ---- layer 3: G3 = dy (no ReLU behind the last layer)
    floatx16 G3[T3], H2[T2];
    if constexpr (C3 >= 32) {
      pm_dma_wait();
      pm_get_sw<T3, C3>(P, G3, o3);
      db3 += pm_colsum<C3>(P, lane, o3);
    } else {      // two columns: registers 0, 1 of the lanes with h = 0; the tile [32 points][2] for the point fragments
      ZERO16(G3[0]);
      const float2 d = h == 0 ? *reinterpret_cast<const float2*>(a.dy + (row0 + r) * 2) : make_float2(0.f, 0.f);
      G3[0][0] = d.x, G3[0][1] = d.y;
      if (h == 0) *reinterpret_cast<float2*>(P + 2 * r) = d;
      pm_dma_wait();
      pzn::wave_lds_sync();
      if (lane < 2) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < 32; ++p) s += P[2 * p + lane];
        db3 += s;
      }
    }
    pm_get_sw<T2, C2>(Q, H2, o2);
    // dW3[o][i] += sum_p G3[p][o] h2[p][i]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[T3][3], fb[T2][3];
#pragma unroll
      for (int ot = 0; ot < T3; ++ot) {
        if constexpr (C3 >= 32) {
          pm_point_frag<C3>(P, ot, s, o3, fa[ot]);
        } else {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = r < 2 ? P[2 * (16 * s + 8 * h + j) + r] : 0.f;
          split8(v, fa[ot]);
        }
      }
#pragma unroll
      for (int it = 0; it < T2; ++it) pm_point_frag<C2>(Q, it, s, o2, fb[it]);
#pragma unroll
      for (int ot = 0; ot < T3; ++ot)
#pragma unroll
        for (int it = 0; it < T2; ++it) dW3[ot][it] = mma6(fa[ot], fb[it], dW3[ot][it]);
    }
    pzn::wave_lds_sync();                     // both tiles are dead: h1 may land in Q
    pm_dma_tile<64>(a.h1 + row0 * 64, Q, lane);
    // dH2 = W3^T G3, gated by h2 > 0
    floatx16 G2[T2];
#pragma unroll
    for (int i = 0; i < T2; ++i) ZERO16(G2[i]);
    pm_layer<T2, S::K3>(G2, G3, img3, lane);
#pragma unroll
    for (int i = 0; i < T2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) G2[i][e] = H2[i][e] > 0.f ? G2[i][e] : 0.f;
    pm_put_sw<T2, C2>(P, G2, o2);
    pm_dma_wait();
    pzn::wave_lds_sync();
    // ---- layer 2
    floatx16 H1[2];
    pm_get_sw<2, 64>(Q, H1, o1);
    db2 += pm_colsum<C2>(P, lane, o2);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[T2][3], fb[2][3];
#pragma unroll
      for (int ot = 0; ot < T2; ++ot) pm_point_frag<C2>(P, ot, s, o2, fa[ot]);
#pragma unroll
      for (int it = 0; it < 2; ++it) pm_point_frag<64>(Q, it, s, o1, fb[it]);
#pragma unroll
      for (int ot = 0; ot < T2; ++ot)
#pragma unroll
        for (int it = 0; it < 2; ++it) dW2[ot][it] = mma6(fa[ot], fb[it], dW2[ot][it]);
    }
    pzn::wave_lds_sync();
    pm_dma_tile<64>(a.x + row0 * 64, Q, lane);
    floatx16 G1[2];
    ZERO16(G1[0]);
    ZERO16(G1[1]);
    pm_layer<2, C2 / 16>(G1, G2, img2, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) G1[i][e] = H1[i][e] > 0.f ? G1[i][e] : 0.f;
    pm_put_sw<2, 64>(P, G1, o1);
    pm_dma_wait();
    pzn::wave_lds_sync();
    // ---- layer 1
    db1 += pm_colsum<64>(P, lane, o1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int ot = 0; ot < 2; ++ot) pm_point_frag<64>(P, ot, s, o1, fa[ot]);
#pragma unroll
      for (int it = 0; it < 2; ++it) pm_point_frag<64>(Q, it, s, o1, fb[it]);
#pragma unroll
      for (int ot = 0; ot < 2; ++ot)
#pragma unroll
        for (int it = 0; it < 2; ++it) dW1[ot][it] = mma6(fa[ot], fb[it], dW1[ot][it]);
    }
    pzn::wave_lds_sync();
    if (tile + 1 < t1) pm_dma_tile<C2>(a.h2 + (row0 + 32) * C2, Q, lane);     // the next tile's first operands
    floatx16 DX[2];
    ZERO16(DX[0]);
    ZERO16(DX[1]);
    pm_layer<2, 4>(DX, G1, img1, lane);
    pm_put_sw<2, 64>(P, DX, o1);
    pzn::wave_lds_sync();
    pm_sw_to_rows<64>(P, a.dx + row0 * 64, lane);
    pzn::wave_lds_sync();
    if constexpr (C3 >= 32) {
      if (tile + 1 < t1) pm_dma_tile<C3>(a.dy + (row0 + 32) * C3, P, lane);
    }
  }

  // ---- the workgroup's four wavefronts meet in LDS (fixed order), one partial per workgroup
  float* bp = a.part_b + (long)wg * 192;
  bp[lane] = db1, bp[64 + lane] = db2, bp[128 + lane] = db3;
  float* red = reinterpret_cast<float*>(lds);
  auto for_tiles = [&](auto&& f) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) f(dW1[i][j], n++);
#pragma unroll
    for (int i = 0; i < T2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) f(dW2[i][j], n++);
#pragma unroll
    for (int i = 0; i < T3; ++i)
#pragma unroll
      for (int j = 0; j < T2; ++j) f(dW3[i][j], n++);
  };
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
      for_tiles([&](const floatx16& t, int n) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float* q = red + (n * 16 + e) * 64 + lane;
          *q = w == 0 ? t[e] : *q + t[e];
        }
      });
    }
  }
  __syncthreads();
  float* pw = a.part_w + (long)blockIdx.x * (S::NACC * 1024);
  for (int i = tid; i < S::NACC * 1024 / 4; i += 256)
    reinterpret_cast<pm_f4*>(pw)[i] = reinterpret_cast<const pm_f4*>(red)[i];
}

// partial sums -> dW1[64, ldw1], dW2[C2, 64], dW3[C3, C2], db2[C2], db3[C3] and db1: [64] (waves_per_cloud = 0) or
// [clouds, 64]; fixed summation order
struct PmRedArgs {
  const float *part_w, *part_b;
  float *dW1, *dW2, *dW3, *db1, *db2, *db3;
  int ldw1, nwg, nwaves, waves_per_cloud, nclouds;
  int accumulate;      // != 0: the parameter gradients are ADDED to what is there (a per-cloud db1 is always overwritten)
};
// A workgroup = 64 consecutive outputs x 16 slices of the partial index; a thread adds its slice with four independent
// chains (the loads of a chain do not wait for each other), the slices meet in LDS in a fixed order.
template <int C2, int C3>
__global__ __launch_bounds__(1024) void point_mlp3_reduce_kernel(PmRedArgs a) {
  using S = PmShape<C2, C3>;
  constexpr int T2 = S::T2;
  constexpr int NW = S::NACC * 1024;
  __shared__ float red[16][64];
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int nb1 = (a.waves_per_cloud ? a.nclouds : 1) * 64;
  int grp = blockIdx.x;
  const float* base;
  int n;
  long stride;
  int kind, idx;       // 0: dW element idx, 1: db1[idx], 2: db2[idx], 3: db3[idx]
  bool valid = true;
  if (grp < NW / 64) {
    kind = 0, idx = grp * 64 + o;
    base = a.part_w + idx, n = a.nwg, stride = NW;
  } else if ((grp -= NW / 64) < nb1 / 64) {
    kind = 1, idx = grp * 64 + o;
    const int w0 = a.waves_per_cloud ? grp * a.waves_per_cloud : 0;
    base = a.part_b + (long)w0 * 192 + o, n = a.waves_per_cloud ? a.waves_per_cloud : a.nwaves, stride = 192;
  } else {
    grp -= nb1 / 64;
    kind = 2 + grp, idx = o;
    valid = o < (grp == 0 ? C2 : C3);
    base = a.part_b + 64 * (1 + grp) + (valid ? o : 0), n = a.nwaves, stride = 192;
  }
  const int per = (n + 15) / 16, i0 = q * per, i1 = min(n, i0 + per);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int i = i0;
  for (; i + 3 < i1; i += 4) {
    s0 += base[(long)i * stride], s1 += base[(long)(i + 1) * stride];
    s2 += base[(long)(i + 2) * stride], s3 += base[(long)(i + 3) * stride];
  }
  for (; i < i1; ++i) s0 += base[(long)i * stride];
  red[q][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (q != 0 || !valid) return;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) s += red[j][o];
  if (kind == 0) {
    const int t = idx >> 10, e = (idx >> 6) & 15, l = idx & 63;
    const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), col = l & 31;
    float* dst = nullptr;
    if (t < 4) {
      dst = a.dW1 + (long)(32 * (t >> 1) + row) * a.ldw1 + 32 * (t & 1) + col;
    } else if (t < 4 + 2 * T2) {
      const int m = t - 4;
      dst = a.dW2 + (32 * (m >> 1) + row) * 64 + 32 * (m & 1) + col;
    } else {
      const int m = t - 4 - 2 * T2, oo = 32 * (m / T2) + row;
      if (oo < C3) dst = a.dW3 + oo * C2 + 32 * (m % T2) + col;
    }
    if (dst) *dst = a.accumulate ? *dst + s : s;
  } else if (kind == 1) {
    a.db1[idx] = (a.accumulate && !a.waves_per_cloud) ? a.db1[idx] + s : s;
  } else {
    float* dst = (kind == 2 ? a.db2 : a.db3) + idx;
    *dst = a.accumulate ? *dst + s : s;
  }
}

bool pm_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename K>
int pm_set_lds(K k, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
             ? PZN_OK
             : PZN_ELAUNCH;
}

bool pm_enabled() {
  constexpr bool v = true;
  return v;
}

}  // namespace

// 1 when (C1, C2, C3) is a chain this build fuses: 64 -> 64 -> {64 | 32} -> {64 | 2} on rows of 64 floats
PZN_EXPORT int pzn_point_mlp3_supported(int C0, int C1, int C2, int C3) {
  return pm_enabled() && C0 == 64 && C1 == 64 && ((C2 == 64 && C3 == 64) || (C2 == 32 && C3 == 2));
}

// y = (relu(relu(x W1^T + b1) W2^T + b2)) W3^T + b3 for M rows of 64 floats; h1, h2 = the two hidden activations (kept
// for the backward).  W1[64, ldw1] may be a column slice (ldw1 >= 64); b1 is [64] (b1_per_cloud = 0) or one row of 64
// per cloud of rows_per_cloud rows (the folded global half of a boundary head's first layer).  M % 32 == 0,
// rows_per_cloud % 32 == 0.
PZN_EXPORT int pzn_point_mlp3_fwd_f32(const float* x, long long M, int rows_per_cloud, const float* W1, int ldw1, const float* b1,
                                      int b1_per_cloud, const float* W2, const float* b2, const float* W3, const float* b3,
                                      int C2, int C3, float* h1, float* h2, float* y, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && W1 && b1 && W2 && b2 && W3 && b3 && h1 && h2 && y && M > 0 && ldw1 >= 64);
  if (!pzn_point_mlp3_supported(64, 64, C2, C3) || M % 32 != 0 || M / 32 > 0x7fffffffL) return PZN_EUNSUPPORTED;
  PZN_CHECK_ARG(!b1_per_cloud || (rows_per_cloud > 0 && rows_per_cloud % 32 == 0 && M % rows_per_cloud == 0));
  PZN_CHECK_ARG(pm_aligned16(x) && pm_aligned16(b1) && pm_aligned16(b2) && pm_aligned16(h1) && pm_aligned16(h2) &&
                pm_aligned16(y) && (C3 == 2 || pm_aligned16(b3)));
  PmFwdArgs a{x, W1, b1, W2, b2, W3, b3, h1, h2, y, ldw1, b1_per_cloud ? 64 : 0, b1_per_cloud ? rows_per_cloud : 1,
              (int)(M / 32)};
  const int nwg = (int)((a.ntiles + PM_FWD_WAVES - 1) / PM_FWD_WAVES);
  const dim3 grid(nwg < 256 ? nwg : 256), block(PM_FWD_WAVES * 64);
  hipStream_t st = pzn_hip_stream(stream);
  if (C2 == 64) {
    constexpr int lds = pm_fwd_lds<64, 64>();
    if (pm_set_lds(point_mlp3_fwd_kernel<64, 64>, lds) != PZN_OK) return PZN_ELAUNCH;
    PZN_LAUNCH((point_mlp3_fwd_kernel<64, 64>), grid, block, lds, st, a);
  } else {
    constexpr int lds = pm_fwd_lds<32, 2>();
    if (pm_set_lds(point_mlp3_fwd_kernel<32, 2>, lds) != PZN_OK) return PZN_ELAUNCH;
    PZN_LAUNCH((point_mlp3_fwd_kernel<32, 2>), grid, block, lds, st, a);
  }
  PZN_RETURN_LAUNCH_STATUS();
}

namespace {

struct PmGeom {
  int ntiles, T, nwg;
};
// tiles per wavefront: ~1024 wavefront slots (one per SIMD); with a per-cloud first bias a wavefront stays inside ONE cloud
bool pm_geometry(long long M, int rows_per_cloud, int per_cloud, PmGeom* g) {
  if (M <= 0 || M % 32 != 0 || M / 32 > 0x3fffffffLL) return false;
  g->ntiles = (int)(M / 32);
  int T = (g->ntiles + 1023) / 1024;
  if (per_cloud) {
    if (rows_per_cloud <= 0 || rows_per_cloud % 32 != 0 || M % rows_per_cloud != 0) return false;
    const int tpc = rows_per_cloud / 32;
    if (T > tpc) T = tpc;
    while (tpc % T != 0) ++T;
  }
  g->T = T;
  const int nwaves = (g->ntiles + T - 1) / T;
  g->nwg = (nwaves + 3) / 4;
  return true;
}

template <int C2, int C3>
int pm_launch_bwd(const PmBwdArgs& a, const PmRedArgs& ra, int nwg, hipStream_t st) {
  using S = PmShape<C2, C3>;
  if (pm_set_lds(point_mlp3_bwd_kernel<C2, C3>, S::LDS) != PZN_OK) return PZN_ELAUNCH;
  PZN_LAUNCH((point_mlp3_bwd_kernel<C2, C3>), dim3(nwg), dim3(256), S::LDS, st, a);
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  const int ngroups = S::NACC * 16 + (ra.waves_per_cloud ? ra.nclouds : 1) + 2;     // of 64 outputs: dW | db1 | db2 | db3
  PZN_LAUNCH((point_mlp3_reduce_kernel<C2, C3>), dim3(ngroups), dim3(1024), 0, st, ra);
  PZN_RETURN_LAUNCH_STATUS();
}

}  // namespace

PZN_EXPORT size_t pzn_point_mlp3_bwd_workspace_bytes(long long M, int rows_per_cloud, int b1_per_cloud, int C2, int C3) {
  PmGeom g;
  if (!pzn_point_mlp3_supported(64, 64, C2, C3) || !pm_geometry(M, rows_per_cloud, b1_per_cloud, &g)) return 0;
  const size_t nacc = C2 == 64 ? PmShape<64, 64>::NACC : PmShape<32, 2>::NACC;
  return (size_t)g.nwg * (nacc * 4096 + 4 * 192 * sizeof(float));
}

// Backward of pzn_point_mlp3_fwd_f32: dx[M, 64], dW1[64, ldw1] (columns 0..63 written), dW2[C2, 64], dW3[C3, C2], db2[C2],
// db3[C3] and db1 — [64], or with b1_per_cloud one row per cloud: the gradient of the per-cloud bias —: OVERWRITTEN, or with
// accumulate != 0 ADDED to (one owner per element, no atomics: e.g. straight into a flat gradient bucket; dx and a per-cloud
// db1 are always overwritten).  x, h1, h2 as the forward left them.  Two launches (the pass and the fixed-order sum of its
// partial results).
PZN_EXPORT int pzn_point_mlp3_bwd_f32(const float* dy, const float* x, const float* h1, const float* h2, long long M,
                                      int rows_per_cloud, const float* W1, int ldw1, int b1_per_cloud, const float* W2,
                                      const float* W3, int C2, int C3, float* dx, float* dW1, float* db1, float* dW2,
                                      float* db2, float* dW3, float* db3, int accumulate, void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(dy && x && h1 && h2 && W1 && W2 && W3 && dx && dW1 && db1 && dW2 && db2 && dW3 && db3 && workspace && ldw1 >= 64);
  PmGeom g;
  if (!pzn_point_mlp3_supported(64, 64, C2, C3)) return PZN_EUNSUPPORTED;
  if (!pm_geometry(M, rows_per_cloud, b1_per_cloud, &g)) return b1_per_cloud && M > 0 && M % 32 == 0 ? PZN_EINVAL : PZN_EUNSUPPORTED;
  PZN_CHECK_ARG(pm_aligned16(dy) && pm_aligned16(x) && pm_aligned16(h1) && pm_aligned16(h2) && pm_aligned16(dx) &&
                pm_aligned16(workspace));
  const size_t nacc = C2 == 64 ? PmShape<64, 64>::NACC : PmShape<32, 2>::NACC;
  float* part_w = static_cast<float*>(workspace);
  float* part_b = part_w + (size_t)g.nwg * nacc * 1024;
  PmBwdArgs a{dy, x, h1, h2, W1, W2, W3, dx, part_w, part_b, ldw1, g.ntiles, g.T};
  PmRedArgs ra{part_w, part_b, dW1, dW2, dW3, db1, db2, db3, ldw1, g.nwg, g.nwg * 4,
               b1_per_cloud ? (rows_per_cloud / 32) / g.T : 0, b1_per_cloud ? (int)(M / rows_per_cloud) : 0, accumulate};
  hipStream_t st = pzn_hip_stream(stream);
  return C2 == 64 ? pm_launch_bwd<64, 64>(a, ra, g.nwg, st) : pm_launch_bwd<32, 2>(a, ra, g.nwg, st);
}
