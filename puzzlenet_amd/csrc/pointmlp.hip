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

bool pm_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename K>
int pm_set_lds(K k, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
             ? PZN_OK
             : PZN_ELAUNCH;
}

bool pm_enabled() {
  static const bool v = [] { const char* e = getenv("PZN_POINT_MLP"); return !(e && e[0] == '0'); }();
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
    hipLaunchKernelGGL((point_mlp3_fwd_kernel<64, 64>), grid, block, lds, st, a);
  } else {
    constexpr int lds = pm_fwd_lds<32, 2>();
    if (pm_set_lds(point_mlp3_fwd_kernel<32, 2>, lds) != PZN_OK) return PZN_ELAUNCH;
    hipLaunchKernelGGL((point_mlp3_fwd_kernel<32, 2>), grid, block, lds, st, a);
  }
  PZN_RETURN_LAUNCH_STATUS();
}
