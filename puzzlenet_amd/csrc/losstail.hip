// losstail.hip — the small ops of training_step's loss tail as single launches.
//
// Each of these was a chain of ATen kernels (and hipBLASLt GEMMs for the 3x3 / 4x4 products) of a few microseconds of
// work apiece — together ~60 launches and ~0.5 ms of a 13 ms step:
//   se3.transform      se_math/se3.py:110-120     R a + p on [B,N,3] points (was: two slices, bmm, add, two permute copies)
//   comp               model5_b.py:1512-1519      16 * mean((g igt - I)^2)   (was: bmm, eye, repeat, mse, mul)
//   boundary CE        model5_b.py:1063-1064, 1085-1090   cross_entropy([B,2,N] logits, labels) and the class-1
//                                                  probability the top-128 selection ranks by (was: log_softmax,
//                                                  nll_loss2d, a second softmax, a slice copy; and their backward)
//   top-k of rows      model5_b.py:1089-1091      indices of the K largest entries per row, descending
//   mean of 4 maps     model5_b.py:468-469        (a1 + a2 + a3 + a4) / 4
//   column mean + argmax  model5_b.py:937-942     attention.mean(dim=1) and the index of its largest entry
#include <math.h>

#include "pzn_common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* sh) {  // blockDim.x <= 1024; every thread gets the total
  v = pzn::wave_sum_f32(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += sh[i];
  return t;
}

// ------------------------------------------------------------------------------------ se3.transform
__global__ __launch_bounds__(256) void se3_transform_fwd_kernel(const float* __restrict__ g, const float* __restrict__ p,
                                                                int N, float* __restrict__ out) {
  const int b = blockIdx.y;
  const float* G = g + (size_t)b * 16;
  const float r00 = G[0], r01 = G[1], r02 = G[2], t0 = G[3], r10 = G[4], r11 = G[5], r12 = G[6], t1 = G[7], r20 = G[8],
              r21 = G[9], r22 = G[10], t2 = G[11];
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const float* q = p + ((size_t)b * N + n) * 3;
    const float x = q[0], y = q[1], z = q[2];
    float* o = out + ((size_t)b * N + n) * 3;
    o[0] = (r00 * x + r01 * y + r02 * z) + t0;  // R a, then + p (se3.py:116-118)
    o[1] = (r10 * x + r11 * y + r12 * z) + t1;
    o[2] = (r20 * x + r21 * y + r22 * z) + t2;
  }
}

// dp = R^T dout;  dg[0:3,0:3] = sum_n dout_n p_n^T,  dg[0:3,3] = sum_n dout_n,  dg[3,:] = 0.  One workgroup per b.
__global__ __launch_bounds__(256) void se3_transform_bwd_kernel(const float* __restrict__ g, const float* __restrict__ p,
                                                                const float* __restrict__ dout, int N,
                                                                float* __restrict__ dp, float* __restrict__ dg) {
  __shared__ float sh[16];
  const int b = blockIdx.x;
  const float* G = g + (size_t)b * 16;
  const float r00 = G[0], r01 = G[1], r02 = G[2], r10 = G[4], r11 = G[5], r12 = G[6], r20 = G[8], r21 = G[9], r22 = G[10];
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.f;
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const size_t o = ((size_t)b * N + n) * 3;
    const float d0 = dout[o], d1 = dout[o + 1], d2 = dout[o + 2];
    if (dp) {
      dp[o] = r00 * d0 + r10 * d1 + r20 * d2;
      dp[o + 1] = r01 * d0 + r11 * d1 + r21 * d2;
      dp[o + 2] = r02 * d0 + r12 * d1 + r22 * d2;
    }
    if (dg) {
      const float x = p[o], y = p[o + 1], z = p[o + 2];
      acc[0] += d0 * x, acc[1] += d0 * y, acc[2] += d0 * z, acc[3] += d0;
      acc[4] += d1 * x, acc[5] += d1 * y, acc[6] += d1 * z, acc[7] += d1;
      acc[8] += d2 * x, acc[9] += d2 * y, acc[10] += d2 * z, acc[11] += d2;
    }
  }
  if (dg) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const float t = block_sum(acc[i], sh);
      if (threadIdx.x == 0) dg[(size_t)b * 16 + i] = t;
    }
    if (threadIdx.x < 4) dg[(size_t)b * 16 + 12 + threadIdx.x] = 0.f;
  }
}

// ------------------------------------------------------------------------------------ comp
__device__ __forceinline__ void mat4_mul_minus_I(const float* a, const float* b, float* c) {  // c = a b - I
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
      c[i * 4 + j] = s - (i == j ? 1.f : 0.f);
    }
}

__global__ __launch_bounds__(256) void comp_fwd_kernel(const float* __restrict__ g, const float* __restrict__ igt, int B,
                                                       float* __restrict__ loss) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float a[16], m[16], c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = g[(size_t)b * 16 + i], m[i] = igt[(size_t)b * 16 + i];
    mat4_mul_minus_I(a, m, c);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i] * c[i];
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) loss[0] = s / (float)B;  // mean over B*16 elements, * 16 (model5_b.py:1519)
}

// dg_b = dloss * (2 / B) * (g_b igt_b - I) igt_b^T
__global__ __launch_bounds__(256) void comp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ igt,
                                                       const float* __restrict__ dloss, int B, float* __restrict__ dg) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float a[16], m[16], c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = g[(size_t)b * 16 + i], m[i] = igt[(size_t)b * 16 + i];
  mat4_mul_minus_I(a, m, c);
  const float sc = dloss[0] * 2.f / (float)B;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += c[i * 4 + j] * m[k * 4 + j];
      dg[(size_t)b * 16 + i * 4 + k] = sc * s;
    }
}

// ------------------------------------------------------------------------------------ boundary cross-entropy
// logits [B,2,N] as a view: logit of class c at point n of cloud b = logits[b 2N + c sc + n sn] ((sc, sn) = (N, 1): the tensor as
// the reference holds it; (1, 2): the heads' [B,N,2] output seen through permute(0,2,1), model5_b.py:751-754, with no copy),
// labels [B,N] (0 / 1 as floats, dataset.py:1363-1366).  prob1[b,n] = softmax over the two classes,
// class 1 (model5_b.py:1085-1090); loss += sum over the block of (logsumexp - logit[label]) / (B N).
__global__ __launch_bounds__(256) void boundary_ce_fwd_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ labels, int N, long total, int sc, int sn,
                                                              float* __restrict__ prob1, float* __restrict__ loss) {
  __shared__ float sh[16];
  float s = 0.f;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / N;
    const int n = (int)(e - b * N);
    const float* lp = logits + b * 2 * N + (long)n * sn;
    const float l0 = lp[0], l1 = lp[sc];
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m), sum = e0 + e1;
    prob1[e] = e1 / sum;
    const float lse = m + logf(sum);
    s += lse - (labels[e] != 0.f ? l1 : l0);
  }
  s = block_sum(s, sh);
  if (threadIdx.x == 0) loss[1 + blockIdx.x] = s;     // per-workgroup partial; summed in a fixed order below
}

// loss[0] = (sum of the nparts partials in index order) / total: bit-reproducible, unlike float atomics across workgroups
__global__ __launch_bounds__(64) void boundary_ce_reduce_kernel(float* __restrict__ loss, int nparts, float inv_total) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 64) s += loss[1 + i];
  s = pzn::wave_sum_f32(s);
  if (threadIdx.x == 0) loss[0] = s * inv_total;
}

__global__ __launch_bounds__(256) void boundary_ce_bwd_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ labels,
                                                              const float* __restrict__ dloss, int N, long total, int sc,
                                                              int sn, float* __restrict__ dlogits) {
  const float scale = dloss[0] / (float)total;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long b = e / N;
    const int n = (int)(e - b * N);
    const long o = b * 2 * N + (long)n * sn;
    const float l0 = logits[o], l1 = logits[o + sc];
    const float m = fmaxf(l0, l1);
    const float e0 = expf(l0 - m), e1 = expf(l1 - m), sum = e0 + e1;
    const bool one = labels[e] != 0.f;
    dlogits[o] = scale * (e0 / sum - (one ? 0.f : 1.f));      // (same strides as logits)
    dlogits[o + sc] = scale * (e1 / sum - (one ? 1.f : 0.f));
  }
}

// ------------------------------------------------------------------------------------ top-k of rows
// One workgroup per row: 4-pass radix select (8 bits a pass, LDS histogram) of the K-th largest key, ordered
// compaction (keys above the threshold, then as many threshold keys as are missing, lowest index first), bitonic sort
// of the K winners by (value descending, index ascending).  N <= 16384, K <= 256.
constexpr int TK_T = 256;

__device__ __forceinline__ uint32_t order_key(float v) {  // larger float -> larger key; -0 and +0 share a key (they compare equal)
  uint32_t u = __float_as_uint(v);
  u = u == 0x80000000u ? 0u : u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(TK_T) void topk_rows_kernel(const float* __restrict__ x, int N, int K,
                                                         int64_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem);  // [N]
  __shared__ int hist[256];
  __shared__ int wsum[TK_T / 64];
  __shared__ uint64_t win[TK_T];  // (key << 32) | ~index : descending sort = value descending, index ascending
  __shared__ uint32_t s_prefix;
  __shared__ int s_need, s_base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* row = x + (size_t)blockIdx.x * N;
  for (int i = tid; i < N; i += TK_T) keys[i] = order_key(row[i]);
  if (tid == 0) s_prefix = 0, s_need = K;
  __syncthreads();
  // radix select: after pass p the top 8(p+1) bits of the K-th largest key are known
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix;
    const uint32_t pmask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    for (int i = tid; i < N; i += TK_T) {
      const uint32_t k = keys[i];
      if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int need = s_need, b = 255;
      for (; b > 0; --b) {
        if (hist[b] >= need) break;
        need -= hist[b];
      }
      s_need = need;  // how many keys of bin b (with this prefix) are still wanted
      s_prefix = prefix | ((uint32_t)b << shift);
    }
    __syncthreads();
  }
  const uint32_t T = s_prefix;  // the K-th largest key; s_need of the keys equal to it are taken
  const int need_eq = s_need;
  if (tid == 0) s_base = 0;
  win[tid] = 0;  // (lanes >= K stay at 0: they sort to the end)
  __syncthreads();
  // ordered compaction, two sweeps: keys > T (any order is fine, the sort follows), then the first need_eq keys == T
  for (int sweep = 0; sweep < 2; ++sweep) {
    const int limit = sweep == 0 ? K : need_eq;
    const int start = s_base;
    __syncthreads();
    int taken = 0;  // block-uniform running count within the sweep
    for (int i0 = 0; i0 < N && taken < limit; i0 += TK_T) {
      const int i = i0 + tid;
      const uint32_t k = i < N ? keys[i] : 0u;
      const bool on = i < N && (sweep == 0 ? k > T : k == T);
      const uint64_t bal = __ballot(on);
      const int before = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
      if (lane == 0) wsum[wv] = __builtin_popcountll(bal);
      __syncthreads();
      int off = taken, tot = 0;
      for (int q = 0; q < TK_T / 64; ++q) {
        const int c = wsum[q];
        off += q < wv ? c : 0;
        tot += c;
      }
      const int pos = off + before;
      if (on && pos < limit) win[start + pos] = ((uint64_t)k << 32) | (uint32_t)(~(uint32_t)i);
      taken += tot;
      __syncthreads();
    }
    if (tid == 0) s_base = start + (taken < limit ? taken : limit);
    __syncthreads();
  }
  // bitonic sort of 256 entries, descending
  for (int k2 = 2; k2 <= TK_T; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      const int partner = tid ^ j;
      const uint64_t a = win[tid], b = win[partner];
      __syncthreads();
      const bool up = (tid & k2) == 0;  // descending overall: "up" blocks keep the larger element first
      const bool keep_max = (tid < partner) == up;
      win[tid] = keep_max ? (a > b ? a : b) : (a > b ? b : a);
      __syncthreads();
    }
  if (tid < K) idx[(size_t)blockIdx.x * K + tid] = (int64_t)(~(uint32_t)win[tid]);
}

// ------------------------------------------------------------------------------------ mean of four maps
__global__ __launch_bounds__(256) void avg4_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                   const float4* __restrict__ c, const float4* __restrict__ d, long n4,
                                                   float4* __restrict__ out) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = a[i], y = b[i], z = c[i], w = d[i];
    out[i] = make_float4((((x.x + y.x) + z.x) + w.x) / 4.f, (((x.y + y.y) + z.y) + w.y) / 4.f,
                         (((x.z + y.z) + z.z) + w.z) / 4.f, (((x.w + y.w) + z.w) + w.w) / 4.f);
  }
}

// ------------------------------------------------------------------------------------ per-cloud bias + ReLU
// The first layer of the boundary heads (model5_b.py:745-752) is Linear(cat([g.repeat(1,N,1), x], -1)): its product splits
// into x W[:,Cg:]^T per point and (g W[:,:Cg]^T + b) per CLOUD.  y[b,n,:] = relu(y[b,n,:] + cb[b,:]) in place, and for
// the backward the per-cloud column sums of the gated gradient: dcb[b,c] = sum_n (y[b,n,c] > 0 ? dy[b,n,c] : 0).
__global__ __launch_bounds__(256) void cloud_bias_relu_kernel(float4* __restrict__ y, const float4* __restrict__ cb, long n4,
                                                              int c4, long per_cloud4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 b = cb[(i / per_cloud4) * c4 + (i % c4)];
    float4 v = y[i];
    v.x = fmaxf(v.x + b.x, 0.f), v.y = fmaxf(v.y + b.y, 0.f), v.z = fmaxf(v.z + b.z, 0.f), v.w = fmaxf(v.w + b.w, 0.f);
    y[i] = v;
  }
}

// grid (B); block 1024 = c4 column quads x 1024 / c4 row lanes: ONE workgroup per cloud walks all N rows, the row lanes'
// partial sums meet in LDS in a fixed order and the result is stored (no atomics: bit-reproducible run to run, like
// boundary_ce_reduce_kernel and point_mlp3_reduce_kernel; this is the boundary heads' layer-by-layer path, PZN_POINT_MLP=0)
__global__ __launch_bounds__(1024) void cloud_gated_colsum_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                                  int N, int c4, float* __restrict__ dcb) {
  __shared__ float4 part[1024];
  const int b = blockIdx.x, q = threadIdx.x % c4, rl = threadIdx.x / c4, nrl = 1024 / c4;
  const long base = (long)b * N * c4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rl < nrl)
    for (int n = rl; n < N; n += nrl) {
      const float4 g = dy[base + (long)n * c4 + q], v = y[base + (long)n * c4 + q];
      s.x += v.x > 0.f ? g.x : 0.f, s.y += v.y > 0.f ? g.y : 0.f, s.z += v.z > 0.f ? g.z : 0.f, s.w += v.w > 0.f ? g.w : 0.f;
    }
  part[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0) {
    for (int j = 1; j < nrl; ++j) {
      const float4 o = part[j * c4 + q];
      s.x += o.x, s.y += o.y, s.z += o.z, s.w += o.w;
    }
    *reinterpret_cast<float4*>(dcb + ((long)b * c4 + q) * 4) = s;
  }
}

// ------------------------------------------------------------------------------------ column mean + argmax
// a [B, R, C] -> mean[b, c] = (sum_r a[b, r, c]) / R;  arg[b] = first index of the largest mean.  Two launches so that
// the 16.8 MB of a [64,256,256] map are read by 8 workgroups per cloud instead of one (108 -> ~15 us): partial column
// sums of CM_CHUNKS row ranges (no atomics: the order of the sum is fixed), then one workgroup per b adds the partials
// in order, divides, stores the mean and reduces the arg-max.  Thread per column (C <= 1024).
constexpr int CM_CHUNKS = 8;

__global__ __launch_bounds__(1024) void colsum_partial_kernel(const float* __restrict__ a, int R, int C,
                                                              float* __restrict__ part) {
  const int b = blockIdx.x, ch = blockIdx.y, c = threadIdx.x;
  if (c >= C) return;
  const int per = (R + CM_CHUNKS - 1) / CM_CHUNKS, r0 = ch * per, r1 = min(R, r0 + per);
  const float* p = a + (size_t)b * R * C + c;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += p[(size_t)r * C];
  part[((size_t)b * CM_CHUNKS + ch) * C + c] = s;
}

__global__ __launch_bounds__(1024) void colmean_argmax_kernel(const float* __restrict__ part, int R, int C,
                                                              float* __restrict__ mean, int64_t* __restrict__ arg) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const int b = blockIdx.x, c = threadIdx.x;
  float s = 0.f;
  if (c < C) {
#pragma unroll
    for (int ch = 0; ch < CM_CHUNKS; ++ch) s += part[((size_t)b * CM_CHUNKS + ch) * C + c];
    s /= (float)R;
    mean[(size_t)b * C + c] = s;
  }
  float best = c < C ? s : -INFINITY;
  int bi = c < C ? c : 0x7fffffff;
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const float ov = __shfl_xor(best, m, PZN_WAVE);
    const int oi = __shfl_xor(bi, m, PZN_WAVE);
    if (ov > best || (ov == best && oi < bi)) best = ov, bi = oi;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 0) sv[wave] = best, si[wave] = bi;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nw; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) best = sv[w], bi = si[w];
    arg[b] = bi;
  }
}

inline int grid_for(long total, int block, int cap = 2048) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

PZN_EXPORT int pzn_se3_transform_fwd_f32(const float* g, const float* p, int B, int N, float* out, pzn_stream_t stream) {
  PZN_CHECK_ARG(g && p && out && B > 0 && B <= 65535 && N > 0);
  PZN_LAUNCH(se3_transform_fwd_kernel, dim3((unsigned)grid_for(N, 256, 64), (unsigned)B), dim3(256), 0,
                     pzn_hip_stream(stream), g, p, N, out);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_se3_transform_bwd_f32(const float* g, const float* p, const float* dout, int B, int N, float* dp,
                                         float* dg, pzn_stream_t stream) {
  PZN_CHECK_ARG(g && p && dout && (dp || dg) && B > 0 && N > 0);
  PZN_LAUNCH(se3_transform_bwd_kernel, dim3((unsigned)B), dim3(256), 0, pzn_hip_stream(stream), g, p, dout, N, dp,
                     dg);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_comp_fwd_f32(const float* g, const float* igt, int B, float* loss, pzn_stream_t stream) {
  PZN_CHECK_ARG(g && igt && loss && B > 0);
  PZN_LAUNCH(comp_fwd_kernel, dim3(1), dim3(256), 0, pzn_hip_stream(stream), g, igt, B, loss);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_comp_bwd_f32(const float* g, const float* igt, const float* dloss, int B, float* dg,
                                pzn_stream_t stream) {
  PZN_CHECK_ARG(g && igt && dloss && dg && B > 0);
  PZN_LAUNCH(comp_bwd_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, pzn_hip_stream(stream), g, igt, dloss,
                     B, dg);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_boundary_ce_fwd_f32(const float* logits, const float* labels, int B, int N, int points_major, float* prob1,
                                       float* loss, pzn_stream_t stream) {
  PZN_CHECK_ARG(logits && labels && prob1 && loss && B > 0 && N > 0);
  const int sc = points_major ? 1 : N, sn = points_major ? 2 : 1;
  hipStream_t st = pzn_hip_stream(stream);
  const long total = (long)B * N;
  const int nparts = (int)grid_for(total, 256, 512);
  PZN_LAUNCH(boundary_ce_fwd_kernel, dim3((unsigned)nparts), dim3(256), 0, st, logits, labels, N, total, sc, sn, prob1, loss);
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  PZN_LAUNCH(boundary_ce_reduce_kernel, dim3(1), dim3(64), 0, st, loss, nparts, 1.f / (float)total);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_boundary_ce_bwd_f32(const float* logits, const float* labels, const float* dloss, int B, int N,
                                       int points_major, float* dlogits, pzn_stream_t stream) {
  PZN_CHECK_ARG(logits && labels && dloss && dlogits && B > 0 && N > 0);
  const int sc = points_major ? 1 : N, sn = points_major ? 2 : 1;
  const long total = (long)B * N;
  PZN_LAUNCH(boundary_ce_bwd_kernel, dim3((unsigned)grid_for(total, 256, 512)), dim3(256), 0, pzn_hip_stream(stream),
                     logits, labels, dloss, N, total, sc, sn, dlogits);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_topk_rows_f32(const float* x, int R, int N, int K, int64_t* idx, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && idx && R > 0 && N > 0 && K > 0 && K <= N);
  if (K > TK_T || N > 16384) return PZN_EUNSUPPORTED;
  PZN_LAUNCH(topk_rows_kernel, dim3((unsigned)R), dim3(TK_T), (size_t)N * sizeof(uint32_t), pzn_hip_stream(stream), x,
                     N, K, idx);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_avg4_f32(const float* a, const float* b, const float* c, const float* d, size_t n, float* out,
                            pzn_stream_t stream) {
  PZN_CHECK_ARG(a && b && c && d && out && n > 0);
  if ((n & 3) || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
                   reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(out)) & 15))
    return PZN_EUNSUPPORTED;
  PZN_LAUNCH(avg4_kernel, dim3((unsigned)grid_for((long)(n >> 2), 256, 4096)), dim3(256), 0, pzn_hip_stream(stream),
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<const float4*>(c),
                     reinterpret_cast<const float4*>(d), (long)(n >> 2), reinterpret_cast<float4*>(out));
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_cloud_bias_relu_f32(float* y, const float* cb, int B, int N, int C, pzn_stream_t stream) {
  PZN_CHECK_ARG(y && cb && B > 0 && N > 0 && C > 0);
  if ((C & 3) || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(cb)) & 15)) return PZN_EUNSUPPORTED;
  const long n4 = (long)B * N * (C / 4);
  PZN_LAUNCH(cloud_bias_relu_kernel, dim3((unsigned)grid_for(n4, 256, 4096)), dim3(256), 0, pzn_hip_stream(stream),
                     reinterpret_cast<float4*>(y), reinterpret_cast<const float4*>(cb), n4, C / 4, (long)N * (C / 4));
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_cloud_gated_colsum_f32(const float* dy, const float* y, int B, int N, int C, float* dcb,
                                          pzn_stream_t stream) {
  PZN_CHECK_ARG(dy && y && dcb && B > 0 && B <= 65535 && N > 0 && C > 0);
  if ((C & 3) || C > 1024 || 1024 % (C / 4) != 0 ||
      ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dcb)) & 15))
    return PZN_EUNSUPPORTED;
  hipStream_t st = pzn_hip_stream(stream);
  PZN_LAUNCH(cloud_gated_colsum_kernel, dim3((unsigned)B), dim3(1024), 0, st,
                     reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(y), N, C / 4, dcb);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT size_t pzn_colmean_workspace_bytes(int B, int C) {
  return B > 0 && C > 0 ? sizeof(float) * (size_t)B * CM_CHUNKS * C : 0;
}

PZN_EXPORT int pzn_colmean_argmax_f32(const float* a, int B, int R, int C, float* mean, int64_t* arg, void* workspace,
                                      pzn_stream_t stream) {
  PZN_CHECK_ARG(a && mean && arg && workspace && B > 0 && B <= 65535 && R > 0 && C > 0);
  if (C > 1024) return PZN_EUNSUPPORTED;
  const int threads = ((C + 63) / 64) * 64;
  hipStream_t st = pzn_hip_stream(stream);
  float* part = static_cast<float*>(workspace);
  PZN_LAUNCH(colsum_partial_kernel, dim3((unsigned)B, CM_CHUNKS), dim3((unsigned)threads), 0, st, a, R, C, part);
  PZN_LAUNCH(colmean_argmax_kernel, dim3((unsigned)B), dim3((unsigned)threads), 0, st, part, R, C, mean, arg);
  PZN_RETURN_LAUNCH_STATUS();
}
