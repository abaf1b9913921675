// sapoint.hip — first shared-MLP layer of a set-abstraction level computed PER POINT instead of per grouped row.
//
// Reference (pointnet_util.py:123-132 + model5_b.py:452 / :459): every one of the B*S*32 grouped rows is
// {xyz[j] - centre, feat[j]} with j = idx[b,s,k], and the first 1x1 convolution multiplies each row by W1[C1, 3+D].
// The feature block of that product depends on the POINT only, and a point is gathered by 8-16 rows:
//     h[b,s,k,:] = relu( P'[b,j,:] + Q[b,s,:] ),   P' = feat W1[:,3:]^T + W1[:,0:3] xyz[j],   Q = b1 - W1[:,0:3] centre_s.
// So the product shrinks from B*S*32 rows to B*N rows (8x / 16x fewer flops, same result up to the order of the
// fp32 sum), the grouped tensor [B,S,32,3+D] is never written, and the rows themselves are generated where they are
// consumed: forward inside the matrix-core kernel (csrc/salevel.hip), backward inside the weight-gradient pass
// (csrc/poolbwd.hip) and the walk by point (csrc/sapool.hip).  This file: the P' / Q tables (sa_prep_kernel) and the
// inverse neighbour lists the walk by point follows (sa_inverse_lists_kernel).
// (Rounds 1-4 wrote the rows h and read their gradient dh back: sa_point_l1_{fwd,bwd}_kernel, removed in round 6.)
#include <stdlib.h>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

// Inverse neighbour lists of one cloud per workgroup: off[b][0..N] (exclusive prefix of the reference counts) and
// rows[b][.] = the in-cloud row numbers (s*32 + k) grouped by the point they gathered, ascending within a point.  Counters live in LDS.
constexpr int INV_T = 1024;
// LROWS: the lists are built and sorted in LDS ([SK] ints behind the counters) and leave in one coalesced pass; otherwise (a
// cloud whose S * K rows do not fit) they are built in global memory and sorted there.
template <bool LROWS>
__global__ __launch_bounds__(INV_T) void sa_inverse_lists_kernel(const int64_t* __restrict__ idx, int N, int SK,
                                                                 int32_t* __restrict__ off,
                                                                 int32_t* rows,
                                                                 int32_t* __restrict__ pts) {
  extern __shared__ int cnt[];  // [N] counters, then [INV_T] scan scratch, then (LROWS) [SK] rows
  int* scan = cnt + N;
  int* lrows = scan + INV_T;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t* ib = idx + (size_t)b * SK;
  for (int j = tid; j < N; j += INV_T) cnt[j] = 0;
  // this thread's entries i = tid, tid + 1024, ...: the first INV_R of them are kept in registers for the fill pass
  // (all loads of the pass in flight at once; 16 x 1024 covers the model's 512 x 32 rows per cloud)
  constexpr int INV_R = 16;
  int mine[INV_R];
#pragma unroll
  for (int u = 0; u < INV_R; ++u) {
    const int i = tid + u * INV_T;
    mine[u] = i < SK ? (int)ib[i] : -1;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < INV_R; ++u)
    if (mine[u] >= 0) atomicAdd(&cnt[mine[u]], 1);
  for (int i = tid + INV_R * INV_T; i < SK; i += INV_T) atomicAdd(&cnt[(int)ib[i]], 1);
  __syncthreads();
  // exclusive scan of cnt[0..N): thread t owns the contiguous chunk [t*per, (t+1)*per)
  const int per = (N + INV_T - 1) / INV_T;
  const int j0 = min(N, tid * per), j1 = min(N, j0 + per);
  int local = 0;
  for (int j = j0; j < j1; ++j) local += cnt[j];
  scan[tid] = local;
  __syncthreads();
  for (int o = 1; o < INV_T; o <<= 1) {
    const int v = tid >= o ? scan[tid - o] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  int run = scan[tid] - local;
  const int first = run;              // this thread's points own the entries [first, last)
  int32_t* ob = off + (size_t)b * (N + 1);
  for (int j = j0; j < j1; ++j) {
    const int c = cnt[j];
    ob[j] = run;
    cnt[j] = run;  // becomes the fill cursor
    run += c;
  }
  const int last = run;
  if (tid == INV_T - 1) ob[N] = scan[INV_T - 1];
  __syncthreads();
  int32_t* rb = rows + (size_t)b * SK;
  int* dst = LROWS ? lrows : rb;
#pragma unroll
  for (int u = 0; u < INV_R; ++u) {
    const int j = mine[u];
    if (j >= 0) dst[atomicAdd(&cnt[j], 1)] = tid + u * INV_T;
  }
  for (int i = tid + INV_R * INV_T; i < SK; i += INV_T) dst[atomicAdd(&cnt[(int)ib[i]], 1)] = i;
  // The fill places a point's rows in the order the atomic cursors were taken; sorted ascending, the per-point sums that walk
  // these lists (csrc/sapool.hip) add in the same order in every run.  Thread t sorts the lists of its own points (8-16 rows
  // each in the model: an insertion sort in place) and names the point of each of its entries.
  __syncthreads();
  int at = first;
  for (int j = j0; j < j1; ++j) {
    const int hi = cnt[j];            // (the cursor stands at the end of point j's list)
    for (int i = at + 1; i < hi; ++i) {
      const int v = dst[i];
      int k = i - 1;
      while (k >= at && dst[k] > v) dst[k + 1] = dst[k], --k;
      dst[k + 1] = v;
    }
    if (pts)
      for (int i = at; i < hi; ++i) pts[(size_t)b * SK + i] = j;
    at = hi;
  }
  (void)last;
  if (LROWS) {
    __syncthreads();
    for (int i = tid; i < SK; i += INV_T) rb[i] = lrows[i];
  }
}

// P[row, :] += W1[:, 0:3] xyz[row]  (rows = B*N points)   and   Q[g, :] = b1 - W1[:, 0:3] new_xyz[g]  (g = B*S groups):
// with the coordinate term split this way a generated row is relu(P[idx] + Q[g]) — one add and one max per element
// where the round-1 form needed three fmas with per-row broadcasts.  Thread = 4 channels of one row / group.
// One multiply-add of the coordinate term kept out of the packed-fp32 unit.  Under plain -O3 the compiler pairs the four
// channels of a thread into v_pk_mul_f32 / v_pk_fma_f32 with op_sel modifiers (one coordinate against two weights).
// Measured on MI355X: while a workgroup of the general matrix-core engine (gemm_kernel) or of the generated-row kernel
// shares the CU — the other encoder's stream — that packed form returned sums with the y term missing in lanes 48-63
// (a few rows per launch, 0.1-0.2 absolute in P; tests/test_gpu_concurrency.py reproduces it).  The same arithmetic as
// single v_fma_f32 / v_mul_f32 (bit-identical results) is not affected, and this kernel is bound by its row traffic anyway.
__device__ __forceinline__ float prep_mul(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float prep_fma(float a, float b, float c) {
  float r;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float prep_add(float a, float b) {
  float r;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float prep_sub(float a, float b) {
  float r;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__global__ __launch_bounds__(256) void sa_prep_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                      const float* __restrict__ W1, int ldw, const float* __restrict__ b1,
                                                      long prow, long groups, int C1, float* __restrict__ P,
                                                      float* __restrict__ Q) {
  // the three coordinate columns of W1 (and b1), once per workgroup, as planes {wx[C1], wy[C1], wz[C1], b[C1]}: a thread
  // then reads its four channels of each plane with one 16-byte LDS load instead of twelve scattered global dwords
  extern __shared__ __attribute__((aligned(16))) float sw[];
  for (int t = threadIdx.x; t < C1; t += blockDim.x) {
    const float* w = W1 + (size_t)t * ldw;
    sw[t] = w[0], sw[C1 + t] = w[1], sw[2 * C1 + t] = w[2], sw[3 * C1 + t] = b1 ? b1[t] : 0.f;
  }
  __syncthreads();
  const int c4 = C1 >> 2;
  const long total = (prow + groups) * c4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / c4;
    const int c = (int)(e - r * c4) * 4;
    const bool point = r < prow;
    const float* q = point ? xyz + (size_t)r * 3 : new_xyz + (size_t)(r - prow) * 3;
    const float x = q[0], y = q[1], z = q[2];
    const float4 wx = *reinterpret_cast<const float4*>(sw + c), wy = *reinterpret_cast<const float4*>(sw + C1 + c);
    const float4 wz = *reinterpret_cast<const float4*>(sw + 2 * C1 + c);
    float t[4];      // fmaf(wz, z, fmaf(wy, y, wx * x))
    t[0] = prep_fma(wz.x, z, prep_fma(wy.x, y, prep_mul(wx.x, x)));
    t[1] = prep_fma(wz.y, z, prep_fma(wy.y, y, prep_mul(wx.y, x)));
    t[2] = prep_fma(wz.z, z, prep_fma(wy.z, y, prep_mul(wx.z, x)));
    t[3] = prep_fma(wz.w, z, prep_fma(wy.w, y, prep_mul(wx.w, x)));
    if (point) {
      float4* o = reinterpret_cast<float4*>(P + (size_t)r * C1 + c);
      float4 v = *o;
      v.x = prep_add(v.x, t[0]), v.y = prep_add(v.y, t[1]), v.z = prep_add(v.z, t[2]), v.w = prep_add(v.w, t[3]);
      *o = v;
    } else {
      const float4 bb = *reinterpret_cast<const float4*>(sw + 3 * C1 + c);
      float4 v;
      v.x = prep_sub(bb.x, t[0]), v.y = prep_sub(bb.y, t[1]), v.z = prep_sub(bb.z, t[2]), v.w = prep_sub(bb.w, t[3]);
      *reinterpret_cast<float4*>(Q + (size_t)(r - prow) * C1 + c) = v;
    }
  }
}

}  // namespace

PZN_EXPORT int pzn_sa_prep_f32(const float* xyz, const float* new_xyz, const float* W1, const float* b1, int B, int N, int S,
                               int D, int C1, float* P, float* Q, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && new_xyz && W1 && P && Q && B > 0 && N > 0 && S > 0 && D >= 0 && C1 > 0);
  if ((C1 & 3) || (reinterpret_cast<uintptr_t>(P) & 15) || (reinterpret_cast<uintptr_t>(Q) & 15)) return PZN_EUNSUPPORTED;
  const long prow = (long)B * N, groups = (long)B * S;
  const long total = (prow + groups) * (C1 >> 2);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (C1 > 4096) return PZN_EUNSUPPORTED;      // (4 planes of C1 floats in LDS)
  if (blocks > 1024) blocks = 1024;             // a few rows per thread: the LDS prologue is paid per workgroup
  PZN_LAUNCH(sa_prep_kernel, dim3((unsigned)blocks), dim3(256), (size_t)4 * C1 * sizeof(float), pzn_hip_stream(stream),
                     xyz, new_xyz, W1, 3 + D, b1, prow, groups, C1, P, Q);
  PZN_RETURN_LAUNCH_STATUS();
}

int pzn_sa_prep_q(const float* new_xyz, const float* W1, const float* b1, int B, int S, int D, int C1, float* Q, hipStream_t st) {
  if (!new_xyz || !W1 || !Q || B <= 0 || S <= 0 || C1 <= 0) return PZN_EINVAL;
  if ((C1 & 3) || C1 > 4096 || (reinterpret_cast<uintptr_t>(Q) & 15)) return PZN_EUNSUPPORTED;
  const long groups = (long)B * S;
  long blocks = (groups * (C1 >> 2) + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  PZN_LAUNCH(sa_prep_kernel, dim3((unsigned)blocks), dim3(256), (size_t)4 * C1 * sizeof(float), st, new_xyz, new_xyz, W1, 3 + D, b1,
             0L, groups, C1, Q, Q);      // no point rows: every row of the walk is a row of Q
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_knn_inverse_lists(const int64_t* idx, int B, int N, int S, int K, int32_t* off, int32_t* rows,
                                     int32_t* pts, pzn_stream_t stream) {
  PZN_CHECK_ARG(idx && off && rows && B > 0 && N > 0 && S > 0 && K > 0 && B <= 65535);
  PZN_CHECK_ARG((long)S * K < 2147483647L);
  const size_t base = sizeof(int) * ((size_t)N + INV_T);
  if (base > 150 * 1024) return PZN_EUNSUPPORTED;
  const size_t with_rows = base + sizeof(int) * (size_t)S * K;
  const bool lrows = with_rows <= 150 * 1024;
  const size_t lds = lrows ? with_rows : base;
  const void* fn = lrows ? (const void*)sa_inverse_lists_kernel<true> : (const void*)sa_inverse_lists_kernel<false>;
  if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PZN_ELAUNCH;
  if (lrows)
    PZN_LAUNCH(sa_inverse_lists_kernel<true>, dim3((unsigned)B), dim3(INV_T), lds, pzn_hip_stream(stream), idx, N, S * K, off,
               rows, pts);
  else
    PZN_LAUNCH(sa_inverse_lists_kernel<false>, dim3((unsigned)B), dim3(INV_T), lds, pzn_hip_stream(stream), idx, N, S * K, off,
               rows, pts);
  PZN_RETURN_LAUNCH_STATUS();
}
