// sapoint.hip — first shared-MLP layer of a set-abstraction level computed PER POINT instead of per grouped row.
//
// Reference (pointnet_util.py:123-132 + model5_b.py:452 / :459): every one of the B*S*32 grouped rows is
// {xyz[j] - centre, feat[j]} with j = idx[b,s,k], and the first 1x1 convolution multiplies each row by W1[C1, 3+D].
// The feature block of that product depends on the POINT only, and a point is gathered by 8-16 rows:
//     h[b,s,k,:] = relu( W1[:,0:3] (xyz[j] - centre_s)  +  P[b,j,:]  +  b1 ),     P = feat W1[:,3:]^T  per point.
// So the product shrinks from B*S*32 rows to B*N rows (8x / 16x fewer flops, same result up to the order of the
// fp32 sum), the grouped tensor [B,S,32,3+D] is never written, and the layer becomes a gather: read P rows
// (L2 / MALL resident: 67 MB at level 1), write h.  Backward, with dh = the (ReLU-masked) gradient of h:
//     dP[b,j,:] = sum over the rows that gathered j of dh[row,:]      (inverse neighbour lists, no atomics on rows)
//     dW1[:,0:3] += dh^T (xyz[j] - centre),   db1 += column sums of dh            (same pass over dh)
//     dfeat = dP W1[:,3:],   dW1[:,3:] += dP^T feat                               (two per-point GEMMs, callers)
// The HBM bill of the layer: forward 1 write of h; backward 1 read of dh.  (The grouped-row path: forward reads
// the grouped rows and writes h; backward reads dh twice, the grouped rows once, and scatter-adds B*S*32*D floats.)
#include "pzn_common.h"

namespace {

constexpr int SP_T = 256;  // 4 wavefronts

template <int V>
struct VecT;
template <>
struct VecT<1> {
  typedef float type;
};
template <>
struct VecT<2> {
  typedef float2 type;
};
template <>
struct VecT<4> {
  typedef float4 type;
};

template <int V>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[V]) {
  typename VecT<V>::type t = *reinterpret_cast<const typename VecT<V>::type*>(p);
  const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
  for (int i = 0; i < V; ++i) v[i] = f[i];
}
template <int V>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[V]) {
  typename VecT<V>::type t;
  float* f = reinterpret_cast<float*>(&t);
#pragma unroll
  for (int i = 0; i < V; ++i) f[i] = v[i];
  *reinterpret_cast<typename VecT<V>::type*>(p) = t;
}

__device__ __forceinline__ float bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// h rows.  A wavefront takes 64 consecutive rows: lane l fetches the index / offset of row l, then the rows are
// written one after the other, all 64 lanes on the C1 = 64*V channels of one row (coalesced P read, h write).
template <int V>
__global__ __launch_bounds__(SP_T) void sa_point_l1_fwd_kernel(const float* __restrict__ xyz,
                                                               const float* __restrict__ new_xyz,
                                                               const int64_t* __restrict__ idx,
                                                               const float* __restrict__ P,
                                                               const float* __restrict__ W1, int ldw,
                                                               const float* __restrict__ b1, int N, int S, long rows,
                                                               float* __restrict__ h) {
  constexpr int C1 = 64 * V;
  const int lane = threadIdx.x & 63;
  const long gw = (long)blockIdx.x * (SP_T / 64) + (threadIdx.x >> 6), nw = (long)gridDim.x * (SP_T / 64);
  float wx[V], wy[V], wz[V], bb[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = lane * V + i;
    wx[i] = W1[(size_t)c * ldw], wy[i] = W1[(size_t)c * ldw + 1], wz[i] = W1[(size_t)c * ldw + 2];
    bb[i] = b1 ? b1[c] : 0.f;
  }
  const long nbatch = (rows + 63) >> 6;
  for (long bt = gw; bt < nbatch; bt += nw) {
    const long row = bt * 64 + lane;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    int prow = 0;  // row of P (b*N + j); fits 32 bits: B*N*C1 floats are addressed through size_t below
    if (row < rows) {
      const long grp = row >> 5;         // b*S + s
      const long b = grp / S;
      const int j = (int)idx[row];
      const float* q = xyz + ((size_t)b * N + j) * 3;
      const float* c = new_xyz + (size_t)grp * 3;
      dx = q[0] - c[0], dy = q[1] - c[1], dz = q[2] - c[2];  // pointnet_util.py:124
      prow = (int)(b * N + j);
    }
    const int nr = (int)min((long)64, rows - bt * 64);
    auto finish = [&](float (&v)[V], int r) {
      const float rx = bcast(dx, r), ry = bcast(dy, r), rz = bcast(dz, r);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float t = fmaf(wz[i], rz, fmaf(wy[i], ry, wx[i] * rx)) + v[i] + bb[i];
        v[i] = t > 0.f ? t : 0.f;
      }
      store_vec<V>(h + ((size_t)bt * 64 + r) * C1 + lane * V, v);
    };
    int r = 0;
    for (; r + 4 <= nr; r += 4) {  // four P rows in flight
      float v0[V], v1[V], v2[V], v3[V];
      load_vec<V>(P + (size_t)__builtin_amdgcn_readlane(prow, r) * C1 + lane * V, v0);
      load_vec<V>(P + (size_t)__builtin_amdgcn_readlane(prow, r + 1) * C1 + lane * V, v1);
      load_vec<V>(P + (size_t)__builtin_amdgcn_readlane(prow, r + 2) * C1 + lane * V, v2);
      load_vec<V>(P + (size_t)__builtin_amdgcn_readlane(prow, r + 3) * C1 + lane * V, v3);
      finish(v0, r);
      finish(v1, r + 1);
      finish(v2, r + 2);
      finish(v3, r + 3);
    }
    for (; r < nr; ++r) {
      float v0[V];
      load_vec<V>(P + (size_t)__builtin_amdgcn_readlane(prow, r) * C1 + lane * V, v0);
      finish(v0, r);
    }
  }
}

// Inverse neighbour lists of one cloud per workgroup: off[b][0..N] (exclusive prefix of the reference counts) and
// rows[b][.] = the in-cloud row numbers (s*32 + k) grouped by the point they gathered.  Counters live in LDS.
constexpr int INV_T = 1024;
__global__ __launch_bounds__(INV_T) void sa_inverse_lists_kernel(const int64_t* __restrict__ idx, int N, int SK,
                                                                 int32_t* __restrict__ off,
                                                                 int32_t* __restrict__ rows) {
  extern __shared__ int cnt[];  // [N] counters, then [INV_T] scan scratch
  int* scan = cnt + N;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t* ib = idx + (size_t)b * SK;
  for (int j = tid; j < N; j += INV_T) cnt[j] = 0;
  __syncthreads();
  for (int i = tid; i < SK; i += INV_T) atomicAdd(&cnt[(int)ib[i]], 1);
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
  int32_t* ob = off + (size_t)b * (N + 1);
  for (int j = j0; j < j1; ++j) {
    const int c = cnt[j];
    ob[j] = run;
    cnt[j] = run;  // becomes the fill cursor
    run += c;
  }
  if (tid == INV_T - 1) ob[N] = scan[INV_T - 1];
  __syncthreads();
  int32_t* rb = rows + (size_t)b * SK;
  for (int i = tid; i < SK; i += INV_T) {
    const int pos = atomicAdd(&cnt[(int)ib[i]], 1);
    rb[pos] = i;
  }
}

// dP and the xyz / bias part of the first layer's gradients.  A wavefront per point: the rows that gathered it come
// out of the inverse list 64 at a time (lane l holds row l and its centre offset), then one coalesced read of
// dh[row, :] per row, all lanes on the C1 = 64*V channels.  The four per-channel sums for dW1[:,0:3] and db1 stay in
// registers over all points of the wavefront and meet in LDS at the end: one set of atomics per workgroup.
template <int V>
__global__ __launch_bounds__(SP_T) void sa_point_l1_bwd_kernel(const float* __restrict__ dh,
                                                               const float* __restrict__ xyz,
                                                               const float* __restrict__ new_xyz,
                                                               const int32_t* __restrict__ off,
                                                               const int32_t* __restrict__ rows, int N, int S,
                                                               long npoints, float* __restrict__ dP,
                                                               float* __restrict__ dW1, int ldw,
                                                               float* __restrict__ db1) {
  constexpr int C1 = 64 * V;
  __shared__ float red[SP_T / 64][4][C1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * (SP_T / 64) + wave, nw = (long)gridDim.x * (SP_T / 64);
  const int SK = S * 32;
  float ax[V], ay[V], az[V], ab[V];
#pragma unroll
  for (int i = 0; i < V; ++i) ax[i] = ay[i] = az[i] = ab[i] = 0.f;
  for (long p = gw; p < npoints; p += nw) {
    const long b = p / N;
    const int j = (int)(p - b * N);
    const int32_t* ob = off + (size_t)b * (N + 1) + j;
    const int o0 = ob[0], o1 = ob[1];
    const float px = xyz[(size_t)p * 3], py = xyz[(size_t)p * 3 + 1], pz = xyz[(size_t)p * 3 + 2];
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    for (int base = o0; base < o1; base += 64) {
      const int m = min(64, o1 - base);
      int rid = 0;
      float dx = 0.f, dy = 0.f, dz = 0.f;
      if (lane < m) {
        rid = rows[(size_t)b * SK + base + lane];
        const float* c = new_xyz + ((size_t)b * S + (rid >> 5)) * 3;
        dx = px - c[0], dy = py - c[1], dz = pz - c[2];
      }
      auto take = [&](const float (&g)[V], int r) {
        const float rx = bcast(dx, r), ry = bcast(dy, r), rz = bcast(dz, r);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          acc[i] += g[i];
          ax[i] = fmaf(g[i], rx, ax[i]);
          ay[i] = fmaf(g[i], ry, ay[i]);
          az[i] = fmaf(g[i], rz, az[i]);
        }
      };
      const float* dhb = dh + (size_t)b * SK * C1 + lane * V;
      int r = 0;
      for (; r + 4 <= m; r += 4) {  // four dh rows in flight
        float g0[V], g1[V], g2[V], g3[V];
        load_vec<V>(dhb + (size_t)__builtin_amdgcn_readlane(rid, r) * C1, g0);
        load_vec<V>(dhb + (size_t)__builtin_amdgcn_readlane(rid, r + 1) * C1, g1);
        load_vec<V>(dhb + (size_t)__builtin_amdgcn_readlane(rid, r + 2) * C1, g2);
        load_vec<V>(dhb + (size_t)__builtin_amdgcn_readlane(rid, r + 3) * C1, g3);
        take(g0, r);
        take(g1, r + 1);
        take(g2, r + 2);
        take(g3, r + 3);
      }
      for (; r < m; ++r) {
        float g0[V];
        load_vec<V>(dhb + (size_t)__builtin_amdgcn_readlane(rid, r) * C1, g0);
        take(g0, r);
      }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) ab[i] += acc[i];
    store_vec<V>(dP + (size_t)p * C1 + lane * V, acc);
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    red[wave][0][lane * V + i] = ax[i];
    red[wave][1][lane * V + i] = ay[i];
    red[wave][2][lane * V + i] = az[i];
    red[wave][3][lane * V + i] = ab[i];
  }
  __syncthreads();
  for (int f = threadIdx.x; f < 4 * C1; f += SP_T) {
    const int q = f / C1, c = f - q * C1;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SP_T / 64; ++w) t += red[w][q][c];
    if (q < 3)
      atomicAdd(dW1 + (size_t)c * ldw + q, t);
    else if (db1)
      atomicAdd(db1 + c, t);
  }
}

}  // namespace

PZN_EXPORT int pzn_sa_point_l1_fwd_f32(const float* xyz, const float* new_xyz, const int64_t* idx, const float* P,
                                       const float* W1, const float* b1, int B, int N, int S, int D, int C1, float* h,
                                       pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && new_xyz && idx && P && W1 && h && B > 0 && N > 0 && S > 0 && D >= 0);
  PZN_CHECK_ARG((long)B * N < 2147483647L);
  if (C1 != 64 && C1 != 128 && C1 != 256) return PZN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(P) & 15) || (reinterpret_cast<uintptr_t>(h) & 15)) return PZN_EUNSUPPORTED;
  const long rows = (long)B * S * 32;
  const long nbatch = (rows + 63) / 64;
  long blocks = (nbatch + 3) / 4;
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = pzn_hip_stream(stream);
  const dim3 grid((unsigned)blocks), block(SP_T);
  const int ldw = 3 + D;
  if (C1 == 64)
    hipLaunchKernelGGL(sa_point_l1_fwd_kernel<1>, grid, block, 0, st, xyz, new_xyz, idx, P, W1, ldw, b1, N, S, rows, h);
  else if (C1 == 128)
    hipLaunchKernelGGL(sa_point_l1_fwd_kernel<2>, grid, block, 0, st, xyz, new_xyz, idx, P, W1, ldw, b1, N, S, rows, h);
  else
    hipLaunchKernelGGL(sa_point_l1_fwd_kernel<4>, grid, block, 0, st, xyz, new_xyz, idx, P, W1, ldw, b1, N, S, rows, h);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_knn_inverse_lists(const int64_t* idx, int B, int N, int S, int K, int32_t* off, int32_t* rows,
                                     pzn_stream_t stream) {
  PZN_CHECK_ARG(idx && off && rows && B > 0 && N > 0 && S > 0 && K > 0 && B <= 65535);
  PZN_CHECK_ARG((long)S * K < 2147483647L);
  const size_t lds = sizeof(int) * ((size_t)N + INV_T);
  if (lds > 150 * 1024) return PZN_EUNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)sa_inverse_lists_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
          hipSuccess)
    return PZN_ELAUNCH;
  hipLaunchKernelGGL(sa_inverse_lists_kernel, dim3((unsigned)B), dim3(INV_T), lds, pzn_hip_stream(stream), idx, N, S * K,
                     off, rows);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_sa_point_l1_bwd_f32(const float* dh, const float* xyz, const float* new_xyz, const int32_t* off,
                                       const int32_t* rows, int B, int N, int S, int D, int C1, float* dP, float* dW1,
                                       float* db1, pzn_stream_t stream) {
  PZN_CHECK_ARG(dh && xyz && new_xyz && off && rows && dP && dW1 && B > 0 && N > 0 && S > 0 && D >= 0);
  if (C1 != 64 && C1 != 128 && C1 != 256) return PZN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(dP) & 15) || (reinterpret_cast<uintptr_t>(dh) & 15)) return PZN_EUNSUPPORTED;
  const long npoints = (long)B * N;
  long blocks = (npoints + 15) / 16;  // >= 4 points per wavefront
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipStream_t st = pzn_hip_stream(stream);
  const dim3 grid((unsigned)blocks), block(SP_T);
  const int ldw = 3 + D;
  if (C1 == 64)
    hipLaunchKernelGGL(sa_point_l1_bwd_kernel<1>, grid, block, 0, st, dh, xyz, new_xyz, off, rows, N, S, npoints, dP, dW1,
                       ldw, db1);
  else if (C1 == 128)
    hipLaunchKernelGGL(sa_point_l1_bwd_kernel<2>, grid, block, 0, st, dh, xyz, new_xyz, off, rows, N, S, npoints, dP, dW1,
                       ldw, db1);
  else
    hipLaunchKernelGGL(sa_point_l1_bwd_kernel<4>, grid, block, 0, st, dh, xyz, new_xyz, off, rows, N, S, npoints, dP, dW1,
                       ldw, db1);
  PZN_RETURN_LAUNCH_STATUS();
}
