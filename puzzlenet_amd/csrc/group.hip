// group.hip — neighbourhood gather / group (+ backward), index_points, square_distance.
//
// Replaces the tail of pointnet_util.sample_and_group (pointnet_util.py:123-132):
//   grouped_xyz   = index_points(xyz, idx)            # gather 12-B rows
//   grouped_norm  = grouped_xyz - new_xyz[:, :, None]
//   grouped_feats = index_points(points, idx)         # gather 4*D-B rows
//   new_points    = cat([grouped_norm, grouped_feats], -1)   # [B,S,K,3+D]
// i.e. four materialised intermediates, by ONE kernel whose only HBM stream is
// the [B,S,K,3+D] output (the per-cloud feature table is N*D*4 B <= 1 MB and is
// re-read ~S*K/N times: it lives in the XCD's L2).
//
// HBM-bound design: one 64-lane wavefront owns one (cloud, centroid) pair, i.e.
// one contiguous K*(3+D)*4-byte chunk of the output (8,576 B at K=32, D=64).
// The chunk is assembled in LDS from 16-byte reads of whole feature rows, then
// streamed out with 16-byte-per-lane fully coalesced stores — the rows
// themselves start at 268-byte offsets, so storing them directly would split
// every 256-B wave store over three cache lines.
#include "pzn_common.h"

namespace {

constexpr int GRP_WAVES = 4;

// Indices come from our own kNN / ball-query kernels, but a ball query with no
// hit yields N (pointnet_util.py:91-95) on which the reference's gather raises;
// device code cannot raise, so indices are clamped to stay memory-safe.
__device__ __forceinline__ int clamp_idx(int64_t j, int N) {
  return j < 0 ? 0 : (j >= N ? N - 1 : (int)j);
}

// Fast path: D % 4 == 0 and K*(3+PAD+D) % 4 == 0.  PAD = 1 is the model-internal layout
// {dx, dy, dz, 0, f_0 .. f_{D-1}}: rows are 16-byte aligned, so the dense engine reads them with
// 16-byte loads and the feature block starts on a 16-byte boundary.
template <int PAD>
__global__ __launch_bounds__(GRP_WAVES* PZN_WAVE) void group_fwd_vec_kernel(
    const float* __restrict__ xyz, const float* __restrict__ feat, const float* __restrict__ new_xyz,
    const int64_t* __restrict__ idx, int N, int S, int K, int D, long total_q, float* __restrict__ out,
    float* __restrict__ grouped_xyz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int W = 3 + PAD + D;
  const int lane = threadIdx.x & (PZN_WAVE - 1);
  const int wave = threadIdx.x / PZN_WAVE;
  float* chunk = reinterpret_cast<float*>(smem_raw) + (size_t)wave * K * W;
  const int V = D >> 2;
  const long q_stride = (long)gridDim.x * GRP_WAVES;
  for (long qi = (long)blockIdx.x * GRP_WAVES + wave; qi < total_q; qi += q_stride) {
    const long b = qi / S;
    const int64_t* qidx = idx + qi * K;
    const float* ctr = new_xyz + qi * 3;
    const float* cf = feat + (size_t)b * N * D;
    const float* cx = xyz + (size_t)b * N * 3;
    // features: K rows x V float4
    for (int t = lane; t < K * V; t += PZN_WAVE) {
      int k = t / V, v = t - k * V;
      int j = clamp_idx(qidx[k], N);
      float4 f = *reinterpret_cast<const float4*>(cf + (size_t)j * D + 4 * v);
      float* dst = chunk + k * W + 3 + PAD + 4 * v;
      if (PAD) {
        *reinterpret_cast<float4*>(dst) = f;
      } else {
        dst[0] = f.x;
        dst[1] = f.y;
        dst[2] = f.z;
        dst[3] = f.w;
      }
    }
    if (PAD)
      for (int k = lane; k < K; k += PZN_WAVE) chunk[k * W + 3] = 0.f;
    // coordinates: K rows x 3
    for (int t = lane; t < K * 3; t += PZN_WAVE) {
      int k = t / 3, c = t - 3 * k;
      int j = clamp_idx(qidx[k], N);
      float p = cx[(size_t)j * 3 + c];
      chunk[k * W + c] = __fsub_rn(p, ctr[c]);  // pointnet_util.py:125
      if (grouped_xyz) grouped_xyz[qi * K * 3 + t] = p;
    }
    pzn::wave_lds_sync();
    float4* o4 = reinterpret_cast<float4*>(out + qi * K * W);
    const float4* c4 = reinterpret_cast<const float4*>(chunk);
    for (int t = lane; t < (K * W) >> 2; t += PZN_WAVE) o4[t] = c4[t];
    pzn::wave_lds_sync();
  }
}

// Generic path: any K, D (incl. D == 0): one flat element per lane per step.
__global__ __launch_bounds__(256) void group_fwd_any_kernel(
    const float* __restrict__ xyz, const float* __restrict__ feat, const float* __restrict__ new_xyz,
    const int64_t* __restrict__ idx, int N, int S, int K, int D, long total, float* __restrict__ out,
    float* __restrict__ grouped_xyz) {
  const int W = 3 + D;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long row = e / W;  // (b*S + s)*K + k
    int c = (int)(e - row * W);
    long qi = row / K;
    long b = qi / S;
    int j = clamp_idx(idx[row], N);
    float v;
    if (c < 3) {
      float p = xyz[((size_t)b * N + j) * 3 + c];
      v = __fsub_rn(p, new_xyz[qi * 3 + c]);
      if (grouped_xyz) grouped_xyz[row * 3 + c] = p;
    } else {
      v = feat[((size_t)b * N + j) * D + (c - 3)];
    }
    out[e] = v;
  }
}

// Backward: scatter-add of grad_out rows into the per-cloud tables.  One wave
// per (cloud, centroid); a 64-lane atomic instruction covers 256 contiguous
// bytes of one destination row (the shape global_atomic_add_f32 runs fastest at).
__global__ __launch_bounds__(GRP_WAVES* PZN_WAVE) void group_bwd_kernel(
    const float* __restrict__ grad_out, const int64_t* __restrict__ idx, int N, int S, int K, int D, long total_q,
    float* __restrict__ grad_xyz, float* __restrict__ grad_feat, float* __restrict__ grad_new_xyz) {
  const int W = 3 + D;
  const int lane = threadIdx.x & (PZN_WAVE - 1);
  const int wave = threadIdx.x / PZN_WAVE;
  const long q_stride = (long)gridDim.x * GRP_WAVES;
  for (long qi = (long)blockIdx.x * GRP_WAVES + wave; qi < total_q; qi += q_stride) {
    const long b = qi / S;
    const int64_t* qidx = idx + qi * K;
    const float* go = grad_out + qi * K * W;
    float acc = 0.f;  // lanes 0..2: sum_k grad_out[..., lane]
    for (int k = 0; k < K; ++k) {
      int j = clamp_idx(qidx[k], N);
      const float* g = go + (size_t)k * W;
      if (lane < 3) {
        float v = g[lane];
        acc += v;
        if (grad_xyz) atomicAdd(grad_xyz + ((size_t)b * N + j) * 3 + lane, v);
      }
      if (grad_feat) {
        float* dst = grad_feat + ((size_t)b * N + j) * D;
        for (int c = lane; c < D; c += PZN_WAVE) atomicAdd(dst + c, g[3 + c]);
      }
    }
    if (grad_new_xyz && lane < 3) grad_new_xyz[qi * 3 + lane] = -acc;
  }
}

// index_points (pointnet_util.py:39-50) on a flattened index: out[b,m,:] = points[b, idx[b,m], :].
__global__ __launch_bounds__(256) void gather_fwd_kernel(const float* __restrict__ points,
                                                         const int64_t* __restrict__ idx, int N, int M, int C,
                                                         long total, float* __restrict__ out) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long row = e / C;
    int c = (int)(e - row * C);
    long b = row / M;
    int j = clamp_idx(idx[row], N);
    out[e] = points[((size_t)b * N + j) * C + c];
  }
}

__global__ __launch_bounds__(256) void gather_bwd_kernel(const float* __restrict__ grad_out,
                                                         const int64_t* __restrict__ idx, int N, int M, int C,
                                                         long total, float* __restrict__ grad_points) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long row = e / C;
    int c = (int)(e - row * C);
    long b = row / M;
    int j = clamp_idx(idx[row], N);
    atomicAdd(grad_points + ((size_t)b * N + j) * C + c, grad_out[e]);
  }
}

// square_distance (pointnet_util.py:22-36), materialising form.
__global__ __launch_bounds__(256) void sqdist_kernel(const float* __restrict__ src, const float* __restrict__ dst,
                                                     int S, int N, long total, float* __restrict__ out) {
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    long row = e / N;  // b*S + s
    int j = (int)(e - row * N);
    long b = row / S;
    const float* a = src + row * 3;
    const float* p = dst + ((size_t)b * N + j) * 3;
    out[e] = pzn::sqdist3(a[0], a[1], a[2], p[0], p[1], p[2]);
  }
}

inline int flat_grid(long total, int block) {
  long g = (total + block - 1) / block;
  long cap = 256L * 16;  // 256 CUs x 16 resident blocks, grid-stride beyond
  return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

}  // namespace

PZN_EXPORT int pzn_group_fwd_f32(const float* xyz, const float* feat, const float* new_xyz, const int64_t* idx, int B,
                                 int N, int S, int K, int D, float* out, float* grouped_xyz, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && new_xyz && idx && out && B > 0 && N > 0 && S > 0 && K > 0 && D >= 0);
  PZN_CHECK_ARG(D == 0 || feat);
  hipStream_t st = pzn_hip_stream(stream);
  const int W = 3 + D;
  const long total_q = (long)B * S;
  size_t lds = (size_t)GRP_WAVES * K * W * sizeof(float);
  if (D > 0 && (D & 3) == 0 && ((K * W) & 3) == 0 && lds <= 150 * 1024 &&
      (reinterpret_cast<uintptr_t>(feat) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(&group_fwd_vec_kernel<0>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return PZN_ELAUNCH;
    long blocks = (total_q + GRP_WAVES - 1) / GRP_WAVES;
    long cap = 256L * 8;
    int grid = (int)(blocks < cap ? blocks : cap);
    PZN_LAUNCH(group_fwd_vec_kernel<0>, dim3(grid), dim3(GRP_WAVES * PZN_WAVE), lds, st, xyz, feat, new_xyz, idx,
                       N, S, K, D, total_q, out, grouped_xyz);
  } else {
    long total = total_q * K * W;
    PZN_LAUNCH(group_fwd_any_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, st, xyz, feat, new_xyz, idx, N,
                       S, K, D, total, out, grouped_xyz);
  }
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_group_bwd_f32(const float* grad_out, const int64_t* idx, int B, int N, int S, int K, int D,
                                 float* grad_xyz, float* grad_feat, float* grad_new_xyz, pzn_stream_t stream) {
  PZN_CHECK_ARG(grad_out && idx && B > 0 && N > 0 && S > 0 && K > 0 && D >= 0);
  if (!grad_xyz && !grad_feat && !grad_new_xyz) return PZN_OK;
  if (D == 0) grad_feat = nullptr;
  hipStream_t st = pzn_hip_stream(stream);
  const long total_q = (long)B * S;
  long blocks = (total_q + GRP_WAVES - 1) / GRP_WAVES;
  long cap = 256L * 8;
  int grid = (int)(blocks < cap ? blocks : cap);
  PZN_LAUNCH(group_bwd_kernel, dim3(grid), dim3(GRP_WAVES * PZN_WAVE), 0, st, grad_out, idx, N, S, K, D,
                     total_q, grad_xyz, grad_feat, grad_new_xyz);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_gather_fwd_f32(const float* points, const int64_t* idx, int B, int N, int M, int C, float* out,
                                  pzn_stream_t stream) {
  PZN_CHECK_ARG(points && idx && out && B > 0 && N > 0 && M > 0 && C > 0);
  long total = (long)B * M * C;
  PZN_LAUNCH(gather_fwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, pzn_hip_stream(stream), points, idx,
                     N, M, C, total, out);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_gather_bwd_f32(const float* grad_out, const int64_t* idx, int B, int N, int M, int C,
                                  float* grad_points, pzn_stream_t stream) {
  PZN_CHECK_ARG(grad_out && idx && grad_points && B > 0 && N > 0 && M > 0 && C > 0);
  long total = (long)B * M * C;
  PZN_LAUNCH(gather_bwd_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, pzn_hip_stream(stream), grad_out,
                     idx, N, M, C, total, grad_points);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_square_distance_f32(const float* src, const float* dst, int B, int S, int N, float* out,
                                       pzn_stream_t stream) {
  PZN_CHECK_ARG(src && dst && out && B > 0 && S > 0 && N > 0);
  long total = (long)B * S * N;
  PZN_LAUNCH(sqdist_kernel, dim3(flat_grid(total, 256)), dim3(256), 0, pzn_hip_stream(stream), src, dst, S, N,
                     total, out);
  PZN_RETURN_LAUNCH_STATUS();
}

// ------------------------------------------------------- max over the point axis (model5_b.py:475, :741)
// out[b, c] = max_l x[b, l, c] with the arg-max row (lowest row on ties), and its backward
// dx[b, l, c] = (l == idx[b, c]) ? dout[b, c] : 0 written as one streaming pass (no separate zero fill).
namespace {

// One workgroup = 128 channels of one cloud: thread (row lane r = tid / 32, channel group q = tid % 32) walks rows r, r + 8,
// ... reading 16 bytes (4 channels) per row — 32 threads fetch 512 contiguous bytes of a row (the first version read 128-byte
// pieces at a 4-KB stride and took 84 us per launch inside the step for the 67 MB encoder output).  Rows ascend within a
// thread, so a strict > keeps the first maximum; the eight row lanes meet in LDS, ties to the lower row.
constexpr int MAXPTS_T = 256, MAXPTS_RL = MAXPTS_T / 32;
__global__ __launch_bounds__(MAXPTS_T) void maxpts_fwd_kernel(const float* __restrict__ x, int L, int C,
                                                              float* __restrict__ out, int32_t* __restrict__ idx) {
  __shared__ float sv[MAXPTS_RL][128];
  __shared__ int si[MAXPTS_RL][128];
  const int q = threadIdx.x & 31, r = threadIdx.x >> 5;
  const int b = blockIdx.y, c = blockIdx.x * 128 + 4 * q;
  float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  int bi[4] = {0, 0, 0, 0};
  if (c < C) {      // (C % 4 == 0: a thread's four channels are all inside or all outside)
    const float* p = x + (size_t)b * L * C + c;
    int l = r;
    for (; l + 3 * MAXPTS_RL < L; l += 4 * MAXPTS_RL) {      // four rows in flight
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(l + u * MAXPTS_RL) * C);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ll = l + u * MAXPTS_RL;
        if (v[u].x > best[0]) best[0] = v[u].x, bi[0] = ll;
        if (v[u].y > best[1]) best[1] = v[u].y, bi[1] = ll;
        if (v[u].z > best[2]) best[2] = v[u].z, bi[2] = ll;
        if (v[u].w > best[3]) best[3] = v[u].w, bi[3] = ll;
      }
    }
    for (; l < L; l += MAXPTS_RL) {
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)l * C);
      if (v.x > best[0]) best[0] = v.x, bi[0] = l;
      if (v.y > best[1]) best[1] = v.y, bi[1] = l;
      if (v.z > best[2]) best[2] = v.z, bi[2] = l;
      if (v.w > best[3]) best[3] = v.w, bi[3] = l;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) sv[r][4 * q + i] = best[i], si[r][4 * q + i] = bi[i];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int cc = blockIdx.x * 128 + threadIdx.x;
    if (cc < C) {
      float bv = sv[0][threadIdx.x];
      int bl = si[0][threadIdx.x];
#pragma unroll
      for (int rr = 1; rr < MAXPTS_RL; ++rr) {
        const float v = sv[rr][threadIdx.x];
        const int i = si[rr][threadIdx.x];
        if (v > bv || (v == bv && i < bl)) bv = v, bl = i;
      }
      out[(size_t)b * C + cc] = bv;
      idx[(size_t)b * C + cc] = bl;
    }
  }
}

__global__ __launch_bounds__(256) void maxpts_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                         int L, int C, float* __restrict__ dx) {
  // one thread per 4 consecutive channels of one row (C % 4 == 0), grid-stride over rows
  const int c4 = C >> 2, b = blockIdx.y;
  const size_t per_batch = (size_t)L * c4;
  for (size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x; f < per_batch; f += (size_t)gridDim.x * blockDim.x) {
    const int l = (int)(f / c4), c = (int)(f % c4) * 4;
    const int4 a = *reinterpret_cast<const int4*>(idx + (size_t)b * C + c);
    const float4 g = *reinterpret_cast<const float4*>(dout + (size_t)b * C + c);
    *reinterpret_cast<float4*>(dx + ((size_t)b * L + l) * C + c) =
        make_float4(a.x == l ? g.x : 0.f, a.y == l ? g.y : 0.f, a.z == l ? g.z : 0.f, a.w == l ? g.w : 0.f);
  }
}

}  // namespace

// any C, any alignment: one thread per (cloud, channel), consecutive threads on consecutive channels (the round-1 form;
// the vector kernels above need C % 4 == 0 and 16-byte aligned pointers)
__global__ __launch_bounds__(256) void maxpts_fwd_scalar_kernel(const float* __restrict__ x, int L, int C,
                                                                float* __restrict__ out, int32_t* __restrict__ idx) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  const float* p = x + (size_t)b * L * C + c;
  float best = p[0];
  int bi = 0;
  for (int l = 1; l < L; ++l) {
    const float v = p[(size_t)l * C];
    if (v > best) best = v, bi = l;     // first maximum wins, as torch.max
  }
  out[(size_t)b * C + c] = best;
  idx[(size_t)b * C + c] = bi;
}
__global__ __launch_bounds__(256) void maxpts_bwd_scalar_kernel(const float* __restrict__ dout, const int32_t* __restrict__ idx,
                                                                int L, int C, float* __restrict__ dx) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  const int sel = idx[(size_t)b * C + c];
  const float g = dout[(size_t)b * C + c];
  float* p = dx + (size_t)b * L * C + c;
  for (int l = 0; l < L; ++l) p[(size_t)l * C] = l == sel ? g : 0.f;
}

PZN_EXPORT int pzn_maxpool_points_fwd_f32(const float* x, int B, int L, int C, float* out, int32_t* idx,
                                          pzn_stream_t stream) {
  PZN_CHECK_ARG(x && out && idx && B > 0 && B <= 65535 && L > 0 && C > 0);
  if ((C & 3) != 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) {
    PZN_LAUNCH(maxpts_fwd_scalar_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)B), dim3(256), 0,
                       pzn_hip_stream(stream), x, L, C, out, idx);
    PZN_RETURN_LAUNCH_STATUS();
  }
  PZN_LAUNCH(maxpts_fwd_kernel, dim3((unsigned)((C + 127) / 128), (unsigned)B), dim3(MAXPTS_T), 0,
                     pzn_hip_stream(stream), x, L, C, out, idx);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_maxpool_points_bwd_f32(const float* dout, const int32_t* idx, int B, int L, int C, float* dx,
                                          pzn_stream_t stream) {
  PZN_CHECK_ARG(dout && idx && dx && B > 0 && B <= 65535 && L > 0 && C > 0);
  if ((C & 3) != 0 ||
      ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(dx)) & 15) != 0) {
    PZN_LAUNCH(maxpts_bwd_scalar_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)B), dim3(256), 0,
                       pzn_hip_stream(stream), dout, idx, L, C, dx);
    PZN_RETURN_LAUNCH_STATUS();
  }
  const size_t per_batch = (size_t)L * (C >> 2);
  size_t gx = (per_batch + 255) / 256;
  if (gx > 64) gx = 64;
  PZN_LAUNCH(maxpts_bwd_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, pzn_hip_stream(stream), dout, idx, L,
                     C, dx);
  PZN_RETURN_LAUNCH_STATUS();
}
