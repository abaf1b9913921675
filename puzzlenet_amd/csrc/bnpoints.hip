// bnpoints.hip — BatchNorm1d(num_points) + ReLU of the per-point feature MLP (model5_b.py:424, :447-448).
//
// The reference applies nn.BatchNorm1d(num_points) to a [B, N, C] tensor, so the "channel" axis is the POINT index n
// and the statistics of point n run over its B*C values x[:, n, :].  B <= 64, C <= 64 (the model): one workgroup per
// point with the values in registers (one HBM read); other shapes: one wavefront per point, the B rows of C floats
// read once from HBM and twice more out of L1 / L2 (mean, centred second moment, output);
// normalisation + affine + ReLU fused, running statistics updated as torch does (momentum on the mean and on the
// UNBIASED variance).  Backward recomputes the ReLU gate from x and the saved statistics.
#include "pzn_common.h"

namespace {

constexpr int BN_T = 256;  // 4 points per workgroup

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, PZN_WAVE);
  return v;
}

__global__ __launch_bounds__(BN_T) void bn_points_relu_fwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, int training,
                                                                  float momentum, float eps, int B, int N, int C,
                                                                  float* __restrict__ y, float* __restrict__ save_mean,
                                                                  float* __restrict__ save_invstd) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (BN_T / 64) + (threadIdx.x >> 6);
  if (n >= N) return;
  const size_t bstride = (size_t)N * C;
  const float* xn = x + (size_t)n * C;
  float mean, invstd;
  if (training) {
    const float cnt = (float)B * (float)C;
    float s = 0.f;
    for (int b = 0; b < B; ++b)
      for (int c = lane; c < C; c += 64) s += xn[b * bstride + c];
    mean = wave_sum(s) / cnt;
    float q = 0.f;
    for (int b = 0; b < B; ++b)
      for (int c = lane; c < C; c += 64) {
        const float d = xn[b * bstride + c] - mean;
        q = fmaf(d, d, q);
      }
    const float var = wave_sum(q) / cnt;  // biased: what normalises the batch
    invstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
      if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mean;
      if (running_var) {
        const float unbiased = cnt > 1.f ? var * (cnt / (cnt - 1.f)) : var;
        running_var[n] = (1.f - momentum) * running_var[n] + momentum * unbiased;
      }
    }
  } else {
    mean = running_mean[n];
    invstd = 1.0f / sqrtf(running_var[n] + eps);
  }
  if (lane == 0) {
    if (save_mean) save_mean[n] = mean;
    if (save_invstd) save_invstd[n] = invstd;
  }
  const float w = weight ? weight[n] : 1.f, bb = bias ? bias[n] : 0.f;
  float* yn = y + (size_t)n * C;
  for (int b = 0; b < B; ++b)
    for (int c = lane; c < C; c += 64) {
      const float t = fmaf((xn[b * bstride + c] - mean) * invstd, w, bb);
      yn[b * bstride + c] = t > 0.f ? t : 0.f;
    }
}

__global__ __launch_bounds__(BN_T) void bn_points_relu_bwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ dy,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ save_mean,
                                                                  const float* __restrict__ save_invstd, int training,
                                                                  int B, int N, int C, float* __restrict__ dx,
                                                                  float* __restrict__ dweight,
                                                                  float* __restrict__ dbias) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * (BN_T / 64) + (threadIdx.x >> 6);
  if (n >= N) return;
  const size_t bstride = (size_t)N * C;
  const float* xn = x + (size_t)n * C;
  const float* gn = dy + (size_t)n * C;
  const float mean = save_mean[n], invstd = save_invstd[n];
  const float w = weight ? weight[n] : 1.f, bb = bias ? bias[n] : 0.f;
  float sg = 0.f, sgx = 0.f;
  for (int b = 0; b < B; ++b)
    for (int c = lane; c < C; c += 64) {
      const float xh = (xn[b * bstride + c] - mean) * invstd;
      const float g = fmaf(xh, w, bb) > 0.f ? gn[b * bstride + c] : 0.f;  // ReLU gate recomputed
      sg += g;
      sgx = fmaf(g, xh, sgx);
    }
  sg = wave_sum(sg);
  sgx = wave_sum(sgx);
  if (lane == 0) {
    if (dweight) atomicAdd(dweight + n, sgx);
    if (dbias) atomicAdd(dbias + n, sg);
  }
  if (!dx) return;
  const float cnt = (float)B * (float)C;
  const float k = w * invstd;
  // eval mode: the statistics are constants, dx = g * w * invstd
  const float m1 = training ? sg / cnt : 0.f, m2 = training ? sgx / cnt : 0.f;
  float* dn = dx + (size_t)n * C;
  for (int b = 0; b < B; ++b)
    for (int c = lane; c < C; c += 64) {
      const float xh = (xn[b * bstride + c] - mean) * invstd;
      const float g = fmaf(xh, w, bb) > 0.f ? gn[b * bstride + c] : 0.f;
      dn[b * bstride + c] = k * (g - m1 - xh * m2);
    }
}

// B <= 64, C <= 64 (the model's shapes): one WORKGROUP per point, the point's values live in registers (wavefront w
// holds rows w, w+4, ...; lane = feature), one HBM read, the two reductions meet through LDS.
constexpr int BN_R = 16;  // rows per wavefront

__device__ __forceinline__ float block_sum4(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();  // red may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(BN_T) void bn_point_block_fwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ running_mean,
                                                                  float* __restrict__ running_var, int training,
                                                                  float momentum, float eps, int B, int N, int C,
                                                                  float* __restrict__ y, float* __restrict__ save_mean,
                                                                  float* __restrict__ save_invstd) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = blockIdx.x;
  const size_t bstride = (size_t)N * C;
  const float* xn = x + (size_t)n * C + lane;
  float v[BN_R];
#pragma unroll
  for (int i = 0; i < BN_R; ++i) {
    const int b = w + 4 * i;
    v[i] = (b < B && lane < C) ? xn[b * bstride] : 0.f;
  }
  float mean, invstd;
  if (training) {
    const float cnt = (float)B * (float)C;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < BN_R; ++i) s += v[i];
    mean = block_sum4(s, red) / cnt;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < BN_R; ++i) {
      const float d = v[i] - mean;
      q = (w + 4 * i < B && lane < C) ? fmaf(d, d, q) : q;
    }
    const float var = block_sum4(q, red) / cnt;
    invstd = 1.0f / sqrtf(var + eps);
    if (threadIdx.x == 0) {
      if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * mean;
      if (running_var) {
        const float unbiased = cnt > 1.f ? var * (cnt / (cnt - 1.f)) : var;
        running_var[n] = (1.f - momentum) * running_var[n] + momentum * unbiased;
      }
    }
  } else {
    mean = running_mean[n];
    invstd = 1.0f / sqrtf(running_var[n] + eps);
  }
  if (threadIdx.x == 0) {
    if (save_mean) save_mean[n] = mean;
    if (save_invstd) save_invstd[n] = invstd;
  }
  const float wt = weight ? weight[n] : 1.f, bb = bias ? bias[n] : 0.f;
  float* yn = y + (size_t)n * C + lane;
#pragma unroll
  for (int i = 0; i < BN_R; ++i) {
    const int b = w + 4 * i;
    if (b < B && lane < C) {
      const float t = fmaf((v[i] - mean) * invstd, wt, bb);
      yn[b * bstride] = t > 0.f ? t : 0.f;
    }
  }
}

__global__ __launch_bounds__(BN_T) void bn_point_block_bwd_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ dy,
                                                                  const float* __restrict__ weight,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ save_mean,
                                                                  const float* __restrict__ save_invstd, int training,
                                                                  int B, int N, int C, float* __restrict__ dx,
                                                                  float* __restrict__ dweight,
                                                                  float* __restrict__ dbias) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = blockIdx.x;
  const size_t bstride = (size_t)N * C;
  const float* xn = x + (size_t)n * C + lane;
  const float* gn = dy + (size_t)n * C + lane;
  const float mean = save_mean[n], invstd = save_invstd[n];
  const float wt = weight ? weight[n] : 1.f, bb = bias ? bias[n] : 0.f;
  float xh[BN_R], g[BN_R];
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int i = 0; i < BN_R; ++i) {
    const int b = w + 4 * i;
    const bool ok = b < B && lane < C;
    xh[i] = ok ? (xn[b * bstride] - mean) * invstd : 0.f;
    const float gi = ok ? gn[b * bstride] : 0.f;
    g[i] = fmaf(xh[i], wt, bb) > 0.f ? gi : 0.f;  // ReLU gate recomputed
    sg += g[i];
    sgx = fmaf(g[i], xh[i], sgx);
  }
  sg = block_sum4(sg, red);
  sgx = block_sum4(sgx, red);
  if (threadIdx.x == 0) {
    if (dweight) atomicAdd(dweight + n, sgx);
    if (dbias) atomicAdd(dbias + n, sg);
  }
  if (!dx) return;
  const float cnt = (float)B * (float)C;
  const float k = wt * invstd;
  const float m1 = training ? sg / cnt : 0.f, m2 = training ? sgx / cnt : 0.f;
  float* dn = dx + (size_t)n * C + lane;
#pragma unroll
  for (int i = 0; i < BN_R; ++i) {
    const int b = w + 4 * i;
    if (b < B && lane < C) dn[b * bstride] = k * (g[i] - m1 - xh[i] * m2);
  }
}

}  // namespace

PZN_EXPORT int pzn_bn_points_relu_fwd_f32(const float* x, const float* weight, const float* bias, float* running_mean,
                                          float* running_var, int training, float momentum, float eps, int B, int N,
                                          int C, float* y, float* save_mean, float* save_invstd, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && y && B > 0 && N > 0 && C > 0 && eps >= 0.f);
  PZN_CHECK_ARG(training || (running_mean && running_var));
  if (B <= 4 * BN_R && C <= 64)
    PZN_LAUNCH(bn_point_block_fwd_kernel, dim3((unsigned)N), dim3(BN_T), 0, pzn_hip_stream(stream), x, weight, bias,
                       running_mean, running_var, training, momentum, eps, B, N, C, y, save_mean, save_invstd);
  else
    PZN_LAUNCH(bn_points_relu_fwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(BN_T), 0, pzn_hip_stream(stream), x,
                       weight, bias, running_mean, running_var, training, momentum, eps, B, N, C, y, save_mean,
                       save_invstd);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_bn_points_relu_bwd_f32(const float* x, const float* dy, const float* weight, const float* bias,
                                          const float* save_mean, const float* save_invstd, int training, int B, int N,
                                          int C, float* dx, float* dweight, float* dbias, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && dy && save_mean && save_invstd && B > 0 && N > 0 && C > 0);
  if (B <= 4 * BN_R && C <= 64)
    PZN_LAUNCH(bn_point_block_bwd_kernel, dim3((unsigned)N), dim3(BN_T), 0, pzn_hip_stream(stream), x, dy, weight,
                       bias, save_mean, save_invstd, training, B, N, C, dx, dweight, dbias);
  else
    PZN_LAUNCH(bn_points_relu_bwd_kernel, dim3((unsigned)((N + 3) / 4)), dim3(BN_T), 0, pzn_hip_stream(stream), x,
                       dy, weight, bias, save_mean, save_invstd, training, B, N, C, dx, dweight, dbias);
  PZN_RETURN_LAUNCH_STATUS();
}
