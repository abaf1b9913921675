// optim.hip — Adam update of the whole model in one launch.
//
// The reference trains with torch.optim.Adam (model5_b.py:1453-1457).  With every parameter, gradient and
// moment living in one flat buffer each (distributed.FlatGradAllReduce / optim.FlatAdam) the update is a
// single streaming pass over 8 M floats: 16 bytes read + 12 written per element, HBM-bound (~50 us), instead
// of a multi-tensor kernel chain over 104 tensors (~0.4 ms).  Same arithmetic as torch's non-capturable,
// non-amsgrad, zero-weight-decay path (torch/optim/adam.py _single_tensor_adam):
//   m += (g - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g * g;
//   p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
#include <math.h>

#include "pzn_common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   float step_size, float inv_bc2_sqrt, float beta1, float beta2,
                                                   float eps) {
  const size_t n4 = n >> 2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
#define PZN_ADAM1(c)                                                         \
  mm.c = mm.c + (gg.c - mm.c) * (1.f - beta1);                               \
  vv.c = vv.c * beta2 + (1.f - beta2) * gg.c * gg.c;                         \
  pp.c = pp.c - step_size * (mm.c / (sqrtf(vv.c) * inv_bc2_sqrt + eps));
    PZN_ADAM1(x) PZN_ADAM1(y) PZN_ADAM1(z) PZN_ADAM1(w)
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n % 4 elements)
  for (size_t i = (n4 << 2) + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float4 pp, mm, vv, gg;
    pp.x = p[i], mm.x = m[i], vv.x = v[i], gg.x = g[i];
    PZN_ADAM1(x)
    p[i] = pp.x, m[i] = mm.x, v[i] = vv.x;
  }
#undef PZN_ADAM1
}

}  // namespace

PZN_EXPORT int pzn_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n,
                                 float lr, float beta1, float beta2, float eps, int step, pzn_stream_t stream) {
  PZN_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1);
  PZN_CHECK_ARG(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) |
                  reinterpret_cast<uintptr_t>(exp_avg) | reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0);
  // bias corrections in double on the host, as torch computes them in Python floats
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  size_t blocks = ((n >> 2) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  PZN_LAUNCH(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, pzn_hip_stream(stream), param, grad, exp_avg,
                     exp_avg_sq, n, step_size, inv_bc2_sqrt, beta1, beta2, eps);
  PZN_RETURN_LAUNCH_STATUS();
}
