// se3.hip — SE(3) exponential map of the pose head, forward and backward (se_math/se3.py:57-80 `exp`,
// so3.py `mat`, sinc.py:6-18 / 96-108 / 126-138 `sinc1/2/3` with their |t| < 0.01 Taylor branches).
//
// A few hundred flops per pair — but as tensor glue it was ~150 launches of 64-element kernels per step
// (Rodrigues terms, Taylor guards and their autograd), 2.3 ms of a 21 ms step.  One thread per twist:
//   w = x[0:3], v = x[3:6], t = |w|, W = skew(w), S = W W
//   R = I + a W + b S,  V = I + b W + c S,  p = V v,  g = [[R, p], [0 0 0 1]]
//   a = sin t / t, b = (1 - cos t) / t^2, c = (t - sin t) / t^3  (series below |t| = 0.01, as the reference)
// Backward from dg: dv = V^T dp;  M = dp v^T;  GW = a dR + b M;  GS = b dR + c M;
//   dt = a' <dR,W> + b' (<dR,S> + <M,W>) + c' <M,S>;  G = GW + GS W^T + W^T GS;
//   dw = (G21 - G12, G02 - G20, G10 - G01) + dt * w / t  (0 at t = 0, as torch's norm backward).
// Arithmetic in double, results rounded to fp32 once.
#include <math.h>

#include "pzn_common.h"

namespace {

struct Sinc {
  double a, b, c, da, db, dc;
};

__device__ Sinc sincs(double t) {
  Sinc s;
  const double u = t * t;
  if (fabs(t) < 0.01) {  // the reference's nested series and their exact derivatives
    s.a = 1 - u / 6 * (1 - u / 20 * (1 - u / 42));
    s.b = 0.5 * (1 - u / 12 * (1 - u / 30 * (1 - u / 56)));
    s.c = 1.0 / 6 * (1 - u / 20 * (1 - u / 42 * (1 - u / 72)));
    s.da = 2 * t * (-1.0 / 6 + u / 60 - u * u / 1680);
    s.db = 2 * t * (-1.0 / 24 + u / 360 - u * u / 13440);
    s.dc = 2 * t * (-1.0 / 120 + u / 2520 - u * u / 120960);
  } else {
    const double sn = sin(t), cs = cos(t);
    s.a = sn / t;
    s.b = (1 - cs) / u;
    s.c = (t - sn) / (u * t);
    s.da = (t * cs - sn) / u;
    s.db = (t * sn - 2 * (1 - cs)) / (u * t);
    s.dc = ((1 - cs) * t - 3 * (t - sn)) / (u * u);
  }
  return s;
}

__device__ void skew(const double* w, double W[3][3]) {
  W[0][0] = 0, W[0][1] = -w[2], W[0][2] = w[1];
  W[1][0] = w[2], W[1][1] = 0, W[1][2] = -w[0];
  W[2][0] = -w[1], W[2][1] = w[0], W[2][2] = 0;
}

__device__ void matmul3(const double A[3][3], const double B[3][3], double C[3][3], bool ta, bool tb) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += (ta ? A[k][i] : A[i][k]) * (tb ? B[j][k] : B[k][j]);
      C[i][j] = acc;
    }
}

__global__ void se3_exp_fwd_kernel(const float* __restrict__ x, int B, float* __restrict__ g) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double w[3], v[3], W[3][3], S[3][3];
  for (int i = 0; i < 3; ++i) w[i] = x[b * 6 + i], v[i] = x[b * 6 + 3 + i];
  const double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const Sinc s = sincs(t);
  skew(w, W);
  matmul3(W, W, S, false, false);
  float* o = g + (size_t)b * 16;
  for (int i = 0; i < 3; ++i) {
    double p = 0;
    for (int j = 0; j < 3; ++j) {
      const double I = i == j ? 1.0 : 0.0;
      o[i * 4 + j] = (float)(I + s.a * W[i][j] + s.b * S[i][j]);
      p += (I + s.b * W[i][j] + s.c * S[i][j]) * v[j];
    }
    o[i * 4 + 3] = (float)p;
  }
  o[12] = 0.f, o[13] = 0.f, o[14] = 0.f, o[15] = 1.f;
}

__global__ void se3_exp_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dg, int B,
                                   float* __restrict__ dx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double w[3], v[3], W[3][3], S[3][3], dR[3][3], dp[3], M[3][3], GW[3][3], GS[3][3], T1[3][3], T2[3][3];
  for (int i = 0; i < 3; ++i) w[i] = x[b * 6 + i], v[i] = x[b * 6 + 3 + i];
  const double t = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const Sinc s = sincs(t);
  skew(w, W);
  matmul3(W, W, S, false, false);
  const float* d = dg + (size_t)b * 16;
  for (int i = 0; i < 3; ++i) {
    dp[i] = d[i * 4 + 3];
    for (int j = 0; j < 3; ++j) dR[i][j] = d[i * 4 + j];
  }
  double da = 0, db = 0, dc = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      M[i][j] = dp[i] * v[j];
      GW[i][j] = s.a * dR[i][j] + s.b * M[i][j];
      GS[i][j] = s.b * dR[i][j] + s.c * M[i][j];
      da += dR[i][j] * W[i][j];
      db += dR[i][j] * S[i][j] + M[i][j] * W[i][j];
      dc += M[i][j] * S[i][j];
    }
  const double dt = s.da * da + s.db * db + s.dc * dc;
  matmul3(GS, W, T1, false, true);   // GS W^T
  matmul3(W, GS, T2, true, false);   // W^T GS
  double G[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) G[i][j] = GW[i][j] + T1[i][j] + T2[i][j];
  const double k = t > 0 ? dt / t : 0.0;
  float* o = dx + (size_t)b * 6;
  o[0] = (float)(G[2][1] - G[1][2] + k * w[0]);
  o[1] = (float)(G[0][2] - G[2][0] + k * w[1]);
  o[2] = (float)(G[1][0] - G[0][1] + k * w[2]);
  for (int j = 0; j < 3; ++j) {  // dv = V^T dp
    double acc = 0;
    for (int i = 0; i < 3; ++i) acc += ((i == j ? 1.0 : 0.0) + s.b * W[i][j] + s.c * S[i][j]) * dp[i];
    o[3 + j] = (float)acc;
  }
}

}  // namespace

PZN_EXPORT int pzn_se3_exp_fwd_f32(const float* twist, int B, float* g, pzn_stream_t stream) {
  PZN_CHECK_ARG(twist && g && B > 0);
  PZN_LAUNCH(se3_exp_fwd_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, pzn_hip_stream(stream), twist, B, g);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_se3_exp_bwd_f32(const float* twist, const float* dg, int B, float* dtwist, pzn_stream_t stream) {
  PZN_CHECK_ARG(twist && dg && dtwist && B > 0);
  PZN_LAUNCH(se3_exp_bwd_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, pzn_hip_stream(stream), twist, dg,
                     B, dtwist);
  PZN_RETURN_LAUNCH_STATUS();
}
