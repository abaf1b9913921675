// emd64.hip — the double instantiation of the approximate Earth Mover's Distance (PyTorchEMD/cuda/emd_kernel.cu
// dispatches on the floating type: AT_DISPATCH_FLOATING_TYPES at :187, :273, :391), gfx950.
//
// model5_b never calls EMD in double; this is the call surface's other half, written for correctness and plain
// throughput, not tuned like emd.hip: every one of the 10 levels x 3 passes of approxmatch (:25-158) is its own launch
// over all (pair, 256-row tile) workgroups, a thread owns one row (k in passes A and C, l in pass B) and walks the other
// cloud through LDS in tiles of 256 points {x, y, z, weight} in the reference's order, so the sums are the sequential
// sums of the reference's loop nest up to the exponential's last bit.  Pass C adds its
// weights to match[B,m,n] with the row index k contiguous across the lanes.
//
//   A (:51-84)   ratioL_k  = remainL_k / (1e-9f + sum_l e_kl remainR_l)
//   B (:86-119)  s_l = remainR_l sum_k e_kl ratioL_k;
//                ratioR_l = min(remainR_l / (s_l + 1e-9f), 1) remainR_l;  remainR_l = max(0, remainR_l - s_l)
//   C (:121-154) w_kl = e_kl ratioL_k ratioR_l;  match_lk += w_kl;  remainL_k = max(0, remainL_k - sum_l w_kl)
// with e_kl = exp(level * |x1_k - x2_l|^2), level = -4^j for j = 7 .. -1 and 0 for the last pass (:46-50).
#include "pzn_common.h"

namespace {

constexpr int T64 = 256;       // threads per workgroup = rows per workgroup = points per LDS tile

struct Emd64Ws {
  double* remainL;   // [B*n]
  double* ratioL;    // [B*n]
  double* remainR;   // [B*m]
  double* ratioR;    // [B*m]
};

Emd64Ws carve64(void* ws, int B, int n, int m) {
  double* p = static_cast<double*>(ws);
  Emd64Ws w;
  w.remainL = p;
  w.ratioL = p + (size_t)B * n;
  w.remainR = p + (size_t)2 * B * n;
  w.ratioR = w.remainR + (size_t)B * m;
  return w;
}

__global__ __launch_bounds__(T64) void emd64_fill_kernel(double* a, size_t na, double va, double* b, size_t nb, double vb) {
  const size_t i = (size_t)blockIdx.x * T64 + threadIdx.x;
  if (i < na) a[i] = va;
  if (i < nb) b[i] = vb;
}

__device__ __forceinline__ double sq3d(double dx, double dy, double dz) { return dx * dx + dy * dy + dz * dz; }

// One pass.  rows: the cloud the threads own ([B, nr, 3]), walked: the other one ([B, nw, 3]) with per-point weights wgt.
//   PASS 0 (A): rows = cloud 1, out[k] = rrem[k] / (1e-9f + sum_l e wgt[l])                      (wgt = remainR, out = ratioL)
//   PASS 1 (B): rows = cloud 2, s = rrem[l] * sum_k e wgt[k]; out[l] = min(rrem / (s + 1e-9f), 1) rrem; rrem[l] = max(0, rrem - s)
//   PASS 2 (C): rows = cloud 1, w = e rowv[k] wgt[l] added to match[l*n + k]; rrem[k] = max(0, rrem[k] - sum_l w)   (rowv = ratioL)
template <int PASS>
__global__ __launch_bounds__(T64) void emd64_pass_kernel(const double* __restrict__ rows, const double* __restrict__ walked,
                                                         int nr, int nw, double level, const double* __restrict__ wgt,
                                                         double* __restrict__ rrem, const double* __restrict__ rowv,
                                                         double* __restrict__ out, double* __restrict__ match) {
  __shared__ double tile[T64][4];
  const int b = blockIdx.y, r = blockIdx.x * T64 + threadIdx.x;
  const bool live = r < nr;
  const double* rp = rows + ((size_t)b * nr + (live ? r : 0)) * 3;
  const double x = rp[0], y = rp[1], z = rp[2];
  const double rv = PASS == 2 && live ? rowv[(size_t)b * nr + r] : 0.0;
  double sum = PASS == 0 ? (double)1e-9f : 0.0;              // (:59: the float literal, also in the double instantiation)
  for (int t0 = 0; t0 < nw; t0 += T64) {
    const int j = t0 + threadIdx.x;
    if (j < nw) {
      const double* wp = walked + ((size_t)b * nw + j) * 3;
      tile[threadIdx.x][0] = wp[0], tile[threadIdx.x][1] = wp[1], tile[threadIdx.x][2] = wp[2];
      tile[threadIdx.x][3] = wgt[(size_t)b * nw + j];
    }
    __syncthreads();
    const int cnt = min(T64, nw - t0);
    if (live) {
      for (int i = 0; i < cnt; ++i) {
        const double e = exp(level * sq3d(tile[i][0] - x, tile[i][1] - y, tile[i][2] - z));
        if (PASS == 2) {
          const double w = e * rv * tile[i][3];
          match[((size_t)b * nw + t0 + i) * nr + r] += w;     // match[b][l][k], k = this thread's row
          sum += w;
        } else {
          sum += e * tile[i][3];
        }
      }
    }
    __syncthreads();
  }
  if (!live) return;
  const size_t o = (size_t)b * nr + r;
  if (PASS == 0) {
    out[o] = rrem[o] / sum;                                                     // :83
  } else if (PASS == 1) {
    const double rem = rrem[o];
    const double s = sum * rem;                                                 // :114
    out[o] = fmin(rem / (s + (double)1e-9f), 1.0) * rem;                       // :115-116
    rrem[o] = fmax(0.0, rem - s);                                               // :117
  } else {
    rrem[o] = fmax(0.0, rrem[o] - sum);                                         // :153
  }
}

// cost[b] = sum_kl |x1_k - x2_l|^2 match[b][l][k] (:200-243): one workgroup per pair, each thread a strided share in a
// fixed order, then a fixed-order tree
__global__ __launch_bounds__(T64) void emd64_cost_kernel(const double* __restrict__ x1, const double* __restrict__ x2,
                                                         const double* __restrict__ match, int n, int m,
                                                         double* __restrict__ cost) {
  __shared__ double red[T64];
  const int b = blockIdx.x;
  double acc = 0.0;
  for (int l = 0; l < m; ++l) {
    const double* q = x2 + ((size_t)b * m + l) * 3;
    const double qx = q[0], qy = q[1], qz = q[2];
    const double* mr = match + ((size_t)b * m + l) * n;
    for (int k = threadIdx.x; k < n; k += T64) {
      const double* p = x1 + ((size_t)b * n + k) * 3;
      acc += sq3d(qx - p[0], qy - p[1], qz - p[2]) * mr[k];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = T64 / 2; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) cost[b] = red[0];
}

// grad1[b][k] = grad_cost[b] * 2 sum_l match[b][l][k] (x1_k - x2_l)  (:333-355): a thread per k, match read with k contiguous
__global__ __launch_bounds__(T64) void emd64_grad1_kernel(const double* __restrict__ gc, const double* __restrict__ x1,
                                                          const double* __restrict__ x2, const double* __restrict__ match,
                                                          int n, int m, double* __restrict__ g1) {
  const int b = blockIdx.y, k = blockIdx.x * T64 + threadIdx.x;
  if (k >= n) return;
  const double* p = x1 + ((size_t)b * n + k) * 3;
  const double x = p[0], y = p[1], z = p[2];
  double dx = 0.0, dy = 0.0, dz = 0.0;
  for (int l = 0; l < m; ++l) {
    const double* q = x2 + ((size_t)b * m + l) * 3;       // (wave-uniform address: one request)
    const double d = match[((size_t)b * m + l) * n + k] * 2.0;
    dx += (x - q[0]) * d, dy += (y - q[1]) * d, dz += (z - q[2]) * d;
  }
  double* o = g1 + ((size_t)b * n + k) * 3;
  o[0] = dx * gc[b], o[1] = dy * gc[b], o[2] = dz * gc[b];
}

// grad2[b][l] = grad_cost[b] * 2 sum_k match[b][l][k] (x2_l - x1_k)  (:286-327): a wavefront per l, lanes stride over k
__global__ __launch_bounds__(T64) void emd64_grad2_kernel(const double* __restrict__ gc, const double* __restrict__ x1,
                                                          const double* __restrict__ x2, const double* __restrict__ match,
                                                          int n, int m, double* __restrict__ g2) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, l = blockIdx.x * (T64 / 64) + (threadIdx.x >> 6);
  if (l >= m) return;
  const double* q = x2 + ((size_t)b * m + l) * 3;
  const double x = q[0], y = q[1], z = q[2];
  const double* mr = match + ((size_t)b * m + l) * n;
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int k = lane; k < n; k += 64) {
    const double* p = x1 + ((size_t)b * n + k) * 3;
    const double d = mr[k] * 2.0;
    sx += (x - p[0]) * d, sy += (y - p[1]) * d, sz += (z - p[2]) * d;
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    sx += __shfl_xor(sx, s, 64);
    sy += __shfl_xor(sy, s, 64);
    sz += __shfl_xor(sz, s, 64);
  }
  if (lane == 0) {
    double* o = g2 + ((size_t)b * m + l) * 3;
    o[0] = sx * gc[b], o[1] = sy * gc[b], o[2] = sz * gc[b];
  }
}

bool aligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

}  // namespace

PZN_EXPORT size_t pzn_emd_workspace_bytes_f64(int B, int n, int m) {
  if (B <= 0 || n <= 0 || m <= 0) return 0;
  return sizeof(double) * 2 * (size_t)B * ((size_t)n + m);
}

PZN_EXPORT int pzn_emd_approxmatch_f64(const double* xyz1, const double* xyz2, int B, int n, int m, double* match,
                                       void* workspace, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz1 && xyz2 && match && workspace && B > 0 && n > 0 && m > 0 && B <= 65535);
  PZN_CHECK_ARG(aligned8(xyz1) && aligned8(xyz2) && aligned8(match) && aligned8(workspace));
  hipStream_t st = pzn_hip_stream(stream);
  const Emd64Ws w = carve64(workspace, B, n, m);
  // :29-35 integer division; :39-44 initial state
  const double multiL = n >= m ? 1.0 : (double)(m / n), multiR = n >= m ? (double)(n / m) : 1.0;
  if (hipMemsetAsync(match, 0, sizeof(double) * (size_t)B * n * m, st) != hipSuccess) return PZN_ELAUNCH;
  const size_t nl = (size_t)B * n, nr = (size_t)B * m, nmax = nl > nr ? nl : nr;
  PZN_LAUNCH(emd64_fill_kernel, dim3((unsigned)((nmax + T64 - 1) / T64)), dim3(T64), 0, st, w.remainL, nl, multiL,
                     w.remainR, nr, multiR);
  const dim3 g1((n + T64 - 1) / T64, B), g2((m + T64 - 1) / T64, B);
  for (int j = 7; j >= -2; --j) {
    const double level = j == -2 ? 0.0 : -(double)powf(4.0f, (float)j);        // :47-50
    PZN_LAUNCH(emd64_pass_kernel<0>, g1, dim3(T64), 0, st, xyz1, xyz2, n, m, level, w.remainR, w.remainL, nullptr,
                       w.ratioL, nullptr);
    PZN_LAUNCH(emd64_pass_kernel<1>, g2, dim3(T64), 0, st, xyz2, xyz1, m, n, level, w.ratioL, w.remainR, nullptr,
                       w.ratioR, nullptr);
    PZN_LAUNCH(emd64_pass_kernel<2>, g1, dim3(T64), 0, st, xyz1, xyz2, n, m, level, w.ratioR, w.remainL, w.ratioL,
                       nullptr, match);
  }
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_emd_matchcost_f64(const double* xyz1, const double* xyz2, const double* match, int B, int n, int m,
                                     double* cost, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz1 && xyz2 && match && cost && B > 0 && n > 0 && m > 0);
  PZN_CHECK_ARG(aligned8(xyz1) && aligned8(xyz2) && aligned8(match) && aligned8(cost));
  PZN_LAUNCH(emd64_cost_kernel, dim3(B), dim3(T64), 0, pzn_hip_stream(stream), xyz1, xyz2, match, n, m, cost);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_emd_matchcost_grad_f64(const double* grad_cost, const double* xyz1, const double* xyz2,
                                          const double* match, int B, int n, int m, double* grad1, double* grad2,
                                          pzn_stream_t stream) {
  PZN_CHECK_ARG(grad_cost && xyz1 && xyz2 && match && grad1 && grad2 && B > 0 && n > 0 && m > 0 && B <= 65535);
  PZN_CHECK_ARG(aligned8(grad_cost) && aligned8(xyz1) && aligned8(xyz2) && aligned8(match) && aligned8(grad1) && aligned8(grad2));
  hipStream_t st = pzn_hip_stream(stream);
  PZN_LAUNCH(emd64_grad1_kernel, dim3((n + T64 - 1) / T64, B), dim3(T64), 0, st, grad_cost, xyz1, xyz2, match, n, m,
                     grad1);
  PZN_LAUNCH(emd64_grad2_kernel, dim3((m + T64 / 64 - 1) / (T64 / 64), B), dim3(T64), 0, st, grad_cost, xyz1, xyz2,
                     match, n, m, grad2);
  PZN_RETURN_LAUNCH_STATUS();
}
