// sachain.hip — one set-abstraction level of the encoder (model5_b.py:449-454 / :456-461 for given centroids) behind one entry
// point each way.
//
// The level is five launches forward (the per-point table P' = feat W1[:,3:]^T, the neighbour search when no indices were
// prefetched, P' += W1[:,0:3] xyz and Q = b1 - W1[:,0:3] centre, the split of W2 into planes, the generated-row max-pool
// level) and four backward (inverse neighbour lists, the pooled layer's weight gradients + hit lists + walk by point, the
// feature gradient, the first layer's feature-weight gradient), through kernels of gemm.hip / knn.hip / sapoint.hip /
// salevel.hip / poolbwd.hip / sapool.hip.  As in attnchain.hip, this file only moves that sequence across the ABI: the library
// enqueues it on one caller-owned buffer per direction (a training step runs four levels: 36 ctypes calls and ~50
// allocations become 8 and 8), the kernels, their order and their operands are those of the caller-composed form.
//
// forward buffer ("saved"):  P' [B*N, C1] | Q [B*S, C1] | W1[:, 3:] dense [C1, D] | idx [B*S*32] int64 (when searched here) |
//   planes of W2.        backward buffer:  off [B*(N+1)] | rows [B*S*32] | pts [B*S*32] (int32) | dP [B*N, C1] | hit lists.
#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

size_t up256(size_t n) { return (n + 255) / 256 * 256; }

struct SaFwdLayout {
  size_t P, Q, wf, wx, idx, planes, total;
};

SaFwdLayout sa_fwd_layout(int B, int N, int S, int D, int C1, int C2) {
  SaFwdLayout f;
  size_t at = 0;
  f.P = at, at += up256((size_t)B * N * C1 * 4);
  f.Q = at, at += up256((size_t)B * S * C1 * 4);
  f.wf = at, at += up256((size_t)C1 * D * 4);
  f.wx = at, at += up256((size_t)3 * C1 * 4);
  f.idx = at, at += up256((size_t)B * S * 32 * 8);
  f.planes = at, at += up256(pzn_sa_level_fwd_workspace_bytes(C1, C2));
  f.total = at;
  return f;
}

struct SaBwdLayout {
  size_t off, rows, pts, dP, hits, total;
};

SaBwdLayout sa_bwd_layout(int B, int N, int S, int C1, int C2) {
  SaBwdLayout b;
  size_t at = 0;
  b.off = at, at += up256((size_t)B * (N + 1) * 4);
  b.rows = at, at += up256((size_t)B * S * 32 * 4);
  b.pts = at, at += up256((size_t)B * S * 32 * 4);
  b.dP = at, at += up256((size_t)B * N * C1 * 4);
  b.hits = at, at += up256(pzn_sa_level_bwd_pt_workspace_bytes(B, S, C2));
  b.total = at;
  return b;
}

// dst[c, 0:D] = W1[c, 3:3+D]: the feature block of the first layer's weight as a dense matrix (the products on it want
// 16-byte aligned rows; W1 + 3 is not); wx[q][c] = W1[c, q], q = 0..2: its coordinate columns as planes
__global__ __launch_bounds__(256) void w1_features_kernel(const float* __restrict__ W1, int C1, int D, float* __restrict__ dst,
                                                          float* __restrict__ wx) {
  const int n = C1 * D;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + 3 * C1; i += gridDim.x * blockDim.x) {
    if (i < n) {
      const int c = i / D, d = i - c * D;
      dst[i] = W1[(size_t)c * (3 + D) + 3 + d];
    } else {
      const int q = (i - n) / C1, c = (i - n) - q * C1;
      wx[q * C1 + c] = W1[(size_t)c * (3 + D) + q];
    }
  }
}

}  // namespace

PZN_EXPORT size_t pzn_sa_level_chain_saved_bytes(int B, int N, int S, int D, int C1, int C2) {
  return (B > 0 && N > 0 && S > 0 && D > 0 && C1 > 0 && C2 > 0) ? sa_fwd_layout(B, N, S, D, C1, C2).total : 0;
}
PZN_EXPORT size_t pzn_sa_level_chain_scratch_bytes(int B, int N, int S, int C1, int C2) {
  return (B > 0 && N > 0 && S > 0 && C1 > 0 && C2 > 0) ? sa_bwd_layout(B, N, S, C1, C2).total : 0;
}

// xyz[B,N,3], feat[B,N,D], new_xyz[B,S,3], idx[B,S,32] int64 or NULL (the 32 nearest neighbours are then searched here and kept
// in `saved`), W1[C1, 3+D], b1[C1], W2[C2, C1], b2[C2] -> out[B*S, C2], argmax[B*S, C2].  saved: pzn_sa_level_chain_saved_bytes()
// bytes, 256-byte aligned, kept by the caller for the backward.  Shapes: those of pzn_sa_level_bwd_pt_f32 (C1 % 128 == 0,
// C2 in {64, 128, 256}); PZN_EUNSUPPORTED otherwise, before anything is launched.
PZN_EXPORT int pzn_sa_level_chain_fwd_f32(const float* xyz, const float* feat, const float* new_xyz, const int64_t* idx,
                                          const float* W1, const float* b1, const float* W2, const float* b2, int B, int N, int S,
                                          int D, int C1, int C2, float* out, int32_t* argmax, void* saved, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && feat && new_xyz && W1 && b1 && W2 && b2 && out && argmax && saved);
  PZN_CHECK_ARG(B > 0 && N > 0 && S > 0 && D > 0 && C1 > 0 && C2 > 0 && (reinterpret_cast<uintptr_t>(saved) & 255) == 0);
  if (C1 % 128 != 0 || !(C2 == 64 || C2 == 128 || C2 == 256)) return PZN_EUNSUPPORTED;
  const SaFwdLayout f = sa_fwd_layout(B, N, S, D, C1, C2);
  unsigned char* base = static_cast<unsigned char*>(saved);
  float* P = reinterpret_cast<float*>(base + f.P);
  float* Q = reinterpret_cast<float*>(base + f.Q);
  float* wf = reinterpret_cast<float*>(base + f.wf);
  hipStream_t st = pzn_hip_stream(stream);
  float* wx = reinterpret_cast<float*>(base + f.wx);
  PZN_LAUNCH(w1_features_kernel, dim3((unsigned)((C1 * (D + 3) + 255) / 256)), dim3(256), 0, st, W1, C1, D, wf, wx);
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  // P' = feat W1[:, 3:]^T + xyz W1[:, 0:3]^T: the coordinate columns in the product's store epilogue where the skinny-layer
  // kernel takes the shape (one pass over the table instead of the product's write + pzn_sa_prep_f32's read-modify-write:
  // 4 x 25 -> 4 x 6 us per step, the same bits), Q by the small pass that is left
  int rc = pzn_ws_gemm_r3(feat, D, wf, D, P, C1, B * N, C1, D, xyz, wx, st);
  const bool fused_xyz = rc == PZN_OK;
  if (rc == PZN_EUNSUPPORTED) rc = pzn_linear_fwd_f32(feat, wf, nullptr, B * N, D, C1, 0, P, stream);
  if (rc != PZN_OK) return rc;
  if (!idx) {
    int64_t* mine = reinterpret_cast<int64_t*>(base + f.idx);
    rc = pzn_knn_f32(xyz, new_xyz, B, N, S, 32, mine, stream);
    if (rc != PZN_OK) return rc;
    idx = mine;
  }
  rc = fused_xyz ? pzn_sa_prep_q(new_xyz, W1, b1, B, S, D, C1, Q, st) : pzn_sa_prep_f32(xyz, new_xyz, W1, b1, B, N, S, D, C1, P, Q, stream);
  if (rc != PZN_OK) return rc;
  void* planes = base + f.planes;
  if (pzn_sa_level_fwd_workspace_bytes(C1, C2) > 0) {
    rc = pzn_sa_level_prep_weights_f32(W2, C1, C2, planes, stream);
    if (rc == PZN_OK) rc = pzn_sa_level_fwd_packed_f32(P, Q, idx, b2, B, N, S, C1, C2, out, argmax, planes, stream);
    if (rc != PZN_EUNSUPPORTED) return rc;
  }
  return pzn_sa_level_fwd_ws_f32(P, Q, idx, W2, b2, B, N, S, C1, C2, out, argmax, planes, stream);
}

// dout[B*S, C2] -> dfeat[B,N,D] (overwritten; may be NULL) and the four parameter gradients: accumulate != 0 (the flat gradient
// bucket): all ADDED to; 0: all overwritten.  idx: what the forward was given (NULL: the indices it searched, in `saved`).
// scratch: pzn_sa_level_chain_scratch_bytes() bytes, 256-byte aligned.
PZN_EXPORT int pzn_sa_level_chain_bwd_f32(const float* dout, const int32_t* argmax, const float* out, const float* xyz,
                                          const float* feat, const float* new_xyz, const int64_t* idx, const float* W1,
                                          const float* W2, const void* saved, int B, int N, int S, int D, int C1, int C2,
                                          float* dfeat, float* dW1, float* db1, float* dW2, float* db2, int accumulate,
                                          void* scratch, pzn_stream_t stream) {
  PZN_CHECK_ARG(dout && argmax && out && xyz && feat && new_xyz && W1 && W2 && saved && dW1 && db1 && dW2 && db2 && scratch);
  PZN_CHECK_ARG(B > 0 && N > 0 && S > 0 && D > 0 && C1 > 0 && C2 > 0);
  PZN_CHECK_ARG((reinterpret_cast<uintptr_t>(saved) & 255) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 255) == 0);
  if (C1 % 128 != 0 || !(C2 == 64 || C2 == 128 || C2 == 256)) return PZN_EUNSUPPORTED;
  const SaFwdLayout f = sa_fwd_layout(B, N, S, D, C1, C2);
  const SaBwdLayout b = sa_bwd_layout(B, N, S, C1, C2);
  const unsigned char* sbase = static_cast<const unsigned char*>(saved);
  unsigned char* w = static_cast<unsigned char*>(scratch);
  const float* P = reinterpret_cast<const float*>(sbase + f.P);
  const float* Q = reinterpret_cast<const float*>(sbase + f.Q);
  const float* wf = reinterpret_cast<const float*>(sbase + f.wf);
  if (!idx) idx = reinterpret_cast<const int64_t*>(sbase + f.idx);
  int32_t* off = reinterpret_cast<int32_t*>(w + b.off);
  int32_t* rows = reinterpret_cast<int32_t*>(w + b.rows);
  int32_t* pts = reinterpret_cast<int32_t*>(w + b.pts);
  float* dP = reinterpret_cast<float*>(w + b.dP);
  hipStream_t st = pzn_hip_stream(stream);
  if (!accumulate) {      // the walk by point and the slice product ADD into dW1 / db1
    if (pzn_zero_async(dW1, (size_t)C1 * (3 + D), st) != PZN_OK) return PZN_ELAUNCH;
    if (pzn_zero_async(db1, (size_t)C1, st) != PZN_OK) return PZN_ELAUNCH;
  }
  int rc = pzn_knn_inverse_lists(idx, B, N, S, 32, off, rows, pts, stream);
  if (rc != PZN_OK) return rc;
  rc = pzn_sa_level_bwd_pt_f32(dout, argmax, out, W2, P, Q, idx, xyz, new_xyz, off, rows, pts, B, N, S, D, C1, C2, dP, dW2, db2,
                               dW1, db1, accumulate, w + b.hits, stream);
  if (rc != PZN_OK) return rc;
  if (dfeat) {
    rc = pzn_linear_dgrad_f32(dP, nullptr, wf, B * N, D, C1, nullptr, dfeat, stream);
    if (rc != PZN_OK) return rc;
  }
  // dW1[:, 3:] += dP^T feat, straight into the parameter's column slice
  return pzn_linear_slice_wgrad_f32(dP, feat, B * N, D, C1, dW1 + 3, 3 + D, nullptr, stream);
}
