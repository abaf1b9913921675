/*
 * pzn.h — C ABI of libpzn.so, the MI355X (gfx950) hot path of PuzzleNet's
 * pairwise point-cloud forward/backward.
 *
 * Everything below is `extern "C"`, takes raw DEVICE pointers, sizes and a HIP
 * stream, and returns an int status (0 = PZN_OK, negative = error).  No torch
 * types, no exceptions across the boundary, no hidden allocation: the caller
 * owns every buffer (workspace sizes are queried with the *_workspace_bytes
 * helpers).  All kernels are enqueued on `stream` and return immediately.
 *
 * Each entry point cites the reference interface it replaces
 * (paths relative to the reference checkout, Gibbs-liu/PuzzleNet @ 2024_10_08).
 *
 * Conventions
 *   - float tensors are fp32, contiguous, row-major; index tensors are int64.
 *   - `B` batch (clouds), `N` points per cloud, `S` centroids/queries per
 *     cloud, `K` neighbours per query, `D` feature channels.
 *   - distances are ((dx*dx + dy*dy) + dz*dz) in fp32 WITHOUT fma contraction,
 *     which is bit-identical to pointnet_util.square_distance on CPU.
 */
#ifndef PZN_H_
#define PZN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* pzn_stream_t; /* hipStream_t, passed opaquely (0 = null stream) */

enum {
  PZN_OK = 0,
  PZN_EINVAL = -1,       /* bad shape / null pointer / unsupported size */
  PZN_ELAUNCH = -2,      /* hipGetLastError() != hipSuccess after a launch */
  PZN_EUNSUPPORTED = -3, /* valid request this build cannot serve */
  PZN_ENODEVICE = -4     /* no HIP device / device is not gfx950 */
};

/* Library identification and error text. */
int pzn_version(void);
const char* pzn_strerror(int status);
/* 0 when a gfx950 device is usable by this process, PZN_ENODEVICE otherwise. */
int pzn_device_check(void);

/* ------------------------------------------------------------------------ */
/* Point ops: pointnet_util.py                                              */
/* ------------------------------------------------------------------------ */

/* pointnet_util.py:22-36  square_distance(src[B,S,3], dst[B,N,3]) -> [B,S,N]
 * Materialising form kept for API parity; the kNN / ball-query kernels never
 * build this tensor. */
int pzn_square_distance_f32(const float* src, const float* dst, int B, int S,
                            int N, float* out, pzn_stream_t stream);

/* pointnet_util.py:53-73  farthest_point_sample(xyz[B,N,3], npoint)
 * `start_idx[B]` carries the reference's torch.randint draw (line 65) so the
 * kernel itself is deterministic.  out_idx is int64 [B,npoint]. Ties in the
 * arg-max resolve to the LOWEST index (torch.max on CPU). */
int pzn_fps_f32(const float* xyz, int B, int N, int npoint,
                const int64_t* start_idx, int64_t* out_idx,
                pzn_stream_t stream);

/* pointnet_util.py:118-119  dists.argsort()[:, :, :K] fused with the distance:
 * idx[B,S,K] = the K nearest points of xyz[B,N,3] to each new_xyz[B,S,3],
 * ascending by (distance, index) == a stable ascending sort.  K <= N. */
int pzn_knn_f32(const float* xyz, const float* new_xyz, int B, int N, int S,
                int K, int64_t* idx, pzn_stream_t stream);

/* pointnet_util.py:76-96  query_ball_point(radius, nsample, xyz, new_xyz):
 * first `nsample` indices (ascending) with d <= radius^2, slots beyond the hit
 * count are filled with the first hit; a row with no hit is filled with N
 * (what the reference's sort/pad produces). radius2 = (float)(radius*radius). */
int pzn_ball_query_f32(float radius2, int nsample, const float* xyz,
                       const float* new_xyz, int B, int N, int S, int64_t* idx,
                       pzn_stream_t stream);

/* pointnet_util.py:39-50  index_points(points[B,N,C], idx[B,M]) -> [B,M,C]
 * (M = S or S*K; the rank-3 form is the same gather on a flattened idx). */
int pzn_gather_fwd_f32(const float* points, const int64_t* idx, int B, int N,
                       int M, int C, float* out, pzn_stream_t stream);
/* autograd of the gather: grad_points[B,N,C] += scatter(grad_out[B,M,C]).
 * grad_points must be zero-initialised by the caller. */
int pzn_gather_bwd_f32(const float* grad_out, const int64_t* idx, int B, int N,
                       int M, int C, float* grad_points, pzn_stream_t stream);

/* pointnet_util.py:123-132  the grouping tail of sample_and_group:
 * out[B,S,K,3+D] = cat(xyz[idx] - new_xyz[:, :, None], feat[idx]).
 * feat may be NULL with D = 0 (points=None, pointnet_util.py:131-132).
 * grouped_xyz[B,S,K,3] (un-normalised, returnfps=True) is written when
 * non-NULL. */
int pzn_group_fwd_f32(const float* xyz, const float* feat, const float* new_xyz,
                      const int64_t* idx, int B, int N, int S, int K, int D,
                      float* out, float* grouped_xyz, pzn_stream_t stream);
/* Backward of the above.  Any of the three gradient outputs may be NULL.
 * grad_xyz[B,N,3] and grad_feat[B,N,D] must be zero-initialised;
 * grad_new_xyz[B,S,3] is overwritten (= -sum_k grad_out[...,0:3]). */
int pzn_group_bwd_f32(const float* grad_out, const int64_t* idx, int B, int N,
                      int S, int K, int D, float* grad_xyz, float* grad_feat,
                      float* grad_new_xyz, pzn_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Earth Mover's Distance: PyTorchEMD/cuda/emd.cpp:23-27, emd_kernel.cu     */
/* ------------------------------------------------------------------------ */

/* Bytes of scratch the EMD entry points need for (B,n,m). */
size_t pzn_emd_workspace_bytes(int B, int n, int m);

/* emd_kernel.cu:171-193  ApproxMatchForward(xyz1[B,n,3], xyz2[B,m,3])
 *   -> match[B,m,n]   (kernel emd_kernel.cu:25-158).  */
int pzn_emd_approxmatch_f32(const float* xyz1, const float* xyz2, int B, int n,
                            int m, float* match, void* workspace,
                            pzn_stream_t stream);
/* emd_kernel.cu:257-279  MatchCostForward -> cost[B] (kernel :200-243). */
int pzn_emd_matchcost_f32(const float* xyz1, const float* xyz2,
                          const float* match, int B, int n, int m, float* cost,
                          pzn_stream_t stream);
/* emd_kernel.cu:373-398  MatchCostBackward -> grad1[B,n,3], grad2[B,m,3]
 * (kernels :333-355 and :286-327). */
int pzn_emd_matchcost_grad_f32(const float* grad_cost, const float* xyz1,
                               const float* xyz2, const float* match, int B,
                               int n, int m, float* grad1, float* grad2,
                               pzn_stream_t stream);
/* Fused replacement of the three calls above for PyTorchEMD/emd.py:5-21:
 * runs the 10-level auction and accumulates cost[B] and the UNSCALED
 * gradients g1[B,n,3] = d cost/d xyz1, g2[B,m,3] = d cost/d xyz2 on the fly,
 * so match[B,m,n] is never written to HBM.  backward = grad_cost[b] * g{1,2}. */
int pzn_emd_fused_f32(const float* xyz1, const float* xyz2, int B, int n, int m,
                      float* cost, float* g1, float* g2, void* workspace,
                      pzn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PZN_H_ */
