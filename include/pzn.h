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

/* Measurement only (bench.py; no counterpart in the reference): per-KERNEL device time.  pzn_ktimer_enable(1): every kernel
 * the library launches from now on is bracketed by a HIP event pair on its own stream, under the name a rocprofv3 kernel
 * trace shows (template arguments included).  pzn_ktimer_collect() waits for the recorded launches and folds them into
 * rows -> number of rows; pzn_ktimer_row(i, ...) reads one (name, launches, summed milliseconds).  Off by default. */
int pzn_ktimer_enable(int on);
int pzn_ktimer_collect(void);
int pzn_ktimer_row(int i, char* name, int cap, int* launches, double* ms);

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
/* dataset.py:1147-1163 (`fps` of a raw piece in the loader processes) as a background job beside a training step: the same
 * picks bit for bit, but no LDS image of the cloud (1.2 KB of LDS per workgroup instead of up to 150 KB), so the step's
 * LDS-tiled kernels keep their CUs.  N <= 32768.  counts (NULL or int64 [B]): rows >= counts[b] of cloud b are padding
 * (copies of row 0, which never win) and are left out of the rounds.  max_count (0: unknown): the caller's promise that
 * counts[b] <= max_count for every cloud - a cut piece holds at most M - n of the raw cloud's M points - which lets the
 * launch hold only that many rows in registers (fewer, faster wavefronts); rows beyond it are never read. */
int pzn_fps_background_f32(const float* xyz, int B, int N, int npoint,
                           const int64_t* start_idx, int64_t* out_idx,
                           const int64_t* counts, int max_count, pzn_stream_t stream);

/* pointnet_util.py:118-119  dists.argsort()[:, :, :K] fused with the distance:
 * idx[B,S,K] = the K nearest points of xyz[B,N,3] to each new_xyz[B,S,3],
 * ascending by (distance, index) == a stable ascending sort.  K <= N. */
int pzn_knn_f32(const float* xyz, const float* new_xyz, int B, int N, int S,
                int K, int64_t* idx, pzn_stream_t stream);

/* pointnet_util.py:117-132 (knn=True, nsample = 32) in ONE launch and in the reference's layout: the
 * neighbour search of pzn_knn_f32 followed, per query and in the same wavefront, by the grouping of
 * pzn_group_fwd_f32: idx[B,S,32] and out[B,S,32,3+D] = cat(xyz[idx] - new_xyz, feat[idx]); grouped_xyz
 * [B,S,32,3] is written when non-NULL (returnfps=True).  64 <= N <= 8192, D > 0, D % 4 == 0, feat and out
 * 16-byte aligned; PZN_EUNSUPPORTED otherwise (compose the two single entry points then). */
int pzn_knn_group_f32(const float* xyz, const float* feat, const float* new_xyz, int B, int N,
                      int S, int D, int64_t* idx, float* out, float* grouped_xyz,
                      pzn_stream_t stream);

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
/* 1..4 SMALL calls of the function above (n, m <= 256 each: model5_b.py:1012 and :1123-1125, the three small terms of the loss) as
 * ONE launch of single-workgroup auctions; HOST arrays of device pointers / sizes, one entry per call; results bit-identical to
 * the separate calls.  PZN_EUNSUPPORTED when one of them is larger. */
int pzn_emd_fused_small_multi_f32(int count, const float* const* xyz1, const float* const* xyz2,
                                  const int* B, const int* n, const int* m, float* const* cost,
                                  float* const* g1, float* const* g2, pzn_stream_t stream);

/* The double instantiation of the three calls (emd_kernel.cu:187, :273, :391 dispatch on the floating type;
 * csrc/emd64.hip): same layouts in double, workspace of pzn_emd_workspace_bytes_f64(B,n,m) bytes.  model5_b never
 * calls EMD in double; PyTorchEMD.emd.earth_mover_distance routes double clouds here. */
size_t pzn_emd_workspace_bytes_f64(int B, int n, int m);
int pzn_emd_approxmatch_f64(const double* xyz1, const double* xyz2, int B, int n, int m, double* match,
                            void* workspace, pzn_stream_t stream);
int pzn_emd_matchcost_f64(const double* xyz1, const double* xyz2, const double* match, int B, int n, int m,
                          double* cost, pzn_stream_t stream);
int pzn_emd_matchcost_grad_f64(const double* grad_cost, const double* xyz1, const double* xyz2, const double* match,
                               int B, int n, int m, double* grad1, double* grad2, pzn_stream_t stream);

/* Measurement aid (bench.py's EMD roofline): byte offset inside the workspace of pzn_emd_walk_counter_count() uint64
 * counters the fused entry point leaves behind; their sum x 64 = (row, point) evaluations executed (points of exhausted mass
 * and points outside a level's x window are not walked); counters (16 i + s) * 16, s = 0..15, belong to launch i of the call
 * (0 = first pass A, 1 + 2 k = pass B and 2 + 2 k = pass C of the k-th level); (size_t)-1 on the single-workgroup path
 * (n, m <= 256: 30 n m). */
size_t pzn_emd_walk_counter_offset(int B, int n, int m);
int pzn_emd_walk_counter_count(void);

/* ------------------------------------------------------------------------ */
/* Loss tail: model5_b.py:1495-1505 chamfer_loss                            */
/* ------------------------------------------------------------------------ */

size_t pzn_chamfer_workspace_bytes(int B, int n, int m);
/* chamfer_loss(a[B,n,3], b[B,m,3]) -> (torch.min(P,1): min over a for each b
 * point [B,m],  torch.min(P,2): min over b for each a point [B,n]) with the
 * reference's expansion P = |a|^2 + |b|^2 - 2 a.b, never materialising
 * P[B,n,m].  arg-min indices (int32) are written for the backward. */
int pzn_chamfer_fwd_f32(const float* a, const float* b, int B, int n, int m,
                        float* min_over_a, int32_t* arg_over_a,
                        float* min_over_b, int32_t* arg_over_b, void* workspace,
                        pzn_stream_t stream);
/* grad_a[B,n,3], grad_b[B,m,3] (zero-initialised by the caller) from the
 * upstream gradients g_over_a[B,m], g_over_b[B,n] (either may be NULL). */
int pzn_chamfer_bwd_f32(const float* a, const float* b, int B, int n, int m,
                        const float* g_over_a, const int32_t* arg_over_a,
                        const float* g_over_b, const int32_t* arg_over_b,
                        float* grad_a, float* grad_b, pzn_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Dense path on the matrix cores (fp32 results; default operand path = bf16x3 */
/* split precision, exact-fp32 MFMA on request: pzn_gemm_set_precision)       */
/* ------------------------------------------------------------------------ */

/* nn.Linear as used throughout model5_b.py (e.g. :417-422, :559-599):
 * y[M,Nout] = act(x[M,Kin] W[Nout,Kin]^T + bias[Nout]); act = ReLU if relu.
 * bias may be NULL. */
int pzn_linear_fwd_f32(const float* x, const float* W, const float* bias, int M,
                       int Kin, int Nout, int relu, float* y,
                       pzn_stream_t stream);
/* Same product with the epilogue of model5_b.py:453-454 / :460-461: ReLU then
 * max over each group of 32 consecutive rows (the K=32 neighbours of one
 * centroid): out[R,Nout], argmax[R,Nout] (row 0..31 of the maximum). */
int pzn_linear_maxpool_fwd_f32(const float* x, const float* W, const float* bias,
                               int R, int Kin, int Nout, float* out,
                               int32_t* argmax, pzn_stream_t stream);
/* dx[M,Kin] = (dy * [y_relu > 0]) W, then * [x_relu > 0].  y_relu / x_relu are
 * the forward outputs of this / the previous layer when they ended in a ReLU,
 * NULL otherwise. */
int pzn_linear_dgrad_f32(const float* dy, const float* y_relu, const float* W,
                         int M, int Kin, int Nout, const float* x_relu,
                         float* dx, pzn_stream_t stream);
/* dW[Nout,Kin] = (dy * [y_relu > 0])^T x,  db[Nout] = its column sums (db may
 * be NULL).  Split over the row range, accumulated with fp32 atomics: results are
 * reproducible to rounding, not bitwise.  accumulate == 0: both outputs are overwritten;
 * accumulate != 0: added to (e.g. directly into a zeroed flat gradient bucket, which
 * saves the zero-fill launches and autograd's separate "grad += dW" pass). */
int pzn_linear_wgrad_f32(const float* dy, const float* y_relu, const float* x,
                         int M, int Kin, int Nout, float* dW, float* db,
                         int accumulate, pzn_stream_t stream);
/* The three products of nn.Linear on a COLUMN SLICE of a wider weight matrix: W points at column k0 of
 * W_full[Nout, ldw] and is Kin columns wide.  cat(x_1 .. x_n) W_full^T (model5_b.py:466-474) then is sum_i x_i W_i^T + b
 * without the concatenation ever being built, and a weight gradient can be added straight into a slice of a parameter.
 *   fwd:   y[M,Nout] = x[M,Kin] W^T + bias   (accumulate == 0)   or   y += x W^T   (accumulate != 0, bias ignored)
 *   dgrad: dx[M,Kin] = dy[M,Nout] W (+ addend[M,Kin] when non-NULL)
 *   wgrad: dW[Nout, ldw-strided, Kin columns] += dy^T x,  db[Nout] += column sums of dy (db may be NULL) */
int pzn_linear_slice_fwd_f32(const float* x, const float* W, int ldw, const float* bias, int M, int Kin,
                             int Nout, int accumulate, float* y, pzn_stream_t stream);
int pzn_linear_slice_dgrad_f32(const float* dy, const float* W, int ldw, int M, int Kin, int Nout,
                               const float* addend, float* dx, pzn_stream_t stream);
int pzn_linear_slice_wgrad_f32(const float* dy, const float* x, int M, int Kin, int Nout, float* dW, int ldw,
                               float* db, pzn_stream_t stream);
/* Backward through the max-pool epilogue: the [R*32,Nout] gradient
 * dy[g*32+k, c] = (argmax[g,c]==k && out[g,c]>0) ? dout[g,c] : 0 is generated
 * inside the operand loader, never materialised. */
int pzn_linear_maxpool_dgrad_f32(const float* dout, const int32_t* argmax,
                                 const float* out, const float* W, int R, int Kin,
                                 int Nout, const float* x_relu, float* dx,
                                 pzn_stream_t stream);
int pzn_linear_maxpool_wgrad_f32(const float* dout, const int32_t* argmax,
                                 const float* out, const float* x, int R, int Kin,
                                 int Nout, float* dW, float* db, int accumulate,
                                 pzn_stream_t stream);
/* Matrix-core path of every dense entry point below: 0 = exact fp32 (v_mfma_f32_32x32x2_f32),
 * 1 = bf16x3 split precision (x = x1+x2+x3 in bf16, six v_mfma_f32_32x32x16_bf16 products,
 * fp32 accumulate: fp32-GEMM accuracy at up to 2.67x the matrix-pipe rate), 2 = auto (default;
 * currently bf16x3 for every product).
 * Process-wide; also PZN_GEMM_PRECISION=f32|x3|auto in the environment. */
int pzn_gemm_set_precision(int mode);
int pzn_gemm_get_precision(void);
/* Precision of the attention contractions (model5_b.py:67-75: q k^T, attn v, and their four backward products) inside
 * pzn_attn_* / pzn_attn_block_*: 0 (default) = the matrix-core path selected above (fp32 results), 1 = operands rounded
 * to bf16 once and ONE v_mfma_f32_32x32x16_bf16 per product, fp32 accumulation, fp32 softmax (BASELINE configs[4] "bf16
 * attn with MFMA").  The q / k / v / out projections keep the fp32-accurate path.  Also PZN_ATTN_PRECISION=bf16. */
int pzn_attn_set_precision(int mode);
int pzn_attn_get_precision(void);

/* Shared MLP + max over the K=32 neighbours, model5_b.py:452-454 / :459-461:
 *   h = relu(x[R*32,C0] W1^T + b1)  (kept: the backward reads it);
 *   out[R,C2] = max_k relu(h W2^T + b2),  argmax[R,C2]. */
int pzn_sharedmlp_max_fwd_f32(const float* x, const float* W1, const float* b1,
                              const float* W2, const float* b2, int R, int C0,
                              int C1, int C2, float* h, float* out,
                              int32_t* argmax, pzn_stream_t stream);
/* dh_ws: [R*32,C1] scratch.  dx[R*32,C0] may be NULL.  dW1,db1,dW2,db2 are
 * overwritten. */
int pzn_sharedmlp_max_bwd_f32(const float* x, const float* W1, const float* W2,
                              const float* h, const float* out,
                              const int32_t* argmax, const float* dout, int R,
                              int C0, int C1, int C2, float* dh_ws, float* dx,
                              float* dW1, float* db1, float* dW2, float* db2,
                              int accumulate, pzn_stream_t stream);

/* The same set-abstraction level with its first layer computed PER POINT (csrc/sapoint.hip): a grouped row is
 * {xyz[j] - centre, feat[j]} (pointnet_util.py:123-132), so its product with W1[C1,3+D] (model5_b.py:452 / :459)
 * is a per-point term plus a per-group term (below); the grouped tensor is never materialised.
 *   pzn_knn_inverse_lists: the B*S*K (row, point) pairs of idx sorted by point, per cloud: rows[B, S*K] = in-cloud
 *         row numbers (s*K + k), ascending within a point, pts[B, S*K] = the point each gathered (may be NULL),
 *         off[B, N+1] = where every point's rows start (index_points backward without atomics on rows). */
int pzn_knn_inverse_lists(const int64_t* idx, int B, int N, int S, int K, int32_t* off,
                          int32_t* rows, int32_t* pts, pzn_stream_t stream);

/* The same level with the first layer's rows NEVER in memory (what model5_b's encoder runs by default).  The coordinate
 * term is split:  W1[:,0:3] (xyz[j] - centre) = W1[:,0:3] xyz[j] - W1[:,0:3] centre,  so with
 *   Pp[B*N, C1] = feat W1[:,3:]^T + W1[:,0:3] xyz      (pzn_linear_fwd_f32, then pzn_sa_prep_f32 adds the xyz term in place)
 *   Q [B*S, C1] = b1 - W1[:,0:3] new_xyz                (pzn_sa_prep_f32)
 * a grouped row of the first layer is relu(Pp[idx] + Q[group]) — generated inside the operand loader of the
 * weight-stationary matrix-core kernel (fwd) and inside the backward passes; same result as the grouped-row form above up to
 * the order of the fp32 sum.  K = 32, C1 % 128 == 0 backward (C1 % 32 == 0 forward), C2 in {64,128,256}.
 *   fwd:  out[B*S, C2] = max_k relu(relu(Pp[idx[.,k]] + Q) W2^T + b2), argmax[B*S, C2]
 *   bwd:  pzn_sa_level_bwd_pt_f32 below (the rows' gradient dh[B*S*32, C1] is summed per point where it is computed and
 *         never written; rounds 2-4 wrote it: pzn_sa_level_bwd_f32 + pzn_sa_point_l1_bwd_f32, removed in round 6). */
int pzn_sa_prep_f32(const float* xyz, const float* new_xyz, const float* W1, const float* b1, int B, int N, int S,
                    int D, int C1, float* P, float* Q, pzn_stream_t stream);
int pzn_sa_level_fwd_f32(const float* Pp, const float* Q, const int64_t* idx, const float* W2, const float* b2,
                         int B, int N, int S, int C1, int C2, float* out, int32_t* argmax, pzn_stream_t stream);
/* same, with a caller-owned 16-byte aligned workspace of pzn_sa_level_fwd_workspace_bytes(C1, C2) bytes (0 = no streamed
 * form for this shape; workspace may then be NULL): C1 = C2 in {128, 256} run the streamed-weights kernel, which
 * generates each grouped row once and streams the split planes of W2 through LDS; identical results (same products,
 * same tie rule of the arg-max). */
size_t pzn_sa_level_fwd_workspace_bytes(int C1, int C2);
int pzn_sa_level_fwd_ws_f32(const float* Pp, const float* Q, const int64_t* idx, const float* W2, const float* b2,
                            int B, int N, int S, int C1, int C2, float* out, int32_t* argmax, void* workspace,
                            pzn_stream_t stream);
/* The same in two launches that a caller times (or schedules) separately, as pzn_attn_fused_prep_weights is for the
 * attention blocks: the split of W2 into the streamed kernel's plane image (workspace of
 * pzn_sa_level_fwd_workspace_bytes() bytes), and the level on a workspace that holds it.  PZN_EUNSUPPORTED where the
 * shape has no streamed form (pzn_sa_level_fwd_ws_f32 serves every shape). */
int pzn_sa_level_prep_weights_f32(const float* W2, int C1, int C2, void* workspace, pzn_stream_t stream);
int pzn_sa_level_fwd_packed_f32(const float* Pp, const float* Q, const int64_t* idx, const float* b2, int B, int N,
                                int S, int C1, int C2, float* out, int32_t* argmax, const void* workspace,
                                pzn_stream_t stream);

/* The level's backward BY POINT (csrc/sapool.hip): dW2[C2,C1] / db2[C2] (overwritten, or added to when accumulate), the
 * per-point sums dP[B*N, C1] of the rows' gradient dh (overwritten), dW1[:, 0:3] += dh^T (xyz[idx] - centre) and db1 += column
 * sums of dh (both ADDED to, db1 may be NULL).  dh itself is computed by point (hit lists per group, sorted by arg-max row;
 * W2 slice in LDS; gate from Pp and Q) and never written.  off / rows / pts: pzn_knn_inverse_lists of idx.  A wavefront owns
 * whole points: every row of dP is written exactly once by plain stores and the rows of a point are summed in ascending row
 * order, so dP is the same bit for bit in every run.  workspace: pzn_sa_level_bwd_pt_workspace_bytes(B, S, C2) bytes,
 * 16-byte aligned.  PZN_EUNSUPPORTED for C1 % 128 != 0 or C2 not in {64, 128, 256}. */
size_t pzn_sa_level_bwd_pt_workspace_bytes(int B, int S, int C2);
int pzn_sa_level_bwd_pt_f32(const float* dout, const int32_t* argmax, const float* out, const float* W2,
                            const float* Pp, const float* Q, const int64_t* idx, const float* xyz, const float* new_xyz,
                            const int32_t* off, const int32_t* rows, const int32_t* pts, int B, int N, int S, int D, int C1,
                            int C2, float* dP, float* dW2, float* db2, float* dW1, float* db1, int accumulate, void* workspace,
                            pzn_stream_t stream);

/* One set-abstraction level of the encoder behind one entry point each way (csrc/sachain.hip): the sequences
 *   fwd:  P' = feat W1[:,3:]^T (pzn_linear_fwd_f32) -> neighbour search when idx == NULL (pzn_knn_f32) -> pzn_sa_prep_f32 ->
 *         pzn_sa_level_prep_weights_f32 -> pzn_sa_level_fwd_packed_f32
 *   bwd:  pzn_knn_inverse_lists -> pzn_sa_level_bwd_pt_f32 -> dfeat = dP W1[:,3:] (pzn_linear_dgrad_f32) ->
 *         dW1[:,3:] += dP^T feat (pzn_linear_slice_wgrad_f32)
 * enqueued by the library on one caller-owned buffer per direction (same kernels, order and operands as the caller-composed
 * form: 2 calls and 2 allocations per level instead of 9 and ~13).  xyz[B,N,3], feat[B,N,D], new_xyz[B,S,3], idx[B,S,32] or NULL,
 * W1[C1,3+D], b1[C1], W2[C2,C1], b2[C2]; out[B*S,C2], argmax[B*S,C2]; saved / scratch: pzn_sa_level_chain_saved_bytes /
 * _scratch_bytes, 256-byte aligned; the backward gets the idx the forward got (NULL: the searched indices live in `saved`).
 * accumulate != 0: the four parameter gradients are ADDED to; 0: overwritten.  dfeat[B,N,D] overwritten, may be NULL.
 * C1 % 128 == 0, C2 in {64,128,256}: PZN_EUNSUPPORTED otherwise, before anything is launched. */
size_t pzn_sa_level_chain_saved_bytes(int B, int N, int S, int D, int C1, int C2);
size_t pzn_sa_level_chain_scratch_bytes(int B, int N, int S, int C1, int C2);
int pzn_sa_level_chain_fwd_f32(const float* xyz, const float* feat, const float* new_xyz, const int64_t* idx,
                               const float* W1, const float* b1, const float* W2, const float* b2, int B, int N, int S,
                               int D, int C1, int C2, float* out, int32_t* argmax, void* saved, pzn_stream_t stream);
int pzn_sa_level_chain_bwd_f32(const float* dout, const int32_t* argmax, const float* out, const float* xyz,
                               const float* feat, const float* new_xyz, const int64_t* idx, const float* W1,
                               const float* W2, const void* saved, int B, int N, int S, int D, int C1, int C2,
                               float* dfeat, float* dW1, float* db1, float* dW2, float* db2, int accumulate,
                               void* scratch, pzn_stream_t stream);

/* The per-point stem of the encoder in one launch each way (csrc/stem.hip, model5_b.py:447-448):
 *   out = relu(bn2(mlp2(relu(bn1(mlp1(xyz))))))   xyz[B,N,3], mlp1 = (W1[64,3], b1[64]), mlp2 = (W2[64,64], b2[64]),
 * bn1 / bn2 = BatchNorm1d(N) over the POINT axis of [B,N,64] (weight / bias / running buffers [N]; any may be NULL as the module
 * has them; training != 0: batch statistics, running buffers updated in place as torch does).  out[B,N,64]; mean1 / invstd1 /
 * mean2 / invstd2 [N] (written) feed the backward, which recomputes every activation from xyz.  bwd: dW1[64,3], db1[64],
 * dW2[64,64], db2[64] and the BatchNorm weight / bias gradients [N] (these may be NULL) are ADDED to; no gradient for xyz;
 * workspace = pzn_stem_bwd_workspace_bytes(N) bytes, 16-byte aligned, need not be cleared (the workgroups' partial sums);
 * dout2 (may be NULL): a second gradient of the same output, added to dout while loading (two consumers of the features).
 * PZN_EUNSUPPORTED for B > 64 or W2 not 16-byte aligned (compose pzn_linear_* + pzn_bn_points_relu_* then). */
int pzn_stem_fwd_f32(const float* xyz, const float* W1, const float* b1, const float* bn1_weight, const float* bn1_bias,
                     float* bn1_running_mean, float* bn1_running_var, float bn1_momentum, float bn1_eps, const float* W2,
                     const float* b2, const float* bn2_weight, const float* bn2_bias, float* bn2_running_mean,
                     float* bn2_running_var, float bn2_momentum, float bn2_eps, int training, int B, int N, float* out,
                     float* mean1, float* invstd1, float* mean2, float* invstd2, pzn_stream_t stream);
size_t pzn_stem_bwd_workspace_bytes(int N);
int pzn_stem_bwd_f32(const float* xyz, const float* dout, const float* dout2, const float* W1, const float* b1, const float* W2, const float* b2,
                     const float* bn1_weight, const float* bn1_bias, const float* bn2_weight, const float* bn2_bias,
                     const float* mean1, const float* invstd1, const float* mean2, const float* invstd2, int training, int B,
                     int N, float* dW1, float* db1, float* dW2, float* db2, float* dbn1_weight, float* dbn1_bias,
                     float* dbn2_weight, float* dbn2_bias, void* workspace, pzn_stream_t stream);

/* relu(BatchNorm1d(num_points)(x)) of the per-point feature MLP (model5_b.py:424, :447-448): x[B,N,C], the BN
 * "channel" axis is the POINT index, statistics over the B*C values of a point.  training != 0: batch statistics
 * (biased variance), running_mean / running_var updated in place as torch does (momentum, unbiased variance; may be
 * NULL); else the running statistics normalise.  save_mean / save_invstd [N] feed the backward.
 * bwd: dx[B,N,C] (may be NULL), dweight[N] / dbias[N] ADDED to (may be NULL); the ReLU gate is recomputed from x. */
int pzn_bn_points_relu_fwd_f32(const float* x, const float* weight, const float* bias,
                               float* running_mean, float* running_var, int training,
                               float momentum, float eps, int B, int N, int C, float* y,
                               float* save_mean, float* save_invstd, pzn_stream_t stream);
int pzn_bn_points_relu_bwd_f32(const float* x, const float* dy, const float* weight,
                               const float* bias, const float* save_mean,
                               const float* save_invstd, int training, int B, int N, int C,
                               float* dx, float* dweight, float* dbias, pzn_stream_t stream);

/* scaled_dot_production of layerAttention, model5_b.py:67-75:
 * attn[B,L,L] = softmax(q[B,L,dk] k[B,L,dk]^T / sqrt(dk)), out[B,L,dv] = attn v.
 * attn is an output because the reference returns it (model5_b.py:97,101). */
int pzn_attn_fwd_f32(const float* q, const float* k, const float* v, int B,
                     int L, int dk, int dv, float* attn, float* out,
                     pzn_stream_t stream);
size_t pzn_attn_bwd_workspace_bytes(int B, int L, int dk, int dv);
/* dq, dk_out, dv_out from d_out[B,L,dv] and d_attn[B,L,L] (may be NULL). */
int pzn_attn_bwd_f32(const float* q, const float* k, const float* v,
                     const float* attn, const float* d_out, const float* d_attn,
                     int B, int L, int dk, int dv, float* dq, float* dk_out,
                     float* dv_out, void* workspace, pzn_stream_t stream);

/* layerAttention as one unit (model5_b.py:83-101): q,k,v = Linear(x[B*L,E]); attn = softmax(q k^T /
 * sqrt(dk)); r = x - attn v; yo = relu(r Wo^T + bo); out = x + yo.  q[B*L,dk], k, v[B*L,E], attn
 * [B,L,L], r, yo are outputs the backward needs (attn is also the block's second result).  The
 * backward takes dout (and d_attn or NULL), returns dx and the eight parameter gradients
 * (overwritten, or added to when accumulate != 0).  Returns PZN_EUNSUPPORTED for shapes the
 * weight-stationary kernel does not take: compose the block from the single entry points then. */
int pzn_attn_block_fwd_f32(const float* x, const float* Wq, const float* bq, const float* Wk,
                           const float* bk, const float* Wv, const float* bv, const float* Wo,
                           const float* bo, int B, int L, int E, int dk, float* q, float* k,
                           float* v, float* attn, float* r, float* yo, float* out,
                           pzn_stream_t stream);
size_t pzn_attn_block_bwd_workspace_bytes(int B, int L, int E, int dk);
int pzn_attn_block_bwd_f32(const float* x, const float* Wq, const float* Wk, const float* Wv,
                           const float* Wo, const float* q, const float* k, const float* v,
                           const float* attn, const float* r, const float* yo, const float* dout,
                           const float* d_attn, int B, int L, int E, int dk, void* workspace,
                           float* dx, float* dWq, float* dbq, float* dWk, float* dbk, float* dWv,
                           float* dbv, float* dWo, float* dbo, int accumulate,
                           pzn_stream_t stream);

/* layerAttention (model5_b.py:67-75, 83-101) as CHAINED matrix-core kernels (csrc/attnfused.hip) for the model's shape
 * L = 256 points, E = 256 channels, dk = 64 (pzn_attn_fused_supported).  The point sits on the MFMA lane, so every
 * product's accumulator is the next product's operand in registers: no score matrix and no q / k / v tensors in
 * memory.  Shared operands live as bf16x3 "plane images" in MFMA-fragment order, written by their producer:
 *   weights:  pzn_attn_fused_prep_weights -> one buffer of pzn_attn_fused_weight_bytes() per block;
 *   q, k, v:  pzn_attn_fused_proj -> three images per problem (q, k: *_qk_image_bytes(B); v: *_v_image_bytes(B)); one
 *             image serves the products that sum over its columns (plain slabs) and those that sum over its rows
 *             (the same bytes fetched in a transposed-read cut by the LDS-DMA: no second copy since round 4).
 * Every entry point takes nprob = 1 or 2 independent problems as arrays of nprob pointers (the two encoders of
 * predict5, model5_b.py:700-707, in one launch); B clouds each; all buffers 16-byte aligned.
 *   fwd:    r = x + relu(Wo (x - softmax(q k^T / 8) v) + bo) [B*L,E];  t = x - attn v [B*L,E];  mask [B*L,8] u32 = gate
 *           bits of the ReLU;  lse [B*L];  map[i] (may be NULL): map_mode 0: [B,L,L] = scale P; 1: += scale P (the mean of the
 *           four blocks' maps, model5_b.py:468-469); 2 / 3: the same for [B,L/16,L] = the column sums of P over each strip
 *           of 16 query rows - all that a consumer of the map's mean over its rows needs (model5_b.py:937-942: 1/16 of the
 *           bytes, no [B,L,L] tensor).
 *   bwd_q:  query side of the backward from dr (+ dr2 when non-NULL: rows of ld_dr / ld_dr2 floats, so a column slice of
 *           a wider gradient needs no copy and the sum of two gradients no add): dz = dr . gate, dq [B*L,dk], delta [B*L],
 *           the image of da = -dz Wo, and for bwd_k only: u = dr + dz Wo [B*L,E] and dq_tiles [B*L,dk] in the kernels'
 *           own register-tile order (same sizes as the row tensors; not meant to be read by anything else).
 *   bwd_k:  key side (u, dq = the two tile tensors of bwd_q): dk [B*L,dk], dv [B*L,E],
 *           dx = u + dq Wq + dk Wk + dv Wv [B*L,E] = the block's input gradient.
 *   wgrads: the eight parameter gradients from dz, t, dq, dk, dv and the block input x (one problem per call). */
int pzn_attn_fused_supported(int L, int E, int dk);
size_t pzn_attn_fused_weight_bytes(void);
size_t pzn_attn_fused_qk_image_bytes(int B);
size_t pzn_attn_fused_v_image_bytes(int B);
int pzn_attn_fused_prep_weights(const float* Wq, const float* Wk, const float* Wv, const float* Wo,
                                void* planes, pzn_stream_t stream);
/* the same for n <= 4 blocks in one launch (an encoder's four layers): arrays of n pointers */
int pzn_attn_fused_prep_weights_n(int n, const float* const* Wq, const float* const* Wk,
                                  const float* const* Wv, const float* const* Wo,
                                  void* const* planes, pzn_stream_t stream);
int pzn_attn_fused_proj(int nprob, const float* const* x, const void* const* w,
                        const float* const* bq, const float* const* bk, const float* const* bv, int B,
                        void* const* qrp, void* const* krp, void* const* vrp, pzn_stream_t stream);
int pzn_attn_fused_fwd(int nprob, const float* const* x, const void* const* qrp,
                       const void* const* krp, const void* const* vrp, const void* const* w,
                       const float* const* bo, int B, float* const* r, float* const* t,
                       void* const* mask, float* const* map, float* const* lse, int map_mode,
                       float map_scale, pzn_stream_t stream);
int pzn_attn_fused_bwd_q(int nprob, const float* const* dr, int ld_dr, const float* const* dr2,
                         int ld_dr2, const void* const* mask,
                         const void* const* qrp, const void* const* krp, const void* const* vrp,
                         const void* const* w, int B, float* const* dz, float* const* u,
                         float* const* dq, float* const* dq_tiles, void* const* darp,
                         float* const* delta, pzn_stream_t stream);
int pzn_attn_fused_bwd_k(int nprob, const void* const* qrp, const void* const* krp,
                         const void* const* vrp, const void* const* darp, const void* const* w,
                         const float* const* lse, const float* const* delta, const float* const* u,
                         const float* const* dq, int B, float* const* dk, float* const* dv,
                         float* const* dx, pzn_stream_t stream);
/* The four weight gradients and four bias gradients of one block from what the chained kernels leave in memory: dWo (+)= dz^T t,
 * dWq/k/v (+)= dq/dk/dv^T x, the biases' = column sums (accumulate == 0: overwritten).  E = 256, dk = 64, M % 64 == 0: one launch
 * of LDS-shared 128 x 128 tiles over 24 row ranges (csrc/attnwgrad.hip), which meet in fp32 atomics here and in a fixed-order sum
 * inside pzn_attn_chain_bwd_f32 (its scratch holds their partial tiles); other shapes: the general weight-gradient kernels. */
int pzn_attn_fused_wgrads(const float* dz, const float* t, const float* dq, const float* dk,
                          const float* dv, const float* x, int M, int E, int dk_dim, float* dWq,
                          float* dbq, float* dWk, float* dbk, float* dWv, float* dbv, float* dWo,
                          float* dbo, int accumulate, pzn_stream_t stream);

/* model5_b.py:462-475 of ONE encoder behind one entry point each way (csrc/attnchain.hip): the four layerAttention blocks, the
 * mean of their maps, the out projection over the five un-concatenated slices and the max over the points, enqueued by the
 * library on one caller-owned buffer per direction (the same kernels, in the same order, on the same operands as the entry
 * points above composed by the caller: 2 calls and 2 allocations per encoder and training step instead of 24 and ~100).
 *   params: HOST array of 34 device pointers in the module's order: per block (Wq, bq, Wk, bk, Wv, bv, Wo, bo) x 4, then the out
 *           projection's W[1024, 1280] and bias[1024].   x[B*256, 256].
 *   fwd:  map = the mean map, [B,256,256] (strips == 0) or its strip column sums [B,16,256] (strips != 0, see pzn_attn_fused_fwd);
 *         out[B*256, 1024] or NULL; fmax[B,1024], arg[B,1024]; saved: pzn_attn_chain_saved_bytes(B) bytes, 256-byte aligned,
 *         kept for the backward.
 *   bwd:  the case predict5 creates - only the maximum carries a gradient, dfg[B,1024]; grads: HOST array of 34 device pointers
 *         (accumulate != 0: all ADDED to; 0: the blocks' overwritten, the out projection's two zero-initialised by the caller);
 *         dx[B*256, 256] overwritten; scratch: pzn_attn_chain_scratch_bytes(B) bytes, 256-byte aligned.
 * The precision mode (pzn_attn_set_precision) must be the same for both calls. */
size_t pzn_attn_chain_saved_bytes(int B);
size_t pzn_attn_chain_scratch_bytes(int B);
int pzn_attn_chain_fwd_f32(const float* x, const float* const* params, int B, int strips, float* map, float* out,
                           float* fmax, int32_t* arg, void* saved, pzn_stream_t stream);
int pzn_attn_chain_bwd_f32(const float* x, const float* const* params, const void* saved, const int32_t* arg,
                           const float* dfg, int B, float* const* grads, int accumulate, float* dx, void* scratch,
                           pzn_stream_t stream);

/* First layer of the boundary heads without the concatenation (model5_b.py:745-752: Linear(cat([g.repeat(1,N,1), x], -1))
 * = x W[:,Cg:]^T per point + (g W[:,:Cg]^T + b) per cloud):
 *   pzn_cloud_bias_relu_f32:     y[b,n,:] = relu(y[b,n,:] + cb[b,:])  in place       (C % 4 == 0, 16-byte aligned)
 *   pzn_cloud_gated_colsum_f32:  dcb[b,c] = sum_n (y[b,n,c] > 0 ? dy[b,n,c] : 0)    (overwritten; fixed summation order) */
int pzn_cloud_bias_relu_f32(float* y, const float* cb, int B, int N, int C, pzn_stream_t stream);
int pzn_cloud_gated_colsum_f32(const float* dy, const float* y, int B, int N, int C, float* dcb, pzn_stream_t stream);

/* The per-point MLP chains of the boundary heads as ONE launch each way (csrc/pointmlp.hip; replaces the three
 * nn.Linear (+ ReLU) launches of nn.Sequential at model5_b.py:571-592 as called at :738-739, :751-754):
 *   y = (relu(relu(x W1^T + b1) W2^T + b2)) W3^T + b3  on M rows of 64 floats, h1 / h2 = the hidden activations.
 * Chains: 64 -> 64 -> 64 -> 64 (C2 = C3 = 64) and 64 -> 64 -> 32 -> 2 (C2 = 32, C3 = 2); W1[64, ldw1] may be a column
 * slice of a wider matrix (ldw1 >= 64); b1 is [64], or with b1_per_cloud != 0 one row of 64 per cloud of rows_per_cloud
 * rows (the folded global half of a head's first layer, see pzn_cloud_bias_relu_f32).  M % 32 == 0, 16-byte aligned
 * operands; PZN_EUNSUPPORTED for other chains (compose pzn_linear_* then). */
int pzn_point_mlp3_supported(int C0, int C1, int C2, int C3);
int pzn_point_mlp3_fwd_f32(const float* x, long long M, int rows_per_cloud, const float* W1, int ldw1,
                           const float* b1, int b1_per_cloud, const float* W2, const float* b2,
                           const float* W3, const float* b3, int C2, int C3, float* h1, float* h2,
                           float* y, pzn_stream_t stream);
/* Backward of the chain in one pass + a fixed-order sum of its partial results (timing-independent output):
 * dx[M, 64], dW1[64, ldw1] (columns 0..63 written), dW2[C2, 64], dW3[C3, C2], db2[C2], db3[C3] and db1 — [64], or with
 * b1_per_cloud != 0 one row of 64 per cloud (the gradient of the per-cloud bias) —: OVERWRITTEN (accumulate == 0) or, the
 * parameter gradients, ADDED to (accumulate != 0: one owner per element, no atomics - e.g. straight into a flat gradient
 * bucket; dx and a per-cloud db1 are overwritten either way); x, h1, h2 as the forward left them; workspace of
 * pzn_point_mlp3_bwd_workspace_bytes() bytes (0 = unsupported shape). */
size_t pzn_point_mlp3_bwd_workspace_bytes(long long M, int rows_per_cloud, int b1_per_cloud, int C2, int C3);
int pzn_point_mlp3_bwd_f32(const float* dy, const float* x, const float* h1, const float* h2, long long M,
                           int rows_per_cloud, const float* W1, int ldw1, int b1_per_cloud, const float* W2,
                           const float* W3, int C2, int C3, float* dx, float* dW1, float* db1, float* dW2,
                           float* db2, float* dW3, float* db3, int accumulate, void* workspace, pzn_stream_t stream);

/* The encoder's out projection and the max over the points in ONE launch (model5_b.py:466-475):
 *   out[b,l,:] = cat(x[0] .. x[nslice-1])[b,l,:] W^T + bias   (W[Nout, nslice*E]; the concatenation is never built),
 *   fmax[b,:] = max_l out[b,l,:],  arg[b,:] = its point (the lowest one on ties, as torch.max).
 * out may be NULL: predict5 uses only the maximum (model5_b.py:723).  Shape taken: L = 256, E = 256, nslice = 5,
 * Nout = 1024 (pzn_outproj_maxpts_workspace_bytes > 0; the workspace, 16-byte aligned, holds the split planes of W);
 * everything else, and the exact-fp32 engine, returns PZN_EUNSUPPORTED (compose pzn_linear_slice_fwd_f32 +
 * pzn_maxpool_points_fwd_f32 then).  arg feeds pzn_linear_maxpts_dgrad/wgrad_f32. */
size_t pzn_outproj_maxpts_workspace_bytes(int L, int E, int nslice, int Nout);
int pzn_outproj_maxpts_fwd_f32(const float* const* x, int nslice, const float* W, const float* bias, int B, int L, int E,
                               int Nout, float* out, float* fmax, int32_t* arg, void* workspace, pzn_stream_t stream);

/* torch.max(x, dim=1) over the point axis (model5_b.py:475 global feature, :741): out[b,c] =
 * max_l x[b,l,c], idx[b,c] = its row (the lowest one on ties); backward dx[b,l,c] =
 * (l == idx[b,c]) ? dout[b,c] : 0, every element of dx written (any C; C % 4 == 0 with 16-byte aligned
 * pointers takes the vector kernels). */
int pzn_maxpool_points_fwd_f32(const float* x, int B, int L, int C, float* out, int32_t* idx,
                               pzn_stream_t stream);
int pzn_maxpool_points_bwd_f32(const float* dout, const int32_t* idx, int B, int L, int C,
                               float* dx, pzn_stream_t stream);

/* Backward of "linear, then max over the points" (model5_b.py:474-475: out = self.out(att); f_global = max over
 * dim 1) when only the maximum is used downstream (predict5, model5_b.py:723): the gradient of the [B*L, Nout]
 * product has one non-zero per (cloud, channel), at row arg[b,c] (pzn_maxpool_points_fwd_f32's idx), so both
 * products are B*Nout row operations instead of B*L*Nout*Kin multiply-adds:
 *   dgrad:  dx[b,l,:] = sum over {c: arg[b,c]==l} dg[b,c] W[c,:]        (every element of dx written; Kin % 64 == 0,
 *           L <= 600, Nout <= 16384, W[Nout,Kin] dense; workspace of pzn_linear_maxpts_workspace_bytes(B, Nout)
 *           bytes, 16-byte aligned: the channels of each cloud sorted by selected row; summation order is fixed,
 *           results are reproducible)
 *   wgrad:  dW[c, s*seg_cols + k] += sum_b dg[b,c] x_s[b, arg[b,c], k],  db[c] += sum_b dg[b,c]  (db may be NULL);
 *           x is given as nseg <= 8 separate [B*L, seg_cols] tensors — the column blocks of a concatenation that
 *           need not exist (x_segs: HOST array of nseg device pointers; seg_cols % 4 == 0; dW[Nout, nseg*seg_cols]).
 * PZN_EUNSUPPORTED outside these shapes (callers then take pzn_maxpool_points_bwd_f32 + the dense products). */
size_t pzn_linear_maxpts_workspace_bytes(int B, int Nout);
int pzn_linear_maxpts_dgrad_f32(const float* dg, const int32_t* arg, const float* W, int B, int L, int Kin,
                                int Nout, void* workspace, float* dx, pzn_stream_t stream);
int pzn_linear_maxpts_wgrad_f32(const float* dg, const int32_t* arg, const float* const* x_segs, int nseg,
                                int seg_cols, int B, int L, int Nout, float* dW, float* db, pzn_stream_t stream);

/* SE(3) exponential of the pose head (se_math/se3.py:57-80 with so3.mat and the Taylor-guarded
 * sinc1/2/3 of se_math/sinc.py): twist[B,6] = (w, v) -> g[B,4,4]; backward dg[B,4,4] ->
 * dtwist[B,6] (the last row of dg is ignored: it is constant). */
int pzn_se3_exp_fwd_f32(const float* twist, int B, float* g, pzn_stream_t stream);
int pzn_se3_exp_bwd_f32(const float* twist, const float* dg, int B, float* dtwist,
                        pzn_stream_t stream);

/* se3.transform on points (se_math/se3.py:110-120 as model5_b.py:948-952, :1116 use it): out[b,n,:] = R_b p[b,n,:] + t_b
 * with g[B,4,4] = [[R, t], [0 0 0 1]] and p[B,N,3].  bwd: dp[B,N,3] = R^T dout (may be NULL), dg[B,4,4] (may be NULL;
 * overwritten: dR = sum_n dout p^T, dt = sum_n dout, last row 0). */
int pzn_se3_transform_fwd_f32(const float* g, const float* p, int B, int N, float* out, pzn_stream_t stream);
int pzn_se3_transform_bwd_f32(const float* g, const float* p, const float* dout, int B, int N, float* dp,
                              float* dg, pzn_stream_t stream);
/* TouchedRegraster.comp (model5_b.py:1512-1519): loss[0] = 16 * mean((g igt - I)^2) over [B,4,4];
 * bwd: dg[B,4,4] = dloss[0] * d loss / d g (igt is data). */
int pzn_comp_fwd_f32(const float* g, const float* igt, int B, float* loss, pzn_stream_t stream);
int pzn_comp_bwd_f32(const float* g, const float* igt, const float* dloss, int B, float* dg, pzn_stream_t stream);
/* Boundary head losses (model5_b.py:1063-1064 F.cross_entropy(logits[B,2,N], labels) with mean reduction, and :1085-1090
 * softmax(logits, dim=1)[:, 1, :]): labels[B,N] hold 0 / 1 as floats (dataset.py:1363-1366).  fwd: prob1[B,N] (class-1
 * probability, what the top-128 selection ranks by), loss[0] (overwritten); loss points at PZN_BOUNDARY_CE_LOSS_FLOATS floats:
 * the value and the per-workgroup partial sums it is reduced from in a fixed order (bit-reproducible).
 * bwd: dlogits[B,2,N] = dloss[0] * d loss / d logits.
 * points_major != 0: logits (and dlogits) are the heads' [B,N,2] output as it lies in memory - the [B,2,N] tensor the reference
 * builds with permute(0,2,1) (model5_b.py:751-754) read through its strides, no transposed copy either way. */
#define PZN_BOUNDARY_CE_LOSS_FLOATS 513
int pzn_boundary_ce_fwd_f32(const float* logits, const float* labels, int B, int N, int points_major, float* prob1,
                            float* loss, pzn_stream_t stream);
int pzn_boundary_ce_bwd_f32(const float* logits, const float* labels, const float* dloss, int B, int N,
                            int points_major, float* dlogits, pzn_stream_t stream);
/* torch.topk(x[R,N], K, dim=1)[1] (model5_b.py:1089-1091): indices of the K largest entries of every row, value
 * descending, equal values by ascending index.  K <= 256, N <= 16384 (PZN_EUNSUPPORTED beyond). */
int pzn_topk_rows_f32(const float* x, int R, int N, int K, int64_t* idx, pzn_stream_t stream);
/* (((a + b) + c) + d) / 4 elementwise (model5_b.py:468-469: the mean of the four attention maps); n % 4 == 0, 16-byte
 * aligned pointers (PZN_EUNSUPPORTED otherwise). */
int pzn_avg4_f32(const float* a, const float* b, const float* c, const float* d, size_t n, float* out,
                 pzn_stream_t stream);
/* a[B,R,C] -> mean[B,C] = a.mean(dim=1) and arg[B] = index of its largest entry (the lowest on ties): model5_b.py:937-942,
 * `x2[:, topk(attention.mean(dim=1), 32)[1][:, 0]]` needs only the first of the 32.  C <= 1024; workspace of
 * pzn_colmean_workspace_bytes(B, C) bytes (partial column sums, fixed summation order). */
size_t pzn_colmean_workspace_bytes(int B, int C);
int pzn_colmean_argmax_f32(const float* a, int B, int R, int C, float* mean, int64_t* arg, void* workspace,
                           pzn_stream_t stream);

/* torch.optim.Adam step (model5_b.py:1453-1457: Adam(lr), no weight decay, no amsgrad) over flat
 * buffers of n floats: param, exp_avg, exp_avg_sq updated in place from grad; step = 1, 2, ...
 * (bias corrections 1 - beta^step).  All four buffers 16-byte aligned. */
int pzn_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                      size_t n, float lr, float beta1, float beta2, float eps, int step,
                      pzn_stream_t stream);

/* ------------------------------------------------------------------------ */
/* Data pipeline on the GPU: dataset.py                        */
/* ------------------------------------------------------------------------ */

/* dataset.py:761-775 + 1176-1180 (CADDataset.slice with its re-draw loop) for a batch, one launch: K candidate planes per sample
 * (normals float64 [B,K,3] = np.random.rand(3,1), zs float64 [B,K] = np.random.rand(1)/3), the FIRST that leaves >= n_min points
 * on both sides is taken (none: the most balanced one, ok = 0).  pieces [2B,cap,3]: rows 0..B-1 the `up` pieces
 * (points . normal + z >= 0, float64, no fma), rows B..2B-1 the `down` pieces, each in the cloud's point order and padded with
 * copies of its first row; counts int64 [2B]; start int64 [2B] = floor(u * count) (u float64 [B,2]: uniform draws standing in for
 * np.random.randint(0, n_piece), dataset.py:1153); plane float64 [B,4] = (normal, z) taken; ok uint8 [B]. */
int pzn_cut_compact_f32(const float* raw, const double* normals, const double* zs, const double* u, int B, int M,
                        int K, int n_min, int cap, float* pieces, int64_t* counts, int64_t* start, double* plane,
                        uint8_t* ok, pzn_stream_t stream);
/* dataset.py:1363-1366: 0/1 masks [R,N] with ones at the k picked rows idx int64 [R,k] of each cloud. */
int pzn_pick_mask_f32(const int64_t* idx, int R, int k, int N, float* mask, pzn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PZN_H_ */
