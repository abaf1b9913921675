// pzn_internal.h — functions shared between translation units of libpzn.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// gemm.hip: 0 = attention products on the split-precision path (fp32 results), 1 = single bf16 MFMAs (pzn_attn_set_precision)
int pzn_attn_precision_mode();

// poolbwd.hip: weight-gradient pass of "linear + ReLU + max over 32 neighbours" as the sparse problem it is:
//   dW[C2, C1] += scatter(dout)^T h,  db[C2] += column sums (db may be NULL), one non-zero of scatter(dout) per (group, channel).
// h[G*32, C1] = the rows, or NULL with a gate source: the rows are then regenerated from the per-point first layer
// (csrc/sapoint.hip, pzn_sa_prep_f32):  h[(g, k), :] = relu(P[(g / S) * N + idx[g*32 + k], :] + Q[g, :]).
struct PznGateSource {
  const float* P;        // [B*N, C1]
  const int64_t* idx;    // [G*32]
  const float* Q;        // [G, C1]
  int N, S;
};
bool pzn_pool_wgrad_supported(int C1, int C2, const float* h);
// ws (optional, pzn_pool_wgrad_ws_bytes): the workgroups' partial tiles, summed in a fixed order; NULL: fp32 atomics
size_t pzn_pool_wgrad_ws_bytes(int G, int C1, int C2);
int pzn_pool_wgrad_sparse(const float* dout, const int32_t* argmax, const float* out, const float* h, float* dW, float* db,
                          int G, int C1, int C2, hipStream_t st, const PznGateSource* gs = nullptr, void* ws = nullptr,
                          size_t ws_bytes = 0);

// attnwgrad.hip: the four weight gradients of one attention block (E = 256, dk = 64) in one launch of LDS-shared 128 x 128 tiles
// (+ a fixed-order reduction of the row ranges' partial tiles when the caller has a workspace of pzn_attn_wgrad_ws_bytes(M)
// bytes; ws == NULL: fp32 atomics).  PZN_EUNSUPPORTED unless M % 64 == 0.
size_t pzn_attn_wgrad_ws_bytes(int M);
int pzn_attn_wgrad_tiled(const float* dz, const float* t, const float* dq, const float* dkk, const float* dvv, const float* x,
                         int M, float* dWq, float* dbq, float* dWk, float* dbk, float* dWv, float* dbv, float* dWo, float* dbo,
                         int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
// gemm.hip: pzn_attn_fused_wgrads with that workspace (ws_bytes = 0: none)
int pzn_attn_fused_wgrads_ws(const float* dz, const float* t, const float* dq, const float* dkk, const float* dvv, const float* x,
                             int M, int E, int dk, float* dWq, float* dbq, float* dWk, float* dbk, float* dWv, float* dbv,
                             float* dWo, float* dbo, int accumulate, void* ws, size_t ws_bytes, pzn_stream_t stream);

// wsgemm.hip: weight-stationary bf16x3 GEMM for skinny layers.  C[M,N] = epi(A[M,K] W^T), W[n*ldw+k]
// (w_kmajor = 0) or W[k*ldw+n] (w_kmajor = 1); genY masks A by genY > 0, maskH masks C, argmax != NULL
// selects the max-over-32-rows epilogue (C is then [M/32, N]); scat != NULL adds row m atomically into C row
// (m / scat_in) * scat_out + scat[m] instead of storing it.
bool pzn_ws_gemm_supported(int M, int N, int K, const float* A, int lda, const float* genY, bool maxpool);
int pzn_ws_gemm(const float* A, int lda, const float* W, int ldw, int w_kmajor, float* C, int ldc, int M, int N, int K,
                const float* bias, int relu, const float* genY, const float* maskH, int32_t* argmax,
                const int64_t* scat, int scat_in, int scat_out, hipStream_t st);

// same with the extra store-epilogue options: residual != NULL adds residual(m,n) (layout of C) — into C2 when C2 != NULL
// (C then keeps the value without it), else into C; accumulate != 0 adds to what C holds.
int pzn_ws_gemm_ex(const float* A, int lda, const float* W, int ldw, int w_kmajor, float* C, int ldc, int M, int N, int K,
                   const float* bias, int relu, const float* genY, const float* maskH, int32_t* argmax,
                   const int64_t* scat, int scat_in, int scat_out, const float* residual, float* C2, int accumulate,
                   hipStream_t st);

// C[M, N] = A W^T + xyz[M, 3] (wx | wy | wz)[N] (planes contiguous): the coordinate columns of a layer over [xyz | features] in
// the store epilogue, with sa_prep_kernel's arithmetic (bit-identical to the product followed by pzn_sa_prep_f32's pass)
int pzn_ws_gemm_r3(const float* A, int lda, const float* W, int ldw, float* C, int ldc, int M, int N, int K, const float* xyz,
                   const float* planes, hipStream_t st);
// sapoint.hip: only the Q table of pzn_sa_prep_f32 (Q[g, :] = b1 - W1[:, 0:3] new_xyz[g])
int pzn_sa_prep_q(const float* new_xyz, const float* W1, const float* b1, int B, int S, int D, int C1, float* Q, hipStream_t st);

// salevel.hip: the generated-row max-pool level with W2 streamed through LDS; PZN_EUNSUPPORTED for other shapes
size_t pzn_sa_level_stream_workspace_bytes(int C1, int C2);
int pzn_sa_level_stream(const float* Pp, const float* Q, const int64_t* idx, const float* W2, const float* b2, int G, int N,
                        int S, int C1, int C2, float* out, int32_t* argmax, void* workspace, hipStream_t st, int prepacked = 0);
int pzn_sa_level_stream_pack(const float* W2, int C1, int C2, void* workspace, hipStream_t st);
// max-pool variant on a generated activation stream (first set-abstraction layer per point): see wsgemm.hip
int pzn_ws_gemm_gather_maxpool(const float* Pp, const float* Q, const int64_t* idx, const float* W2, const float* b2, int G,
                               int N, int S, int C1, int C2, float* out, int32_t* argmax, hipStream_t st);

// dfgemm.hip: weight gradient dW[N,K'] += dY^T X (+ db += column sums of dY) with MFMA fragments loaded
// straight from global memory; genY masks dY by genY > 0; skip_col >= 0 drops that column of X from the
// output (K' = K - 1: the zero pad of the padded group rows).  Accumulates: zero dW / db first if needed.
bool pzn_df_wgrad_supported(int M, int N, int K);
int pzn_df_wgrad(const float* dy, int ldy, const float* genY, const float* x, int ldx, int M, int N, int K, float* dW,
                 int ldw, float* db, int skip_col, hipStream_t st);
// three weight gradients sharing X in one launch: dW_i[N_i,K] += dY_i^T X, db_i += sums; N_i % 64 == 0, dY_i dense
int pzn_df_wgrad3(const float* const dy[3], const int n[3], float* const dW[3], float* const db[3], const float* x, int ldx,
                  int M, int K, hipStream_t st);
// three linear layers sharing their input in one launch: C_i[M,N_i] = A W_i^T + b_i (N_i % 64 == 0, dense C_i)
int pzn_ws_gemm3(const float* A, int lda, const float* const W[3], const float* const bias[3], float* const C[3],
                 const int N[3], int M, int K, hipStream_t st);
