// attnchain.hip — model5_b.py:462-475 of ONE encoder behind one entry point each way.
//
// The four layerAttention blocks, the running mean of their maps, the out projection over the five un-concatenated slices and
// the max over the points are 10 launches forward and 14 backward through the chained kernels of attnfused.hip / outproj.hip /
// maxptsbwd.hip / dfgemm.hip.  Enqueued one by one from Python each costs a ctypes call and the ~20 tensors a layer leaves for
// its backward cost an allocation each: 48 calls and ~200 allocations per training step for the two encoders, a fifth of the
// host time of a step.  Here the sequence is enqueued by the library on ONE caller-owned buffer per direction whose layout is
// private to this file: 2 calls and 2 allocations per encoder and step.  The kernels, their order and every operand are the
// ones the Python composition (ops._AttnChainFused) uses — this file only moves the loop across the ABI.
//
// forward buffer ("saved": the backward reads it):  per layer  W planes | q image | k image | v image | r [M,E] | t [M,E] |
//   gate bits [M,8] | ln-sum-exp [M];  then the out projection's plane workspace.
// backward buffer (scratch):  G [M,5E] | dz | u | dx0 | dx1 [M,E each] | dq | dqt | dk [M,dk each] | dv [M,E] | da image |
//   delta [M] | sort workspace of the sparse out-projection backward | partial tiles of the weight gradients (attnwgrad.hip).
#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

constexpr int L = 256, E = 256, DK = 64, NOUT = 1024, NLAYER = 4;

size_t up256(size_t n) { return (n + 255) / 256 * 256; }

struct FwdLayout {
  size_t w, q, k, v, r, t, mask, lse, layer;      // offsets inside a layer, bytes per layer
  size_t outproj, total;
};

FwdLayout fwd_layout(int B) {
  const size_t M = (size_t)B * L;
  FwdLayout f;
  size_t at = 0;
  f.w = at, at += up256(pzn_attn_fused_weight_bytes());
  f.q = at, at += up256(pzn_attn_fused_qk_image_bytes(B));
  f.k = at, at += up256(pzn_attn_fused_qk_image_bytes(B));
  f.v = at, at += up256(pzn_attn_fused_v_image_bytes(B));
  f.r = at, at += up256(M * E * 4);
  f.t = at, at += up256(M * E * 4);
  f.mask = at, at += up256(M * 8 * 4);
  f.lse = at, at += up256(M * 4);
  f.layer = at;
  f.outproj = NLAYER * f.layer;
  f.total = f.outproj + up256(pzn_outproj_maxpts_workspace_bytes(L, E, 5, NOUT));
  return f;
}

struct BwdLayout {
  size_t G, dz, u, dx0, dx1, dq, dqt, dk, dv, da, delta, sort, wg, wg_bytes, total;
};

BwdLayout bwd_layout(int B) {
  const size_t M = (size_t)B * L;
  BwdLayout b;
  size_t at = 0;
  b.G = at, at += up256(M * 5 * E * 4);
  b.dz = at, at += up256(M * E * 4);
  b.u = at, at += up256(M * E * 4);
  b.dx0 = at, at += up256(M * E * 4);
  b.dx1 = at, at += up256(M * E * 4);
  b.dq = at, at += up256(M * DK * 4);
  b.dqt = at, at += up256(M * DK * 4);
  b.dk = at, at += up256(M * DK * 4);
  b.dv = at, at += up256(M * E * 4);
  b.da = at, at += up256(pzn_attn_fused_v_image_bytes(B));
  b.delta = at, at += up256(M * 4);
  b.sort = at, at += up256(pzn_linear_maxpts_workspace_bytes(B, NOUT));
  b.wg_bytes = pzn_attn_wgrad_ws_bytes((int)M);
  b.wg = at, at += up256(b.wg_bytes);
  b.total = at;
  return b;
}

// out[m, :] = a[m, :] (row stride lda) + b[m, :] (dense), E columns: the chain's input gradient = its slice of the projection's
// input gradient + what the first block passes back
__global__ __launch_bounds__(256) void add_slice_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                        long rows, float* __restrict__ out) {
  const long n4 = rows * (E / 4);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const long m = i / (E / 4);
    const int c = (int)(i - m * (E / 4)) * 4;
    const float4 x = *reinterpret_cast<const float4*>(a + m * lda + c);
    const float4 y = *reinterpret_cast<const float4*>(b + m * E + c);
    *reinterpret_cast<float4*>(out + m * E + c) = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

}  // namespace

PZN_EXPORT size_t pzn_attn_chain_saved_bytes(int B) { return B > 0 ? fwd_layout(B).total : 0; }
PZN_EXPORT size_t pzn_attn_chain_scratch_bytes(int B) { return B > 0 ? bwd_layout(B).total : 0; }

// params: HOST array of 34 device pointers in the module's order — per block (Wq, bq, Wk, bk, Wv, bv, Wo, bo) x 4, then the out
// projection's W[1024, 1280] and bias.  x[B*256, 256]; map: [B,256,256] (strips == 0) or [B,16,256] (strips != 0), the mean of the
// four blocks' maps (model5_b.py:468-469); out: [B*256, 1024] or NULL (predict5 uses only the maximum, :723); fmax[B,1024],
// arg[B,1024]; saved: pzn_attn_chain_saved_bytes(B) bytes, 256-byte aligned, kept by the caller for the backward.
PZN_EXPORT int pzn_attn_chain_fwd_f32(const float* x, const float* const* params, int B, int strips, float* map, float* out,
                                      float* fmax, int32_t* arg, void* saved, pzn_stream_t stream) {
  PZN_CHECK_ARG(x && params && B > 0 && map && fmax && arg && saved && (reinterpret_cast<uintptr_t>(saved) & 255) == 0);
  for (int i = 0; i < 34; ++i) PZN_CHECK_ARG(params[i] != nullptr);
  const FwdLayout f = fwd_layout(B);
  unsigned char* base = static_cast<unsigned char*>(saved);
  auto at = [&](int layer, size_t off) { return static_cast<void*>(base + (size_t)layer * f.layer + off); };
  {
    const float* wq[NLAYER], *wk[NLAYER], *wv[NLAYER], *wo[NLAYER];
    void* planes[NLAYER];
    for (int i = 0; i < NLAYER; ++i)
      wq[i] = params[8 * i], wk[i] = params[8 * i + 2], wv[i] = params[8 * i + 4], wo[i] = params[8 * i + 6], planes[i] = at(i, f.w);
    int rc = pzn_attn_fused_prep_weights_n(NLAYER, wq, wk, wv, wo, planes, stream);
    if (rc != PZN_OK) return rc;
  }
  const float* cur = x;
  for (int i = 0; i < NLAYER; ++i) {
    const void* w = at(i, f.w);
    void* q = at(i, f.q);
    void* k = at(i, f.k);
    void* v = at(i, f.v);
    const float* bq = params[8 * i + 1], *bk = params[8 * i + 3], *bv = params[8 * i + 5], *bo = params[8 * i + 7];
    int rc = pzn_attn_fused_proj(1, &cur, &w, &bq, &bk, &bv, B, &q, &k, &v, stream);
    if (rc != PZN_OK) return rc;
    float* r = static_cast<float*>(at(i, f.r));
    float* t = static_cast<float*>(at(i, f.t));
    void* mask = at(i, f.mask);
    float* lse = static_cast<float*>(at(i, f.lse));
    const void* qc = q, *kc = k, *vc = v;
    rc = pzn_attn_fused_fwd(1, &cur, &qc, &kc, &vc, &w, &bo, B, &r, &t, &mask, &map, &lse, (i > 0 ? 1 : 0) | (strips ? 2 : 0),
                            strips ? 0.25f / 16 : 0.25f, stream);
    if (rc != PZN_OK) return rc;
    cur = r;
  }
  const float* xsl[5] = {static_cast<const float*>(at(0, f.r)), static_cast<const float*>(at(1, f.r)),
                         static_cast<const float*>(at(2, f.r)), static_cast<const float*>(at(3, f.r)), x};      // att1..att4, f2f (:466)
  return pzn_outproj_maxpts_fwd_f32(xsl, 5, params[32], params[33], B, L, E, NOUT, out, fmax, arg, base + f.outproj, stream);
}

// Backward for the case predict5 creates: only f_global = max over the points carries a gradient (dfg[B,1024]).  grads: HOST array
// of 34 device pointers, the parameters' gradients in the order of `params` (accumulate != 0: every one is ADDED to, the flat
// gradient bucket; 0: the blocks' are overwritten, the out projection's two must be zero-initialised by the caller: its
// sparse backward adds).  dx[B*256, 256] is overwritten; scratch: pzn_attn_chain_scratch_bytes(B) bytes, 256-byte aligned.
PZN_EXPORT int pzn_attn_chain_bwd_f32(const float* x, const float* const* params, const void* saved, const int32_t* arg,
                                      const float* dfg, int B, float* const* grads, int accumulate, float* dx, void* scratch,
                                      pzn_stream_t stream) {
  PZN_CHECK_ARG(x && params && saved && arg && dfg && B > 0 && grads && dx && scratch);
  PZN_CHECK_ARG((reinterpret_cast<uintptr_t>(saved) & 255) == 0 && (reinterpret_cast<uintptr_t>(scratch) & 255) == 0);
  for (int i = 0; i < 34; ++i) PZN_CHECK_ARG(params[i] != nullptr && grads[i] != nullptr);
  const FwdLayout f = fwd_layout(B);
  const BwdLayout b = bwd_layout(B);
  const unsigned char* sbase = static_cast<const unsigned char*>(saved);
  unsigned char* w = static_cast<unsigned char*>(scratch);
  auto sv = [&](int layer, size_t off) { return static_cast<const void*>(sbase + (size_t)layer * f.layer + off); };
  const int M = B * L;
  float* G = reinterpret_cast<float*>(w + b.G);
  const float* xsl[5] = {static_cast<const float*>(sv(0, f.r)), static_cast<const float*>(sv(1, f.r)),
                         static_cast<const float*>(sv(2, f.r)), static_cast<const float*>(sv(3, f.r)), x};
  int rc = pzn_linear_maxpts_wgrad_f32(dfg, arg, xsl, 5, E, B, L, NOUT, grads[32], grads[33], stream);
  if (rc != PZN_OK) return rc;
  rc = pzn_linear_maxpts_dgrad_f32(dfg, arg, params[32], B, L, 5 * E, NOUT, w + b.sort, G, stream);
  if (rc != PZN_OK) return rc;
  float* dz = reinterpret_cast<float*>(w + b.dz);
  float* u = reinterpret_cast<float*>(w + b.u);
  float* dxa = reinterpret_cast<float*>(w + b.dx0);
  float* dxb = reinterpret_cast<float*>(w + b.dx1);
  float* dq = reinterpret_cast<float*>(w + b.dq);
  float* dqt = reinterpret_cast<float*>(w + b.dqt);
  float* dkk = reinterpret_cast<float*>(w + b.dk);
  float* dvv = reinterpret_cast<float*>(w + b.dv);
  void* da = w + b.da;
  float* delta = reinterpret_cast<float*>(w + b.delta);
  const float* g = G + 3 * E;      // gradient of att4: its slice of the projection's input gradient (read in place)
  const float* g2 = nullptr;        // what the block behind passed back
  for (int i = NLAYER - 1; i >= 0; --i) {
    const void* mask = sv(i, f.mask), *q = sv(i, f.q), *k = sv(i, f.k), *v = sv(i, f.v), *wp = sv(i, f.w);
    const float* lse = static_cast<const float*>(sv(i, f.lse));
    const float* t = static_cast<const float*>(sv(i, f.t));
    const float* xin = i > 0 ? static_cast<const float*>(sv(i - 1, f.r)) : x;
    rc = pzn_attn_fused_bwd_q(1, &g, 5 * E, g2 ? &g2 : nullptr, E, &mask, &q, &k, &v, &wp, B, &dz, &u, &dq, &dqt, &da, &delta,
                              stream);
    if (rc != PZN_OK) return rc;
    const void* dac = da;
    const float* deltac = delta, *uc = u, *dqtc = dqt;
    rc = pzn_attn_fused_bwd_k(1, &q, &k, &v, &dac, &wp, &lse, &deltac, &uc, &dqtc, B, &dkk, &dvv, &dxa, stream);
    if (rc != PZN_OK) return rc;
    float* const* gp = grads + 8 * i;      // (gq, gbq, gk, gbk, gv, gbv, go, gbo)
    rc = pzn_attn_fused_wgrads_ws(dz, t, dq, dkk, dvv, xin, M, E, DK, gp[0], gp[1], gp[2], gp[3], gp[4], gp[5], gp[6], gp[7],
                                  accumulate, w + b.wg, b.wg_bytes, stream);
    if (rc != PZN_OK) return rc;
    const int sl = i > 0 ? i - 1 : 4;      // att_i sits in slice i-1 of the concatenation, f2f in slice 4
    g = G + (size_t)sl * E;
    g2 = dxa;
    float* tmp = dxa;
    dxa = dxb, dxb = tmp;
  }
  PZN_LAUNCH(add_slice_kernel, dim3(1024), dim3(256), 0, pzn_hip_stream(stream), g, 5 * E, g2, (long)M, dx);
  PZN_RETURN_LAUNCH_STATUS();
}
