// maxptsbwd.hip — backward of "linear, then max over the points of a cloud" as the sparse problem it is.
//
// model5_b.py:474-475:  out = self.out(att)  ([B,256,1280] -> [B,256,1024]);  f_global = torch.max(out, dim=1)[0].
// predict5 (model5_b.py:723-759) uses f_global only, so the gradient of `out` has ONE non-zero per (cloud, channel):
//   dout[b, l, c] = (l == arg[b,c]) ? dg[b,c] : 0.
// As dense products the two backward GEMMs of that layer are 2 x 43 GFLOP per encoder (16384 x 1024 x 1280); as sparse
// row operations they are B*Nout = 65536 axpys of length 1280 each way (168 MFLOP):
//   dgrad:  dx[b, l, :]  = sum over {c : arg[b,c] == l} of dg[b,c] * W[c, :]
//   wgrad:  dW[c, :]    += sum over b of dg[b,c] * x[b, arg[b,c], :],      db[c] += sum over b of dg[b,c]
// (the same idea as poolbwd.hip for the max over the 32 neighbours of a group, but with L = 256 rows per group and no
// ReLU, so neither the ballot walk nor the 32-row tiles of those kernels apply).
#include "pzn_common.h"

namespace {

constexpr int MD_COLS = 64;      // columns of dx per workgroup (one per lane)
constexpr int MD_THREADS = 1024;  // 16 wavefronts share the channel loop
constexpr int MD_LDS_LIMIT = 150 * 1024;

// dgrad: workgroup = (cloud, 64-column chunk), clouds on the fast grid axis (the workgroups resident at any time share
// one or two 256-KB column chunks of W, which stay in every XCD's L2).  The [L][64] slab of dx lives in LDS.  Rows are
// OWNED by wavefronts (row % 16 == wave), so the slab is updated with plain read-add-write — LDS operations of one
// wavefront execute in order — and in a fixed order (ascending channel): the result is reproducible bit for bit.
// (First version: every wavefront took every 16th channel and added with ds_add_f32; 425 us per launch, slower than
// the dense product.)  Channels are bucketed by owner first: each wavefront finds its channels with a ballot per 64
// (count pass, offsets by a 16-entry scan, fill pass: ascending lists in LDS), then walks its list eight channels at a
// time: eight coalesced 256-byte reads of W[c, chunk] in flight, scaled by dg[b,c], added to row arg[b,c].  The slab is
// streamed out whole, so rows nobody selected are written as zeros and dx needs no zero fill.
__global__ __launch_bounds__(MD_THREADS) void maxpts_lin_dgrad_kernel(const float* __restrict__ dg,
                                                                      const int32_t* __restrict__ arg,
                                                                      const float* __restrict__ W, int L, int Kin, int Nout,
                                                                      float* __restrict__ dx) {
  extern __shared__ float slab[];  // [L][MD_COLS], then arg[Nout], dg[Nout], list[Nout]
  constexpr int NW = MD_THREADS / 64;
  __shared__ int cnt[NW];
  int* sa = reinterpret_cast<int*>(slab + (size_t)L * MD_COLS);
  float* sg = reinterpret_cast<float*>(sa + Nout);
  int* list = reinterpret_cast<int*>(sg + Nout);
  const int b = blockIdx.x, col0 = blockIdx.y * MD_COLS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < L * MD_COLS; i += MD_THREADS) slab[i] = 0.f;
  for (int c = threadIdx.x; c < Nout; c += MD_THREADS) {
    sa[c] = min(max(arg[(size_t)b * Nout + c], 0), L - 1);      // (clamp: a bad index must not leave the slab)
    sg[c] = dg[(size_t)b * Nout + c];
  }
  __syncthreads();
  int mine = 0;
  for (int c0 = 0; c0 < Nout; c0 += 64) {
    const int c = c0 + lane;
    const bool hit = c < Nout && (sa[c] & (NW - 1)) == wave;
    mine += __popcll(__ballot(hit));
  }
  if (lane == 0) cnt[wave] = mine;
  __syncthreads();
  int off = 0;
  for (int v = 0; v < wave; ++v) off += cnt[v];
  int fill = off;
  for (int c0 = 0; c0 < Nout; c0 += 64) {
    const int c = c0 + lane;
    const bool hit = c < Nout && (sa[c] & (NW - 1)) == wave;
    const uint64_t m = __ballot(hit);
    if (hit) list[fill + __popcll(m & ((1ull << lane) - 1ull))] = c;
    fill += __popcll(m);
  }
  pzn::wave_lds_sync();      // the list segment is read by this wavefront only
  const float* w = W + col0 + lane;
  constexpr int U = 8;
  int i = off;
  const int end = off + mine;
  for (; i + U <= end; i += U) {
    float wv[U], gv[U];
    int av[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = __builtin_amdgcn_readfirstlane(list[i + u]);
      wv[u] = w[(size_t)c * Kin], gv[u] = sg[c], av[u] = sa[c];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) slab[av[u] * MD_COLS + lane] += gv[u] * wv[u];
  }
  for (; i < end; ++i) {
    const int c = __builtin_amdgcn_readfirstlane(list[i]);
    slab[sa[c] * MD_COLS + lane] += sg[c] * w[(size_t)c * Kin];
  }
  __syncthreads();
  // 16 lanes x 16 bytes = one 256-byte row segment; 64 rows per pass of the workgroup
  const int q = threadIdx.x & 15, r0 = threadIdx.x >> 4;
  for (int l = r0; l < L; l += MD_THREADS / 16)
    *reinterpret_cast<float4*>(dx + ((size_t)b * L + l) * Kin + col0 + q * 4) =
        *reinterpret_cast<const float4*>(slab + l * MD_COLS + q * 4);
}

constexpr int MW_MAXSEG = 8;
struct MwSegs {
  const float* p[MW_MAXSEG];
};

// wgrad: workgroup = (channel c, 256-column block of one segment of x).  x may be a concatenation that was never built
// (model5_b.py:466/:470: cat([att1..att4, f2f])): segment s is its own [B*L, seg_cols] tensor.  Wavefront w sums clouds
// w, w+4, ...: lane l reads 16 bytes of the selected row (a 1-KB row segment per wavefront, 8 rows in flight), the four
// partial sums meet in LDS, and the workgroup — sole owner of its dW run — adds them to what dW already holds.
__global__ __launch_bounds__(256) void maxpts_lin_wgrad_kernel(const float* __restrict__ dg, const int32_t* __restrict__ arg,
                                                               MwSegs xs, int seg_cols, int cblocks, int B, int L, int Nout,
                                                               int ldw, float* __restrict__ dW, float* __restrict__ db) {
  __shared__ float4 red[4][64];
  __shared__ float gsum[4];
  const int c = blockIdx.x;
  const int s = blockIdx.y / cblocks, cb = blockIdx.y % cblocks;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = cb * 256 + lane * 4;
  const bool live = col < seg_cols;
  const float* x = xs.p[s] + (live ? col : 0);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float gs = 0.f;
  constexpr int U = 8;
  int b = wave;
  for (; b + (U - 1) * 4 < B; b += U * 4) {
    float4 v[U];
    float gv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int bb = b + u * 4;
      const int a = min(max(arg[(size_t)bb * Nout + c], 0), L - 1);
      gv[u] = dg[(size_t)bb * Nout + c];
      v[u] = *reinterpret_cast<const float4*>(x + ((size_t)bb * L + a) * seg_cols);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc.x += gv[u] * v[u].x, acc.y += gv[u] * v[u].y, acc.z += gv[u] * v[u].z, acc.w += gv[u] * v[u].w;
      gs += gv[u];
    }
  }
  for (; b < B; b += 4) {
    const int a = min(max(arg[(size_t)b * Nout + c], 0), L - 1);
    const float gv = dg[(size_t)b * Nout + c];
    const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * L + a) * seg_cols);
    acc.x += gv * v.x, acc.y += gv * v.y, acc.z += gv * v.z, acc.w += gv * v.w;
    gs += gv;
  }
  red[wave][lane] = acc;
  if (lane == 0) gsum[wave] = gs;
  __syncthreads();
  if (wave == 0) {
    if (live) {
      const float4 r1 = red[1][lane], r2 = red[2][lane], r3 = red[3][lane];
      float4* out = reinterpret_cast<float4*>(dW + (size_t)c * ldw + (size_t)s * seg_cols + col);
      float4 o = *out;
      o.x += (acc.x + r1.x) + (r2.x + r3.x), o.y += (acc.y + r1.y) + (r2.y + r3.y);
      o.z += (acc.z + r1.z) + (r2.z + r3.z), o.w += (acc.w + r1.w) + (r2.w + r3.w);
      *out = o;
    }
    if (db != nullptr && blockIdx.y == 0 && lane == 0) db[c] += (gsum[0] + gsum[1]) + (gsum[2] + gsum[3]);
  }
}

}  // namespace

PZN_EXPORT int pzn_linear_maxpts_dgrad_f32(const float* dg, const int32_t* arg, const float* W, int B, int L, int Kin,
                                           int Nout, float* dx, pzn_stream_t stream) {
  PZN_CHECK_ARG(dg && arg && W && dx && B > 0 && B <= 65535 && L > 0 && Kin > 0 && Nout > 0);
  const size_t lds = (size_t)L * MD_COLS * sizeof(float) + (size_t)Nout * 12;
  if (Kin % MD_COLS != 0 || lds > (size_t)MD_LDS_LIMIT || (reinterpret_cast<uintptr_t>(dx) & 15) != 0) return PZN_EUNSUPPORTED;
  static bool attr_set = false;      // (benign race: the attribute is idempotent)
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(maxpts_lin_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            MD_LDS_LIMIT) != hipSuccess)
      return PZN_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(maxpts_lin_dgrad_kernel, dim3((unsigned)B, (unsigned)(Kin / MD_COLS)), dim3(MD_THREADS), lds,
                     pzn_hip_stream(stream), dg, arg, W, L, Kin, Nout, dx);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_linear_maxpts_wgrad_f32(const float* dg, const int32_t* arg, const float* const* x_segs, int nseg,
                                           int seg_cols, int B, int L, int Nout, float* dW, float* db,
                                           pzn_stream_t stream) {
  PZN_CHECK_ARG(dg && arg && x_segs && dW && nseg > 0 && seg_cols > 0 && B > 0 && L > 0 && Nout > 0);
  if (nseg > MW_MAXSEG || seg_cols % 4 != 0 || (reinterpret_cast<uintptr_t>(dW) & 15) != 0) return PZN_EUNSUPPORTED;
  MwSegs xs;
  for (int s = 0; s < MW_MAXSEG; ++s) {
    xs.p[s] = s < nseg ? x_segs[s] : nullptr;
    if (s < nseg) {
      PZN_CHECK_ARG(x_segs[s] != nullptr);
      if ((reinterpret_cast<uintptr_t>(x_segs[s]) & 15) != 0) return PZN_EUNSUPPORTED;
    }
  }
  const int cblocks = (seg_cols + 255) / 256;
  const long long gy = (long long)nseg * cblocks;
  PZN_CHECK_ARG(gy <= 65535);
  hipLaunchKernelGGL(maxpts_lin_wgrad_kernel, dim3((unsigned)Nout, (unsigned)gy), dim3(256), 0, pzn_hip_stream(stream), dg,
                     arg, xs, seg_cols, cblocks, B, L, Nout, nseg * seg_cols, dW, db);
  PZN_RETURN_LAUNCH_STATUS();
}
