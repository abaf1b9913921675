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

// dgrad, step 1: per cloud, the channels sorted by (selected row, channel): keys (row << 16 | c) are unique, so a
// bitonic sort in LDS gives the stable order; one workgroup per cloud, 55 compare-exchange stages for 1024 channels.
__global__ __launch_bounds__(1024) void maxpts_sort_kernel(const int32_t* __restrict__ arg, int L, int Nout, int npad,
                                                           uint32_t* __restrict__ sorted) {
  extern __shared__ uint32_t keys[];  // [npad]
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < npad; c += blockDim.x)
    keys[c] = c < Nout ? ((uint32_t)min(max(arg[(size_t)b * Nout + c], 0), L - 1) << 16) | (uint32_t)c : 0xFFFFFFFFu;
  __syncthreads();
  for (int k = 2; k <= npad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < npad; i += blockDim.x) {
        const int p = i ^ j;
        if (p > i) {
          const uint32_t x = keys[i], y = keys[p];
          if ((x > y) == ((i & k) == 0)) keys[i] = y, keys[p] = x;
        }
      }
      __syncthreads();
    }
  for (int c = threadIdx.x; c < Nout; c += blockDim.x) sorted[(size_t)b * Nout + c] = keys[c];
}

// dgrad, step 2: workgroup = (cloud, 64-column chunk), clouds on the fast grid axis (the workgroups resident at any
// time share one or two 256-KB column chunks of W, which stay in every XCD's L2).  The [L][64] slab of dx lives in LDS.
// The sorted channel list is cut into 16 equal segments, one per wavefront, whatever the rows are — arg-max rows pile
// up (in the training step most channels of a cloud select a handful of points), so ownership by row would leave one
// wavefront with most of the work.  A wavefront walks its segment eight channels at a time (eight coalesced 256-byte
// reads of W[c, chunk] in flight), sums runs of equal row in registers and stores a finished run to the slab; only its
// first and last run can continue in a neighbour's segment: those go to a side buffer and one wavefront adds them in
// segment order afterwards.  No atomics (ds_add_f32 made a first version 7x slower than this), and a fixed summation
// order: results are reproducible bit for bit.  The slab is streamed out whole, so rows nobody selected are written as
// zeros and dx needs no zero fill.
__global__ __launch_bounds__(MD_THREADS) void maxpts_lin_dgrad_kernel(const float* __restrict__ dg,
                                                                      const uint32_t* __restrict__ sorted,
                                                                      const float* __restrict__ W, int L, int Kin, int Nout,
                                                                      float* __restrict__ dx) {
  extern __shared__ float slab[];  // [L][MD_COLS]
  constexpr int NW = MD_THREADS / 64;
  __shared__ float bnd[NW][2][MD_COLS];
  __shared__ int bnd_row[NW][2];
  const int b = blockIdx.x, col0 = blockIdx.y * MD_COLS;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < L * MD_COLS; i += MD_THREADS) slab[i] = 0.f;
  if (lane < 2) bnd_row[wave][lane] = -1;
  __syncthreads();
  const int seg = (Nout + NW - 1) / NW;
  const int i0 = wave * seg, i1 = min(Nout, i0 + seg);
  const float* w = W + col0 + lane;
  int cur = -1, runs = 0;
  float acc = 0.f;
  for (int base = i0; base < i1; base += 64) {
    const int n = min(64, i1 - base);
    uint32_t key = 0;
    float gl = 0.f;
    if (lane < n) {
      key = sorted[(size_t)b * Nout + base + lane];
      gl = dg[(size_t)b * Nout + (key & 0xFFFFu)];
    }
    for (int t0 = 0; t0 < n; t0 += 8) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const uint32_t ku = (uint32_t)__builtin_amdgcn_readlane((int)key, min(t0 + u, n - 1));
        wv[u] = w[(size_t)(ku & 0xFFFFu) * Kin];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (t0 + u < n) {
          const uint32_t ku = (uint32_t)__builtin_amdgcn_readlane((int)key, t0 + u);
          const float gu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gl), t0 + u));
          const int row = (int)(ku >> 16);
          if (row != cur) {
            if (cur >= 0) {
              if (runs == 0) {      // the segment's first run may have begun in the previous segment
                bnd[wave][0][lane] = acc;
                if (lane == 0) bnd_row[wave][0] = cur;
              } else {
                slab[cur * MD_COLS + lane] = acc;
              }
              ++runs;
            }
            cur = row, acc = 0.f;
          }
          acc += gu * wv[u];
        }
      }
    }
  }
  if (cur >= 0) {      // the last run may continue in the next segment (it is also the first one when runs == 0)
    bnd[wave][runs == 0 ? 0 : 1][lane] = acc;
    if (lane == 0) bnd_row[wave][runs == 0 ? 0 : 1] = cur;
  }
  __syncthreads();
  if (wave == 0) {
    for (int v = 0; v < NW; ++v)
      for (int e = 0; e < 2; ++e) {
        const int r = bnd_row[v][e];
        if (r >= 0) slab[r * MD_COLS + lane] += bnd[v][e][lane];
      }
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

PZN_EXPORT size_t pzn_linear_maxpts_workspace_bytes(int B, int Nout) {
  return B > 0 && Nout > 0 ? (size_t)B * Nout * sizeof(uint32_t) : 0;
}

PZN_EXPORT int pzn_linear_maxpts_dgrad_f32(const float* dg, const int32_t* arg, const float* W, int B, int L, int Kin,
                                           int Nout, void* workspace, float* dx, pzn_stream_t stream) {
  PZN_CHECK_ARG(dg && arg && W && dx && workspace && B > 0 && B <= 65535 && L > 0 && Kin > 0 && Nout > 0);
  const size_t lds = (size_t)L * MD_COLS * sizeof(float);
  int npad = 64;
  while (npad < Nout) npad <<= 1;
  if (Kin % MD_COLS != 0 || lds > (size_t)MD_LDS_LIMIT || L > 65535 || Nout > 16384 ||
      ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
    return PZN_EUNSUPPORTED;
  static bool attr_set_dev[64] = {};      // per device: the attribute belongs to the device's copy of the kernel
  int devid = 0;                          // (benign race: the attribute is idempotent)
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return PZN_ELAUNCH;
  bool& attr_set = attr_set_dev[devid];
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(maxpts_lin_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            MD_LDS_LIMIT) != hipSuccess)
      return PZN_ELAUNCH;
    attr_set = true;
  }
  uint32_t* sorted = static_cast<uint32_t*>(workspace);
  PZN_LAUNCH(maxpts_sort_kernel, dim3((unsigned)B), dim3(1024), (size_t)npad * sizeof(uint32_t), pzn_hip_stream(stream),
                     arg, L, Nout, npad, sorted);
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  PZN_LAUNCH(maxpts_lin_dgrad_kernel, dim3((unsigned)B, (unsigned)(Kin / MD_COLS)), dim3(MD_THREADS), lds,
                     pzn_hip_stream(stream), dg, sorted, W, L, Kin, Nout, dx);
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
  PZN_LAUNCH(maxpts_lin_wgrad_kernel, dim3((unsigned)Nout, (unsigned)gy), dim3(256), 0, pzn_hip_stream(stream), dg,
                     arg, xs, seg_cols, cblocks, B, L, Nout, nseg * seg_cols, dW, db);
  PZN_RETURN_LAUNCH_STATUS();
}
