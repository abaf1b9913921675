// stem.hip — the per-point stem of the encoder in ONE launch each way (round 5).
//
// model5_b.py:447-448:   x_feature = relu(bn2(mlp2(relu(bn1(mlp1(xyz))))))
// with mlp1 = Linear(3, 64), mlp2 = Linear(64, 64) and bn1, bn2 = BatchNorm1d(num_points) applied to [B, N, 64] tensors:
// the BatchNorm "channel" is the POINT index n, its statistics run over the B x 64 values of the point (model5_b.py:424).
// Everything a point needs is therefore local to the point: its B coordinates (B x 12 bytes), the two small weight
// matrices, its own BatchNorm parameters.  Rounds 1-4 ran this as 2 linear + 2 BatchNorm launches forward and 5 launches
// backward per encoder, ~9 passes over 33.5 MB tensors (0.42 ms of launches per step).  Here one workgroup = one point at a
// time (512 threads: wavefront w holds the samples b = w, w + 8, ... of the point, lane = channel), persistent over the
// points of its share:
//   forward : y1 = W1 xyz + b1 (3 fma per value) -> statistics 1 -> a1 = relu(bn1(y1)) -> LDS tile [B][64] ->
//             y2 = W2 a1 + b2 (the wavefront's sample row broadcast from LDS against the lane's W2 row in registers,
//             64 fma per value) -> statistics 2 -> a2 = relu(bn2(y2)) stored.  Reads xyz, writes a2: nothing else.
//   backward: y1, a1, y2 RECOMPUTED from xyz and the saved statistics (no activation is kept), then
//             g2 = da2 gated -> BatchNorm-2 backward (two sums over the point) -> dy2;  dW2 += dy2^T a1 (accumulators of
//             the lane's W2 row stay in registers over all points of the workgroup);  da1 = dy2 W2 (dy2 rows broadcast
//             from LDS against the lane's W2 column);  BatchNorm-1 backward -> dy1;  dW1 += dy1^T xyz, db1, db2; the
//             BatchNorm weight / bias gradients of the point.  Reads xyz and da2, writes parameter gradients only.
// Arithmetic: fp32 vector fma in the reference's order of operations per element (products summed over the input channel
// in ascending order); BatchNorm as torch computes it (biased variance for the batch, unbiased for the running buffer).
#include "pzn_common.h"

namespace {

constexpr int ST_W = 8;     // wavefronts per workgroup: wavefront w holds the samples w, w + 8, ... (8 per lane: with 4 wavefronts
                            // and 16 samples per lane the backward needed more than the 512 registers a lane can have)
constexpr int ST_T = 64 * ST_W;
constexpr int ST_R = 64 / ST_W;    // samples per wavefront: B <= 64
constexpr int ST_C = 64;    // hidden = output width

__device__ __forceinline__ float st_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, PZN_WAVE);
  return v;
}
__device__ __forceinline__ float st_block_sum(float v, float* red) {
  v = st_wave_sum(v);
  __syncthreads();  // red may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int q = 0; q < ST_W; ++q) t += red[q];
  return t;
}

struct StemBn {
  const float* weight;   // [N] or NULL
  const float* bias;     // [N] or NULL
  float* running_mean;   // [N] or NULL
  float* running_var;    // [N] or NULL
  float momentum, eps;
};

struct StemFwdArgs {
  const float* xyz;      // [B, N, 3]
  const float *W1, *b1;  // [64, 3], [64]
  const float *W2, *b2;  // [64, 64], [64]
  StemBn bn1, bn2;
  int training, B, N;
  float* out;            // [B, N, 64]
  float *mean1, *invstd1, *mean2, *invstd2;   // [N] each: what normalised (batch or running statistics)
};

// What a point needs from global memory, fetched ONE POINT AHEAD (a workgroup walks its points one after the other: every load
// issued at the place of use was a full memory latency on the critical path, a dozen per point in the first version):
//   crd: lane 3 i + k (i < ST_R, k < 3) holds coordinate k of the wavefront's sample w + ST_W i;
//   par: lanes 0..7 hold the point's eight per-point scalars (which ones: the caller's table).
struct StemPoint {
  float crd, par;
};

__device__ __forceinline__ float st_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

__device__ __forceinline__ StemPoint stem_fetch(const float* __restrict__ xyz, int B, int N, int n, int w, int lane,
                                                const float* sp, float dflt) {
  StemPoint r;
  const int i = lane / 3, k = lane - 3 * i, b = w + ST_W * i;
  r.crd = (lane < 3 * ST_R && b < B) ? xyz[((size_t)b * N + n) * 3 + k] : 0.f;
  r.par = sp ? sp[n] : dflt;
  return r;
}

// statistics of the point's B x 64 values held as v[i] (sample w + ST_W i, channel lane); -> mean, invstd.  old_mean / old_var:
// the running buffers' values for this point (prefetched).
__device__ __forceinline__ void stem_stats(const float (&v)[ST_R], int B, int w, int n, const StemBn& bn, int training, float* red,
                                           float old_mean, float old_var, float& mean, float& invstd) {
  if (training) {
    const float cnt = (float)B * (float)ST_C;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ST_R; ++i) s += (w + ST_W * i < B) ? v[i] : 0.f;
    mean = st_block_sum(s, red) / cnt;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < ST_R; ++i) {
      const float d = v[i] - mean;
      q = (w + ST_W * i < B) ? fmaf(d, d, q) : q;
    }
    const float var = st_block_sum(q, red) / cnt;  // biased: what normalises the batch
    invstd = 1.0f / sqrtf(var + bn.eps);
    if (threadIdx.x == 0) {
      if (bn.running_mean) bn.running_mean[n] = (1.f - bn.momentum) * old_mean + bn.momentum * mean;
      if (bn.running_var) {
        const float unbiased = cnt > 1.f ? var * (cnt / (cnt - 1.f)) : var;
        bn.running_var[n] = (1.f - bn.momentum) * old_var + bn.momentum * unbiased;
      }
    }
  } else {
    mean = old_mean;
    invstd = 1.0f / sqrtf(old_var + bn.eps);
  }
}

// y2[i] = b2[lane] + sum_c W2[lane][c] a[b][c]: the sample's row of the LDS tile is a broadcast read, the W2 row sits in registers
__device__ __forceinline__ void stem_layer2(const float* __restrict__ tile, const float (&wrow)[ST_C], float bb, int w, int B,
                                            float (&y2)[ST_R]) {
#pragma unroll
  for (int i = 0; i < ST_R; ++i) {
    y2[i] = 0.f;
    if (ST_W * i >= B) continue;      // a whole round of the wavefronts beyond the batch (uniform)
    const float4* row = reinterpret_cast<const float4*>(tile + (w + ST_W * i) * ST_C);
    float acc = bb;
#pragma unroll
    for (int c4 = 0; c4 < ST_C / 4; ++c4) {
      const float4 a = row[c4];
      acc = fmaf(wrow[4 * c4], a.x, acc);
      acc = fmaf(wrow[4 * c4 + 1], a.y, acc);
      acc = fmaf(wrow[4 * c4 + 2], a.z, acc);
      acc = fmaf(wrow[4 * c4 + 3], a.w, acc);
      if ((c4 & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four row reads in flight, not all 16 x ST_R of the unrolled nest
    }
    y2[i] = acc;
  }
}

__global__ __launch_bounds__(ST_T) void stem_fwd_kernel(StemFwdArgs p) {
  __shared__ __attribute__((aligned(16))) float tile[64 * ST_C];
  __shared__ float red[ST_W];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int B = p.B, N = p.N;
  const float wx = p.W1[lane * 3], wy = p.W1[lane * 3 + 1], wz = p.W1[lane * 3 + 2], bb1 = p.b1[lane], bb2 = p.b2[lane];
  float wrow[ST_C];
#pragma unroll
  for (int c4 = 0; c4 < ST_C / 4; ++c4) {
    const float4 t = *reinterpret_cast<const float4*>(p.W2 + (size_t)lane * ST_C + 4 * c4);
    wrow[4 * c4] = t.x, wrow[4 * c4 + 1] = t.y, wrow[4 * c4 + 2] = t.z, wrow[4 * c4 + 3] = t.w;
  }
  // the per-point scalars of lanes 0..7: running mean / variance of bn1, of bn2, then weight / bias of bn1, of bn2
  const float* sp = lane == 0 ? p.bn1.running_mean : lane == 1 ? p.bn1.running_var : lane == 2 ? p.bn2.running_mean
                    : lane == 3 ? p.bn2.running_var : lane == 4 ? p.bn1.weight : lane == 5 ? p.bn1.bias
                    : lane == 6 ? p.bn2.weight : lane == 7 ? p.bn2.bias : nullptr;
  const float dflt = (lane == 1 || lane == 3 || lane == 4 || lane == 6) ? 1.f : 0.f;
  StemPoint nxt = stem_fetch(p.xyz, B, N, blockIdx.x, w, lane, sp, dflt);
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    const StemPoint cur = nxt;
    if (n + (int)gridDim.x < N) nxt = stem_fetch(p.xyz, B, N, n + gridDim.x, w, lane, sp, dflt);
    float y1[ST_R];
#pragma unroll
    for (int i = 0; i < ST_R; ++i)      // (((b + wx x) + wy y) + wz z)
      y1[i] = fmaf(wz, st_lane(cur.crd, 3 * i + 2), fmaf(wy, st_lane(cur.crd, 3 * i + 1), fmaf(wx, st_lane(cur.crd, 3 * i), bb1)));
    float mean, invstd;
    stem_stats(y1, B, w, n, p.bn1, p.training, red, st_lane(cur.par, 0), st_lane(cur.par, 1), mean, invstd);
    if (threadIdx.x == 0) p.mean1[n] = mean, p.invstd1[n] = invstd;
    const float g1 = st_lane(cur.par, 4), o1 = st_lane(cur.par, 5);
    __syncthreads();      // the previous point's tile reads are done
#pragma unroll
    for (int i = 0; i < ST_R; ++i) {
      const float t = fmaf((y1[i] - mean) * invstd, g1, o1);
      tile[(w + ST_W * i) * ST_C + lane] = t > 0.f ? t : 0.f;
    }
    __syncthreads();
    float y2[ST_R];
    stem_layer2(tile, wrow, bb2, w, B, y2);
    stem_stats(y2, B, w, n, p.bn2, p.training, red, st_lane(cur.par, 2), st_lane(cur.par, 3), mean, invstd);
    if (threadIdx.x == 0) p.mean2[n] = mean, p.invstd2[n] = invstd;
    const float g2 = st_lane(cur.par, 6), o2 = st_lane(cur.par, 7);
#pragma unroll
    for (int i = 0; i < ST_R; ++i) {
      const int b = w + ST_W * i;
      if (b < B) {
        const float t = fmaf((y2[i] - mean) * invstd, g2, o2);
        p.out[((size_t)b * N + n) * ST_C + lane] = t > 0.f ? t : 0.f;
      }
    }
  }
}

struct StemBwdArgs {
  const float* xyz;      // [B, N, 3]
  const float* dout;     // [B, N, 64]
  const float *W1, *b1, *W2, *b2;
  const float *bn1w, *bn1b, *bn2w, *bn2b;      // [N] or NULL
  const float *mean1, *invstd1, *mean2, *invstd2;
  int training, B, N;
  float *dW1, *db1, *dW2, *db2;                // [64,3], [64], [64,64], [64]: ADDED to
  float *dbn1w, *dbn1b, *dbn2w, *dbn2b;        // [N]: ADDED to (may be NULL)
  float* part;                                 // [gridDim.x][ST_PART]: the workgroups' sums of dW2 | db2 | db1 | dW1, for stem_reduce_kernel
};

// One workgroup's parameter-gradient sums: dW2 row-major [64][64], then db2[64], db1[64], dW1 [64][3].  (256 workgroups adding
// these with atomics put a million adds on 128 cache lines: ~0.4 ms, ten times the arithmetic of the kernel.)
// LDS row stride of W2 in the backward kernel: a lane reads ITS row 16 bytes at a time; rows 64 floats apart put all 64 lanes on the
// same four banks (a 16-way conflict on each of the 16 reads per point and wavefront — it was most of the kernel), 68 apart spread
// each group of 16 lanes over all 64 banks.  Column reads (lane = column) are conflict-free either way.
constexpr int ST_WS = ST_C + 4;
constexpr int ST_PART = ST_C * ST_C + 2 * ST_C + 3 * ST_C;

// Per-sample values of the point live in LDS tiles [sample][channel] and the loops over the wavefront's samples are ROLLED: with
// the samples in register arrays (as in the forward kernel) the unrolled nest of the three 64 x 64 x 64 products needed more
// registers than a lane has (512 at four wavefronts, 256 + 1.9 KB of spills at eight).
__global__ __launch_bounds__(ST_T) void stem_bwd_kernel(StemBwdArgs p) {
  __shared__ __attribute__((aligned(16))) float t0[64 * ST_C];      // a1, later the normalised y1
  __shared__ __attribute__((aligned(16))) float t1[64 * ST_C];      // gated dout, then dy2
  __shared__ __attribute__((aligned(16))) float t2[64 * ST_C];      // normalised y2, later gated da1
  __shared__ __attribute__((aligned(16))) float w2s[ST_C * ST_WS];  // W2 row-major, rows ST_WS apart (see there)
  __shared__ float xs[ST_W * ST_R * 3];                             // the point's coordinates, [wavefront][round][3]
  __shared__ float red[ST_W];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int B = p.B, N = p.N, Bp = (p.B + ST_W - 1) / ST_W * ST_W;      // samples, rounded up to whole rounds of the wavefronts
  const float wx = p.W1[lane * 3], wy = p.W1[lane * 3 + 1], wz = p.W1[lane * 3 + 2], bb1 = p.b1[lane], bb2 = p.b2[lane];
  for (int f = threadIdx.x; f < ST_C * ST_C / 4; f += ST_T)
    *reinterpret_cast<float4*>(w2s + (f / (ST_C / 4)) * ST_WS + 4 * (f % (ST_C / 4))) = reinterpret_cast<const float4*>(p.W2)[f];
  float dw2[ST_C];      // dW2[lane][:] over the points of this workgroup
#pragma unroll
  for (int c = 0; c < ST_C; ++c) dw2[c] = 0.f;
  float db2 = 0.f, db1 = 0.f, dwx = 0.f, dwy = 0.f, dwz = 0.f;
  const float cnt = (float)B * (float)ST_C;
  __syncthreads();
  // the per-point scalars of lanes 0..7: mean / invstd of layer 1, of layer 2, then weight / bias of bn1, of bn2
  const float* sp = lane == 0 ? p.mean1 : lane == 1 ? p.invstd1 : lane == 2 ? p.mean2 : lane == 3 ? p.invstd2 : lane == 4 ? p.bn1w
                    : lane == 5 ? p.bn1b : lane == 6 ? p.bn2w : lane == 7 ? p.bn2b : nullptr;
  const float dflt = (lane == 4 || lane == 6) ? 1.f : 0.f;
  StemPoint nxt = stem_fetch(p.xyz, B, N, blockIdx.x, w, lane, sp, dflt);
  float nd[ST_R];      // the output gradient of the NEXT point for the wavefront's samples
#pragma unroll
  for (int i = 0; i < ST_R; ++i) nd[i] = (w + ST_W * i < B) ? p.dout[((size_t)(w + ST_W * i) * N + blockIdx.x) * ST_C + lane] : 0.f;
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    const float m1 = st_lane(nxt.par, 0), is1 = st_lane(nxt.par, 1), m2 = st_lane(nxt.par, 2), is2 = st_lane(nxt.par, 3);
    const float g1 = st_lane(nxt.par, 4), o1 = st_lane(nxt.par, 5), g2 = st_lane(nxt.par, 6), o2 = st_lane(nxt.par, 7);
    // (W2's row and column of the lane are re-read from LDS for every point — 16 + 64 reads — behind an offset the optimiser
    //  cannot see through: hoisted out of the point loop they would be 128 registers held beside the 64 of dw2)
    int zoff = 0;
    asm volatile("" : "+v"(zoff));
    __syncthreads();      // the previous point's tiles are done with
    // ---- stage what was fetched (own elements / own rows only: no barrier), then fetch the next point's
    if (lane < 3 * ST_R) xs[w * 3 * ST_R + lane] = nxt.crd;      // [wavefront][round][3]
#pragma unroll
    for (int i = 0; i < ST_R; ++i)
      if (ST_W * i < Bp) t1[(w + ST_W * i) * ST_C + lane] = nd[i];
    if (n + (int)gridDim.x < N) {
      const int n2 = n + gridDim.x;
      nxt = stem_fetch(p.xyz, B, N, n2, w, lane, sp, dflt);
#pragma unroll
      for (int i = 0; i < ST_R; ++i) nd[i] = (w + ST_W * i < B) ? p.dout[((size_t)(w + ST_W * i) * N + n2) * ST_C + lane] : 0.f;
    }
    // ---- A: recompute layer 1: a1 -> t0
#pragma unroll 1
    for (int b = w, i = 0; b < Bp; b += ST_W, ++i) {
      const float* q = xs + (w * ST_R + i) * 3;
      const float y1 = fmaf(wz, q[2], fmaf(wy, q[1], fmaf(wx, q[0], bb1)));
      const float t = fmaf((y1 - m1) * is1, g1, o1);
      t0[b * ST_C + lane] = t > 0.f ? t : 0.f;
    }
    __syncthreads();
    // ---- B: y2 = W2 a1 + b2, its normalised value -> t2, the gated output gradient -> t1, the two sums of BatchNorm 2
    float sg = 0.f, sgx = 0.f;
    {
      float wrow[ST_C];      // W2[lane][:]
#pragma unroll
      for (int c4 = 0; c4 < ST_C / 4; ++c4) {
        const float4 t = *reinterpret_cast<const float4*>(w2s + zoff + lane * ST_WS + 4 * c4);
        wrow[4 * c4] = t.x, wrow[4 * c4 + 1] = t.y, wrow[4 * c4 + 2] = t.z, wrow[4 * c4 + 3] = t.w;
      }
#pragma unroll 2
      for (int b = w; b < Bp; b += ST_W) {
        const float4* row = reinterpret_cast<const float4*>(t0 + b * ST_C);
        float acc = bb2;
#pragma unroll
        for (int c4 = 0; c4 < ST_C / 4; ++c4) {
          const float4 a = row[c4];
          acc = fmaf(wrow[4 * c4], a.x, acc);
          acc = fmaf(wrow[4 * c4 + 1], a.y, acc);
          acc = fmaf(wrow[4 * c4 + 2], a.z, acc);
          acc = fmaf(wrow[4 * c4 + 3], a.w, acc);
        }
        const float xh = b < B ? (acc - m2) * is2 : 0.f;
        const float gi = t1[b * ST_C + lane];      // (zero beyond the batch)
        const float g = (b < B && fmaf(xh, g2, o2) > 0.f) ? gi : 0.f;      // ReLU gate recomputed
        t1[b * ST_C + lane] = g;
        t2[b * ST_C + lane] = xh;
        sg += g;
        sgx = fmaf(g, xh, sgx);
      }
    }
    sg = st_block_sum(sg, red);
    sgx = st_block_sum(sgx, red);
    if (threadIdx.x == 0) {
      if (p.dbn2w) atomicAdd(p.dbn2w + n, sgx);
      if (p.dbn2b) atomicAdd(p.dbn2b + n, sg);
    }
    // ---- C: dy2 -> t1 (own elements), db2, dW2[lane][c] += sum_b dy2[b][lane] a1[b][c]   (a1 rows broadcast from t0)
    {
      const float k = g2 * is2, a1_ = p.training ? sg / cnt : 0.f, a2_ = p.training ? sgx / cnt : 0.f;
#pragma unroll 2
      for (int b = w; b < Bp; b += ST_W) {
        const float d = b < B ? k * (t1[b * ST_C + lane] - a1_ - t2[b * ST_C + lane] * a2_) : 0.f;      // dy2[b][lane]
        t1[b * ST_C + lane] = d;
        db2 += d;
        const float4* row = reinterpret_cast<const float4*>(t0 + b * ST_C);
#pragma unroll
        for (int c4 = 0; c4 < ST_C / 4; ++c4) {
          const float4 a = row[c4];
          dw2[4 * c4] = fmaf(d, a.x, dw2[4 * c4]);
          dw2[4 * c4 + 1] = fmaf(d, a.y, dw2[4 * c4 + 1]);
          dw2[4 * c4 + 2] = fmaf(d, a.z, dw2[4 * c4 + 2]);
          dw2[4 * c4 + 3] = fmaf(d, a.w, dw2[4 * c4 + 3]);
        }
      }
    }
    __syncthreads();      // dy2 complete in t1; t0 (a1) and t2 (normalised y2) are free
    // ---- D: da1[b][lane] = sum_c' dy2[b][c'] W2[c'][lane], gated by layer 1's ReLU -> t2; normalised y1 -> t0; BatchNorm 1's sums
    sg = 0.f, sgx = 0.f;
    {
      float wcol[ST_C];      // W2[:][lane]
#pragma unroll
      for (int c = 0; c < ST_C; ++c) wcol[c] = w2s[zoff + c * ST_WS + lane];
#pragma unroll 2
      for (int b = w, i = 0; b < Bp; b += ST_W, ++i) {
        const float4* row = reinterpret_cast<const float4*>(t1 + b * ST_C);
        float acc = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < ST_C / 4; ++c4) {
          const float4 d = row[c4];
          acc = fmaf(d.x, wcol[4 * c4], acc);
          acc = fmaf(d.y, wcol[4 * c4 + 1], acc);
          acc = fmaf(d.z, wcol[4 * c4 + 2], acc);
          acc = fmaf(d.w, wcol[4 * c4 + 3], acc);
        }
        const float* q = xs + (w * ST_R + i) * 3;
        const float y1 = fmaf(wz, q[2], fmaf(wy, q[1], fmaf(wx, q[0], bb1)));
        const float xh = b < B ? (y1 - m1) * is1 : 0.f;
        const float g = (b < B && fmaf(xh, g1, o1) > 0.f) ? acc : 0.f;
        t2[b * ST_C + lane] = g;
        t0[b * ST_C + lane] = xh;
        sg += g;
        sgx = fmaf(g, xh, sgx);
      }
    }
    sg = st_block_sum(sg, red);
    sgx = st_block_sum(sgx, red);
    if (threadIdx.x == 0) {
      if (p.dbn1w) atomicAdd(p.dbn1w + n, sgx);
      if (p.dbn1b) atomicAdd(p.dbn1b + n, sg);
    }
    // ---- E: dy1, db1, dW1 += dy1^T xyz   (own elements of t2 / t0)
    {
      const float k = g1 * is1, a1_ = p.training ? sg / cnt : 0.f, a2_ = p.training ? sgx / cnt : 0.f;
#pragma unroll 1
      for (int b = w, i = 0; b < B; b += ST_W, ++i) {
        const float d = k * (t2[b * ST_C + lane] - a1_ - t0[b * ST_C + lane] * a2_);      // dy1[b][lane]
        const float* q = xs + (w * ST_R + i) * 3;
        db1 += d;
        dwx = fmaf(d, q[0], dwx), dwy = fmaf(d, q[1], dwy), dwz = fmaf(d, q[2], dwz);
      }
    }
  }
  // ---- the workgroup's parameter gradients: the wavefronts meet in LDS (W2 is not needed any more), one set of atomics
  __syncthreads();
  float* part = p.part + (size_t)blockIdx.x * ST_PART;
  float* acc = w2s;      // [ST_W][CB][64]: CB columns of dW2 at a time
  constexpr int CB = ST_C * ST_C / (ST_W * 64);
  // dW2: wavefront w holds partial dW2[lane][:] of its samples: sum the wavefronts column block by column block
#pragma unroll
  for (int c0 = 0; c0 < ST_C; c0 += CB) {      // (unrolled: dw2 is a register array)
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[(w * CB + c) * ST_C + lane] = dw2[c0 + c];      // [w][c][lane]
    __syncthreads();
    for (int f = threadIdx.x; f < CB * ST_C; f += ST_T) {
      const int c = f / ST_C, l = f % ST_C;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < ST_W; ++q) t += acc[(q * CB + c) * ST_C + l];
      part[l * ST_C + c0 + c] = t;
    }
  }
  __syncthreads();
  float* r5 = t0;      // [5][ST_W][64]: db2, db1, dwx, dwy, dwz
  r5[(0 * ST_W + w) * 64 + lane] = db2;
  r5[(1 * ST_W + w) * 64 + lane] = db1;
  r5[(2 * ST_W + w) * 64 + lane] = dwx;
  r5[(3 * ST_W + w) * 64 + lane] = dwy;
  r5[(4 * ST_W + w) * 64 + lane] = dwz;
  __syncthreads();
  for (int f = threadIdx.x; f < 5 * 64; f += ST_T) {
    const int q = f / 64, l = f % 64;
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < ST_W; ++u) t += r5[(q * ST_W + u) * 64 + l];
    if (q < 2)
      part[ST_C * ST_C + q * ST_C + l] = t;
    else
      part[ST_C * ST_C + 2 * ST_C + l * 3 + (q - 2)] = t;
  }
}

// dW2 | db2 | db1 | dW1 += the sum of the workgroups' parts.  64 outputs per workgroup, sixteen wavefronts each summing a sixteenth of
// the parts (coalesced across the 64 outputs), met in LDS.
__global__ __launch_bounds__(1024) void stem_reduce_kernel(const float* __restrict__ part, int nparts, float* dW2, float* db2, float* db1,
                                                           float* dW1) {
  __shared__ float red[16][64];
  const int l = threadIdx.x & 63, g = threadIdx.x >> 6, o = blockIdx.x * 64 + l;
  float t = 0.f;
#pragma unroll 4
  for (int q = g; q < nparts; q += 16) t += part[(size_t)q * ST_PART + o];
  red[g][l] = t;
  __syncthreads();
  if (g == 0) {
    t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][l];
    float* dst = o < ST_C * ST_C ? dW2 + o : o < ST_C * ST_C + ST_C ? db2 + (o - ST_C * ST_C)
                 : o < ST_C * ST_C + 2 * ST_C ? db1 + (o - ST_C * ST_C - ST_C) : dW1 + (o - ST_C * ST_C - 2 * ST_C);
    *dst += t;
  }
}

int stem_grid(int N) {
  int g = N < 256 ? N : 256;      // one workgroup of 8 wavefronts per CU, each walking N / 256 points with its W2 row / gradient in registers
  return g < 1 ? 1 : g;
}

}  // namespace

// relu(bn2(mlp2(relu(bn1(mlp1(xyz)))))) for xyz[B, N, 3], mlp1 = (W1[64,3], b1[64]), mlp2 = (W2[64,64], b2[64]), bn1 / bn2 =
// BatchNorm1d(N) over the point axis (weight / bias / running buffers [N], any of them NULL as the module has them).
// training != 0: batch statistics, running buffers updated in place as torch does.  out[B, N, 64]; mean1 / invstd1 / mean2 /
// invstd2 [N] feed the backward.  B <= 64.
PZN_EXPORT int pzn_stem_fwd_f32(const float* xyz, const float* W1, const float* b1, const float* bn1_weight, const float* bn1_bias,
                                float* bn1_running_mean, float* bn1_running_var, float bn1_momentum, float bn1_eps,
                                const float* W2, const float* b2, const float* bn2_weight, const float* bn2_bias,
                                float* bn2_running_mean, float* bn2_running_var, float bn2_momentum, float bn2_eps, int training,
                                int B, int N, float* out, float* mean1, float* invstd1, float* mean2, float* invstd2,
                                pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && W1 && b1 && W2 && b2 && out && mean1 && invstd1 && mean2 && invstd2 && B > 0 && N > 0);
  PZN_CHECK_ARG(training || (bn1_running_mean && bn1_running_var && bn2_running_mean && bn2_running_var));
  if (B > 64 || (reinterpret_cast<uintptr_t>(W2) & 15)) return PZN_EUNSUPPORTED;
  StemFwdArgs a{xyz, W1, b1, W2, b2,
                {bn1_weight, bn1_bias, bn1_running_mean, bn1_running_var, bn1_momentum, bn1_eps},
                {bn2_weight, bn2_bias, bn2_running_mean, bn2_running_var, bn2_momentum, bn2_eps},
                training, B, N, out, mean1, invstd1, mean2, invstd2};
  hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)stem_grid(N)), dim3(ST_T), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT size_t pzn_stem_bwd_workspace_bytes(int N) { return (size_t)stem_grid(N) * ST_PART * sizeof(float); }

// Backward of pzn_stem_fwd_f32 from dout[B, N, 64]: dW1[64,3], db1[64], dW2[64,64], db2[64] and the BatchNorm weight / bias
// gradients [N] are ADDED to (the BatchNorm ones may be NULL); nothing is returned for xyz.  Activations are recomputed from xyz
// and the saved statistics.  workspace: pzn_stem_bwd_workspace_bytes(N) bytes, 16-byte aligned (need not be cleared).
PZN_EXPORT int pzn_stem_bwd_f32(const float* xyz, const float* dout, const float* W1, const float* b1, const float* W2,
                                const float* b2, const float* bn1_weight, const float* bn1_bias, const float* bn2_weight,
                                const float* bn2_bias, const float* mean1, const float* invstd1, const float* mean2,
                                const float* invstd2, int training, int B, int N, float* dW1, float* db1, float* dW2, float* db2,
                                float* dbn1_weight, float* dbn1_bias, float* dbn2_weight, float* dbn2_bias, void* workspace,
                                pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && dout && W1 && b1 && W2 && b2 && mean1 && invstd1 && mean2 && invstd2 && dW1 && db1 && dW2 && db2 && B > 0 &&
                N > 0 && workspace && !(reinterpret_cast<uintptr_t>(workspace) & 15));
  if (B > 64 || (reinterpret_cast<uintptr_t>(W2) & 15)) return PZN_EUNSUPPORTED;
  StemBwdArgs a{xyz, dout, W1, b1, W2, b2, bn1_weight, bn1_bias, bn2_weight, bn2_bias, mean1, invstd1, mean2, invstd2, training, B, N,
                dW1, db1, dW2, db2, dbn1_weight, dbn1_bias, dbn2_weight, dbn2_bias, static_cast<float*>(workspace)};
  const int grid = stem_grid(N);
  hipLaunchKernelGGL(stem_bwd_kernel, dim3((unsigned)grid), dim3(ST_T), 0, pzn_hip_stream(stream), a);
  static_assert(ST_PART % 64 == 0, "stem_reduce_kernel: 64 outputs per workgroup");
  hipLaunchKernelGGL(stem_reduce_kernel, dim3(ST_PART / 64), dim3(1024), 0, pzn_hip_stream(stream), a.part, grid, dW2, db2, db1, dW1);
  PZN_RETURN_LAUNCH_STATUS();
}
