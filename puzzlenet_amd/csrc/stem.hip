// stem.hip — the per-point stem of the encoder in ONE launch each way (round 5).
//
// model5_b.py:447-448:   x_feature = relu(bn2(mlp2(relu(bn1(mlp1(xyz))))))
// with mlp1 = Linear(3, 64), mlp2 = Linear(64, 64) and bn1, bn2 = BatchNorm1d(num_points) applied to [B, N, 64] tensors:
// the BatchNorm "channel" is the POINT index n, its statistics run over the B x 64 values of the point (model5_b.py:424).
// Everything a point needs is therefore local to the point: its B coordinates (B x 12 bytes), the two small weight
// matrices, its own BatchNorm parameters.  Rounds 1-4 ran this as 2 linear + 2 BatchNorm launches forward and 5 launches
// backward per encoder, ~9 passes over 33.5 MB tensors.  Here one workgroup (8 wavefronts) = one point at a time, persistent over
// the points of its share; the point's [B x 64] tiles live in LDS ([sample][channel], rows ST_LD floats apart):
//   forward : y1 = W1 xyz + b1 (3 fma per value; wavefront w holds samples w, w + 8, ..., lane = channel) -> statistics 1 ->
//             a1 = relu(bn1(y1)) -> tile -> y2 = a1 W2^T + b2 on the matrix cores in EXACT fp32 (v_mfma_f32_16x16x4_f32: a
//             k-ordered fmaf chain; wavefront = one 16-channel column block x the 16-sample row blocks) -> statistics 2 ->
//             a2 = relu(bn2(y2)) stored from the accumulator layout.  Reads xyz, writes a2: nothing else.
//   backward: y1, a1, y2 RECOMPUTED from xyz and the saved statistics (no activation is kept), then
//             g2 = da2 gated -> BatchNorm-2 backward (two sums over the point) -> dy2 -> tile;  dW2 += dy2^T a1 (16 output
//             tiles, two per wavefront, accumulators in registers over all points of the workgroup);  da1 = dy2 W2;
//             BatchNorm-1 backward -> dy1;  dW1 += dy1^T xyz, db1, db2; the BatchNorm weight / bias gradients of the point.
//             Reads xyz and da2, writes parameter gradients only (per-workgroup parts, summed by stem_reduce_kernel).
// The first version of this file fed fp32 vector FMAs from LDS broadcasts (one 16-byte broadcast read per four FMAs): the LDS
// pipe bounded it at 24 + 55 us for B = 32, N = 1024 and it lost to the four launches it replaces; the matrix-core form reads
// each tile element once per wavefront that needs it.
// Arithmetic: fp32 throughout; the products of a row are summed over the input channel in the order 0, 16, 32, 48, 1, 17, ...
// (the MFMA's k order with lane group q holding channels 16 q ... 16 q + 15); BatchNorm as torch computes it (biased variance
// for the batch, unbiased for the running buffer).
#include "pzn_common.h"

namespace {

typedef float st_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float st_lane0(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

constexpr int ST_W = 8;     // wavefronts per workgroup
constexpr int ST_T = 64 * ST_W;
constexpr int ST_R = 64 / ST_W;    // samples per wavefront in the per-sample phases: B <= 64
constexpr int ST_C = 64;    // hidden = output width
// Row stride of the LDS tiles: an MFMA A-fragment is read as lane (row m = lane & 15, k-group q = lane >> 4) taking 16
// consecutive floats at [row m][16 q]: 16-byte reads whose 16 lanes of a pass sit 68 floats apart = 4 banks apart, all 64 banks.
constexpr int ST_LD = ST_C + 4;

// sum over the wavefront, the same value in every lane: DPP / swizzle steps inside the halves, two v_readlane for the halves
// (no ds_bpermute round trips: the sums sit between barriers on the per-point critical path)
__device__ __forceinline__ float st_x(float v, uint32_t bits) { return v + __builtin_bit_cast(float, bits); }
__device__ __forceinline__ float st_wave_sum(float v) {
  v = st_x(v, pzn::xor_lane<1>(__builtin_bit_cast(uint32_t, v)));
  v = st_x(v, pzn::xor_lane<2>(__builtin_bit_cast(uint32_t, v)));
  v = st_x(v, pzn::xor_lane<4>(__builtin_bit_cast(uint32_t, v)));
  v = st_x(v, pzn::xor_lane<8>(__builtin_bit_cast(uint32_t, v)));
  v = st_x(v, pzn::xor_lane<16>(__builtin_bit_cast(uint32_t, v)));
  return st_lane0(v, 0) + st_lane0(v, 32);
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
// two sums over a workgroup of W wavefronts behind one pair of barriers (red: 2 * W floats)
template <int W>
__device__ __forceinline__ void st_block_sum2(float& a, float& b, float* red) {
  a = st_wave_sum(a), b = st_wave_sum(b);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a, red[W + (threadIdx.x >> 6)] = b;
  __syncthreads();
  a = b = 0.f;
#pragma unroll
  for (int q = 0; q < W; ++q) a += red[q], b += red[W + q];
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
// issued at the place of use is a full memory latency on the critical path):
//   crd: lane 3 i + k (i < 64 / W, k < 3) holds coordinate k of the wavefront's sample w + W i (W wavefronts per workgroup);
//   par: lanes 0..7 hold the point's eight per-point scalars (which ones: the caller's table).
struct StemPoint {
  float crd, par;
};

__device__ __forceinline__ float st_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

template <int W>
__device__ __forceinline__ StemPoint stem_fetch(const float* __restrict__ xyz, int B, int N, int n, int w, int lane,
                                                const float* sp, float dflt) {
  StemPoint r;
  const int i = lane / 3, k = lane - 3 * i, b = w + W * i;
  r.crd = (lane < 3 * (64 / W) && b < B) ? xyz[((size_t)b * N + n) * 3 + k] : 0.f;
  r.par = sp ? sp[n] : dflt;
  return r;
}

// statistics of the point's B x 64 values, of which this lane holds v[i] (bit i of `valid`: v[i] is a value of the batch; nw =
// how many values the WAVEFRONT holds); -> mean, invstd.  Each wavefront takes mean and sum of squared deviations of its own
// values (two passes, no barrier), the eight (count, mean, M2) triples meet in LDS once and every thread combines them
// (M2 = sum of M2_w + n_w (mean_w - mean)^2: the two-pass result to rounding, one barrier pair instead of two).
// old_mean / old_var: the running buffers' values for this point (prefetched).  red: 3 * ST_W floats.
template <int NV>
__device__ __forceinline__ void stem_stats(const float (&v)[NV], uint32_t valid, float nw, int B, int n, const StemBn& bn,
                                           int training, float* red, float old_mean, float old_var, float& mean, float& invstd) {
  if (training) {
    const float cnt = (float)B * (float)ST_C;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += ((valid >> i) & 1u) ? v[i] : 0.f;
    const float mw = nw > 0.f ? st_wave_sum(s) / nw : 0.f;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float d = v[i] - mw;
      q = ((valid >> i) & 1u) ? fmaf(d, d, q) : q;
    }
    q = st_wave_sum(q);
    const int w = threadIdx.x >> 6;
    __syncthreads();      // red may still be read from the previous exchange
    if ((threadIdx.x & 63) == 0) red[w] = nw, red[ST_W + w] = mw, red[2 * ST_W + w] = q;
    __syncthreads();
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < ST_W; ++k) sm = fmaf(red[k], red[ST_W + k], sm);
    mean = sm / cnt;
    float m2 = 0.f;
#pragma unroll
    for (int k = 0; k < ST_W; ++k) {
      const float d = red[ST_W + k] - mean;
      m2 += fmaf(red[k] * d, d, red[2 * ST_W + k]);
    }
    const float var = m2 / cnt;  // biased: what normalises the batch
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

// 16 consecutive floats of an LDS tile row (an MFMA operand fragment of this lane: k = 16 q + s, s = 0..15)
__device__ __forceinline__ void st_frag(const float* __restrict__ p, float (&f)[16]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 t = *reinterpret_cast<const float4*>(p + 4 * j);
    f[4 * j] = t.x, f[4 * j + 1] = t.y, f[4 * j + 2] = t.z, f[4 * j + 3] = t.w;
  }
}

// The matrix-core geometry shared by the two kernels.  v_mfma_f32_16x16x4_f32: A[m = lane & 15][k = lane >> 4],
// B[k = lane >> 4][n = lane & 15], D[m = 4 (lane >> 4) + r][n = lane & 15] in register r.  A product over 64 channels takes 16
// instructions; at instruction s lane group q = lane >> 4 contributes channel 16 q + s (both operands: any order of k is the
// same sum), so a lane's fragment is 16 consecutive floats.  Wavefront w owns the 16-wide column block ct = w & 3 of every
// [sample][channel] product and the 16-sample row blocks bt = (w >> 2), (w >> 2) + 2 (B <= 64: at most 4 row blocks).

// (128 registers: two workgroups per CU - each marches through its barriers in lock step, the other one fills the gaps)
__global__ __launch_bounds__(ST_T, 4) void stem_fwd_kernel(StemFwdArgs p) {
  __shared__ __attribute__((aligned(16))) float t0[64 * ST_LD];      // a1 [sample][channel]
  __shared__ float red[3 * ST_W];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), q = lane >> 4, col = lane & 15;
  const int B = p.B, N = p.N, nbt = (B + 15) >> 4;
  const int ct = w & 3, bt0 = w >> 2;
  const float wx = p.W1[lane * 3], wy = p.W1[lane * 3 + 1], wz = p.W1[lane * 3 + 2], bb1 = p.b1[lane];
  float bw[16];      // B operand of y2 = a1 W2^T: W2[16 ct + col][16 q + s]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 t = *reinterpret_cast<const float4*>(p.W2 + (size_t)(16 * ct + col) * ST_C + 16 * q + 4 * j);
    bw[4 * j] = t.x, bw[4 * j + 1] = t.y, bw[4 * j + 2] = t.z, bw[4 * j + 3] = t.w;
  }
  const float bias2 = p.b2[16 * ct + col];
  for (int i = B * ST_LD + threadIdx.x; i < 16 * nbt * ST_LD; i += ST_T) t0[i] = 0.f;      // the rows past the batch: zero operands
  // the per-point scalars of lanes 0..7: running mean / variance of bn1, of bn2, then weight / bias of bn1, of bn2
  const float* sp = lane == 0 ? p.bn1.running_mean : lane == 1 ? p.bn1.running_var : lane == 2 ? p.bn2.running_mean
                    : lane == 3 ? p.bn2.running_var : lane == 4 ? p.bn1.weight : lane == 5 ? p.bn1.bias
                    : lane == 6 ? p.bn2.weight : lane == 7 ? p.bn2.bias : nullptr;
  const float dflt = (lane == 1 || lane == 3 || lane == 4 || lane == 6) ? 1.f : 0.f;
  uint32_t valid1 = 0, valid2 = 0;      // which of the lane's values exist: per-sample phase, accumulator layout
#pragma unroll
  for (int i = 0; i < ST_R; ++i) valid1 |= (w + ST_W * i < B ? 1u : 0u) << i;
#pragma unroll
  for (int i = 0; i < 8; ++i) valid2 |= (16 * (bt0 + 2 * (i >> 2)) + 4 * q + (i & 3) < B ? 1u : 0u) << i;
  // values of the batch held by this wavefront in the two layouts (for the statistics)
  const float nw1 = 64.f * (float)__popc(valid1);
  float nw2 = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rows = B - 16 * (bt0 + 2 * j);
    nw2 += 16.f * (float)(rows < 0 ? 0 : rows > 16 ? 16 : rows);
  }
  StemPoint nxt = stem_fetch<ST_W>(p.xyz, B, N, blockIdx.x, w, lane, sp, dflt);
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    const StemPoint cur = nxt;
    if (n + (int)gridDim.x < N) nxt = stem_fetch<ST_W>(p.xyz, B, N, n + gridDim.x, w, lane, sp, dflt);
    float y1[ST_R];
#pragma unroll
    for (int i = 0; i < ST_R; ++i)      // (((b + wx x) + wy y) + wz z)
      y1[i] = fmaf(wz, st_lane(cur.crd, 3 * i + 2), fmaf(wy, st_lane(cur.crd, 3 * i + 1), fmaf(wx, st_lane(cur.crd, 3 * i), bb1)));
    float mean, invstd;
    stem_stats(y1, valid1, nw1, B, n, p.bn1, p.training, red, st_lane(cur.par, 0), st_lane(cur.par, 1), mean, invstd);
    if (threadIdx.x == 0) p.mean1[n] = mean, p.invstd1[n] = invstd;
    const float g1 = st_lane(cur.par, 4), o1 = st_lane(cur.par, 5);
    __syncthreads();      // the previous point's tile reads are done
#pragma unroll
    for (int i = 0; i < ST_R; ++i) {
      const float t = fmaf((y1[i] - mean) * invstd, g1, o1);
      if ((valid1 >> i) & 1u) t0[(w + ST_W * i) * ST_LD + lane] = t > 0.f ? t : 0.f;
    }
    __syncthreads();
    // ---- y2 = a1 W2^T + b2 for the wavefront's (up to) two 16 x 16 blocks
    float y2[8];
    {
      st_f4 acc[2];
      float af[2][16];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[j] = st_f4{bias2, bias2, bias2, bias2};
        if (bt0 + 2 * j < nbt) st_frag(t0 + (16 * (bt0 + 2 * j) + col) * ST_LD + 16 * q, af[j]);
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (bt0 + 2 * j < nbt) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][s], bw[s], acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) y2[i] = acc[i >> 2][i & 3];
    }
    stem_stats(y2, valid2, nw2, B, n, p.bn2, p.training, red, st_lane(cur.par, 2), st_lane(cur.par, 3), mean, invstd);
    if (threadIdx.x == 0) p.mean2[n] = mean, p.invstd2[n] = invstd;
    const float g2 = st_lane(cur.par, 6), o2 = st_lane(cur.par, 7);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if ((valid2 >> i) & 1u) {
        const int b = 16 * (bt0 + 2 * (i >> 2)) + 4 * q + (i & 3);
        const float t = fmaf((y2[i] - mean) * invstd, g2, o2);
        p.out[((size_t)b * N + n) * ST_C + 16 * ct + col] = t > 0.f ? t : 0.f;
      }
    }
  }
}

struct StemBwdArgs {
  const float* xyz;      // [B, N, 3]
  const float* dout;     // [B, N, 64]
  const float* dout2;    // a second gradient of the same output (another consumer's) or NULL: added while loading
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
constexpr int ST_PART = ST_C * ST_C + 2 * ST_C + 3 * ST_C;

// Four wavefronts per workgroup here, two workgroups per CU: the same eight wavefronts a CU holds at this register count, but as two
// barrier domains (a workgroup marches through seven barriers per point in lock step; the other one fills its gaps: the
// forward gained a third that way).  Wavefront w owns column block ct = w of every product and all (up to four) 16-sample row
// blocks, two at a time; of dW2 it owns the four 16 x 16 blocks of its 16 columns.
constexpr int SB_W = 4, SB_T = 64 * SB_W, SB_R = 64 / SB_W;

__global__ __launch_bounds__(SB_T, 2) void stem_bwd_kernel(StemBwdArgs p) {
  __shared__ __attribute__((aligned(16))) float t0[64 * ST_LD];      // a1 [sample][channel]
  __shared__ __attribute__((aligned(16))) float t1[64 * ST_LD];      // dy2 [sample][channel]
  __shared__ __attribute__((aligned(16))) float w2s[64 * ST_LD];     // W2 [output channel][input channel]
  __shared__ __attribute__((aligned(16))) float w2t[64 * ST_LD];     // W2 transposed: [input channel][output channel]
  __shared__ float xs[64 * 3];                                        // the point's coordinates [sample][3]
  __shared__ float red[2 * SB_W];
  __shared__ float colsum[5 * 4 * ST_C];                              // end of kernel: db2 | db1 | dW1 x, y, z per lane group and channel
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), q = lane >> 4, col = lane & 15;
  const int B = p.B, N = p.N, nbt = (B + 15) >> 4;
  const int ch = 16 * w + col;      // the channel of this lane in the accumulator layout
  // per-sample phase (lane = channel) and accumulator-layout (channel ch) copies of layer 1
  const float wx = p.W1[lane * 3], wy = p.W1[lane * 3 + 1], wz = p.W1[lane * 3 + 2], bb1 = p.b1[lane];
  const float wxc = p.W1[ch * 3], wyc = p.W1[ch * 3 + 1], wzc = p.W1[ch * 3 + 2], bb1c = p.b1[ch];
  // The B operands - y2 = a1 W2^T: W2[ch][16 q + s]; da1 = dy2 W2: W2[16 q + s][ch] - are re-read from two LDS copies of W2
  // at every point (four 16-byte reads each): held in registers beside the rest they pushed the kernel past the 256
  // registers that let two workgroups share a CU.
  for (int i = threadIdx.x; i < ST_C * ST_C; i += SB_T) {
    const float v = p.W2[i];
    w2s[(i / ST_C) * ST_LD + i % ST_C] = v;
    w2t[(i % ST_C) * ST_LD + i / ST_C] = v;
  }
  const float bias2 = p.b2[ch];
  for (int i = B * ST_LD + threadIdx.x; i < 16 * nbt * ST_LD; i += SB_T) t0[i] = 0.f, t1[i] = 0.f;
  st_f4 dw2[4];      // dW2[16 m + 4 q + r][ch], m = 0..3, over the points of this workgroup
#pragma unroll
  for (int m = 0; m < 4; ++m) dw2[m] = st_f4{0.f, 0.f, 0.f, 0.f};
  float db2 = 0.f, db1 = 0.f, dwx = 0.f, dwy = 0.f, dwz = 0.f;      // this lane's part of channel ch
  const float cnt = (float)B * (float)ST_C;
  // the per-point scalars of lanes 0..7: mean / invstd of layer 1, of layer 2, then weight / bias of bn1, of bn2
  const float* sp = lane == 0 ? p.mean1 : lane == 1 ? p.invstd1 : lane == 2 ? p.mean2 : lane == 3 ? p.invstd2 : lane == 4 ? p.bn1w
                    : lane == 5 ? p.bn1b : lane == 6 ? p.bn2w : lane == 7 ? p.bn2b : nullptr;
  const float dflt = (lane == 4 || lane == 6) ? 1.f : 0.f;
  uint32_t valid1 = 0, valid2 = 0;      // per-sample phase: sample w + 4 i; accumulator layout: value i = row block i >> 2, register i & 3
#pragma unroll
  for (int i = 0; i < SB_R; ++i) valid1 |= (w + SB_W * i < B ? 1u : 0u) << i;
#pragma unroll
  for (int i = 0; i < 16; ++i) valid2 |= (16 * (i >> 2) + 4 * q + (i & 3) < B ? 1u : 0u) << i;
  auto fetch_dout = [&](int n, float (&d)[16]) {      // the output gradient in the accumulator layout (64-byte row segments)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int b = 16 * (i >> 2) + 4 * q + (i & 3);
      d[i] = ((valid2 >> i) & 1u) ? p.dout[((size_t)b * N + n) * ST_C + ch] : 0.f;
    }
    if (p.dout2) {      // (uniform)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int b = 16 * (i >> 2) + 4 * q + (i & 3);
        if ((valid2 >> i) & 1u) d[i] += p.dout2[((size_t)b * N + n) * ST_C + ch];
      }
    }
  };
  StemPoint nxt = stem_fetch<SB_W>(p.xyz, B, N, blockIdx.x, w, lane, sp, dflt);
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    const StemPoint cur = nxt;
    // the output gradient of THIS point (then the gated one, then layer 1's): first used behind phase A, a barrier and the 32
    // matrix instructions of phase B, which is where its latency goes (a copy fetched one point ahead cost 16 registers more)
    float g[16];
    fetch_dout(n, g);
    if (n + (int)gridDim.x < N) nxt = stem_fetch<SB_W>(p.xyz, B, N, n + gridDim.x, w, lane, sp, dflt);
    const float m1 = st_lane(cur.par, 0), is1 = st_lane(cur.par, 1), m2 = st_lane(cur.par, 2), is2 = st_lane(cur.par, 3);
    const float g1 = st_lane(cur.par, 4), o1 = st_lane(cur.par, 5), g2 = st_lane(cur.par, 6), o2 = st_lane(cur.par, 7);
    __syncthreads();      // the previous point's tiles and coordinates are done with
    // ---- A: the coordinates -> xs, layer 1 recomputed: a1 -> t0
    if (lane < 3 * SB_R) xs[(w + SB_W * (lane / 3)) * 3 + lane % 3] = cur.crd;      // (zero past the batch)
#pragma unroll
    for (int i = 0; i < SB_R; ++i) {
      const float y1 = fmaf(wz, st_lane(cur.crd, 3 * i + 2), fmaf(wy, st_lane(cur.crd, 3 * i + 1), fmaf(wx, st_lane(cur.crd, 3 * i), bb1)));
      const float t = fmaf((y1 - m1) * is1, g1, o1);
      if ((valid1 >> i) & 1u) t0[(w + SB_W * i) * ST_LD + lane] = t > 0.f ? t : 0.f;
    }
    __syncthreads();
    // ---- B: y2 = a1 W2^T + b2 recomputed; BatchNorm 2 backward: dy2 -> t1, db2
    float xh[16];
    float sg = 0.f, sgx = 0.f;
    int zoff = 0;
    asm volatile("" : "+v"(zoff));      // (keeps the operand loads inside the point loop)
    float bw[16];
    st_frag(w2s + zoff + ch * ST_LD + 16 * q, bw);
#pragma unroll
    for (int h = 0; h < 2; ++h) {      // row blocks 2 h, 2 h + 1
      st_f4 acc[2];
      float af[2][16];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[j] = st_f4{bias2, bias2, bias2, bias2};
        if (2 * h + j < nbt) st_frag(t0 + (16 * (2 * h + j) + col) * ST_LD + 16 * q, af[j]);
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (2 * h + j < nbt) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][s], bw[s], acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = 8 * h + k;
        const bool ok = (valid2 >> i) & 1u;
        xh[i] = ok ? (acc[k >> 2][k & 3] - m2) * is2 : 0.f;
        g[i] = (ok && fmaf(xh[i], g2, o2) > 0.f) ? g[i] : 0.f;      // ReLU gate recomputed
        sg += g[i];
        sgx = fmaf(g[i], xh[i], sgx);
      }
    }
    st_block_sum2<SB_W>(sg, sgx, red);
    if (threadIdx.x == 0) {
      if (p.dbn2w) atomicAdd(p.dbn2w + n, sgx);
      if (p.dbn2b) atomicAdd(p.dbn2b + n, sg);
    }
    {
      const float k = g2 * is2, a1_ = p.training ? sg / cnt : 0.f, a2_ = p.training ? sgx / cnt : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float d = ((valid2 >> i) & 1u) ? k * (g[i] - a1_ - xh[i] * a2_) : 0.f;      // dy2[b][ch]
        db2 += d;
        if ((i >> 2) < nbt) t1[(16 * (i >> 2) + 4 * q + (i & 3)) * ST_LD + ch] = d;
      }
    }
    __syncthreads();      // dy2 complete in t1
    // ---- C: dW2[c'][c] += sum_b dy2[b][c'] a1[b][c]: k = the samples, four per instruction (sample 4 s + q at instruction s);
    //         the wavefront's four output blocks (rows 16 m .. of columns 16 w ..) share the a1 operand
    for (int s = 0; s < 4 * nbt; ++s) {
      const int b = 4 * s + q;
      const float a1v = t0[b * ST_LD + ch];
#pragma unroll
      for (int m = 0; m < 4; ++m) dw2[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(t1[b * ST_LD + 16 * m + col], a1v, dw2[m], 0, 0, 0);
    }
    // ---- D: da1 = dy2 W2 gated by layer 1's ReLU; BatchNorm 1 backward: dy1 -> db1, dW1
    sg = 0.f, sgx = 0.f;
    float bt[16];
    st_frag(w2t + zoff + ch * ST_LD + 16 * q, bt);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      st_f4 acc[2];
      float af[2][16];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[j] = st_f4{0.f, 0.f, 0.f, 0.f};
        if (2 * h + j < nbt) st_frag(t1 + (16 * (2 * h + j) + col) * ST_LD + 16 * q, af[j]);
      }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (2 * h + j < nbt) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][s], bt[s], acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int i = 8 * h + k;
        const bool ok = (valid2 >> i) & 1u;
        const float* c = xs + (16 * (i >> 2) + 4 * q + (i & 3)) * 3;
        const float y1 = ok ? fmaf(wzc, c[2], fmaf(wyc, c[1], fmaf(wxc, c[0], bb1c))) : 0.f;
        xh[i] = ok ? (y1 - m1) * is1 : 0.f;
        g[i] = (ok && fmaf(xh[i], g1, o1) > 0.f) ? acc[k >> 2][k & 3] : 0.f;
        sg += g[i];
        sgx = fmaf(g[i], xh[i], sgx);
      }
    }
    st_block_sum2<SB_W>(sg, sgx, red);
    if (threadIdx.x == 0) {
      if (p.dbn1w) atomicAdd(p.dbn1w + n, sgx);
      if (p.dbn1b) atomicAdd(p.dbn1b + n, sg);
    }
    {
      const float k = g1 * is1, a1_ = p.training ? sg / cnt : 0.f, a2_ = p.training ? sgx / cnt : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if ((valid2 >> i) & 1u) {
          const float* c = xs + (16 * (i >> 2) + 4 * q + (i & 3)) * 3;
          const float d = k * (g[i] - a1_ - xh[i] * a2_);      // dy1[b][ch]
          db1 += d;
          dwx = fmaf(d, c[0], dwx), dwy = fmaf(d, c[1], dwy), dwz = fmaf(d, c[2], dwz);
        }
      }
    }
  }
  // ---- the workgroup's parameter gradients -> its part (summed over the workgroups by stem_reduce_kernel)
  float* part = p.part + (size_t)blockIdx.x * ST_PART;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(16 * m + 4 * q + r) * ST_C + ch] = dw2[m][r];      // every block has one owner
  // four lanes per channel (the lane groups): met in LDS and summed in a fixed order (no atomics: the same bits every run)
  colsum[(0 * 4 + q) * ST_C + ch] = db2;
  colsum[(1 * 4 + q) * ST_C + ch] = db1;
  colsum[(2 * 4 + q) * ST_C + ch] = dwx;
  colsum[(3 * 4 + q) * ST_C + ch] = dwy;
  colsum[(4 * 4 + q) * ST_C + ch] = dwz;
  __syncthreads();
  for (int f = threadIdx.x; f < 5 * ST_C; f += SB_T) {
    const int k = f / ST_C, l = f % ST_C;
    const float* c4 = colsum + k * 4 * ST_C + l;
    const float v = (c4[0] + c4[ST_C]) + (c4[2 * ST_C] + c4[3 * ST_C]);
    if (k < 2)
      part[ST_C * ST_C + k * ST_C + l] = v;
    else
      part[ST_C * ST_C + 2 * ST_C + l * 3 + (k - 2)] = v;
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

int stem_grid(int N, int per_cu) {      // workgroups of 8 wavefronts, per_cu on each CU, each walking its share of the points
  const int cap = 256 * per_cu;
  const int g = N < cap ? N : cap;
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
  PZN_LAUNCH(stem_fwd_kernel, dim3((unsigned)stem_grid(N, 2)), dim3(ST_T), 0, pzn_hip_stream(stream), a);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT size_t pzn_stem_bwd_workspace_bytes(int N) { return (size_t)stem_grid(N, 2) * ST_PART * sizeof(float); }

// Backward of pzn_stem_fwd_f32 from dout[B, N, 64] (+ dout2 when non-NULL: the output had two consumers; their gradients are
// added while loading instead of by a separate pass over two 33.5 MB tensors): dW1[64,3], db1[64], dW2[64,64], db2[64] and the BatchNorm weight / bias
// gradients [N] are ADDED to (the BatchNorm ones may be NULL); nothing is returned for xyz.  Activations are recomputed from xyz
// and the saved statistics.  workspace: pzn_stem_bwd_workspace_bytes(N) bytes, 16-byte aligned (need not be cleared).
PZN_EXPORT int pzn_stem_bwd_f32(const float* xyz, const float* dout, const float* dout2, const float* W1, const float* b1, const float* W2,
                                const float* b2, const float* bn1_weight, const float* bn1_bias, const float* bn2_weight,
                                const float* bn2_bias, const float* mean1, const float* invstd1, const float* mean2,
                                const float* invstd2, int training, int B, int N, float* dW1, float* db1, float* dW2, float* db2,
                                float* dbn1_weight, float* dbn1_bias, float* dbn2_weight, float* dbn2_bias, void* workspace,
                                pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && dout && W1 && b1 && W2 && b2 && mean1 && invstd1 && mean2 && invstd2 && dW1 && db1 && dW2 && db2 && B > 0 &&
                N > 0 && workspace && !(reinterpret_cast<uintptr_t>(workspace) & 15));
  if (B > 64 || (reinterpret_cast<uintptr_t>(W2) & 15)) return PZN_EUNSUPPORTED;
  StemBwdArgs a{xyz, dout, dout2, W1, b1, W2, b2, bn1_weight, bn1_bias, bn2_weight, bn2_bias, mean1, invstd1, mean2, invstd2, training, B, N,
                dW1, db1, dW2, db2, dbn1_weight, dbn1_bias, dbn2_weight, dbn2_bias, static_cast<float*>(workspace)};
  const int grid = stem_grid(N, 2);
  PZN_LAUNCH(stem_bwd_kernel, dim3((unsigned)grid), dim3(SB_T), 0, pzn_hip_stream(stream), a);
  static_assert(ST_PART % 64 == 0, "stem_reduce_kernel: 64 outputs per workgroup");
  PZN_LAUNCH(stem_reduce_kernel, dim3(ST_PART / 64), dim3(1024), 0, pzn_hip_stream(stream), a.part, grid, dW2, db2, db1, dW1);
  PZN_RETURN_LAUNCH_STATUS();
}
