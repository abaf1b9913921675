// attnwgrad.hip — the four weight gradients of one layerAttention block (backward of model5_b.py:83-101) in ONE launch:
//     dWo[256,256] += dz^T t      dWq[64,256] += dq^T x      dWk[64,256] += dk^T x      dWv[256,256] += dv^T x
//     dbo += colsum(dz)           dbq += colsum(dq)          dbk += colsum(dk)          dbv += colsum(dv)
// i.e. a [640 x 256] output whose reduction runs over the M = B*256 ROWS of the operands (16 384 at B = 64): tiny output, long
// reduction.  csrc/dfgemm.hip (two launches per block until round 5) gives every wavefront a 64 x 64 output block and has it
// load its own fragments straight from global memory: each element of dY is fetched and split into bf16 planes by FOUR
// wavefronts, each element of X by four to six, and a wavefront moves ~4 us per 16-row step whatever the grid.  Here a
// workgroup (8 wavefronts, two per SIMD) owns a 128 x 128 output tile over a range of rows and shares the operands through LDS:
//   * loader: wavefront w (0-3: the A tile dY, 4-7: the B tile X) fetches 64 columns, rows 8h .. 8h+7 of every 16-row step, with
//     dword loads - 256 contiguous bytes per row, scalar row address + the lane's offset -, splits the 8 values of a lane into three
//     bf16 planes and writes 3 x 16 bytes to LDS: exactly the MFMA operand of lane (c mod 32, h) of column tile c / 32
//     (v_mfma_f32_32x32x16_bf16 wants 8 consecutive reduction indices of one column per lane).  Every element is fetched and
//     split ONCE per workgroup;
//   * consumer: wavefront w takes the 32 x 64 block (w >> 1, w & 1): 1 + 2 fragments x 3 planes per step by ds_read_b128,
//     12 MFMAs (bf16x3: the six products that give an fp32-accurate result), with the split of the NEXT step's rows issued
//     between them;
//   * the steps are double-buffered in LDS (one barrier per step), the rows of steps s + 2 and s + 3 are in flight;
//   * the partial tiles of the row ranges are stored to a caller-owned workspace and summed in a fixed order by a second small
//     kernel (bit-reproducible; 8 us less than the atomics at 24 row ranges), or - without a workspace - meet in fp32 atomics
//     on dW; column sums for the biases in the A loader wavefronts.
// What bounds it (measured by elimination): vector ISSUE - per step a SIMD issues 24 MFMAs (768 cycles
// of matrix pipe) and ~2 x 70 vector / LDS / memory instructions of the split; deeper load prefetch (4 register sets), a
// three-deep LDS pipeline (fragments read a step ahead) and moving the split under the MFMAs each changed nothing, removing the
// 64-bit per-lane address arithmetic and the atomics did.
// Same products, same bf16x3 arithmetic as the direct-fragment kernel; summation order differs (tests: 1e-5 of float64).
#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int AW_T = 512;          // 8 wavefronts: 4 load the A tile, 4 the B tile; each multiplies a 32 x 64 block
constexpr int AW_TILE = 128;       // output tile: 128 (n) x 128 (k)
constexpr int AW_K = 256;          // columns of X / t (E)
constexpr int AW_NT = 5;           // n tiles: dz 0-127, dz 128-255, dq | dk, dv 0-127, dv 128-255
constexpr int AW_N = AW_NT * AW_TILE;               // 640 output rows
constexpr int AW_PART = AW_N * AW_K + AW_N;         // floats of one split's partial: the [640, 256] tile rows, then 640 column sums
constexpr int AW_FRAG = 64 * 16;   // bytes of one fragment set of a 32-column tile and plane: 64 lanes x 16 B
// LDS per buffer: operand (A, B) x plane (3) x column tile (4) x AW_FRAG = 24 KB; two buffers
constexpr int AW_OPER = 3 * 4 * AW_FRAG;
constexpr int AW_BUF = 2 * AW_OPER;

struct AwHalf {          // a 64-column half of an n tile
  const float* src;      // column 0 of the half: element (m, c) at src[m * ld + c]
  float* dW;             // row 0 of the half in its weight gradient [., 256]
  float* db;             // its bias gradient (64 entries)
  int ld;
};

struct AwArgs {
  AwHalf half[2 * AW_NT];
  const float* b[AW_NT];      // the B operand of each n tile: t (tiles 0, 1) or x (2 .. 4), [M, 256]
  float* partials;            // [splits][AW_PART] or NULL (atomics)
  int M, rows_per_split;
};

__device__ __forceinline__ void aw_split_pair(v2f x, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
  bf16x2 a = __builtin_convertvector(x, bf16x2);
  p1 = __builtin_bit_cast(uint32_t, a);
  v2f fa = v2f{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  v2f r = x - fa;
  bf16x2 b = __builtin_convertvector(r, bf16x2);
  p2 = __builtin_bit_cast(uint32_t, b);
  v2f fb = v2f{__uint_as_float(p2 << 16), __uint_as_float(p2 & 0xffff0000u)};
  v2f r2 = r - fb;
  bf16x2 c = __builtin_convertvector(r2, bf16x2);
  p3 = __builtin_bit_cast(uint32_t, c);
}

// the three planes of 8 consecutive reduction indices of one column -> LDS, as the operand of lane (col & 31, half)
__device__ __forceinline__ void aw_store(const float (&v)[8], unsigned char* oper, int col, int half) {
  uint32_t a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
  aw_split_pair(v2f{v[0], v[1]}, a0, b0, c0);
  aw_split_pair(v2f{v[2], v[3]}, a1, b1, c1);
  aw_split_pair(v2f{v[4], v[5]}, a2, b2, c2);
  aw_split_pair(v2f{v[6], v[7]}, a3, b3, c3);
  unsigned char* dst = oper + (col >> 5) * AW_FRAG + ((half << 5) | (col & 31)) * 16;
  *reinterpret_cast<u32x4*>(dst) = u32x4{a0, a1, a2, a3};
  *reinterpret_cast<u32x4*>(dst + 4 * AW_FRAG) = u32x4{b0, b1, b2, b3};
  *reinterpret_cast<u32x4*>(dst + 8 * AW_FRAG) = u32x4{c0, c1, c2, c3};
}

__global__ __launch_bounds__(AW_T) void attn_wgrad_kernel(AwArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * AW_BUF];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.y >> 1, kt = blockIdx.y & 1;      // output tile (nt, kt)
  const int m_begin = blockIdx.x * p.rows_per_split;
  int m_end = m_begin + p.rows_per_split;
  m_end = m_end < p.M ? m_end : p.M;
  const int nsteps = (m_end - m_begin) >> 4;      // even: rows_per_split % 32 == 0 and M % 32 == 0 (host)
  if (nsteps <= 0) return;

  // loader role: wavefronts 0-3 the A tile (dY), 4-7 the B tile (X): wavefront w loads columns 64 (w & 1) .. + 63 of its tile,
  // rows 8 lh .. 8 lh + 7 of every step (lh = (w >> 1) & 1) - everything but the lane's column is wavefront-uniform, so the row
  // addresses are scalar registers and the lane supplies a 32-bit offset (the 64-bit per-lane address arithmetic was 10 of the
  // ~85 vector instructions of a step, and vector issue, not the matrix pipe, is what bounds this loop)
  const bool lb = wave >= 4;
  const int lch = wave & 1, lh = (wave >> 1) & 1, lc = 64 * lch + lane;
  const AwHalf ha = p.half[2 * nt + lch];
  const int ld = lb ? AW_K : ha.ld;
  const float* ubase = (lb ? p.b[nt] + kt * AW_TILE + 64 * lch : ha.src) + (size_t)(m_begin + 8 * lh) * ld;
  // (uniform by construction; readfirstlane says so to the compiler, which otherwise folds the lane's offset in first and does
  // all the row arithmetic in 64-bit vector adds)
  typedef const __attribute__((address_space(1))) float* gptr;
  const unsigned ub_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(reinterpret_cast<unsigned long long>(ubase) >> 32));
  const unsigned ub_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<unsigned long long>(ubase));
  const unsigned long long ub = ((unsigned long long)ub_hi << 32) | ub_lo;      // (unsigned halves: the builtin returns int)
  const unsigned row_bytes = __builtin_amdgcn_readfirstlane((unsigned)ld * 4u);
  const bool bias = !lb && kt == 0;      // these wavefronts also sum their columns (the bias gradients)
  // two register sets: the loads of step s + 2 are issued while step s is multiplied and s + 1 is split
  float r0[8], r1[8];
  auto issue = [&](int s, float (&r)[8]) {
    const int sc = s < nsteps ? s : nsteps - 1;      // (past the end: re-read the last rows, harmless, keeps the loop branch-free)
    const unsigned long long a = ub + (unsigned long long)sc * 16 * row_bytes;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = reinterpret_cast<gptr>(a + (unsigned long long)e * row_bytes)[(unsigned)lane];      // scalar row address + the lane's 32-bit offset
    // (pins the issue order of the two register sets: left to itself the scheduler interleaves their loads, and the wait for the
    // older set at the top of the loop becomes a wait for both)
    __builtin_amdgcn_sched_barrier(0);
  };
  float dbsum = 0.f;

  // consumer role: wavefront w takes the 32 x 64 block (column tile qi = w >> 1 of A) x (column tiles 2 qj, 2 qj + 1 of B, qj = w & 1):
  // eight wavefronts, two per SIMD
  const int qi = wave >> 1, qj = wave & 1;
  floatx16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // One step, software-pipelined: the fragments of step s are read, its 12 MFMAs are issued with the split of step s + 1 (vector
  // instructions on registers that arrived two steps ago) between them, then step s + 1's planes go to the OTHER LDS buffer and
  // one barrier ends the step.
  const int w_off = (lb ? AW_OPER : 0) + (lc >> 5) * AW_FRAG + ((lh << 5) | (lc & 31)) * 16;
  auto step = [&](int s, float (&rn)[8]) {      // rn: the rows of step s + 1
    const unsigned char* buf = lds + (s & 1) * AW_BUF;
    unsigned char* dst = lds + ((s + 1) & 1) * AW_BUF + w_off;
    bf16x8 fa[3], fb[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      fa[pl] = *reinterpret_cast<const bf16x8*>(buf + (4 * pl + qi) * AW_FRAG + lane * 16);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[j][pl] = *reinterpret_cast<const bf16x8*>(buf + AW_OPER + (4 * pl + 2 * qj + j) * AW_FRAG + lane * 16);
    }
    if (bias && s + 1 < nsteps) {      // (the last step splits clamped re-reads that are never multiplied: not part of the sums)
#pragma unroll
      for (int e = 0; e < 8; ++e) dbsum += rn[e];
    }
    uint32_t a0, a1, a2, a3, b0, b1, b2, b3, c0, c1, c2, c3;
    aw_split_pair(v2f{rn[0], rn[1]}, a0, b0, c0);
    aw_split_pair(v2f{rn[2], rn[3]}, a1, b1, c1);
    aw_split_pair(v2f{rn[4], rn[5]}, a2, b2, c2);
    aw_split_pair(v2f{rn[6], rn[7]}, a3, b3, c3);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      floatx16 c = acc[j];
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[j][0], c, 0, 0, 0);  // small terms first
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j][1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j][2], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[j][0], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j][1], c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[j][0], c, 0, 0, 0);
      acc[j] = c;
    }
    *reinterpret_cast<u32x4*>(dst) = u32x4{a0, a1, a2, a3};
    *reinterpret_cast<u32x4*>(dst + 4 * AW_FRAG) = u32x4{b0, b1, b2, b3};
    *reinterpret_cast<u32x4*>(dst + 8 * AW_FRAG) = u32x4{c0, c1, c2, c3};
    issue(s + 3, rn);
    __syncthreads();      // step s + 1 is in LDS, every wavefront is done with the buffer of step s
  };
  // prologue: step 0's planes into buffer 0; the rows of steps 1 and 2 in flight
  issue(0, r0);
  issue(1, r1);
  if (bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) dbsum += r0[e];
  }
  aw_store(r0, lds + (lb ? AW_OPER : 0), lc, lh);
  issue(2, r0);
  __syncthreads();
  // nsteps is even (host): no branch between the two halves, so that the compiler's vmcnt bookkeeping stays exact (with a
  // conditional second half it waited for vmcnt(0) at the loop top).  Step s splits the rows of step s + 1: r1 in even steps,
  // r0 in odd ones.
  for (int s = 0; s < nsteps; s += 2) {
    step(s, r1);
    step(s + 1, r0);
  }

  // epilogue: element r of lane l = row n = (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the A column tile, column k = l & 31 of B tile j.
  // With a partials buffer (the chained backward owns one) the tile of this row range is STORED - [split][640][256] - and
  // attn_wgrad_reduce_kernel sums the splits in a fixed order; without one the splits meet in fp32 atomics on dW.
  const int l31 = lane & 31, hf = lane >> 5;
  const int n0 = 32 * qi;                               // row of the 128-row tile
  if (p.partials) {
    float* part = p.partials + (size_t)blockIdx.x * AW_PART + (size_t)(nt * AW_TILE + n0) * AW_K;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = kt * AW_TILE + 64 * qj + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) part[(size_t)((r & 3) + 8 * (r >> 2) + 4 * hf) * AW_K + k] = acc[j][r];
    }
  } else {
    const AwHalf ho = p.half[2 * nt + (n0 >> 6)];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = kt * AW_TILE + 64 * qj + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) atomicAdd(ho.dW + (size_t)((n0 & 63) + (r & 3) + 8 * (r >> 2) + 4 * hf) * AW_K + k, acc[j][r]);
    }
  }
  // bias gradients: the A loader wavefronts of the kt == 0 workgroups hold the column sums of their rows (two row halves per column)
  if (kt == 0) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
    if (tid < 256) red[128 * lh + lc] = dbsum;
    __syncthreads();
    if (tid < 128) {
      const float v = red[tid] + red[tid + 128];
      if (p.partials)
        p.partials[(size_t)blockIdx.x * AW_PART + (size_t)AW_N * AW_K + nt * AW_TILE + tid] = v;
      else
        atomicAdd(p.half[2 * nt + (tid >> 6)].db + (tid & 63), v);
    }
  }
}

// dW (+)= sum over the splits of their partial tiles, in a fixed order (bit-reproducible).  A workgroup owns 64 consecutive
// float4 outputs (one row of the [640, 256] tile); its four wavefronts each sum every fourth split - all their loads in flight at
// once: with one thread per output the 16 MB of partials were read at 1.5 TB/s, 640 wavefronts each waiting on its own chain -
// and meet through LDS.  The last 10 workgroups do the same for the 640 bias entries (64 scalars each).
__global__ __launch_bounds__(256) void attn_wgrad_reduce_kernel(AwArgs p, int S, int accumulate) {
  __shared__ f32x4 red[3][64];
  const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int row = blockIdx.x;
  if (row < AW_N) {
    const f32x4* src = reinterpret_cast<const f32x4*>(p.partials + (size_t)row * AW_K) + o;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    int s = g;
    for (; s + 12 < S; s += 16) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + (size_t)(s + 4 * u) * (AW_PART / 4));
#pragma unroll
      for (int u = 0; u < 4; ++u) sum += v[u];
    }
    for (; s < S; s += 4) sum += __builtin_nontemporal_load(src + (size_t)s * (AW_PART / 4));
    if (g) red[g - 1][o] = sum;
    __syncthreads();
    if (g == 0) {
      sum = (sum + red[0][o]) + (red[1][o] + red[2][o]);
      f32x4* dst = reinterpret_cast<f32x4*>(p.half[row >> 6].dW + (size_t)(row & 63) * AW_K) + o;
      if (accumulate) sum += *dst;
      *dst = sum;
    }
  } else {
    const int n = (row - AW_N) * 64 + o;
    const float* src = p.partials + (size_t)AW_N * AW_K + n;
    float sum = 0.f;
    for (int s = g; s < S; s += 4) sum += src[(size_t)s * AW_PART];
    float* redf = reinterpret_cast<float*>(&red[0][0]);
    if (g) redf[(g - 1) * 64 + o] = sum;
    __syncthreads();
    if (g == 0) {
      sum = (sum + redf[o]) + (redf[64 + o] + redf[128 + o]);
      float* dst = p.half[n >> 6].db + (n & 63);
      *dst = accumulate ? *dst + sum : sum;
    }
  }
}

}  // namespace

namespace {
// row ranges ("splits") of a launch: 10 tiles x 24 = 240 workgroups, one round of the 256 CUs.  Measured at M = 16 384 (ktimer,
// us per launch incl. ~5 us of timer floor): atomics 8: 93, 16: 60, 24: 54, 32: 61, 48: 65 (every split adds 0.65 MB of atomics,
// ~1 us); partial tiles + reduction 16: 51 + 8, 24: 40 + 9, 32: 41 + 10, 48: 36 + 11.
constexpr int AW_SPLITS = 24;
int aw_rows_per_split(int M) { return ((M / 64 + AW_SPLITS - 1) / AW_SPLITS) * 64; }
}  // namespace

size_t pzn_attn_wgrad_ws_bytes(int M) {
  if (M < 64 || (M & 63)) return 0;
  const int rows = aw_rows_per_split(M);
  return (size_t)((M + rows - 1) / rows) * AW_PART * sizeof(float);
}

// dz, t, dq, dk, dv, x: [M, 256 | 256 | 64 | 64 | 256 | 256] row-major (E = 256, dk = 64, M % 64 == 0, else PZN_EUNSUPPORTED); the
// eight gradients are added to (accumulate != 0) or overwritten.  ws: pzn_attn_wgrad_ws_bytes(M) bytes (256-byte aligned) for the
// row ranges' partial tiles - the result is then a fixed-order sum (bit-reproducible) - or NULL: fp32 atomics.
int pzn_attn_wgrad_tiled(const float* dz, const float* t, const float* dq, const float* dkk, const float* dvv, const float* x,
                         int M, float* dWq, float* dbq, float* dWk, float* dbk, float* dWv, float* dbv, float* dWo, float* dbo,
                         int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  if (M < 64 || (M & 63)) return PZN_EUNSUPPORTED;
  AwArgs a;
  for (int h = 0; h < 4; ++h) a.half[h] = AwHalf{dz + 64 * h, dWo + (size_t)64 * h * AW_K, dbo + 64 * h, 256};
  a.half[4] = AwHalf{dq, dWq, dbq, 64};
  a.half[5] = AwHalf{dkk, dWk, dbk, 64};
  for (int h = 0; h < 4; ++h) a.half[6 + h] = AwHalf{dvv + 64 * h, dWv + (size_t)64 * h * AW_K, dbv + 64 * h, 256};
  a.b[0] = a.b[1] = t;
  a.b[2] = a.b[3] = a.b[4] = x;
  a.M = M;
  const bool use_ws = ws && ws_bytes >= pzn_attn_wgrad_ws_bytes(M) && (reinterpret_cast<uintptr_t>(ws) & 15) == 0;
  const int rows = aw_rows_per_split(M);
  const int S = (M + rows - 1) / rows;
  a.rows_per_split = rows;
  a.partials = use_ws ? static_cast<float*>(ws) : nullptr;
  if (!use_ws && !accumulate) {      // the atomics add
    float* const outs[8] = {dWq, dbq, dWk, dbk, dWv, dbv, dWo, dbo};
    const size_t ns[8] = {(size_t)64 * AW_K, 64, (size_t)64 * AW_K, 64, (size_t)256 * AW_K, 256, (size_t)256 * AW_K, 256};
    for (int i = 0; i < 8; ++i)
      if (pzn_zero_async(outs[i], ns[i], st) != PZN_OK) return PZN_ELAUNCH;
  }
  PZN_LAUNCH(attn_wgrad_kernel, dim3((unsigned)S, 2 * AW_NT), dim3(AW_T), 0, st, a);
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  if (use_ws) {
    PZN_LAUNCH(attn_wgrad_reduce_kernel, dim3(AW_N + AW_N / 64), dim3(256), 0, st, a, S, accumulate);
    if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  }
  return PZN_OK;
}
