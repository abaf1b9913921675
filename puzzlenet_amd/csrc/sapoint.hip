// sapoint.hip — first shared-MLP layer of a set-abstraction level computed PER POINT instead of per grouped row.
//
// Reference (pointnet_util.py:123-132 + model5_b.py:452 / :459): every one of the B*S*32 grouped rows is
// {xyz[j] - centre, feat[j]} with j = idx[b,s,k], and the first 1x1 convolution multiplies each row by W1[C1, 3+D].
// The feature block of that product depends on the POINT only, and a point is gathered by 8-16 rows:
//     h[b,s,k,:] = relu( W1[:,0:3] (xyz[j] - centre_s)  +  P[b,j,:]  +  b1 ),     P = feat W1[:,3:]^T  per point.
// So the product shrinks from B*S*32 rows to B*N rows (8x / 16x fewer flops, same result up to the order of the
// fp32 sum), the grouped tensor [B,S,32,3+D] is never written, and the layer becomes a gather: read P rows
// (L2 / MALL resident: 67 MB at level 1), write h.  Backward, with dh = the (ReLU-masked) gradient of h:
//     dP[b,j,:] = sum over the rows that gathered j of dh[row,:]      (inverse neighbour lists, no atomics on rows)
//     dW1[:,0:3] += dh^T (xyz[j] - centre),   db1 += column sums of dh            (same pass over dh)
//     dfeat = dP W1[:,3:],   dW1[:,3:] += dP^T feat                               (two per-point GEMMs, callers)
// The HBM bill of the layer: forward 1 write of h; backward 1 read of dh.  (The grouped-row path: forward reads
// the grouped rows and writes h; backward reads dh twice, the grouped rows once, and scatter-adds B*S*32*D floats.)
#include <stdlib.h>

#include "pzn_common.h"

namespace {

constexpr int SP_T = 256;  // 4 wavefronts

template <int V>
struct VecT;
template <>
struct VecT<1> {
  typedef float type;
};
template <>
struct VecT<2> {
  typedef float2 type;
};
template <>
struct VecT<4> {
  typedef float4 type;
};

template <int V>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[V]) {
  typename VecT<V>::type t = *reinterpret_cast<const typename VecT<V>::type*>(p);
  const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
  for (int i = 0; i < V; ++i) v[i] = f[i];
}
// the same, as a streaming (non-temporal) load: rows that are read exactly once
template <int V>
__device__ __forceinline__ void load_vec_nt(const float* p, float (&v)[V]) {
  typedef float vt __attribute__((ext_vector_type(V)));
  const vt t = __builtin_nontemporal_load(reinterpret_cast<const vt*>(p));
#pragma unroll
  for (int i = 0; i < V; ++i) v[i] = t[i];
}
template <int V>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[V]) {
  typename VecT<V>::type t;
  float* f = reinterpret_cast<float*>(&t);
#pragma unroll
  for (int i = 0; i < V; ++i) f[i] = v[i];
  *reinterpret_cast<typename VecT<V>::type*>(p) = t;
}

__device__ __forceinline__ float bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// h rows.  A wavefront takes 64 consecutive rows: lane l fetches the index / offset of row l, then the rows are written
// RPI at a time: 64 / RPI lanes per row, 16 bytes (V = 4 channels) per lane, so every load / store instruction moves
// 1 KB whatever C1 = 256 / RPI is (coalesced P read, h write; 8-byte accesses at C1 = 128 ran at 3.3 TB/s against 3.9).
template <int V, int RPI>
__global__ __launch_bounds__(SP_T) void sa_point_l1_fwd_kernel(const float* __restrict__ xyz,
                                                               const float* __restrict__ new_xyz,
                                                               const int64_t* __restrict__ idx,
                                                               const float* __restrict__ P,
                                                               const float* __restrict__ W1, int ldw,
                                                               const float* __restrict__ b1, int N, int S, long rows,
                                                               float* __restrict__ h, int xcd_map) {
  constexpr int LPR = 64 / RPI;  // lanes per row
  constexpr int C1 = LPR * V;
  const int lane = threadIdx.x & 63;
  const int cl = lane % LPR, sub = lane / LPR;
  const long gw = (long)blockIdx.x * (SP_T / 64) + (threadIdx.x >> 6), nw = (long)gridDim.x * (SP_T / 64);
  float wx[V], wy[V], wz[V], bb[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = cl * V + i;
    wx[i] = W1[(size_t)c * ldw], wy[i] = W1[(size_t)c * ldw + 1], wz[i] = W1[(size_t)c * ldw + 2];
    bb[i] = b1 ? b1[c] : 0.f;
  }
  const long nbatch = (rows + 63) >> 6;
  // Workgroups go to the 8 XCDs round-robin and each XCD has its own 4 MB L2.  A cloud's P rows (N*C1 floats: 0.5-1 MB)
  // are gathered 8-16 times each, so every XCD works through its OWN clouds (b = xcd, xcd + 8, ...) in order: one or
  // two clouds' P stay L2-resident per XCD instead of eight clouds' thrashing it (PMC: 391 -> ~60 MB fetched per
  // launch).  Needs whole batches per cloud and a grid that is a multiple of 8; else the plain strided walk.
  const int bpc = (S * 32) >> 6;  // batches per cloud
  const bool by_xcd = xcd_map && ((S * 32) & 63) == 0 && (gridDim.x & 7) == 0;
  const int xcd = blockIdx.x & 7;
  const long B_ = rows / ((long)S * 32);
  const long ncl = by_xcd ? (B_ - xcd + 7) / 8 : 0;          // clouds of this XCD
  const long lw = (long)(blockIdx.x >> 3) * (SP_T / 64) + (threadIdx.x >> 6), nlw = (long)(gridDim.x >> 3) * (SP_T / 64);
  const long q_end = by_xcd ? ncl * bpc : nbatch;
  for (long q = by_xcd ? lw : gw; q < q_end; q += by_xcd ? nlw : nw) {
    const long bt = by_xcd ? ((long)xcd + 8 * (q / bpc)) * bpc + q % bpc : q;
    const long row = bt * 64 + lane;
    float dx = 0.f, dy = 0.f, dz = 0.f;
    int prow = 0;  // row of P (b*N + j); fits 32 bits: B*N*C1 floats are addressed through size_t below
    if (row < rows) {
      const long grp = row >> 5;         // b*S + s
      const long b = grp / S;
      const int j = (int)idx[row];
      const float* pq = xyz + ((size_t)b * N + j) * 3;
      const float* c = new_xyz + (size_t)grp * 3;
      dx = pq[0] - c[0], dy = pq[1] - c[1], dz = pq[2] - c[2];  // pointnet_util.py:124
      prow = (int)(b * N + j);
    }
    const int nr = (int)min((long)64, rows - bt * 64);
    // value of a per-row quantity for the row this lane works on in step r (row r + sub)
    auto pick_i = [&](int x, int r) {
      if constexpr (RPI == 1) return __builtin_amdgcn_readlane(x, r);
      else return __shfl(x, r + sub, PZN_WAVE);
    };
    auto pick_f = [&](float x, int r) { return __builtin_bit_cast(float, pick_i(__builtin_bit_cast(int, x), r)); };
    auto finish = [&](float (&v)[V], int r) {
      const float rx = pick_f(dx, r), ry = pick_f(dy, r), rz = pick_f(dz, r);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float t = fmaf(wz[i], rz, fmaf(wy[i], ry, wx[i] * rx)) + v[i] + bb[i];
        v[i] = t > 0.f ? t : 0.f;
      }
      if (RPI == 1 || r + sub < nr) store_vec<V>(h + ((size_t)bt * 64 + r + sub) * C1 + cl * V, v);
    };
    auto fetch = [&](float (&v)[V], int r) {
      const int pr = pick_i(prow, RPI == 1 ? r : min(r, 63 - sub));   // (rows past the end re-read a valid row)
      load_vec<V>(P + (size_t)pr * C1 + cl * V, v);
    };
    int r = 0;
    for (; r + 4 * RPI <= nr; r += 4 * RPI) {  // four row groups in flight
      float v0[V], v1[V], v2[V], v3[V];
      fetch(v0, r);
      fetch(v1, r + RPI);
      fetch(v2, r + 2 * RPI);
      fetch(v3, r + 3 * RPI);
      finish(v0, r);
      finish(v1, r + RPI);
      finish(v2, r + 2 * RPI);
      finish(v3, r + 3 * RPI);
    }
    for (; r < nr; r += RPI) {
      float v0[V];
      fetch(v0, r);
      finish(v0, r);
    }
  }
}

// Inverse neighbour lists of one cloud per workgroup: off[b][0..N] (exclusive prefix of the reference counts) and
// rows[b][.] = the in-cloud row numbers (s*32 + k) grouped by the point they gathered, ascending within a point.  Counters live in LDS.
constexpr int INV_T = 1024;
// LROWS: the lists are built and sorted in LDS ([SK] ints behind the counters) and leave in one coalesced pass; otherwise (a
// cloud whose S * K rows do not fit) they are built in global memory and sorted there.
template <bool LROWS>
__global__ __launch_bounds__(INV_T) void sa_inverse_lists_kernel(const int64_t* __restrict__ idx, int N, int SK,
                                                                 int32_t* __restrict__ off,
                                                                 int32_t* rows,
                                                                 int32_t* __restrict__ pts) {
  extern __shared__ int cnt[];  // [N] counters, then [INV_T] scan scratch, then (LROWS) [SK] rows
  int* scan = cnt + N;
  int* lrows = scan + INV_T;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int64_t* ib = idx + (size_t)b * SK;
  for (int j = tid; j < N; j += INV_T) cnt[j] = 0;
  // this thread's entries i = tid, tid + 1024, ...: the first INV_R of them are kept in registers for the fill pass
  // (all loads of the pass in flight at once; 16 x 1024 covers the model's 512 x 32 rows per cloud)
  constexpr int INV_R = 16;
  int mine[INV_R];
#pragma unroll
  for (int u = 0; u < INV_R; ++u) {
    const int i = tid + u * INV_T;
    mine[u] = i < SK ? (int)ib[i] : -1;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < INV_R; ++u)
    if (mine[u] >= 0) atomicAdd(&cnt[mine[u]], 1);
  for (int i = tid + INV_R * INV_T; i < SK; i += INV_T) atomicAdd(&cnt[(int)ib[i]], 1);
  __syncthreads();
  // exclusive scan of cnt[0..N): thread t owns the contiguous chunk [t*per, (t+1)*per)
  const int per = (N + INV_T - 1) / INV_T;
  const int j0 = min(N, tid * per), j1 = min(N, j0 + per);
  int local = 0;
  for (int j = j0; j < j1; ++j) local += cnt[j];
  scan[tid] = local;
  __syncthreads();
  for (int o = 1; o < INV_T; o <<= 1) {
    const int v = tid >= o ? scan[tid - o] : 0;
    __syncthreads();
    scan[tid] += v;
    __syncthreads();
  }
  int run = scan[tid] - local;
  const int first = run;              // this thread's points own the entries [first, last)
  int32_t* ob = off + (size_t)b * (N + 1);
  for (int j = j0; j < j1; ++j) {
    const int c = cnt[j];
    ob[j] = run;
    cnt[j] = run;  // becomes the fill cursor
    run += c;
  }
  const int last = run;
  if (tid == INV_T - 1) ob[N] = scan[INV_T - 1];
  __syncthreads();
  int32_t* rb = rows + (size_t)b * SK;
  int* dst = LROWS ? lrows : rb;
#pragma unroll
  for (int u = 0; u < INV_R; ++u) {
    const int j = mine[u];
    if (j >= 0) dst[atomicAdd(&cnt[j], 1)] = tid + u * INV_T;
  }
  for (int i = tid + INV_R * INV_T; i < SK; i += INV_T) dst[atomicAdd(&cnt[(int)ib[i]], 1)] = i;
  // The fill places a point's rows in the order the atomic cursors were taken; sorted ascending, the per-point sums that walk
  // these lists (csrc/sapool.hip) add in the same order in every run.  Thread t sorts the lists of its own points (8-16 rows
  // each in the model: an insertion sort in place) and names the point of each of its entries.
  __syncthreads();
  int at = first;
  for (int j = j0; j < j1; ++j) {
    const int hi = cnt[j];            // (the cursor stands at the end of point j's list)
    for (int i = at + 1; i < hi; ++i) {
      const int v = dst[i];
      int k = i - 1;
      while (k >= at && dst[k] > v) dst[k + 1] = dst[k], --k;
      dst[k + 1] = v;
    }
    if (pts)
      for (int i = at; i < hi; ++i) pts[(size_t)b * SK + i] = j;
    at = hi;
  }
  (void)last;
  if (LROWS) {
    __syncthreads();
    for (int i = tid; i < SK; i += INV_T) rb[i] = lrows[i];
  }
}

// dP and the xyz / bias part of the first layer's gradients.  The inverse lists are one array of B*S*32 entries
// (row, point) sorted by point; a wavefront takes 64 consecutive entries: lane l fetches entry l (row, point, the
// row's centre offset) up front, then the dh rows are read one after the other, all lanes on the C1 = 64*V
// channels (coalesced), and summed until the point changes.  Points wholly inside the wavefront's range are
// stored, the first and last one (their lists may continue in the neighbouring ranges) are added atomically into
// the zero-initialised dP.  The four per-channel sums for dW1[:,0:3] and db1 stay in registers over the whole
// range and meet in LDS at the end: one set of atomics per workgroup.
// (A first version walked point by point: off -> rows -> centre -> dh is a chain of dependent loads per point,
// 0.30 ms for the 537 MB of dh at level 2.)
template <int V, int G>
__global__ __launch_bounds__(SP_T) void sa_point_l1_bwd_kernel(const float* __restrict__ dh,
                                                               const float* __restrict__ xyz,
                                                               const float* __restrict__ new_xyz,
                                                               const int32_t* __restrict__ rows,
                                                               const int32_t* __restrict__ pts, int N, int S,
                                                               long entries, float* __restrict__ dP,
                                                               float* __restrict__ dW1, int ldw,
                                                               float* __restrict__ db1,
                                                               const uint32_t* __restrict__ rowmask) {
  constexpr int C1 = 64 * V;
  __shared__ float red[SP_T / 64][4][C1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long gw = (long)blockIdx.x * (SP_T / 64) + wave, nw = (long)gridDim.x * (SP_T / 64);
  const int SK = S * 32;
  float ax[V], ay[V], az[V], ab[V];
#pragma unroll
  for (int i = 0; i < V; ++i) ax[i] = ay[i] = az[i] = ab[i] = 0.f;
  const long nbatch = (entries + 63) >> 6;
  for (long bt = gw; bt < nbatch; bt += nw) {
    const long e = bt * 64 + lane;
    int grow = 0, gp = -1;  // global row of dh, global point
    int nz = 1;             // rowmask: rows without their bit are exactly zero and were not written (poolbwd.hip): not read
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (e < entries) {
      const long b = e / SK;
      const int rid = rows[e], pj = pts[e];
      grow = (int)(b * SK + rid);
      if (rowmask) nz = (int)((rowmask[grow >> 5] >> (grow & 31)) & 1u);
      gp = (int)(b * N + pj);
      const float* q = xyz + (size_t)gp * 3;
      if (new_xyz) {
        const float* c = new_xyz + ((size_t)b * S + (rid >> 5)) * 3;
        dx = q[0] - c[0], dy = q[1] - c[1], dz = q[2] - c[2];
      } else {  // coordinate term folded into the per-point table (pzn_sa_prep_f32): the factor is the point itself
        dx = q[0], dy = q[1], dz = q[2];
      }
    }
    const int nr = (int)min((long)64, entries - bt * 64);
    const int first = __builtin_amdgcn_readlane(gp, 0), last = __builtin_amdgcn_readlane(gp, nr - 1);
    int cur = first;
    float acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) acc[i] = 0.f;
    auto flush = [&](int point) {
      float* o = dP + (size_t)point * C1 + lane * V;
      if (point == first || point == last) {
#pragma unroll
        for (int i = 0; i < V; ++i) atomicAdd(o + i, acc[i]);
      } else {
        store_vec<V>(o, acc);
      }
#pragma unroll
      for (int i = 0; i < V; ++i) ab[i] += acc[i], acc[i] = 0.f;
    };
    auto take = [&](const float (&g)[V], int r) {
      const int pj = __builtin_amdgcn_readlane(gp, r);
      if (pj != cur) {  // wave-uniform
        flush(cur);
        cur = pj;
      }
      const float rx = bcast(dx, r), ry = bcast(dy, r), rz = bcast(dz, r);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        acc[i] += g[i];
        ax[i] = fmaf(g[i], rx, ax[i]);
        ay[i] = fmaf(g[i], ry, ay[i]);
        az[i] = fmaf(g[i], rz, az[i]);
      }
    };
    const float* dhl = dh + lane * V;
    // the rows that are read: with a row mask only those whose bit is set (the others are exactly zero: a point all of
    // whose rows are skipped keeps the zero dP was filled with), G of them in flight per trip
    uint64_t todo = __ballot(e < entries && nz != 0);
    while (todo) {      // wave-uniform
      int ru[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        ru[u] = todo ? __builtin_ctzll(todo) : -1;
        todo &= todo - 1;
      }
      float g[G][V];
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (ru[u] >= 0) load_vec_nt<V>(dhl + (size_t)__builtin_amdgcn_readlane(grow, ru[u]) * C1, g[u]);
#pragma unroll
      for (int u = 0; u < G; ++u)
        if (ru[u] >= 0) take(g[u], ru[u]);
    }
    flush(cur);
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    red[wave][0][lane * V + i] = ax[i];
    red[wave][1][lane * V + i] = ay[i];
    red[wave][2][lane * V + i] = az[i];
    red[wave][3][lane * V + i] = ab[i];
  }
  __syncthreads();
  for (int f = threadIdx.x; f < 4 * C1; f += SP_T) {
    const int q = f / C1, c = f - q * C1;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < SP_T / 64; ++w) t += red[w][q][c];
    if (q < 3)
      atomicAdd(dW1 + (size_t)c * ldw + q, t);
    else if (db1)
      atomicAdd(db1 + c, t);
  }
}

// P[row, :] += W1[:, 0:3] xyz[row]  (rows = B*N points)   and   Q[g, :] = b1 - W1[:, 0:3] new_xyz[g]  (g = B*S groups):
// with the coordinate term split this way a generated row is relu(P[idx] + Q[g]) — one add and one max per element
// where the round-1 form needed three fmas with per-row broadcasts.  Thread = 4 channels of one row / group.
// One multiply-add of the coordinate term kept out of the packed-fp32 unit.  Under plain -O3 the compiler pairs the four
// channels of a thread into v_pk_mul_f32 / v_pk_fma_f32 with op_sel modifiers (one coordinate against two weights).
// Measured on MI355X: while a workgroup of the general matrix-core engine (gemm_kernel) or of the generated-row kernel
// shares the CU — the other encoder's stream — that packed form returned sums with the y term missing in lanes 48-63
// (a few rows per launch, 0.1-0.2 absolute in P; tests/test_gpu_concurrency.py reproduces it).  The same arithmetic as
// single v_fma_f32 / v_mul_f32 (bit-identical results) is not affected, and this kernel is bound by its row traffic anyway.
__device__ __forceinline__ float prep_mul(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float prep_fma(float a, float b, float c) {
  float r;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float prep_add(float a, float b) {
  float r;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float prep_sub(float a, float b) {
  float r;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__global__ __launch_bounds__(256) void sa_prep_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                      const float* __restrict__ W1, int ldw, const float* __restrict__ b1,
                                                      long prow, long groups, int C1, float* __restrict__ P,
                                                      float* __restrict__ Q) {
  // the three coordinate columns of W1 (and b1), once per workgroup, as planes {wx[C1], wy[C1], wz[C1], b[C1]}: a thread
  // then reads its four channels of each plane with one 16-byte LDS load instead of twelve scattered global dwords
  extern __shared__ __attribute__((aligned(16))) float sw[];
  for (int t = threadIdx.x; t < C1; t += blockDim.x) {
    const float* w = W1 + (size_t)t * ldw;
    sw[t] = w[0], sw[C1 + t] = w[1], sw[2 * C1 + t] = w[2], sw[3 * C1 + t] = b1 ? b1[t] : 0.f;
  }
  __syncthreads();
  const int c4 = C1 >> 2;
  const long total = (prow + groups) * c4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / c4;
    const int c = (int)(e - r * c4) * 4;
    const bool point = r < prow;
    const float* q = point ? xyz + (size_t)r * 3 : new_xyz + (size_t)(r - prow) * 3;
    const float x = q[0], y = q[1], z = q[2];
    const float4 wx = *reinterpret_cast<const float4*>(sw + c), wy = *reinterpret_cast<const float4*>(sw + C1 + c);
    const float4 wz = *reinterpret_cast<const float4*>(sw + 2 * C1 + c);
    float t[4];      // fmaf(wz, z, fmaf(wy, y, wx * x))
    t[0] = prep_fma(wz.x, z, prep_fma(wy.x, y, prep_mul(wx.x, x)));
    t[1] = prep_fma(wz.y, z, prep_fma(wy.y, y, prep_mul(wx.y, x)));
    t[2] = prep_fma(wz.z, z, prep_fma(wy.z, y, prep_mul(wx.z, x)));
    t[3] = prep_fma(wz.w, z, prep_fma(wy.w, y, prep_mul(wx.w, x)));
    if (point) {
      float4* o = reinterpret_cast<float4*>(P + (size_t)r * C1 + c);
      float4 v = *o;
      v.x = prep_add(v.x, t[0]), v.y = prep_add(v.y, t[1]), v.z = prep_add(v.z, t[2]), v.w = prep_add(v.w, t[3]);
      *o = v;
    } else {
      const float4 bb = *reinterpret_cast<const float4*>(sw + 3 * C1 + c);
      float4 v;
      v.x = prep_sub(bb.x, t[0]), v.y = prep_sub(bb.y, t[1]), v.z = prep_sub(bb.z, t[2]), v.w = prep_sub(bb.w, t[3]);
      *reinterpret_cast<float4*>(Q + (size_t)(r - prow) * C1 + c) = v;
    }
  }
}

}  // namespace

PZN_EXPORT int pzn_sa_prep_f32(const float* xyz, const float* new_xyz, const float* W1, const float* b1, int B, int N, int S,
                               int D, int C1, float* P, float* Q, pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && new_xyz && W1 && P && Q && B > 0 && N > 0 && S > 0 && D >= 0 && C1 > 0);
  if ((C1 & 3) || (reinterpret_cast<uintptr_t>(P) & 15) || (reinterpret_cast<uintptr_t>(Q) & 15)) return PZN_EUNSUPPORTED;
  const long prow = (long)B * N, groups = (long)B * S;
  const long total = (prow + groups) * (C1 >> 2);
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (C1 > 4096) return PZN_EUNSUPPORTED;      // (4 planes of C1 floats in LDS)
  if (blocks > 1024) blocks = 1024;             // a few rows per thread: the LDS prologue is paid per workgroup
  PZN_LAUNCH(sa_prep_kernel, dim3((unsigned)blocks), dim3(256), (size_t)4 * C1 * sizeof(float), pzn_hip_stream(stream),
                     xyz, new_xyz, W1, 3 + D, b1, prow, groups, C1, P, Q);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_sa_point_l1_fwd_f32(const float* xyz, const float* new_xyz, const int64_t* idx, const float* P,
                                       const float* W1, const float* b1, int B, int N, int S, int D, int C1, float* h,
                                       pzn_stream_t stream) {
  PZN_CHECK_ARG(xyz && new_xyz && idx && P && W1 && h && B > 0 && N > 0 && S > 0 && D >= 0);
  PZN_CHECK_ARG((long)B * N < 2147483647L);
  if (C1 != 64 && C1 != 128 && C1 != 256) return PZN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(P) & 15) || (reinterpret_cast<uintptr_t>(h) & 15)) return PZN_EUNSUPPORTED;
  const long rows = (long)B * S * 32;
  const long nbatch = (rows + 63) / 64;
  // Few wavefronts with many rows in flight each: 16 (4) wavefronts per CU at 512-byte (1-KB) rows measured best
  // (level 1: 0.137 ms at 1024 workgroups, 0.163 at 512, 0.170 at 4096; level 2: 0.115 at 256-1024, 0.124 at 2048+):
  // more concurrent random row streams only lengthen the queues.
  static const long fcap = [] { const char* e = getenv("PZN_SP_FGRID"); return e ? atol(e) : 0L; }();  // tuning aid
  long blocks = (nbatch + 3) / 4;
  const long want = fcap ? fcap : (C1 == 256 ? 256 : 1024);
  if (blocks > want) blocks = want;
  hipStream_t st = pzn_hip_stream(stream);
  static const int xmap = [] { const char* e = getenv("PZN_SP_XCD"); return e ? atoi(e) : 1; }();  // tuning aid
  if (blocks >= 8) blocks &= ~7L;
  const dim3 grid((unsigned)blocks), block(SP_T);
  const int ldw = 3 + D;
  if (C1 == 64)
    PZN_LAUNCH((sa_point_l1_fwd_kernel<4, 4>), grid, block, 0, st, xyz, new_xyz, idx, P, W1, ldw, b1, N, S, rows, h,
                       xmap);
  else if (C1 == 128)
    PZN_LAUNCH((sa_point_l1_fwd_kernel<4, 2>), grid, block, 0, st, xyz, new_xyz, idx, P, W1, ldw, b1, N, S, rows, h,
                       xmap);
  else
    PZN_LAUNCH((sa_point_l1_fwd_kernel<4, 1>), grid, block, 0, st, xyz, new_xyz, idx, P, W1, ldw, b1, N, S, rows, h,
                       xmap);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT int pzn_knn_inverse_lists(const int64_t* idx, int B, int N, int S, int K, int32_t* off, int32_t* rows,
                                     int32_t* pts, pzn_stream_t stream) {
  PZN_CHECK_ARG(idx && off && rows && B > 0 && N > 0 && S > 0 && K > 0 && B <= 65535);
  PZN_CHECK_ARG((long)S * K < 2147483647L);
  const size_t base = sizeof(int) * ((size_t)N + INV_T);
  if (base > 150 * 1024) return PZN_EUNSUPPORTED;
  const size_t with_rows = base + sizeof(int) * (size_t)S * K;
  const bool lrows = with_rows <= 150 * 1024;
  const size_t lds = lrows ? with_rows : base;
  const void* fn = lrows ? (const void*)sa_inverse_lists_kernel<true> : (const void*)sa_inverse_lists_kernel<false>;
  if (lds > 64 * 1024 && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return PZN_ELAUNCH;
  if (lrows)
    PZN_LAUNCH(sa_inverse_lists_kernel<true>, dim3((unsigned)B), dim3(INV_T), lds, pzn_hip_stream(stream), idx, N, S * K, off,
               rows, pts);
  else
    PZN_LAUNCH(sa_inverse_lists_kernel<false>, dim3((unsigned)B), dim3(INV_T), lds, pzn_hip_stream(stream), idx, N, S * K, off,
               rows, pts);
  PZN_RETURN_LAUNCH_STATUS();
}

static int sa_point_l1_bwd(const float* dh, const float* xyz, const float* new_xyz, const int32_t* rows,
                           const int32_t* pts, int B, int N, int S, int D, int C1, float* dP, float* dW1,
                           float* db1, const uint32_t* rowmask, pzn_stream_t stream);

PZN_EXPORT int pzn_sa_point_l1_bwd_f32(const float* dh, const float* xyz, const float* new_xyz, const int32_t* rows,
                                       const int32_t* pts, int B, int N, int S, int D, int C1, float* dP, float* dW1,
                                       float* db1, pzn_stream_t stream) {
  return sa_point_l1_bwd(dh, xyz, new_xyz, rows, pts, B, N, S, D, C1, dP, dW1, db1, nullptr, stream);
}

// the same with the row mask of pzn_sa_level_bwd_rm_f32: rows of dh without their bit are not read
PZN_EXPORT int pzn_sa_point_l1_bwd_rm_f32(const float* dh, const float* xyz, const float* new_xyz, const int32_t* rows,
                                          const int32_t* pts, int B, int N, int S, int D, int C1, float* dP, float* dW1,
                                          float* db1, const uint32_t* rowmask, pzn_stream_t stream) {
  return sa_point_l1_bwd(dh, xyz, new_xyz, rows, pts, B, N, S, D, C1, dP, dW1, db1, rowmask, stream);
}

static int sa_point_l1_bwd(const float* dh, const float* xyz, const float* new_xyz, const int32_t* rows,
                           const int32_t* pts, int B, int N, int S, int D, int C1, float* dP, float* dW1,
                           float* db1, const uint32_t* rowmask, pzn_stream_t stream) {
  PZN_CHECK_ARG(dh && xyz && rows && pts && dP && dW1 && B > 0 && N > 0 && S > 0 && D >= 0);      // (new_xyz may be NULL)
  PZN_CHECK_ARG((long)B * N < 2147483647L && (long)B * S * 32 < 2147483647L);
  if (C1 != 64 && C1 != 128 && C1 != 256) return PZN_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(dP) & 15) || (reinterpret_cast<uintptr_t>(dh) & 15)) return PZN_EUNSUPPORTED;
  hipStream_t st = pzn_hip_stream(stream);
  if (pzn_zero_async(dP, (size_t)B * N * C1, st) != PZN_OK) return PZN_ELAUNCH;  // points nobody gathered; list ends add
  const long entries = (long)B * S * 32;
  const long nbatch = (entries + 63) / 64;
  // ~64 KB of dh rows in flight per CU measured best: 16 rows per wavefront, 8 (4) wavefronts per CU at 512-byte
  // (1-KB) rows: 0.14 ms for the 537 MB of either level; 32 wavefronts per CU with 4 rows each took 0.21 / 0.29 ms.
  static const long cap = [] { const char* e = getenv("PZN_SP_GRID"); return e ? atol(e) : 0L; }();  // tuning aid
  static const int g8 = [] { const char* e = getenv("PZN_SP_G"); return e ? atoi(e) : 16; }();       // tuning aid
  long blocks = (nbatch + 3) / 4;
  const long want = cap ? cap : (C1 == 256 ? 256 : (C1 == 128 ? 512 : 1024));
  if (blocks > want) blocks = want;
  const dim3 grid((unsigned)blocks), block(SP_T);
  const int ldw = 3 + D;
#define PZN_SP_BWD(VV, GG)                                                                                              \
  PZN_LAUNCH((sa_point_l1_bwd_kernel<VV, GG>), grid, block, 0, st, dh, xyz, new_xyz, rows, pts, N, S, entries, dP, \
                     dW1, ldw, db1, rowmask)
  if (C1 == 64) {
    if (g8 == 16) PZN_SP_BWD(1, 16); else if (g8 == 8) PZN_SP_BWD(1, 8); else PZN_SP_BWD(1, 4);
  } else if (C1 == 128) {
    if (g8 == 16) PZN_SP_BWD(2, 16); else if (g8 == 8) PZN_SP_BWD(2, 8); else PZN_SP_BWD(2, 4);
  } else {
    if (g8 == 16) PZN_SP_BWD(4, 16); else if (g8 == 8) PZN_SP_BWD(4, 8); else PZN_SP_BWD(4, 4);
  }
#undef PZN_SP_BWD
  PZN_RETURN_LAUNCH_STATUS();
}
