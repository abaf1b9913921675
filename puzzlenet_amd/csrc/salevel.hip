// salevel.hip — second shared-MLP layer + ReLU + max over the 32 neighbours on GENERATED rows, weights STREAMED
// (model5_b.py:452-454 / :459-461 behind the per-point first layer: rows relu(P'[idx[g,k]] + Q[g]), csrc/sapoint.hip).
//
// The weight-stationary kernel (wsgemm.hip, GATH) keeps a column slice of W2 in LDS.  For the second level
// (C1 = C2 = 256) only 64 of the 256 columns fit, so four workgroup slices each regenerate every row: gather + add + ReLU
// + split is ~65 vector instructions per 16 k, 6.7 vector instructions per MFMA (profiles/r2_sq_counters.txt), and the
// kernel is bound by vector issue at 54 % matrix-pipe busy.  Here a wavefront owns ALL C2 columns of its 32-row group
// (accumulators: C2/32 tiles), so a row is generated once, and W2 streams through LDS in 24 KB slabs instead (one
// 16-deep k-step of all 256 columns; two k-steps of 128): the slab ring of pzn_mfma.h — LDS-DMA, three slots, one
// barrier per slab —, fragment reads two tiles ahead with counted waits.  Eight wavefronts per workgroup (two per
// SIMD) walk the k-steps of one "round" (one group each) in lockstep; the weights are re-streamed from L2 every round.
//
// Orientation: the generated rows are the MFMA's A operand (lane = row of the group, 8 consecutive k, loaded straight
// from the per-point table at the neighbour's row: 32 bytes per lane and k-step), the weights its B operand, so the
// result has the output column on the lane and the group's 32 rows in the registers: bias, ReLU, max and arg-max over
// the 32 neighbours are register work plus one half-lane exchange, exactly as in the weight-stationary kernel
// (strict >: the lowest neighbour slot wins a tie).
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

#include "pzn_mfma.h"

// W2[C2][C1] -> planes in the order the slabs are consumed: [k-step][plane][column tile][lane][8 bf16], lane (r, h)
// holding W2[32 ct + r][16 ks + 8 h + 0..7] (natural k order: the A operand is generated in natural order too)
__global__ __launch_bounds__(256) void sa_pack_w_kernel(const float* __restrict__ W, int C1, int C2, unsigned char* __restrict__ dst) {
  const int CT = C2 / 32, total = (C1 / 16) * CT * 64;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    const int lane = c & 63, ct = (c >> 6) % CT, ks = (c >> 6) / CT;
    const int r = lane & 31, h = lane >> 5;
    const float* src = W + (size_t)(32 * ct + r) * C1 + 16 * ks + 8 * h;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[j];
    bf16x8 b[3];
    split8(v, b);
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(dst + ((((size_t)ks * 3 + p) * CT + ct) * 64 + lane) * 16) = b[p];
  }
}

struct SaArgs {
  const float* Pp;        // [B*N, C1] per-point table
  const float* Q;         // [G, C1] per-group offsets
  const int64_t* idx;     // [G, 32] neighbour indices inside the group's cloud
  const unsigned char* w; // weight planes (sa_pack_w_kernel)
  const float* bias;      // [C2]
  float* out;             // [G, C2]
  int32_t* argmax;        // [G, C2]
  int G, N, S;
  long long* dbg;         // (SA_STAMPS builds only)
};

#ifdef SA_STAMPS
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) a.dbg[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i)
#endif

constexpr int SA_WAVES = 8;

template <int C1, int CT>
__global__ __launch_bounds__(SA_WAVES * 64, 2) void sa_level_stream_kernel(SaArgs a) {
  constexpr int KS = C1 / 16;                 // k-steps
  constexpr int KPS = 8 / CT;                 // k-steps per 24 KB slab (CT = 8: 1, CT = 4: 2)
  constexpr int NSL = KS / KPS;               // slabs per round
  constexpr int KBLK = CT * 3 * 1024;         // bytes of one k-step inside a slab
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SLAB];
  __shared__ __attribute__((aligned(16))) float qlds[2][SA_WAVES][256];   // per wavefront: this round's row of Q, the next round's
  __shared__ float sbias[CT * 32];            // (a global load in the epilogue would be waited for with vmcnt(0): the ring's DMA too)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < CT * 32) sbias[tid] = a.bias[tid];
  __syncthreads();
  const int r = lane & 31, h = lane >> 5;
  Ring ring{lds, wave, lane, SLAB};
  ring.nw = SA_WAVES;
  const int per_round = gridDim.x * SA_WAVES;
  // XCD-aware: workgroups bid, bid + 8, ... share an XCD; consecutive LOGICAL ids (= consecutive groups = whole clouds) go
  // to the same XCD, so a cloud's per-point table is gathered through ONE L2 instead of all eight (PMC: the kernel's
  // fetch traffic was 4x the tables it reads)
  const int lb = logical_block((int)blockIdx.x, (int)gridDim.x);
  const int rounds = (a.G + per_round - 1) / per_round;
  // the slab sequence is the same every round: slab c of the kernel = slab c % NSL of the weights, ring slot c % 3.
  // The slab number is a compile-time constant at every call site (a run-time c % NSL made the compiler precompute all
  // 3 NSL source addresses as 64-bit register pairs, spill them, and reload each in front of its DMA instruction behind
  // a vmcnt(0)): uniform base + one per-lane offset register.
  const int total = rounds * NSL;
  const uint32_t voff = (uint32_t)wave * 1024u + (uint32_t)lane * 16u;
  auto issue_piece = [&](int c, auto slab, int i) {
    constexpr int SL = decltype(slab)::value;
    if (c < total) {
      const int slot = c % 3;
      uint32_t vo = voff;
      asm volatile("" : "+v"(vo));        // (opaque per call: nothing that depends on it can be hoisted out of the round loop)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(a.w + ((size_t)SL * SLAB + i * SA_WAVES * 1024) + vo),
          (__attribute__((address_space(3))) void*)(lds + slot * SLAB + (i * SA_WAVES + wave) * 1024), 16, 0, 0);
    }
  };
  auto issue_slab = [&](int c, auto slab) {
#pragma unroll
    for (int i = 0; i < 3; ++i) issue_piece(c, slab, i);
  };
  issue_slab(0, std::integral_constant<int, 0>{});
  issue_slab(1, std::integral_constant<int, 1 % NSL>{});
  // What a round needs besides the weights: the neighbour index of the group (fetched two rounds ahead), the group's row of
  // Q (copied into a wavefront-private kilobyte of LDS one round ahead, by the same DMA path: its 8 values per k-step
  // are then two broadcast LDS reads instead of two vector loads per lane), and the lane's row of P' (two 16-byte
  // loads per k-step, issued one step ahead of the step that splits them).  No dependent index -> address -> row round
  // trip stands at the head of a round.
  auto group_of = [&](int rd) {
    const int g = rd * per_round + lb * SA_WAVES + wave;
    return g < a.G ? g : a.G - 1;                                   // (idle wavefronts of the last round redo the last group)
  };
  auto prow_of = [&](int g, int j) { return a.Pp + ((size_t)(g / a.S) * a.N + j) * C1 + 8 * h; };   // lane's row, its k half
  const uint32_t qlds_wave = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)&qlds[0][wave][0];
  // copies 1 KB holding Q[g] into buffer `buf` of this wavefront (C1 = 128: rows g0, g0 + 1 with g0 = min(g, G - 2), so
  // that the copy never reads past the end of Q) -> LDS byte address of this lane's k half of the row
  auto q_fetch = [&](int g, int buf) -> uint32_t {
    const int g0 = C1 == 256 ? g : (g < a.G - 1 ? g : a.G - 2);
    uint32_t vo = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(vo));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const unsigned char*)(a.Q + (size_t)g0 * C1) + vo),
                                     (__attribute__((address_space(3))) void*)&qlds[buf][wave][0], 16, 0, 0);
    return qlds_wave + (uint32_t)(buf * SA_WAVES * 1024 + (g - g0) * C1 * 4 + 32 * h);
  };
  // Row loads and LDS reads are inline asm, waited for by the step's own waits: the row loads are OLDER than the slab
  // the step's vmcnt wait is for (loads return in order), so no wait of their own exists.  Left to the compiler, they
  // were waited for with vmcnt(0) right behind the DMA issue of every third step (its bookkeeping across the
  // conditional issue), and an LDS read it knows of makes it wait for every DMA in flight.
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int NSET = 2 * KPS;             // k-steps whose rows are in registers or in flight
  f32x4 pa[NSET][2], q0, q1;
#define SA_GLOAD(DST, PTR, OFF) \
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(DST) : "v"(PTR), "n"(OFF) : "memory")
#define SA_GLOAD2(PROW, KS_, S_)                                                                \
  do {                                                                                         \
    (void)&pa; /* (odr-use: asm operands alone do not make a generic lambda capture it) */      \
    SA_GLOAD(pa[S_][0], PROW, 64 * (KS_));                                                     \
    SA_GLOAD(pa[S_][1], PROW, 64 * (KS_) + 16);                                                \
  } while (0)
  // behind the step's wait: ties the consumers of set S_ to it
#define SA_GREADY(S_)                                          \
  do {                                                        \
    (void)&pa;                                                \
    asm volatile("" : "+v"(pa[S_][0]), "+v"(pa[S_][1]));      \
  } while (0)
#define SA_QREAD(ADDR, KS_)                                                                                        \
  do {                                                                                                            \
    (void)&q0, (void)&q1;                                                                                         \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q0) : "v"(ADDR), "n"(64 * (KS_)));                          \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q1) : "v"(ADDR), "n"(64 * (KS_) + 16));                     \
  } while (0)
  int g_cur = group_of(0), g_next = group_of(rounds > 1 ? 1 : 0);
  int j_next = (int)a.idx[(size_t)g_next * 32 + r];
  const float* prow = prow_of(g_cur, (int)a.idx[(size_t)g_cur * 32 + r]);
  const float* prow_n;
  uint32_t qaddr = q_fetch(g_cur, 0), qaddr_n;
  // The fragment of k-step ks+1 is built (add, ReLU, split: ~60 vector instructions) a pair of values at a time BEHIND
  // the MFMAs of k-step ks (kstep_rp's fill slots), and so are the step's memory instructions (row loads first, then the
  // three DMA pieces of slab c+2): issued as a block at the head of the step, the 56 memory instructions of the eight
  // wavefronts queue up in front of the one address unit (~700 of 4300 cycles per step during which neither wavefront
  // of a SIMD feeds the matrix pipe).  Sets: k-step ks lives in set ks % NSET; the row loads issued during step c
  // (k-steps (c+1) KPS + 1 ..) are complete behind the wait of step c+1 (vmcnt(3): only DMA pieces stay in flight) and
  // are split during it.
  bf16x8 af[3];
  BNext bn;
  auto pair_of = [&](auto setc, int j) {        // values 2j, 2j+1 of the set's row: relu(P'[idx] + Q), split
    constexpr int st = decltype(setc)::value;
    const f32x4 pv = pa[st][j >> 1], qv = (j >> 1) ? q1 : q0;
    const float x0 = fmaxf(pv[2 * (j & 1)] + qv[2 * (j & 1)], 0.f), x1 = fmaxf(pv[2 * (j & 1) + 1] + qv[2 * (j & 1) + 1], 0.f);
    uint32_t w0, w1, w2;
    split_pair(x0, x1, w0, w1, w2);
    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));   // (pins the work to its fill slot: the compiler sinks it to its use, the next step's head)
    bn.w[0][j] = w0, bn.w[1][j] = w1, bn.w[2][j] = w2;
  };
  static_for<0, KPS + 1>([&](auto kc) {
    constexpr int ks = decltype(kc)::value;
    SA_GLOAD2(prow, ks, ks);
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SA_GREADY(0);
  SA_QREAD(qaddr, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1));
#pragma unroll
  for (int j = 0; j < 4; ++j) pair_of(std::integral_constant<int, 0>{}, j);
  bn.get(af);
  floatx16 acc[CT];                       // starts from the bias (the column sits on the lane: one value per tile)
  auto acc_init = [&](int ct) __attribute__((always_inline)) {
    const float bv = sbias[32 * ct + r];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ct][e] = bv;
  };
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) acc_init(ct);
  // ReLU + max / arg-max over the 32 rows of group g, one column tile: element e of lane l = row (e&3) + 8 (e>>2) + 4 h,
  // column l & 31.  The max runs first, ReLU after (model5_b.py:453-454 has ReLU first: the same value, and the same
  // arg-max as long as the maximum is positive; if it is not, every ReLU'd row is 0 and the lowest slot wins: the fix-up
  // below), 3 instead of 5 vector instructions per element.
  auto epilogue_tile = [&](int ct, int g, bool live) __attribute__((always_inline)) {
    // (a chain of 15 compare-select pairs; as a tree - depth 4 - the same tile took 1230 instead of 870 cycles: measured)
    float best = acc[ct][0];
    int be = 0;                             // register of the maximum (its row below: no table of 16 row numbers in registers)
#pragma unroll
    for (int e = 1; e < 16; ++e) {
      const float v = acc[ct][e];
      const bool gt = v > best;
      best = gt ? v : best;
      be = gt ? e : be;
    }
    int bi = (be & 3) + 8 * (be >> 2) + 4 * h;
    // the other half's candidate: v_permlane32_swap (vector ALU; __shfl_xor is an LDS instruction, waited for with
    // lgkmcnt(0): every fragment read in flight with it)
    const auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const auto si = __builtin_amdgcn_permlane32_swap((uint32_t)bi, (uint32_t)bi, false, false);
    const float ob = __uint_as_float(h ? sb[0] : sb[1]);
    const int oi = (int)(h ? si[0] : si[1]);
    const bool take = ob > best || (ob == best && oi < bi);
    best = take ? ob : best;
    bi = take ? oi : bi;
    if (!(best > 0.f)) best = 0.f, bi = 0;
    if (live) {   // one store per lane: lower half the value, upper half the index
      float* dst = h ? reinterpret_cast<float*>(a.argmax) : a.out;
      dst[(size_t)g * (CT * 32) + 32 * ct + r] = h ? __int_as_float(bi) : best;
    }
  };
  // The epilogue of a round runs inside the FIRST k-step of the next one: tile 0 in front of it, tile rt+1 (and the reset
  // of its accumulator to the bias) behind the MFMAs of tile rt.  As a block between two rounds its ~50 vector
  // instructions per tile stand in front of the matrix pipe of every wavefront at once (~4000 of a round's 65000 cycles
  // with eight column tiles, of 20000 with four).
  int c = 0, g_prev = 0;
  bool live_prev = false;
  for (int rd = 0; rd < rounds; ++rd) {
    const int g = g_cur;
    const bool live = rd * per_round + lb * SA_WAVES + wave < a.G;
    prow_n = prow_of(g_next, j_next);
    const int g_next2 = group_of(rd + 2 < rounds ? rd + 2 : rounds - 1);
    // (asm for the same reason as the row loads: the step waits of this round cover it; a compiler-managed load is
    // waited for with vmcnt(0) at the end of the round, behind the epilogue's stores)
    int j_next2;
    asm volatile("global_load_dword %0, %1, off" : "=v"(j_next2) : "v"(a.idx + (size_t)g_next2 * 32 + r) : "memory");
    qaddr_n = q_fetch(g_next, (rd + 1) & 1);
    // (the last step of a round leaves the issue of its slab to here, behind the epilogue's stores, the index load and the
    // copy of Q: the DMA pieces of the NEXT slab are then the youngest memory operations at every step's wait)
    if (rd > 0) issue_slab(c + 1, std::integral_constant<int, 1 % NSL>{});
    static_for<0, NSL>([&](auto slc) {
      constexpr int sl = decltype(slc)::value;
      constexpr int S0 = NSL == 4 ? 0 : 2;        // (SA_STAMPS: the four steps whose phases are stamped)
      (void)S0;
      if (rd == 1 && sl >= S0 && sl < S0 + 4) STAMP(4 * (sl - S0));
      if (c + 1 < total)
        wait_vm_sync<3>();                // may stay in flight: the next slab's 3 DMA pieces
      else
        wait_vm_sync<0>();
      if (rd == 1 && sl >= S0 && sl < S0 + 4) STAMP(4 * (sl - S0) + 1);
      static_for<0, KPS>([&](auto kc) {   // the sets split during this step are complete behind that wait
        constexpr int st = (sl * KPS + 1 + decltype(kc)::value) % NSET;
        SA_GREADY(st);
      });
      if constexpr (sl == 0)
        if (rd > 0) epilogue_tile(0, g_prev, live_prev), acc_init(0);
      const uint32_t la = ring.lane_addr(c % 3);
      if (rd == 1 && sl >= S0 && sl < S0 + 4) STAMP(4 * (sl - S0) + 2);
      static_for<0, KPS>([&](auto kc) {
        constexpr int kk = decltype(kc)::value, ksn = sl * KPS + kk + 1, st = ksn % NSET;
        if constexpr (ksn < KS)
          SA_QREAD(qaddr, ksn);
        else
          SA_QREAD(qaddr_n, ksn - KS);
        const auto fill = [&](int rt) {
          const int t = kk * CT + rt;                          // tile of the step, 0..7
          if (rt == 0) {                                        // (the Q reads are older than the fragment reads tile 0 waited for)
            (void)&q0, (void)&q1;
            asm volatile("" : "+v"(q0), "+v"(q1));
          }
          constexpr int STRIDE = CT / 4;                        // the four pairs behind tiles 0, STRIDE, 2 STRIDE, 3 STRIDE
          if (rt % STRIDE == 0) pair_of(std::integral_constant<int, st>{}, rt / STRIDE);
          // memory instruction m of the step behind tile 1 + 2m (one k-step per slab) / 1 + m (two): the KPS row loads of
          // the k-steps split during the NEXT step, then the three DMA pieces of slab c + 2
          const int m = KPS == 1 ? ((t & 1) ? (t - 1) / 2 : -1) : t - 1;
          if constexpr (sl == 0 && kk == 0)
            if (rd > 0 && rt + 1 < CT) epilogue_tile(rt + 1, g_prev, live_prev), acc_init(rt + 1);
          static_for<0, KPS>([&](auto ic) {
            constexpr int i = decltype(ic)::value, ksl = (sl + 1) * KPS + 1 + i, sn = ksl % NSET;
            if (m == i) {
              if constexpr (ksl < KS)
                SA_GLOAD2(prow, ksl, sn);
              else
                SA_GLOAD2(prow_n, ksl - KS, sn);
            }
          });
          if constexpr (sl == 0 && KPS == 1) {   // (behind the epilogue's stores: the pieces stay the youngest memory operations)
            if (t == CT - 1) issue_slab(c + 2, std::integral_constant<int, (sl + 2) % NSL>{});
          } else if constexpr (sl + 1 < NSL) {
            if (m >= KPS && m < KPS + 3) issue_piece(c + 2, std::integral_constant<int, (sl + 2) % NSL>{}, m - KPS);
          }
        };
        kstep_rp<CT, decltype(fill), true>(acc, la + kk * KBLK, af, fill);
        bn.get(af);
      });
      if (rd == 1 && sl >= S0 && sl < S0 + 4) STAMP(4 * (sl - S0) + 3);
      ++c;
    });
    if (rd == 1) STAMP(16);
    asm volatile("" : "+v"(j_next2));
    g_cur = g_next, g_next = g_next2, j_next = j_next2;
    prow = prow_n, qaddr = qaddr_n;
    g_prev = g, live_prev = live;
  }
  // (the row loads and the index load issued during the last steps - for a round that does not exist - are still in flight,
  // and the compiler believes their registers dead: drained before the epilogue's addresses and values reuse them.
  // outproj.hip has the measured failure.)
  asm volatile("s_waitcnt vmcnt(0) ; pzn_drain" ::: "memory");
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) epilogue_tile(ct, g_prev, live_prev);     // the last round's
}

}  // namespace

size_t pzn_sa_level_stream_workspace_bytes(int C1, int C2) {
  // (the two production shapes, model5_b.py:449-461; the mixed shapes compile, but with in-flight row registers spilled:
  // tests/test_isa_forms.py checks the instantiated ones)
  return (C1 == C2 && (C1 == 128 || C1 == 256)) ? (size_t)C1 * C2 * 6 : 0;
}

// -> PZN_EUNSUPPORTED for shapes it does not take (the weight-stationary kernel then)
static bool sa_stream_on() {
  constexpr bool on = true;   // tuning aid
  return on;
}

// the split of W2 into its plane image alone (the workspace a later pzn_sa_level_stream(..., prepacked = 1) reads)
int pzn_sa_level_stream_pack(const float* W2, int C1, int C2, void* workspace, hipStream_t st) {
  if (!sa_stream_on() || !workspace || pzn_sa_level_stream_workspace_bytes(C1, C2) == 0 ||
      (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
    return PZN_EUNSUPPORTED;
  PZN_LAUNCH(sa_pack_w_kernel, dim3(64), dim3(256), 0, st, W2, C1, C2, static_cast<unsigned char*>(workspace));
  PZN_RETURN_LAUNCH_STATUS();
}

int pzn_sa_level_stream(const float* Pp, const float* Q, const int64_t* idx, const float* W2, const float* b2, int G, int N,
                        int S, int C1, int C2, float* out, int32_t* argmax, void* workspace, hipStream_t st, int prepacked) {
  if (!sa_stream_on() || !workspace || pzn_sa_level_stream_workspace_bytes(C1, C2) == 0 || G < SA_WAVES) return PZN_EUNSUPPORTED;
  if (((reinterpret_cast<uintptr_t>(Pp) | reinterpret_cast<uintptr_t>(Q) | reinterpret_cast<uintptr_t>(workspace)) & 15) != 0)
    return PZN_EUNSUPPORTED;
  unsigned char* w = static_cast<unsigned char*>(workspace);
  if (!prepacked) {
    PZN_LAUNCH(sa_pack_w_kernel, dim3(64), dim3(256), 0, st, W2, C1, C2, w);
    if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  }
  SaArgs a{Pp, Q, idx, w, b2, out, argmax, G, N, S, nullptr};
#ifdef SA_STAMPS
  {
    static long long* d = nullptr;
    if (!d) hipMalloc(&d, 32 * sizeof(long long));
    a.dbg = d;
  }
#endif
  int gx = (G + SA_WAVES - 1) / SA_WAVES;
  if (gx > 256) gx = 256;
  if (C1 == 256)
    PZN_LAUNCH((sa_level_stream_kernel<256, 8>), dim3(gx), dim3(SA_WAVES * 64), 0, st, a);
  else
    PZN_LAUNCH((sa_level_stream_kernel<128, 4>), dim3(gx), dim3(SA_WAVES * 64), 0, st, a);
#ifdef SA_STAMPS
  {
    long long hst[17];
    hipStreamSynchronize(st);
    hipMemcpy(hst, a.dbg, sizeof(hst), hipMemcpyDeviceToHost);
    fprintf(stderr, "sa stamps (wait, rows, split+reads+mfma | next):");
    for (int i = 0; i < 16; ++i) fprintf(stderr, " %lld", hst[i + 1] - hst[i]);
    fprintf(stderr, "\n");
  }
#endif
  PZN_RETURN_LAUNCH_STATUS();
}
