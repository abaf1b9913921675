// pzn_mfma.h — building blocks of the chained matrix-core kernels (attnfused.hip via pzn_mfma16.h, salevel.hip, outproj.hip), gfx950 only:
// bf16x3 split precision, fragment reads from LDS as inline asm with counted waits (two tiles ahead), the LDS-DMA slab
// ring (three slots, two slabs in flight, one barrier per slab), compile-time loops.  Included inside an anonymous
// namespace by each translation unit (which includes <type_traits> and pzn_common.h first).
#pragma once


typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16x4 __attribute__((address_space(3))) * lds4_t;

constexpr int SLAB = 24576;               // 3 planes x 8 KB
constexpr int NT = 256;                   // threads per workgroup
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ uint32_t f2bf(float x) {
  __bf16 b = (__bf16)x;
  return (uint32_t)__builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf2f(uint32_t b) { return __uint_as_float(b << 16); }

typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// two fp32 values -> three dwords of two bf16 each (x = x1 + x2 + x3, every xi a round-to-nearest bf16 of the remainder):
// 3 v_cvt_pk_bf16_f32 + 4 unpack + 4 subtract
__device__ __forceinline__ void split_pair(float x0, float x1, uint32_t& a, uint32_t& bq, uint32_t& c) {
  const floatx2 x = {x0, x1};
  a = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
  const floatx2 r = {x0 - __uint_as_float(a << 16), x1 - __uint_as_float(a & 0xffff0000u)};
  bq = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
  const floatx2 t = {r[0] - __uint_as_float(bq << 16), r[1] - __uint_as_float(bq & 0xffff0000u)};
  c = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2));
}

// eight fp32 values -> three bf16x8 fragments (x = b0 + b1 + b2 exactly up to 2^-24)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&b)[3]) {
  uint32_t w[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], w[0][j], w[1][j], w[2][j]);
#pragma unroll
  for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]));
}

// The B fragment of the NEXT k-step, built a pair of values at a time behind the MFMAs of the current one
// (BNext::pair(j), j = 0..3, from the step's fill callback), so that its ~44 vector instructions sit in the shadow of
// the matrix pipe instead of in front of the step (measured: 660 cycles per step when issued as one block).
struct BNext {
  uint32_t w[3][4];
  template <bool NEG = false>
  __device__ __forceinline__ void pair(const floatx16& x, int s, int j) {
    const float x0 = x[8 * s + 2 * j], x1 = x[8 * s + 2 * j + 1];
    split_pair(NEG ? -x0 : x0, NEG ? -x1 : x1, w[0][j], w[1][j], w[2][j]);
  }
  // the same with the values gated by bits (16 s' + 8 s + 2 j) and the next one of `word` (a ReLU mask)
  __device__ __forceinline__ void pair_gated(const floatx16& x, int s, int j, uint32_t word, int bit0) {
    const int i = 8 * s + 2 * j;
    const float x0 = (word >> (bit0 + i)) & 1u ? x[i] : 0.f, x1 = (word >> (bit0 + i + 1)) & 1u ? x[i + 1] : 0.f;
    split_pair(x0, x1, w[0][j], w[1][j], w[2][j]);
  }
  __device__ __forceinline__ void get(bf16x8 (&b)[3]) const {
#pragma unroll
    for (int p = 0; p < 3; ++p) b[p] = __builtin_bit_cast(bf16x8, make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]));
  }
};

// registers 8s .. 8s+7 of an accumulator tile as the B fragment of k-step s (s = 0, 1), optionally negated
template <bool NEG = false>
__device__ __forceinline__ void make_b(const floatx16& x, int s, bf16x8 (&b)[3]) {
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = NEG ? -x[8 * s + j] : x[8 * s + j];
  split8(v, b);
}

// acc += A B with A, B in three planes each: the six products >= 2^-16, small terms first
__device__ __forceinline__ floatx16 mma6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], floatx16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
  return c;
}

struct NoGate {};
struct NoFill {
  __device__ __forceinline__ void operator()(int) const {}
};

// LDS fragment reads are inline asm with explicit waits: one wavefront per SIMD means nobody else hides the LDS latency,
// so the fragments of the NEXT tile must be in flight while the six MFMAs of the current one issue.  Left to the
// compiler (256 VGPRs all in use) the reads are re-issued one by one right in front of their MFMA and waited for with
// lgkmcnt(0) three times per tile (measured: 2.6k cycles per 48-MFMA step against 1.5k of matrix-pipe time).
// An asm read is invisible to the compiler's wait bookkeeping: RP_WAIT / TR_WAIT (s_waitcnt lgkmcnt(0) naming every
// destination register read-write) must precede the first use (cdna_hip_programming.md, 5.7 form (ii)).
#define RP_ISSUE(ADDR, O0, O1, O2, A0, A1, A2)                                                       \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A0) : "v"(ADDR), "n"(O0));                      \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A1) : "v"(ADDR), "n"(O1));                      \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(A2) : "v"(ADDR), "n"(O2))
#define RP_WAITN(N_, A0, A1, A2) asm volatile("s_waitcnt lgkmcnt(" #N_ ")" : "+v"(A0), "+v"(A1), "+v"(A2))

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ floatx16 mma6v(bf16x8 a0, bf16x8 a1, bf16x8 a2, const bf16x8 (&b)[3], floatx16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[0], c, 0, 0, 0);
  return c;
}

// one k-step of acc[rt] += A_rt B, A from an Rp slab in LDS: [plane][RT][lane][16 B].  lane_addr = LDS byte address of
// the slab + 16 lane: ONE address register per ring slot, the tile and plane go into the instruction's offset field
// (left as address arithmetic, the compiler keeps a register per (slot, tile) alive through the whole kernel: ~100
// VGPRs, spilled).  Fragments are requested TWO tiles ahead (three register sets): one tile of MFMAs (192 cycles) does
// not cover the LDS latency when the four wavefronts of the workgroup, in lockstep behind the step's barrier, read at
// the same moment.  The wait before tile rt is counted: lgkmcnt(6 / 3) leaves the reads of the following tiles in
// flight (LDS returns in order).  fill(rt) is called behind the MFMAs of tile rt: the step's DMA instructions and the
// construction of the next B fragment go there, spread under the matrix pipe instead of standing in front of it.
// LDS_IS_B: the register operand is the MFMA's A (rows of the result = ITS rows), the LDS fragments are B.
// AHEAD = 1: one tile ahead, two register sets (12 registers fewer; enough where a second wavefront shares the SIMD).
template <int RT, class Fill = NoFill, bool LDS_IS_B = false, int AHEAD = 2>
__device__ __forceinline__ void kstep_rp(floatx16 (&acc)[RT], uint32_t lane_addr, const bf16x8 (&b)[3],
                                         const Fill& fill = Fill()) {
  static_assert(RT >= 4 && (AHEAD == 1 || AHEAD == 2), "pipeline depth");
  constexpr int NS = AHEAD + 1;
  bf16x8 f[NS][3];
  RP_ISSUE(lane_addr, 0, RT * 1024, 2 * RT * 1024, f[0][0], f[0][1], f[0][2]);
  if constexpr (AHEAD == 2) RP_ISSUE(lane_addr, 1024, (RT + 1) * 1024, (2 * RT + 1) * 1024, f[1][0], f[1][1], f[1][2]);
  static_for<0, RT>([&](auto ic) {
    constexpr int rt = decltype(ic)::value;
    constexpr int cur = rt % NS, nxt = (rt + AHEAD) % NS;
    if constexpr (rt + AHEAD < RT) {
      RP_ISSUE(lane_addr, (rt + AHEAD) * 1024, (RT + rt + AHEAD) * 1024, (2 * RT + rt + AHEAD) * 1024, f[nxt][0], f[nxt][1], f[nxt][2]);
      if constexpr (AHEAD == 2)
        RP_WAITN(6, f[cur][0], f[cur][1], f[cur][2]);
      else
        RP_WAITN(3, f[cur][0], f[cur][1], f[cur][2]);
    } else if constexpr (rt + 1 < RT && AHEAD == 2) {
      RP_WAITN(3, f[cur][0], f[cur][1], f[cur][2]);
    } else {
      RP_WAITN(0, f[cur][0], f[cur][1], f[cur][2]);
    }
    if constexpr (LDS_IS_B)
      acc[rt] = mma6v(b[0], b[1], b[2], f[cur], acc[rt]);
    else
      acc[rt] = mma6v(f[cur][0], f[cur][1], f[cur][2], b, acc[rt]);
    fill(rt);
  });
}

// =====================================================================================================================
// Forms with the number of planes as a template parameter (the attention kernels, pzn_mfma16.h):
//   NPL = 3: split precision (x = x1 + x2 + x3, six MFMAs per product, fp32-GEMM accuracy) - the default path;
//   NPL = 1: every operand rounded to ONE bf16 plane, one MFMA per product, fp32 accumulation (the opt-in bf16 attention
//            mode, pzn_attn_set_precision(1) / BASELINE configs[4]).  Images and slabs keep their layout; only plane 0 is
//            written, fetched and read.

// asm fragment reads (see RP_ISSUE): NPL reads per tile, counted waits in units of reads
template <int NPL, int O0, int O1, int O2>
__device__ __forceinline__ void rp_issue(uint32_t addr, bf16x8 (&a)[3]) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[0]) : "v"(addr), "n"(O0));
  if constexpr (NPL == 3) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[1]) : "v"(addr), "n"(O1));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[2]) : "v"(addr), "n"(O2));
  }
}
template <int NPL, int N>
__device__ __forceinline__ void rp_wait(bf16x8 (&a)[3]) {
  if constexpr (NPL == 3)
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[0]) : "n"(N));
}

// ---- transposed reads of a plane image ("T use") -------------------------------------------------------------------
// One image of M[n][F] serves both kinds of product.  With k = feature it is streamed in plain slabs (ds_read_b128).
// With k = ROW n (the operand is M^T[f][n]) the same bytes are fetched in a different cut - the LDS-DMA's per-lane
// source address does the re-arrangement - and read with ds_read_b64_tr_b16 (hardware transpose of 4 x 16 blocks of
// 16-bit elements: in a 16-lane group lane 4q+p supplies the 8-byte unit (row q, columns 4p..4p+3) and receives column
// (lane & 15), rows 0..3).  The 8-byte halves of an image chunk are exactly such units (pzn_mfma16.h has the cut).
// the fragment of one tile: lo / hi halves of NPL planes (2 NPL reads), counted waits in units of reads
struct TrFrag {
  bf16x4 lo[3], hi[3];
};
template <int NPL, int N>
__device__ __forceinline__ void tr_wait(TrFrag& t) {
  if constexpr (NPL == 3)
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(t.lo[0]), "+v"(t.hi[0]), "+v"(t.lo[1]), "+v"(t.hi[1]), "+v"(t.lo[2]), "+v"(t.hi[2]) : "n"(N));
  else
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(t.lo[0]), "+v"(t.hi[0]) : "n"(N));
}
template <int NPL>
__device__ __forceinline__ void tr_join(const TrFrag& t, bf16x8 (&a)[3]) {
#pragma unroll
  for (int p = 0; p < NPL; ++p) a[p] = __builtin_shufflevector(t.lo[p], t.hi[p], 0, 1, 2, 3, 4, 5, 6, 7);
}

// ---- slab ring: global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave-instruction, destination =
// wave-uniform base + lane * 16), three slots, two slabs in flight.  Step c of a kernel's slab sequence is
//     step_sync(pieces of slab c+1)      (this wave's pieces of slab c have landed; slab c+1 may still be in flight;
//                                         barrier: everybody's pieces have landed, slot (c+2) % 3 is no longer read)
//     ring.issue(slab c+2 -> slot (c+2) % 3)
//     multiply slab c
// vmcnt counts in order, so other loads / stores issued in between only make a wait more conservative.
struct Ring {
  unsigned char* lds;
  int wave, lane;
  int slot_bytes;
  int nw = 4;            // wavefronts of the workgroup that share the issue of a slab's pieces
  // NPW pieces per wavefront; piece j of the slab = bytes [1024 j, 1024 j + 1024) of src
  template <int NPW>
  __device__ __forceinline__ void issue(const unsigned char* src, int slot) const {
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
      const int piece = i * nw + wave;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(lds + slot * slot_bytes + piece * 1024), 16, 0, 0);
    }
  }
  __device__ __forceinline__ const unsigned char* slot(int s) const { return lds + s * slot_bytes; }
  __device__ __forceinline__ uint32_t lane_addr(int s) const { return slot_addr(s) + (uint32_t)lane * 16u; }
  __device__ __forceinline__ uint32_t slot_addr(int s) const {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (uint32_t)(s * slot_bytes);
  }
};

template <int N>
__device__ __forceinline__ void wait_vm_sync() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// younger = this wavefront's vector-memory instructions that may stay in flight: the DMA pieces of the NEXT slab plus
// whatever else it issued after the last piece of the slab waited for (vmcnt counts in order).  The argument folds to a
// constant in the unrolled loops.
__device__ __forceinline__ void step_sync(int younger) {
  switch (younger) {
    case 1: wait_vm_sync<1>(); break;
    case 2: wait_vm_sync<2>(); break;
    case 3: wait_vm_sync<3>(); break;
    case 4: wait_vm_sync<4>(); break;
    case 5: wait_vm_sync<5>(); break;
    case 6: wait_vm_sync<6>(); break;
    case 7: wait_vm_sync<7>(); break;
    case 8: wait_vm_sync<8>(); break;
    case 9: wait_vm_sync<9>(); break;
    case 12: wait_vm_sync<12>(); break;
    default: wait_vm_sync<0>(); break;   // (0 or an unknown count: drain)
  }
}

// XCD-aware block id: consecutive logical ids (the two halves of a cloud, neighbouring clouds) share an XCD's L2
__device__ __forceinline__ int logical_block(int bid, int nb) {
  if (nb & 7) return bid;
  return (bid & 7) * (nb >> 3) + (bid >> 3);
}
