// pzn_mfma16.h — the chained attention kernels' building blocks on 16-row tiles (v_mfma_f32_16x16x32_bf16), gfx950 only.
// Included after pzn_mfma.h inside the same anonymous namespace (split_pair, static_for, Ring, step_sync, rp_issue ...).
//
// Why a second tile shape: with 32-row tiles a wavefront's accumulator sets are 128 registers each, a kernel needs two or
// three of them, and one wavefront fills a SIMD's register file — nobody hides its LDS latencies, its DMA issue, its
// B-fragment splits or its memory phases.  With 16 rows per wavefront the sets are 64 registers, a kernel stays under
// 256, and a workgroup of EIGHT wavefronts (two per SIMD) covers the same 128 rows with the same slabs: same bytes per
// flop from L2, twice the wavefronts in flight.
//
//   v_mfma_f32_16x16x32_bf16: D[i][n] += sum_k A[i][k] B[k][n]; lane (c = l & 15, g = l >> 4) holds A[c][8g .. 8g+7],
//   B[8g .. 8g+7][c] and D[4g + r][c], r = 0..3.  A result tile has its COLUMN (the point) on the lane and four rows in
//   registers; two tiles of 16 rows each (features 16 t0 .. and 16 t1 ..) are the B operand of a following product's
//   k-step of 32: element j of lane group g is feature perm16(g, j) = 16 (j >> 2) + 4 g + (j & 3) of that step.
//
// Rp16 image of M[rows][K] (bf16 planes, 16-byte chunks): [k-step of 32][plane][row tile of 16][lane][8], the chunk of
// lane (c, g) = M[16 rt + c][32 ks + perm16(g, j)].  Consumed as in pzn_mfma.h: plain half-slabs (8 row tiles x 3 planes =
// 24 KB, ds_read_b128), own rows straight from global, or transposed (T use): a k-step of 32 ROWS n is the two row tiles
// 2 kk, 2 kk + 1; the fragment of feature tile ft (16 features) is two ds_read_b64_tr_b16 per plane — in a 16-lane group
// lane 4q+p supplies the 8-byte unit (row 4g + q [+ 16], features 16 ft + 4p ..), which is half (ft & 1) of the chunk of
// lane (c = 4g + q, g' = p) of k-step ft >> 1; the DMA places chunk (c, g') at position 4c + g' of its 1 KB block, so the
// reading lane's address is simply block + 16 lane + 8 (ft & 1).  (Only one half of every chunk belongs to a given feature
// tile: a transposed read uses half the banks, 2 LDS cycles per instruction more than the 32-row form.)
#pragma once

typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int NT16 = 512;                          // threads per workgroup: eight wavefronts, two per SIMD
constexpr int STG16_LD = 136;                      // dwords per staged row (16-byte units: 34 per row -> conflict-free both ways)
constexpr int STG16_BYTES = 16 * STG16_LD * 4;     // 8,704 per wavefront: 16 rows x 128 features per pass

template <int NPL>
__device__ __forceinline__ floatx4 mma16(bf16x8 a0, bf16x8 a1, bf16x8 a2, const bf16x8 (&b)[3], floatx4 c) {
  if constexpr (NPL == 3) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[0], c, 0, 0, 0);
    return c;
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[0], c, 0, 0, 0);
  }
}

// B fragment of a k-step from the two accumulator tiles t0 (features 0..15 of the step) and t1 (16..31), a pair of
// values at a time: pair jj = 0, 1 -> t0 registers 2jj, 2jj+1; jj = 2, 3 -> t1 registers 2(jj-2), 2(jj-2)+1
template <int NPL>
struct BNext16 {
  uint32_t w[3][4];
  __device__ __forceinline__ void put(float x0, float x1, int jj) {
    if constexpr (NPL == 3) {
      split_pair(x0, x1, w[0][jj], w[1][jj], w[2][jj]);
    } else {
      const floatx2 x = {x0, x1};
      w[0][jj] = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
    }
  }
  template <bool NEG = false>
  __device__ __forceinline__ void pair(const floatx4& t0, const floatx4& t1, int jj) {
    const float x0 = jj < 2 ? t0[2 * jj] : t1[2 * jj - 4], x1 = jj < 2 ? t0[2 * jj + 1] : t1[2 * jj - 3];
    put(NEG ? -x0 : x0, NEG ? -x1 : x1, jj);
  }
  // gated by bits (bit0 + 2 jj) and the next one of `word` (bit0 = the first tile's base bit: the second tile follows it)
  __device__ __forceinline__ void pair_gated(const floatx4& t0, const floatx4& t1, int jj, uint32_t word, int bit0) {
    const float x0 = jj < 2 ? t0[2 * jj] : t1[2 * jj - 4], x1 = jj < 2 ? t0[2 * jj + 1] : t1[2 * jj - 3];
    put((word >> (bit0 + 2 * jj)) & 1u ? x0 : 0.f, (word >> (bit0 + 2 * jj + 1)) & 1u ? x1 : 0.f, jj);
  }
  __device__ __forceinline__ void get(bf16x8 (&b)[3]) const {
#pragma unroll
    for (int p = 0; p < NPL; ++p) b[p] = __builtin_bit_cast(bf16x8, make_uint4(w[p][0], w[p][1], w[p][2], w[p][3]));
  }
};
template <int NPL, bool NEG = false>
__device__ __forceinline__ void make_b16(const floatx4& t0, const floatx4& t1, bf16x8 (&b)[3]) {
  BNext16<NPL> t;
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) t.template pair<NEG>(t0, t1, jj);
  t.get(b);
}

// one half-slab of a k-step: acc[t] += A_t B for the 8 row tiles of a plain slab [plane][8][lane][16 B]; two tiles
// ahead, counted waits (see kstep_rp_n)
template <int NPL, class Fill = NoFill>
__device__ __forceinline__ void kstep16(floatx4* acc, uint32_t lane_addr, const bf16x8 (&b)[3], const Fill& fill = Fill()) {
  constexpr int RT = 8;
  bf16x8 f[3][3];
  rp_issue<NPL, 0, RT * 1024, 2 * RT * 1024>(lane_addr, f[0]);
  rp_issue<NPL, 1024, (RT + 1) * 1024, (2 * RT + 1) * 1024>(lane_addr, f[1]);
  static_for<0, RT>([&](auto ic) {
    constexpr int rt = decltype(ic)::value;
    constexpr int cur = rt % 3, nxt = (rt + 2) % 3;
    if constexpr (rt + 2 < RT) {
      rp_issue<NPL, (rt + 2) * 1024, (RT + rt + 2) * 1024, (2 * RT + rt + 2) * 1024>(lane_addr, f[nxt]);
      rp_wait<NPL, 2 * NPL>(f[cur]);
    } else if constexpr (rt + 1 < RT) {
      rp_wait<NPL, NPL>(f[cur]);
    } else {
      rp_wait<NPL, 0>(f[cur]);
    }
    acc[rt] = mma16<NPL>(f[cur][0], f[cur][1], f[cur][2], b, acc[rt]);
    fill(rt);
  });
}

// transposed fragments of a T-use (half-)slab: NT tiles of 16 features, tile t at KOFF + plane PLS + 2048 (t >> 1) + 8 (t & 1),
// lo half of the k-step at that address, hi half (rows 16 ..) 1024 bytes further.  la = slab address + 16 lane.
template <int NPL, int OFF, int PLS>
__device__ __forceinline__ void tr16_issue(uint32_t addr, TrFrag& t) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.lo[0]) : "v"(addr), "n"(OFF));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.hi[0]) : "v"(addr), "n"(OFF + 1024));
  if constexpr (NPL == 3) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.lo[1]) : "v"(addr), "n"(OFF + PLS));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.hi[1]) : "v"(addr), "n"(OFF + PLS + 1024));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.lo[2]) : "v"(addr), "n"(OFF + 2 * PLS));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(t.hi[2]) : "v"(addr), "n"(OFF + 2 * PLS + 1024));
  }
}
template <int NTL, int PLS, int KOFF, int NPL, class Fill = NoFill>
__device__ __forceinline__ void kstep16_tr(floatx4* acc, uint32_t la, const bf16x8 (&b)[3], const Fill& fill = Fill()) {
  TrFrag t[3];
  tr16_issue<NPL, KOFF, PLS>(la, t[0]);
  tr16_issue<NPL, KOFF + 8, PLS>(la, t[1]);
  static_for<0, NTL>([&](auto ic) {
    constexpr int ft = decltype(ic)::value;
    constexpr int cur = ft % 3, nxt = (ft + 2) % 3;
    if constexpr (ft + 2 < NTL) {
      tr16_issue<NPL, KOFF + 2048 * ((ft + 2) >> 1) + 8 * ((ft + 2) & 1), PLS>(la, t[nxt]);
      tr_wait<NPL, 4 * NPL>(t[cur]);
    } else if constexpr (ft + 1 < NTL) {
      tr_wait<NPL, 2 * NPL>(t[cur]);
    } else {
      tr_wait<NPL, 0>(t[cur]);
    }
    bf16x8 a[3];
    tr_join<NPL>(t[cur], a);
    acc[ft] = mma16<NPL>(a[0], a[1], a[2], b, acc[ft]);
    fill(ft);
  });
}
// per-lane part of the source address of a T-use piece (one 1 KB image block): the lane that writes position L fetches
// chunk (c = L >> 2, g' = L & 3), which sits at chunk index 16 g' + c of the block
__device__ __forceinline__ uint32_t tr16_src_lane_off(int lane) { return (uint32_t)((16 * (lane & 3) + (lane >> 2)) * 16); }

// ---- 16 rows of a wavefront <-> row-major fp32 memory through its staging buffer (8 tiles = 128 features per pass)
struct Gate16 {
  uint32_t w[2];      // bit (t & 7) * 4 + r of word t >> 3: register r of tile t is live
};
template <int NTL, class G = NoGate>
__device__ __forceinline__ void stage_put16(float* stg, const floatx4* x, int ft0, int lane, const G& gate = G()) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[r] = x[ft0 + t][r];
      if constexpr (std::is_same<G, Gate16>::value) {
        if (!((gate.w[(ft0 + t) >> 3] >> (((ft0 + t) & 7) * 4 + r)) & 1u)) v[r] = 0.f;
      }
    }
    *reinterpret_cast<float4*>(stg + c * STG16_LD + 16 * t + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
  }
}
template <int NTL, bool ADD = false>
__device__ __forceinline__ void stage_get16(const float* stg, floatx4* x, int ft0, int lane) {
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(stg + c * STG16_LD + 16 * t + 4 * g);
    if (ADD) {
      x[ft0 + t][0] += v.x, x[ft0 + t][1] += v.y, x[ft0 + t][2] += v.z, x[ft0 + t][3] += v.w;
    } else {
      x[ft0 + t][0] = v.x, x[ft0 + t][1] = v.y, x[ft0 + t][2] = v.z, x[ft0 + t][3] = v.w;
    }
  }
}
// FT tiles (16 FT features) of the wavefront's 16 rows.  MODE 0: store (x scale); 1: out = old + scale * tile.
// addsrc != NULL: the stored value is tile + addsrc row (rows of ld_add floats, same columns): a residual added in row
// layout, where its load is coalesced, instead of being kept in registers.
template <int FT, int MODE = 0, class G = NoGate>
__device__ __forceinline__ void store_rows16(float* base, long row0, int ld, const floatx4* x, float* stg, int lane,
                                             float scale = 1.f, const G& gate = G(), const float* addsrc = nullptr, int ld_add = 0) {
  constexpr int PASS = FT >= 8 ? 8 : FT;            // tiles per pass
  constexpr int LPR = PASS * 4;                     // lanes per row (16 bytes each)
  constexpr int RPI = 64 / LPR;                     // rows per instruction
#pragma unroll
  for (int ft0 = 0; ft0 < FT; ft0 += PASS) {
    // what the store adds from memory first, all loads in flight together (behind a store the compiler must assume
    // aliasing: load, wait, store, load ... would be 16 / RPI round trips instead of one)
    constexpr int NI = 16 / RPI;
    float* g0 = base + row0 * ld + 16 * ft0 + 4 * (lane % LPR);
    float4 old[MODE == 1 ? NI : 1], add[NI];
    if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NI; ++i) old[i] = *reinterpret_cast<const float4*>(g0 + (long)(i * RPI + lane / LPR) * ld);
    }
    if (addsrc) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
        add[i] = *reinterpret_cast<const float4*>(addsrc + (row0 + i * RPI + lane / LPR) * ld_add + 16 * ft0 + 4 * (lane % LPR));
    }
    stage_put16<PASS, G>(stg, x, ft0, lane, gate);
    pzn::wave_lds_sync();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int rr = i * RPI + lane / LPR;
      float4 v = *reinterpret_cast<const float4*>(stg + rr * STG16_LD + 4 * (lane % LPR));
      if (MODE == 1) {
        const float4 o = old[i];
        v = make_float4(o.x + scale * v.x, o.y + scale * v.y, o.z + scale * v.z, o.w + scale * v.w);
      } else if (scale != 1.f) {
        v = make_float4(scale * v.x, scale * v.y, scale * v.z, scale * v.w);
      }
      if (addsrc) v = make_float4(v.x + add[i].x, v.y + add[i].y, v.z + add[i].z, v.w + add[i].w);
      *reinterpret_cast<float4*>(g0 + (long)rr * ld) = v;
    }
    pzn::wave_lds_sync();
  }
}
// Column sums of the wavefront's 16 rows over FT = 16 tiles (256 features): dst[f] = (MODE ? dst[f] : 0) + scale * sum over the 16
// rows of x[row][f].  Through the staging buffer like store_rows16 (a pass = 8 tiles = 128 features; lane l sums features
// 2 l, 2 l + 1 of the pass down the 16 staged rows, in row order: the same sum every run).
template <int MODE>
__device__ __forceinline__ void store_colsum16(float* dst, const floatx4* x, float* stg, int lane, float scale) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    float2* d = reinterpret_cast<float2*>(dst + 128 * p + 2 * lane);
    float2 old = make_float2(0.f, 0.f);
    if (MODE == 1) old = *d;
    stage_put16<8>(stg, x, 8 * p, lane);
    pzn::wave_lds_sync();
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float2 v = *reinterpret_cast<const float2*>(stg + c * STG16_LD + 2 * lane);
      sx += v.x, sy += v.y;
    }
    *d = make_float2(old.x + scale * sx, old.y + scale * sy);
    pzn::wave_lds_sync();
  }
}
// The loads of every pass are issued first (one memory round trip); until its pass is staged a loaded 16-byte unit waits
// in the destination tile of the same index (x has exactly FT of them) - `tmp` for the adding form, which needs x itself.
__device__ __forceinline__ float4 as_f4(const floatx4& v) { return make_float4(v[0], v[1], v[2], v[3]); }
template <int FT, bool ADD = false>
__device__ __forceinline__ void load_rows16(const float* base, long row0, int ld, floatx4* x, float* stg, int lane,
                                            floatx4* tmp = nullptr) {
  constexpr int PASS = FT >= 8 ? 8 : FT, NP = FT / PASS;
  constexpr int LPR = PASS * 4, RPI = 64 / LPR, NI = 16 / RPI;
  static_assert(NP * NI == FT, "one waiting unit per tile");
  floatx4* w = ADD ? tmp : x;
#pragma unroll
  for (int k = 0; k < FT; ++k)
    w[k] = *reinterpret_cast<const floatx4*>(base + (row0 + (k % NI) * RPI + lane / LPR) * ld + 16 * PASS * (k / NI) + 4 * (lane % LPR));
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      *reinterpret_cast<floatx4*>(stg + (i * RPI + lane / LPR) * STG16_LD + 4 * (lane % LPR)) = w[p * NI + i];
    pzn::wave_lds_sync();
    stage_get16<PASS, ADD>(stg, x, PASS * p, lane);
    pzn::wave_lds_sync();
  }
}
// x = a + b (rows of two tensors): all loads of both first, b's units wait in tmp (FT tiles the caller has no use for yet)
template <int FT>
__device__ __forceinline__ void load_rows16_sum(const float* a, int lda, const float* b, int ldb, long row0, floatx4* x,
                                                floatx4* tmp, float* stg, int lane) {
  constexpr int PASS = 8, NP = FT / PASS, LPR = 32, RPI = 2, NI = 8;
#pragma unroll
  for (int k = 0; k < FT; ++k) {
    x[k] = *reinterpret_cast<const floatx4*>(a + (row0 + (k % NI) * RPI + lane / LPR) * lda + 128 * (k / NI) + 4 * (lane % LPR));
    tmp[k] = *reinterpret_cast<const floatx4*>(b + (row0 + (k % NI) * RPI + lane / LPR) * ldb + 128 * (k / NI) + 4 * (lane % LPR));
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
#pragma unroll
    for (int i = 0; i < NI; ++i)
      *reinterpret_cast<floatx4*>(stg + (i * RPI + lane / LPR) * STG16_LD + 4 * (lane % LPR)) = x[p * NI + i] + tmp[p * NI + i];
    pzn::wave_lds_sync();
    stage_get16<PASS, false>(stg, x, PASS * p, lane);
    pzn::wave_lds_sync();
  }
}

// register-order tile images (tensors private to two kernels): [16-row tile][feature tile][lane][4]
template <int FT>
__device__ __forceinline__ void store_tiles16(float* img, long tile0, int lane, const floatx4* x) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
    *reinterpret_cast<float4*>(img + ((tile0 * FT + ft) * 64 + lane) * 4) = make_float4(x[ft][0], x[ft][1], x[ft][2], x[ft][3]);
}
template <int FT>
__device__ __forceinline__ void load_tiles16(const float* img, long tile0, int lane, floatx4* x) {
  float4 v[FT];
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) v[ft] = *reinterpret_cast<const float4*>(img + ((tile0 * FT + ft) * 64 + lane) * 4);
#pragma unroll
  for (int ft = 0; ft < FT; ++ft) x[ft][0] = v[ft].x, x[ft][1] = v[ft].y, x[ft][2] = v[ft].z, x[ft][3] = v[ft].w;
}

// Rp16 image of the wavefront's rows: k-step ks = tiles 2 ks, 2 ks + 1; rt = the wavefront's 16-row tile inside the cloud
template <int NKS, int NPL, bool NEG = false>
__device__ __forceinline__ void store_rp16(unsigned char* img, int rt, int lane, const floatx4* x) {
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    bf16x8 b[3];
    make_b16<NPL, NEG>(x[2 * ks], x[2 * ks + 1], b);
#pragma unroll
    for (int p = 0; p < NPL; ++p) *reinterpret_cast<bf16x8*>(img + (((ks * 3 + p) * 16 + rt) * 64 + lane) * 16) = b[p];
    __builtin_amdgcn_sched_barrier(0);
  }
}
// the wavefront's own rows of an Rp16 image as the B operand of k-step ks
template <int NPL>
__device__ __forceinline__ void own_frag16(const unsigned char* img, int ks, int rt, int lane, bf16x8 (&f)[3]) {
#pragma unroll
  for (int p = 0; p < NPL; ++p) f[p] = *reinterpret_cast<const bf16x8*>(img + (((ks * 3 + p) * 16 + rt) * 64 + lane) * 16);
}

// accumulator tiles <- bias (the row of a transposed result is the output feature)
template <int NTL>
__device__ __forceinline__ void bias_tiles16(floatx4* acc, const float* bias, int g) {
  float4 v[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) v[t] = *reinterpret_cast<const float4*>(bias + 16 * t + 4 * g);
#pragma unroll
  for (int t = 0; t < NTL; ++t) acc[t][0] = v[t].x, acc[t][1] = v[t].y, acc[t][2] = v[t].z, acc[t][3] = v[t].w;
}
#define ZERO_TILES16(A, N) _Pragma("unroll") for (int i_ = 0; i_ < (N); ++i_) A[i_] = floatx4{0.f, 0.f, 0.f, 0.f}

// lanes c, c + 16, c + 32, c + 48 hold the four row groups of one column: sums / maxima over a column's rows end with these
__device__ __forceinline__ float col_sum4(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float col_max4(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
