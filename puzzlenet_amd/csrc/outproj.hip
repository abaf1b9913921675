// outproj.hip — the encoder's out projection + max over the points in ONE launch (model5_b.py:466-475):
//     out[b,l,:] = cat(att1, att2, att3, att4, f2f)[b,l,:] W_out^T + b_out        (1280 -> 1024, never concatenated)
//     f_global[b,:] = max_l out[b,l,:],   arg[b,:] = its point (the lowest one on ties, as torch.max)
// predict5 uses only f_global (model5_b.py:723): `out` is written only when the caller asks for it.
//
// Same machine as salevel.hip (pzn_mfma.h): a workgroup = eight wavefronts (two per SIMD) = the 256 points of ONE cloud x
// one 256-column group of W_out; a wavefront owns 32 points x 256 columns (eight accumulator tiles) for the whole
// reduction of 1280 = 5 slices x 16 k-steps.  The rows are the MFMA's A operand, straight from the five activation
// tensors (two 16-byte loads per lane and k-step, one step ahead, inline asm covered by the step's vmcnt wait), split in
// the shadow of the MFMAs; the column group's weights (three bf16 planes in fragment order, split once per launch into
// the caller's workspace) stream through the three-slot LDS ring by LDS-DMA, one barrier per k-step.  A launch is
// 4 B workgroups in XCD-aware order (a column group's 1.97 MB of planes is read by the clouds of two XCDs only).
// Epilogue: bias is the accumulators' start value; the max over a wavefront's 32 points is register work + one
// v_permlane32_swap, the eight wavefronts meet in LDS in ascending order (strict >: first maximum).
// Before: five general-engine launches (67 us each at B = 64: 16384 x 256 x 1024 at 0.31 of the ceiling) + a pass over the
// 67 MB of `out` for the maximum.
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "pzn_common.h"
#include "pzn_internal.h"

namespace {

#include "pzn_mfma.h"

constexpr int OP_WAVES = 8;
constexpr int OP_E = 256;          // columns of one slice of the concatenation
constexpr int OP_SLICES = 5;
constexpr int OP_L = 256;          // points per cloud = rows of a workgroup
constexpr int OP_NOUT = 1024;
constexpr int OP_KS = OP_E / 16;   // k-steps per slice
constexpr int OP_CT = 8;           // column tiles of a workgroup (256 columns)
constexpr int OP_GROUPS = OP_NOUT / (32 * OP_CT);
constexpr int OP_STEPS = OP_SLICES * OP_KS;

// W[Nout][5 E] -> [column group][k-step][plane][tile][lane][8 bf16], lane (r, h) holding W[256 cg + 32 ct + r][16 ks + 8 h ..]
__global__ __launch_bounds__(256) void op_pack_w_kernel(const float* __restrict__ W, unsigned char* __restrict__ dst) {
  const int total = OP_GROUPS * OP_STEPS * OP_CT * 64;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    const int lane = c & 63, ct = (c >> 6) % OP_CT, ks = ((c >> 6) / OP_CT) % OP_STEPS, cg = (c >> 6) / (OP_CT * OP_STEPS);
    const int r = lane & 31, h = lane >> 5;
    const float* src = W + (size_t)(256 * cg + 32 * ct + r) * (OP_SLICES * OP_E) + 16 * ks + 8 * h;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[j];
    bf16x8 b[3];
    split8(v, b);
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<bf16x8*>(dst + (((((size_t)cg * OP_STEPS + ks) * 3 + p) * OP_CT + ct) * 64 + lane) * 16) = b[p];
  }
}

struct OpArgs {
  const float* x[OP_SLICES];   // [B*256, 256] each: slice i of the concatenation (att1..att4, f2f)
  const unsigned char* w;      // op_pack_w_kernel
  const float* bias;           // [1024]
  float* out;                  // [B*256, 1024] or NULL
  float* fmax;                 // [B, 1024]
  int32_t* arg;                // [B, 1024]
  int B;
};

__global__ __launch_bounds__(OP_WAVES * 64, 2) void outproj_maxpts_kernel(OpArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[3 * SLAB];
  __shared__ float sbias[32 * OP_CT];
  __shared__ float smax[OP_WAVES][32 * OP_CT];
  __shared__ int sarg[OP_WAVES][32 * OP_CT];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware: workgroups bid, bid + 8, ... share an XCD (and its L2); XCDs 2 cg, 2 cg + 1 take column group cg
  int cg, cloud;
  {
    const int bid = blockIdx.x, nb = gridDim.x;
    if ((nb & 7) == 0) {
      const int xcd = bid & 7, within = bid >> 3;
      cg = xcd >> 1, cloud = 2 * within + (xcd & 1);
    } else {
      cg = bid & 3, cloud = bid >> 2;
    }
  }
  if (tid < 32 * OP_CT) sbias[tid] = a.bias[256 * cg + tid];
  __syncthreads();
  const size_t row = (size_t)cloud * OP_L + 32 * wave + r;          // this lane's point
  const uint32_t voff = (uint32_t)wave * 1024u + (uint32_t)lane * 16u;
  const unsigned char* wcg = a.w + (size_t)cg * OP_STEPS * SLAB;
  // piece i of slab SL of the slice whose planes start at `base`, into ring slot c % 3 (uniform base + ONE lane offset
  // register, opaque per call: see salevel.hip)
  auto issue_piece = [&](int c, const unsigned char* base, auto slab, int i) {
    constexpr int SL = decltype(slab)::value;
    if (c < OP_STEPS) {
      const int slot = c % 3;
      uint32_t vo = voff;
      asm volatile("" : "+v"(vo));
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(base + ((size_t)SL * SLAB + i * OP_WAVES * 1024) + vo),
          (__attribute__((address_space(3))) void*)(lds + slot * SLAB + (i * OP_WAVES + wave) * 1024), 16, 0, 0);
    }
  };
  const unsigned char* wbase = wcg;                                  // planes of the slice in progress
#pragma unroll
  for (int i = 0; i < 3; ++i) issue_piece(0, wbase, std::integral_constant<int, 0>{}, i);
#pragma unroll
  for (int i = 0; i < 3; ++i) issue_piece(1, wbase, std::integral_constant<int, 1>{}, i);

  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 pa[2][2];                    // k-step ks of a slice lives in set ks & 1
#define OP_GLOAD(DST, PTR, OFF) \
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(DST) : "v"(PTR), "n"(OFF) : "memory")
#define OP_GLOAD2(PROW, KS_, S_)                                                               \
  do {                                                                                         \
    (void)&pa; /* (odr-use: asm operands alone do not make a generic lambda capture it) */      \
    OP_GLOAD(pa[S_][0], PROW, 64 * (KS_));                                                     \
    OP_GLOAD(pa[S_][1], PROW, 64 * (KS_) + 16);                                                \
  } while (0)
#define OP_GREADY(S_)                                          \
  do {                                                        \
    (void)&pa;                                                \
    asm volatile("" : "+v"(pa[S_][0]), "+v"(pa[S_][1]));      \
  } while (0)
  auto prow_of = [&](int slice) { return a.x[slice < OP_SLICES ? slice : OP_SLICES - 1] + row * OP_E + 8 * h; };
  const float* prow = prow_of(0);
  const float* prow_n = prow_of(1);
  bf16x8 af[3];
  BNext bn;
  auto pair_of = [&](auto setc, int j) {        // values 2j, 2j+1 of the set's 8: split
    constexpr int st = decltype(setc)::value;
    const f32x4 pv = pa[st][j >> 1];
    uint32_t w0, w1, w2;
    split_pair(pv[2 * (j & 1)], pv[2 * (j & 1) + 1], w0, w1, w2);
    asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));   // (pins the work to its fill slot)
    bn.w[0][j] = w0, bn.w[1][j] = w1, bn.w[2][j] = w2;
  };
  OP_GLOAD2(prow, 0, 0);
  OP_GLOAD2(prow, 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  OP_GREADY(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) pair_of(std::integral_constant<int, 0>{}, j);
  bn.get(af);

  floatx16 acc[OP_CT];               // starts from the bias (the column sits on the lane: one value per tile)
#pragma unroll
  for (int ct = 0; ct < OP_CT; ++ct) {
    const float bv = sbias[32 * ct + r];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ct][e] = bv;
  }
  int c = 0;
  for (int slice = 0; slice < OP_SLICES; ++slice) {
    const unsigned char* wbase_n = wcg + (size_t)(slice + 1 < OP_SLICES ? slice + 1 : slice) * OP_KS * SLAB;
    static_for<0, OP_KS>([&](auto slc) {
      constexpr int sl = decltype(slc)::value;
      if (c + 1 < OP_STEPS)
        wait_vm_sync<3>();                // may stay in flight: the next slab's 3 DMA pieces
      else
        wait_vm_sync<0>();
      OP_GREADY((sl + 1) & 1);            // the set split during this step is complete behind that wait
      const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds + (uint32_t)((c % 3) * SLAB) +
                          (uint32_t)lane * 16u;
      const auto fill = [&](int rt) {
        if (rt % 2 == 0) pair_of(std::integral_constant<int, (sl + 1) & 1>{}, rt / 2);    // k-step sl + 1 (of the next slice: 0)
        if (rt == 1) {                    // rows of k-step sl + 2: split during the next step
          if constexpr (sl + 2 < OP_KS)
            OP_GLOAD2(prow, sl + 2, sl & 1);
          else
            OP_GLOAD2(prow_n, sl + 2 - OP_KS, sl & 1);
        }
        if (rt == 3 || rt == 5 || rt == 7) {
          if constexpr (sl + 2 < OP_KS)
            issue_piece(c + 2, wbase, std::integral_constant<int, sl + 2>{}, (rt - 3) / 2);
          else
            issue_piece(c + 2, wbase_n, std::integral_constant<int, sl + 2 - OP_KS>{}, (rt - 3) / 2);
        }
      };
      kstep_rp<OP_CT, decltype(fill), true>(acc, la, af, fill);
      bn.get(af);
      ++c;
    });
    prow = prow_n, prow_n = prow_of(slice + 2);
    wbase = wbase_n;
  }
  // The row loads of the last step (k-steps of a slice that does not exist: same addresses as the last one) are still in
  // flight; the compiler believes their registers dead and hands them to the epilogue's addresses.  Measured: a wild
  // store (memory violation) once the loads were slow enough - beside another stream's kernels.  Drain them.
  asm volatile("s_waitcnt vmcnt(0) ; pzn_drain" ::: "memory");
  // ---- epilogue: element e of lane l = point (e&3) + 8 (e>>2) + 4 h of the wavefront's 32, column 32 ct + (l & 31)
  if (a.out) {
    float* o = a.out + ((size_t)cloud * OP_L + 32 * wave + 4 * h) * OP_NOUT + 256 * cg + r;
#pragma unroll
    for (int ct = 0; ct < OP_CT; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[(size_t)((e & 3) + 8 * (e >> 2)) * OP_NOUT + 32 * ct] = acc[ct][e];
  }
#pragma unroll
  for (int ct = 0; ct < OP_CT; ++ct) {
    float best = acc[ct][0];
    int be = 0;
#pragma unroll
    for (int e = 1; e < 16; ++e) {
      const float v = acc[ct][e];
      const bool gt = v > best;          // strict: the lower register (= the lower point) keeps a tie
      best = gt ? v : best;
      be = gt ? e : be;
    }
    int bi = (be & 3) + 8 * (be >> 2) + 4 * h;
    const auto sb = __builtin_amdgcn_permlane32_swap(__float_as_uint(best), __float_as_uint(best), false, false);
    const auto si = __builtin_amdgcn_permlane32_swap((uint32_t)bi, (uint32_t)bi, false, false);
    const float ob = __uint_as_float(h ? sb[0] : sb[1]);
    const int oi = (int)(h ? si[0] : si[1]);
    const bool take = ob > best || (ob == best && oi < bi);
    best = take ? ob : best;
    bi = take ? oi : bi;
    if (h == 0) smax[wave][32 * ct + r] = best, sarg[wave][32 * ct + r] = 32 * wave + bi;
  }
  __syncthreads();
  if (tid < 32 * OP_CT) {
    float best = smax[0][tid];
    int bi = sarg[0][tid];
#pragma unroll
    for (int w = 1; w < OP_WAVES; ++w) {
      const float v = smax[w][tid];
      const bool gt = v > best;          // ascending points, strict: first maximum
      best = gt ? v : best;
      bi = gt ? sarg[w][tid] : bi;
    }
    a.fmax[(size_t)cloud * OP_NOUT + 256 * cg + tid] = best;
    a.arg[(size_t)cloud * OP_NOUT + 256 * cg + tid] = bi;
  }
}

}  // namespace

size_t pzn_outproj_maxpts_ws_bytes(int L, int E, int nslice, int Nout) {
  return (L == OP_L && E == OP_E && nslice == OP_SLICES && Nout == OP_NOUT) ? (size_t)OP_GROUPS * OP_STEPS * SLAB : 0;
}

// -> PZN_EUNSUPPORTED for shapes it does not take (the caller composes slices + pzn_maxpool_points_fwd_f32 then)
int pzn_outproj_maxpts(const float* const* x, const float* W, const float* bias, int B, int L, int E, int nslice, int Nout,
                       float* out, float* fmax, int32_t* arg, void* workspace, hipStream_t st) {
  constexpr bool on = true;   // tuning aid
  if (!on || !workspace || pzn_outproj_maxpts_ws_bytes(L, E, nslice, Nout) == 0) return PZN_EUNSUPPORTED;
  uintptr_t al = reinterpret_cast<uintptr_t>(workspace);
  for (int i = 0; i < OP_SLICES; ++i) al |= reinterpret_cast<uintptr_t>(x[i]);
  if ((al & 15) != 0) return PZN_EUNSUPPORTED;
  unsigned char* w = static_cast<unsigned char*>(workspace);
  PZN_LAUNCH(op_pack_w_kernel, dim3(512), dim3(256), 0, st, W, w);
  if (hipGetLastError() != hipSuccess) return PZN_ELAUNCH;
  OpArgs a;
  for (int i = 0; i < OP_SLICES; ++i) a.x[i] = x[i];
  a.w = w, a.bias = bias, a.out = out, a.fmax = fmax, a.arg = arg, a.B = B;
  PZN_LAUNCH(outproj_maxpts_kernel, dim3(OP_GROUPS * B), dim3(OP_WAVES * 64), 0, st, a);
  PZN_RETURN_LAUNCH_STATUS();
}

PZN_EXPORT size_t pzn_outproj_maxpts_workspace_bytes(int L, int E, int nslice, int Nout) {
  return pzn_outproj_maxpts_ws_bytes(L, E, nslice, Nout);
}

// model5_b.py:466-475 in one launch; out may be NULL (only the maximum is needed).  PZN_EUNSUPPORTED: other shapes, the
// exact-fp32 engine, workspace == NULL.
PZN_EXPORT int pzn_outproj_maxpts_fwd_f32(const float* const* x, int nslice, const float* W, const float* bias, int B, int L,
                                          int E, int Nout, float* out, float* fmax, int32_t* arg, void* workspace,
                                          pzn_stream_t stream) {
  PZN_CHECK_ARG(x && W && bias && fmax && arg && B > 0 && nslice > 0);
  for (int i = 0; i < nslice; ++i) PZN_CHECK_ARG(x[i] != nullptr);
  if (pzn_gemm_get_precision() == 0) return PZN_EUNSUPPORTED;
  return pzn_outproj_maxpts(x, W, bias, B, L, E, nslice, Nout, out, fmax, arg, workspace, pzn_hip_stream(stream));
}
