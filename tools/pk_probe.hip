// pk_probe.hip — which packed-fp32 instruction forms return wrong results while the general matrix-core engine runs on
// another stream.  Stand-alone:
//   hipcc --offload-arch=gfx950 -O3 tools/pk_probe.hip -Lpuzzlenet_amd -lpzn -Wl,-rpath,$PWD/puzzlenet_amd -o /tmp/pk_probe
//   /tmp/pk_probe [launches]
// Victim kernel: no LDS, few registers (so its wavefronts share SIMDs with the aggressor's), every thread evaluates each
// form on hashed inputs with inline asm and compares bit for bit with the same arithmetic as single v_fma_f32 /
// v_mul_f32 / v_add_f32; mismatches are counted per (form, quarter of the wavefront, low / high result).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/pzn.h"

typedef float f2 __attribute__((ext_vector_type(2)));

#define NFORMS 9
static const char* FORM_NAMES[NFORMS] = {
    "v_pk_fma_f32 (no modifiers)",
    "v_pk_fma_f32 op_sel:[0,1,0]        (lo: a.lo*b.HI+c.lo)",
    "v_pk_fma_f32 op_sel_hi:[1,0,1]     (hi: a.hi*b.LO+c.hi)",
    "v_pk_mul_f32 op_sel_hi:[1,0]       (hi: a.hi*b.LO)",
    "v_pk_mul_f32 op_sel:[0,1]          (lo: a.lo*b.HI)",
    "v_pk_add_f32 (no modifiers)",
    "v_pk_fma_f32 op_sel:[1,0,0]        (lo: a.HI*b.lo+c.lo)",
    "v_pk_fma_f32 op_sel:[0,0,1]        (lo: a.lo*b.lo+c.HI)",
    "v_pk_add_f32 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]   (lo: a.HI-b.lo, hi: a.hi-b.hi)",
};

__device__ __forceinline__ float sfma(float a, float b, float c) {
  float r;
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float smul(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sadd(float a, float b) {
  float r;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float ssub(float a, float b) {
  float r;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float hashf(uint32_t x) {
  x ^= x >> 16, x *= 0x7feb352dU, x ^= x >> 15, x *= 0x846ca68bU, x ^= x >> 16;
  return (float)(int32_t)(x & 0xFFFFFF) * (1.0f / 8388608.0f) - 1.0f;      // [-1, 1)
}

__global__ __launch_bounds__(256) void victim_kernel(int iters, uint32_t seed, unsigned long long* __restrict__ counts) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int quarter = (threadIdx.x & 63) >> 4;
  for (int it = 0; it < iters; ++it) {
    const uint32_t h = seed + tid * 977u + (uint32_t)it * 0x9e3779b9u;
    f2 a = {hashf(h), hashf(h + 1)}, b = {hashf(h + 2), hashf(h + 3)}, c = {hashf(h + 4), hashf(h + 5)};
    f2 d[NFORMS], e[NFORMS];
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d[0]) : "v"(a), "v"(b), "v"(c));
    e[0] = f2{sfma(a.x, b.x, c.x), sfma(a.y, b.y, c.y)};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(d[1]) : "v"(a), "v"(b), "v"(c));
    e[1] = f2{sfma(a.x, b.y, c.x), sfma(a.y, b.y, c.y)};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d[2]) : "v"(a), "v"(b), "v"(c));
    e[2] = f2{sfma(a.x, b.x, c.x), sfma(a.y, b.x, c.y)};
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d[3]) : "v"(a), "v"(b));
    e[3] = f2{smul(a.x, b.x), smul(a.y, b.x)};
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d[4]) : "v"(a), "v"(b));
    e[4] = f2{smul(a.x, b.y), smul(a.y, b.y)};
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d[5]) : "v"(a), "v"(b));
    e[5] = f2{sadd(a.x, b.x), sadd(a.y, b.y)};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d[6]) : "v"(a), "v"(b), "v"(c));
    e[6] = f2{sfma(a.y, b.x, c.x), sfma(a.y, b.y, c.y)};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d[7]) : "v"(a), "v"(b), "v"(c));
    e[7] = f2{sfma(a.x, b.x, c.y), sfma(a.y, b.y, c.y)};
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d[8]) : "v"(a), "v"(b));
    e[8] = f2{ssub(a.y, b.x), ssub(a.y, b.y)};
#pragma unroll
    for (int f = 0; f < NFORMS; ++f) {
      if (__float_as_uint(d[f].x) != __float_as_uint(e[f].x)) atomicAdd(&counts[(f * 4 + quarter) * 2 + 0], 1ull);
      if (__float_as_uint(d[f].y) != __float_as_uint(e[f].y)) atomicAdd(&counts[(f * 4 + quarter) * 2 + 1], 1ull);
    }
  }
}

// The set-abstraction prep kernel as it was compiled before the fix (plain -O3 pairs the four channels): victim with
// memory traffic.  Counts elements of P that differ from a run without the aggressor.
__global__ __launch_bounds__(256) void prep_like_kernel(const float* __restrict__ xyz, const float* __restrict__ W1, int ldw,
                                                        long prow, int C1, const float* __restrict__ Pin,
                                                        float* __restrict__ P) {
  const int c4 = C1 >> 2;
  const long total = prow * c4;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long r = e / c4;
    const int c = (int)(e - r * c4) * 4;
    float wx[4], wy[4], wz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      wx[i] = W1[(size_t)(c + i) * ldw], wy[i] = W1[(size_t)(c + i) * ldw + 1], wz[i] = W1[(size_t)(c + i) * ldw + 2];
    const float* q = xyz + (size_t)r * 3;
    const float x = q[0], y = q[1], z = q[2];
    float4 v = *reinterpret_cast<const float4*>(Pin + (size_t)r * C1 + c);
    v.x += fmaf(wz[0], z, fmaf(wy[0], y, wx[0] * x));
    v.y += fmaf(wz[1], z, fmaf(wy[1], y, wx[1] * x));
    v.z += fmaf(wz[2], z, fmaf(wy[2], y, wx[2] * x));
    v.w += fmaf(wz[3], z, fmaf(wy[3], y, wx[3] * x));
    *reinterpret_cast<float4*>(P + (size_t)r * C1 + c) = v;
  }
}

__global__ void diff_kernel(const float* a, const float* b, long n, unsigned long long* cnt) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    if (__float_as_uint(a[i]) != __float_as_uint(b[i])) atomicAdd(cnt, 1ull);
}

__global__ void fill_kernel(float* p, long n, uint32_t seed) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    p[i] = hashf(seed + (uint32_t)i);
}

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));              \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 200;
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0));
  CK(hipStreamCreate(&s1));
  // aggressor: pzn_linear_fwd_f32 4096 x 1280 -> 1024 (general engine, split-K for few rows)
  const int M = 4096, K = 1280, N = 1024;
  float *xg, *wg, *yg;
  CK(hipMalloc(&xg, sizeof(float) * M * K));
  CK(hipMalloc(&wg, sizeof(float) * N * K));
  CK(hipMalloc(&yg, sizeof(float) * M * N));
  fill_kernel<<<1024, 256, 0, s0>>>(xg, (long)M * K, 1u);
  fill_kernel<<<1024, 256, 0, s0>>>(wg, (long)N * K, 2u);
  // prep-like victim data
  const long prow = 4096;
  const int C1 = 128, ldw = 67;
  float *xyz, *W1, *Pin, *Pref, *P;
  CK(hipMalloc(&xyz, sizeof(float) * prow * 3));
  CK(hipMalloc(&W1, sizeof(float) * C1 * ldw));
  CK(hipMalloc(&Pin, sizeof(float) * prow * C1));
  CK(hipMalloc(&Pref, sizeof(float) * prow * C1));
  CK(hipMalloc(&P, sizeof(float) * prow * C1));
  fill_kernel<<<64, 256, 0, s0>>>(xyz, prow * 3, 3u);
  fill_kernel<<<64, 256, 0, s0>>>(W1, (long)C1 * ldw, 4u);
  fill_kernel<<<1024, 256, 0, s0>>>(Pin, prow * C1, 5u);
  unsigned long long* counts;
  CK(hipMalloc(&counts, sizeof(unsigned long long) * (NFORMS * 8 + 1)));
  for (int pass = 0; pass < 2; ++pass) {      // 0: alone, 1: beside the aggressor
    CK(hipMemsetAsync(counts, 0, sizeof(unsigned long long) * (NFORMS * 8 + 1), s0));
    prep_like_kernel<<<(unsigned)((prow * (C1 / 4) + 255) / 256), 256, 0, s0>>>(xyz, W1, ldw, prow, C1, Pin, Pref);
    CK(hipStreamSynchronize(s0));
    for (int l = 0; l < launches; ++l) {
      if (pass == 1)
        for (int k = 0; k < 3; ++k)
          if (pzn_linear_fwd_f32(xg, wg, nullptr, M, K, N, 0, yg, (pzn_stream_t)s1) != 0) {
            fprintf(stderr, "aggressor launch failed\n");
            return 1;
          }
      victim_kernel<<<768, 256, 0, s0>>>(8, 1000u + l, counts);
      prep_like_kernel<<<(unsigned)((prow * (C1 / 4) + 255) / 256), 256, 0, s0>>>(xyz, W1, ldw, prow, C1, Pin, P);
      diff_kernel<<<256, 256, 0, s0>>>(P, Pref, prow * C1, counts + NFORMS * 8);
    }
    CK(hipDeviceSynchronize());
    unsigned long long h[NFORMS * 8 + 1];
    CK(hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost));
    printf("== %s (%d launches; %.1f M evaluations per form)\n", pass ? "beside the general matrix-core engine" : "alone",
           launches, launches * 768.0 * 256 * 8 / 1e6);
    for (int f = 0; f < NFORMS; ++f) {
      unsigned long long tot = 0;
      for (int i = 0; i < 8; ++i) tot += h[f * 8 + i];
      printf("  %-78s wrong: %8llu", FORM_NAMES[f], tot);
      if (tot) {
        printf("   [lo/hi per quarter:");
        for (int qd = 0; qd < 4; ++qd) printf(" q%d %llu/%llu", qd, h[(f * 4 + qd) * 2], h[(f * 4 + qd) * 2 + 1]);
        printf("]");
      }
      printf("\n");
    }
    printf("  prep-like kernel (compiler-paired packed math, loads + stores)                  wrong elements: %llu\n", h[NFORMS * 8]);
  }
  return 0;
}
