// pk_probe2.hip — WHICH instruction class of a co-resident kernel is the other half of the packed-fp32 fault
// (v_pk_fma_f32 / v_pk_mul_f32 with the src1 op_sel bit return a wrong LOW result in lanes 48-63
// while workgroups of gemm_kernel or of the generated-row kernel share the CU).  Stand-alone, no libpzn:
//   hipcc --offload-arch=gfx950 -O3 tools/pk_probe2.hip -o /tmp/pk_probe2 && /tmp/pk_probe2 [launches]
// Victim: register-only kernel evaluating packed forms with inline asm against single-instruction references.
// Aggressors: synthetic kernels, each exercising ONE ingredient of the two triggering kernels, run on a second stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define NFORMS 14
static const char* FORM_NAMES[NFORMS] = {
    "v_pk_fma_f32 (no modifiers)",
    "v_pk_fma_f32 op_sel:[0,1,0]     lo: a.lo*b.HI+c.lo",
    "v_pk_mul_f32 op_sel:[0,1]       lo: a.lo*b.HI",
    "v_pk_add_f32 op_sel:[0,1]       lo: a.lo+b.HI",
    "v_pk_fma_f32 op_sel:[0,1,1]     lo: a.lo*b.HI+c.HI",
    "v_pk_mul_f32 op_sel:[1,1]       lo: a.HI*b.HI",
    "v_pk_fma_f32 op_sel:[1,0,0]     lo: a.HI*b.lo+c.lo",
    "v_pk_fma_f32 op_sel:[0,0,1]     lo: a.lo*b.lo+c.HI",
    "v_pk_add_f32 op_sel:[1,0]       lo: a.HI+b.lo",
    "v_pk_fma_f32 op_sel_hi:[1,0,1]  hi: a.hi*b.LO+c.hi",
    "v_pk_mul_f32 op_sel_hi:[1,0]    hi: a.hi*b.LO",
    "v_pk_add_f32 (no modifiers)",
    "v_pk_add_f32 op_sel_hi:[0,1]    hi: a.LO+b.hi",
    "v_pk_fma_f32 op_sel_hi:[0,1,1]  hi: a.LO*b.hi+c.hi",
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
__device__ __forceinline__ float hashf(uint32_t x) {
  x ^= x >> 16, x *= 0x7feb352dU, x ^= x >> 15, x *= 0x846ca68bU, x ^= x >> 16;
  return (float)(int32_t)(x & 0xFFFFFF) * (1.0f / 8388608.0f) - 1.0f;
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
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d[2]) : "v"(a), "v"(b));
    e[2] = f2{smul(a.x, b.y), smul(a.y, b.y)};
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d[3]) : "v"(a), "v"(b));
    e[3] = f2{sadd(a.x, b.y), sadd(a.y, b.y)};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]" : "=v"(d[4]) : "v"(a), "v"(b), "v"(c));
    e[4] = f2{sfma(a.x, b.y, c.y), sfma(a.y, b.y, c.y)};
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]" : "=v"(d[5]) : "v"(a), "v"(b));
    e[5] = f2{smul(a.y, b.y), smul(a.y, b.y)};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(d[6]) : "v"(a), "v"(b), "v"(c));
    e[6] = f2{sfma(a.y, b.x, c.x), sfma(a.y, b.y, c.y)};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(d[7]) : "v"(a), "v"(b), "v"(c));
    e[7] = f2{sfma(a.x, b.x, c.y), sfma(a.y, b.y, c.y)};
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d[8]) : "v"(a), "v"(b));
    e[8] = f2{sadd(a.y, b.x), sadd(a.y, b.y)};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(d[9]) : "v"(a), "v"(b), "v"(c));
    e[9] = f2{sfma(a.x, b.x, c.x), sfma(a.y, b.x, c.y)};
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d[10]) : "v"(a), "v"(b));
    e[10] = f2{smul(a.x, b.x), smul(a.y, b.x)};
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(d[11]) : "v"(a), "v"(b));
    e[11] = f2{sadd(a.x, b.x), sadd(a.y, b.y)};
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d[12]) : "v"(a), "v"(b));
    e[12] = f2{sadd(a.x, b.x), sadd(a.x, b.y)};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d[13]) : "v"(a), "v"(b), "v"(c));
    e[13] = f2{sfma(a.x, b.x, c.x), sfma(a.x, b.y, c.y)};
#pragma unroll
    for (int f = 0; f < NFORMS; ++f) {
      if (__float_as_uint(d[f].x) != __float_as_uint(e[f].x)) atomicAdd(&counts[(f * 4 + quarter) * 2 + 0], 1ull);
      if (__float_as_uint(d[f].y) != __float_as_uint(e[f].y)) atomicAdd(&counts[(f * 4 + quarter) * 2 + 1], 1ull);
    }
  }
}

// ---------------------------------------------------------------------------------------------- aggressors
enum {
  AG_NONE = 0, AG_MFMA_BF16_V, AG_MFMA_BF16_A, AG_MFMA_F32, AG_ACCVGPR, AG_DS_TR, AG_DS_B128, AG_GLOBAL_LD, AG_BUFFER_LD,
  AG_VALU, AG_PK_VALU, AG_BIGVGPR, AG_MFMA_TR, AG_SETPRIO_MFMA, AG_COUNT
};
static const char* AG_NAMES[AG_COUNT] = {
    "nothing (control)", "v_mfma_f32_32x32x16_bf16, VGPR accumulators", "v_mfma_f32_32x32x16_bf16, AGPR accumulators",
    "v_mfma_f32_32x32x2_f32", "v_accvgpr_write / read", "ds_read_b64_tr_b16", "ds_read_b128 + ds_write_b128",
    "global_load_dwordx4 stream", "buffer_load_dwordx4 stream", "v_fma_f32 loop", "v_pk_fma_f32 loop",
    "v_fma_f32 on ~240 VGPRs", "MFMA bf16 + ds_read_b64_tr_b16 interleaved", "s_setprio 3 + MFMA bf16"};

template <int KIND>
__global__ __launch_bounds__(256) void aggressor_kernel(int iters, const float* __restrict__ src, float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  float acc_out = 0.f;
  if (KIND == AG_MFMA_BF16_V || KIND == AG_MFMA_TR || KIND == AG_SETPRIO_MFMA) {
    floatx16 acc = {0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)(0.01f * (lane + j)), b[j] = (__bf16)(0.02f * (lane - j));
    if (KIND == AG_SETPRIO_MFMA) __builtin_amdgcn_s_setprio(3);
    typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
    for (int i = tid; i < 8192; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      if (KIND == AG_MFMA_TR) {
        bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(lds + ((lane * 8 + it * 512) & 32767 & ~7)));
        a[0] = t[0];
      }
    }
    acc_out = acc[0] + acc[5];
  } else if (KIND == AG_MFMA_BF16_A) {
    floatx16 acc = {0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)(0.01f * (lane + j)), b[j] = (__bf16)(0.02f * (lane - j));
    for (int it = 0; it < iters; ++it) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    acc_out = acc[0] + acc[7];
  } else if (KIND == AG_MFMA_F32) {
    floatx16 acc = {0};
    float a = 0.01f * lane, b = 0.02f * lane;
    for (int it = 0; it < iters; ++it) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    acc_out = acc[0] + acc[3];
  } else if (KIND == AG_ACCVGPR) {
    float v = 0.5f * lane, w = 0.f;
    for (int it = 0; it < iters * 4; ++it) {
      asm volatile("v_accvgpr_write_b32 a0, %1\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0" : "=v"(w) : "v"(v) : "a0");
      v = w + 1.f;
    }
    acc_out = v;
  } else if (KIND == AG_DS_TR || KIND == AG_DS_B128) {
    for (int i = tid; i < 8192; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    typedef bf16x4 __attribute__((address_space(3))) * lds4_t;
    float s = 0.f;
    for (int it = 0; it < iters * 2; ++it) {
      if (KIND == AG_DS_TR) {
        bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4_t)(lds + ((lane * 8 + it * 512) & 32767 & ~7)));
        s += (float)t[0];
      } else {
        float4 v = *reinterpret_cast<const float4*>(lds + ((tid * 16 + it * 4096) & 32767));
        *reinterpret_cast<float4*>(lds + ((tid * 16 + it * 4096 + 16384) & 32767)) = v;
        s += v.x;
      }
    }
    acc_out = s;
  } else if (KIND == AG_GLOBAL_LD || KIND == AG_BUFFER_LD) {
    float s = 0.f;
    const size_t base = ((size_t)blockIdx.x * 256 + tid) * 4;
    if (KIND == AG_BUFFER_LD) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 64 << 20, 0x27000);
      for (int it = 0; it < iters / 4; ++it) {
        floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((base + (size_t)it * 262144) * 4) & 0x3fffff0), 0, 0));
        s += v[0];
      }
    } else {
      for (int it = 0; it < iters / 4; ++it) {
        const float4 v = *reinterpret_cast<const float4*>(src + ((base + (size_t)it * 262144) & 0xfffffc));
        s += v.x;
      }
    }
    acc_out = s;
  } else if (KIND == AG_VALU || KIND == AG_PK_VALU) {
    f2 a = {0.5f + lane, 0.25f}, b = {0.999f, 1.001f}, c = {0.001f, 0.002f};
    for (int it = 0; it < iters * 8; ++it) {
      if (KIND == AG_PK_VALU)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
      else
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a.x) : "v"(b.x), "v"(c.x));
    }
    acc_out = a.x + a.y;
  } else if (KIND == AG_BIGVGPR) {
    float r[240];
#pragma unroll
    for (int i = 0; i < 240; ++i) r[i] = 0.001f * (lane + i);
    for (int it = 0; it < iters / 16; ++it) {
#pragma unroll
      for (int i = 0; i < 240; ++i) r[i] = fmaf(r[i], 0.999f, r[(i + 7) % 240]);
    }
#pragma unroll
    for (int i = 0; i < 240; ++i) acc_out += r[i];
  }
  if (acc_out == 12345.678f) sink[tid] = acc_out;
}

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));              \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

template <int KIND>
static void launch_aggr(hipStream_t s, int iters, const float* src, float* sink) {
  aggressor_kernel<KIND><<<1024, 256, 0, s>>>(iters, src, sink);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 100;
  hipStream_t s0, s1;
  CK(hipStreamCreate(&s0));
  CK(hipStreamCreate(&s1));
  float *src, *sink;
  CK(hipMalloc(&src, 64u << 20));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(src, 0, 64u << 20));
  unsigned long long* counts;
  CK(hipMalloc(&counts, sizeof(unsigned long long) * NFORMS * 8));
  typedef void (*launch_fn)(hipStream_t, int, const float*, float*);
  launch_fn fns[AG_COUNT] = {nullptr, launch_aggr<AG_MFMA_BF16_V>, launch_aggr<AG_MFMA_BF16_A>, launch_aggr<AG_MFMA_F32>,
                             launch_aggr<AG_ACCVGPR>, launch_aggr<AG_DS_TR>, launch_aggr<AG_DS_B128>, launch_aggr<AG_GLOBAL_LD>,
                             launch_aggr<AG_BUFFER_LD>, launch_aggr<AG_VALU>, launch_aggr<AG_PK_VALU>, launch_aggr<AG_BIGVGPR>,
                             launch_aggr<AG_MFMA_TR>, launch_aggr<AG_SETPRIO_MFMA>};
  printf("victim: %d launches x 768 x 256 threads x 8 iterations = %.1f M evaluations per form and aggressor\n", launches,
         launches * 768.0 * 256 * 8 / 1e6);
  for (int ag = 0; ag < AG_COUNT; ++ag) {
    CK(hipMemsetAsync(counts, 0, sizeof(unsigned long long) * NFORMS * 8, s0));
    CK(hipStreamSynchronize(s0));
    for (int l = 0; l < launches; ++l) {
      if (fns[ag]) fns[ag](s1, 4000, src, sink);
      victim_kernel<<<768, 256, 0, s0>>>(8, 1000u + l, counts);
    }
    CK(hipDeviceSynchronize());
    unsigned long long h[NFORMS * 8];
    CK(hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost));
    unsigned long long any = 0;
    for (int i = 0; i < NFORMS * 8; ++i) any += h[i];
    printf("== beside: %-52s total wrong %llu\n", AG_NAMES[ag], any);
    fflush(stdout);
    for (int f = 0; f < NFORMS && any; ++f) {
      unsigned long long tot = 0;
      for (int i = 0; i < 8; ++i) tot += h[f * 8 + i];
      if (!tot) continue;
      printf("     %-52s wrong %8llu  [lo/hi per quarter:", FORM_NAMES[f], tot);
      for (int qd = 0; qd < 4; ++qd) printf(" q%d %llu/%llu", qd, h[(f * 4 + qd) * 2], h[(f * 4 + qd) * 2 + 1]);
      printf("]\n");
    }
  }
  return 0;
}
