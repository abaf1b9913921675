// valu_rate.hip — issue cost of the vector instructions the EMD / chamfer walks are made of, on the whole chip.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o scratch/valu_rate && scratch/valu_rate
// Every kernel: 256 threads x (1024 * WPS) workgroups, each wavefront runs ITERS iterations of 16 independent
// instructions of one kind (inline asm, register operands only).  Reported: SIMD cycles per wave-instruction at the
// clock the run reached (wall time; s_memrealtime is a fixed 100 MHz), for WPS = 1, 2, 4, 8 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096

#define STAMP_BEGIN() const unsigned long long t0_ = __builtin_amdgcn_s_memtime()
#define STAMP_END(out)                                                                                   \
  do {                                                                                                   \
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime();                                         \
    if (blockIdx.x == 0 && threadIdx.x == 0) ((unsigned long long*)(out))[1] = t1_ - t0_;               \
  } while (0)
#define BODY16(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) STMT(8) STMT(9) STMT(10) STMT(11) STMT(12) STMT(13) STMT(14) STMT(15)

#define K_SCALAR(NAME, ASM)                                                              \
  __global__ __launch_bounds__(256) void NAME(float* out, float s) {                     \
    float a[16];                                                                         \
    for (int i = 0; i < 16; ++i) a[i] = s * (float)(threadIdx.x + i);                    \
    const float b = s + 1.0f, c = s - 0.5f;                                              \
    STAMP_BEGIN();                                                                       \
    for (int it = 0; it < ITERS; ++it) {                                                 \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
    }                                                                                    \
    STAMP_END(out);                                                                      \
    float t = 0;                                                                         \
    for (int i = 0; i < 16; ++i) t += a[i];                                              \
    if (t == 12345.f) out[0] = t;                                                        \
  }
#define K_PACKED(NAME, ASM)                                                              \
  __global__ __launch_bounds__(256) void NAME(float* out, float s) {                     \
    f2 a[16];                                                                            \
    for (int i = 0; i < 16; ++i) a[i] = (f2){s * (float)(threadIdx.x + i), s};           \
    const f2 b = {s + 1.0f, s}, c = {s - 0.5f, s};                                       \
    STAMP_BEGIN();                                                                       \
    for (int it = 0; it < ITERS; ++it) {                                                 \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
    }                                                                                    \
    STAMP_END(out);                                                                      \
    f2 t = {0, 0};                                                                       \
    for (int i = 0; i < 16; ++i) t += a[i];                                              \
    if (t.x + t.y == 12345.f) out[0] = t.x;                                              \
  }

K_SCALAR(k_fma, "v_fma_f32 %0, %0, %1, %2")
K_SCALAR(k_mul, "v_mul_f32 %0, %0, %1")
K_SCALAR(k_add, "v_add_f32 %0, %0, %1")
K_SCALAR(k_exp, "v_exp_f32 %0, %0")
K_SCALAR(k_rcp, "v_rcp_f32 %0, %0")
K_PACKED(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")
K_PACKED(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
K_PACKED(k_pk_add, "v_pk_add_f32 %0, %0, %1")
K_PACKED(k_pk_add_neg, "v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]")
K_PACKED(k_pk_fma_bcast, "v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]")

// the EMD pass B evaluation of two walked points, written out (12 packed + 2 exp), operands in registers
__global__ __launch_bounds__(256) void k_evalB(float* out, float s) {
  f2 X = {s, s + 1}, Y = {s + 2, s + 3}, Z = {s + 4, s + 5}, W = {s, s};
  const f2 mx = {s * 3, s * 3}, my = {s * 5, s * 5}, mz = {s * 7, s * 7}, c = {-s, -s};
  f2 ar = {0, 0}, ax = ar, ay = ar, az = ar;
  STAMP_BEGIN();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f2 dx, dy, dz, d, e;
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx) : "v"(mx), "v"(X));
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dy) : "v"(my), "v"(Y));
      asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dz) : "v"(mz), "v"(Z));
      asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(d) : "v"(dx));
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d) : "v"(dy));
      asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(d) : "v"(dz));
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(e) : "v"(d), "v"(c));
      asm volatile("v_exp_f32 %0, %0" : "+v"(e.x));
      asm volatile("v_exp_f32 %0, %0" : "+v"(e.y));
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(e) : "v"(W));
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(ar) : "v"(e));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ax) : "v"(e), "v"(dx));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ay) : "v"(e), "v"(dy));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(az) : "v"(e), "v"(dz));
      X += W;
    }
  }
  STAMP_END(out);
  f2 t = ar + ax + ay + az;
  if (t.x + t.y == 12345.f) out[0] = t.x;
}
// the same evaluation with single fp32 instructions (two points = twice the instructions)
__global__ __launch_bounds__(256) void k_evalB_scalar(float* out, float s) {
  float X = s, Y = s + 2, Z = s + 4, W = s;
  const float mx = s * 3, my = s * 5, mz = s * 7, c = -s;
  float ar = 0, ax = 0, ay = 0, az = 0;
  STAMP_BEGIN();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float dx, dy, dz, d, e;
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dx) : "v"(mx), "v"(X));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dy) : "v"(my), "v"(Y));
      asm volatile("v_sub_f32 %0, %1, %2" : "=v"(dz) : "v"(mz), "v"(Z));
      asm volatile("v_mul_f32 %0, %1, %1" : "=v"(d) : "v"(dx));
      asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(d) : "v"(dy));
      asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(d) : "v"(dz));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e) : "v"(d), "v"(c));
      asm volatile("v_exp_f32 %0, %0" : "+v"(e));
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e) : "v"(W));
      asm volatile("v_add_f32 %0, %0, %1" : "+v"(ar) : "v"(e));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ax) : "v"(e), "v"(dx));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ay) : "v"(e), "v"(dy));
      asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(az) : "v"(e), "v"(dz));
      X += W;
    }
  }
  STAMP_END(out);
  float t = ar + ax + ay + az;
  if (t == 12345.f) out[0] = t;
}

template <typename K>
static void run(const char* name, K kern, double insts_per_iter, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("%-22s", name);
  for (int wps = 1; wps <= 8; wps *= 2) {
    const int grid = 256 * wps;  // x 4 wavefronts per workgroup = 1024 * wps wavefronts = wps per SIMD
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 0.001f);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 0.001f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e-3 / ((double)ITERS * insts_per_iter * wps);  // seconds per wave-instruction per SIMD
    unsigned long long cyc[2];
    hipMemcpy(cyc, d, 16, hipMemcpyDeviceToHost);
    // wave 0's own stamps: cycles it spent in the loop; it shares its SIMD with wps - 1 others
    printf("  wps %d: %6.2f ns, %6.2f cyc (%.2f GHz)", wps, per * 1e9, (double)cyc[1] / ((double)ITERS * insts_per_iter * wps),
           (double)cyc[1] / (ms * 1e6));
  }
  printf("\n");
}

int main() {
  float* d;
  hipMalloc(&d, 1024);
  run("v_fma_f32", k_fma, 16, d);
  run("v_mul_f32", k_mul, 16, d);
  run("v_add_f32", k_add, 16, d);
  run("v_exp_f32", k_exp, 16, d);
  run("v_rcp_f32", k_rcp, 16, d);
  run("v_pk_fma_f32", k_pk_fma, 16, d);
  run("v_pk_mul_f32", k_pk_mul, 16, d);
  run("v_pk_add_f32", k_pk_add, 16, d);
  run("v_pk_add_f32 neg", k_pk_add_neg, 16, d);
  run("v_pk_fma_f32 op_sel_hi", k_pk_fma_bcast, 16, d);
  run("evalB packed (2 pts)", k_evalB, 4, d);          // per 2-point evaluation group (12 pk + 2 exp + 1 pk_add of the X update)
  run("evalB scalar (1 pt)", k_evalB_scalar, 8, d);    // per 1-point evaluation (12 + exp + 1)
  return 0;
}
