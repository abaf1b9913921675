// split_probe.hip — is x - float(bf16(x)) via v_dot2c_f32_bf16 bit-identical to unpack + subtract?  (pzn_mfma.h: split_pair)
//   hipcc --offload-arch=gfx950 -O3 tools/split_probe.hip -o /tmp/split_probe && /tmp/split_probe
// 16.7 M values (exponents 2^-37 .. 2^32, zeros, denormals): 0 pairs differ on MI355X.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_ref(float x0, float x1, uint32_t& a, uint32_t& b, uint32_t& c) {
  const floatx2 x = {x0, x1};
  a = __builtin_bit_cast(uint32_t, __builtin_convertvector(x, bf16x2));
  const floatx2 r = {x0 - __uint_as_float(a << 16), x1 - __uint_as_float(a & 0xffff0000u)};
  b = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
  const floatx2 t = {r[0] - __uint_as_float(b << 16), r[1] - __uint_as_float(b & 0xffff0000u)};
  c = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2));
}
__device__ __forceinline__ void split_dot(float x0, float x1, uint32_t& a, uint32_t& b, uint32_t& c) {
  uint32_t klo = 0x0000bf80u, khi = 0xbf800000u;      // bf16 pairs (-1, 0) and (0, -1); opaque: the compiler folds the
  asm volatile("" : "+v"(klo), "+v"(khi));             // vector constants to the same inline -1.0
  const bf16x2 mlo = __builtin_bit_cast(bf16x2, klo), mhi = __builtin_bit_cast(bf16x2, khi);
  const floatx2 x = {x0, x1};
  const bf16x2 pa = __builtin_convertvector(x, bf16x2);
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(pa, mlo, x0, false), r1 = __builtin_amdgcn_fdot2_f32_bf16(pa, mhi, x1, false);
  const floatx2 r = {r0, r1};
  const bf16x2 pb = __builtin_convertvector(r, bf16x2);
  const float t0 = __builtin_amdgcn_fdot2_f32_bf16(pb, mlo, r0, false), t1 = __builtin_amdgcn_fdot2_f32_bf16(pb, mhi, r1, false);
  const floatx2 t = {t0, t1};
  a = __builtin_bit_cast(uint32_t, pa), b = __builtin_bit_cast(uint32_t, pb);
  c = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2));
}
__global__ void k(const float* x, int n, unsigned long long* bad) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  uint32_t a, b, c, d, e, f;
  split_ref(x[2 * i], x[2 * i + 1], a, b, c);
  split_dot(x[2 * i], x[2 * i + 1], d, e, f);
  if (a != d || b != e || c != f) {
    unsigned long long n_ = atomicAdd(bad, 1ull);
    if (n_ < 6) printf("x0 %a x1 %a  ref %08x %08x %08x  dot %08x %08x %08x\n", x[2 * i], x[2 * i + 1], a, b, c, d, e, f);
  }
}
int main() {
  const int n = 1 << 24;
  float* h = (float*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    uint32_t u = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    uint32_t ex = 90 + (rand() % 70);                 // exponents around 1: 2^-37 .. 2^32
    u = (u & 0x807fffffu) | (ex << 23);
    if (i % 97 == 0) u &= 0x80000000u;                 // zeros
    if (i % 101 == 0) u = (u & 0x807fffffu);           // denormals
    h[i] = *(float*)&u;
  }
  float* dx; unsigned long long* db, hb = 0;
  hipMalloc(&dx, n * 4); hipMalloc(&db, 8);
  hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, &hb, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, db);
  hipMemcpy(&hb, db, 8, hipMemcpyDeviceToHost);
  printf("pairs with a different split: %llu of %d\n", hb, n / 2);
  return 0;
}
