// pzn_common.h — shared device/host helpers for libpzn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pzn.h"

#define PZN_EXPORT extern "C" __attribute__((visibility("default")))

#define PZN_WAVE 64

// Status of the launch that was just enqueued on this thread.
#define PZN_RETURN_LAUNCH_STATUS()                          \
  do {                                                      \
    return hipGetLastError() == hipSuccess ? PZN_OK : PZN_ELAUNCH; \
  } while (0)

#define PZN_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return PZN_EINVAL; \
  } while (0)

static inline hipStream_t pzn_hip_stream(pzn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Per-KERNEL timing (measurement only, core.hip; pzn_ktimer_* in include/pzn.h): every launch of the library goes through
// PZN_LAUNCH, which - when pzn_ktimer_enable(1) was called - brackets it with a HIP event pair on the launch's own stream
// under the kernel's name (hipKernelNameRefByPtr: the name a rocprofv3 kernel trace shows, template arguments included).
// Off (the default) it costs one relaxed load per launch.
extern "C" int pzn_ktimer_is_on;
namespace pzn {
struct KSpan {
  void* rec = nullptr;
  hipStream_t st;
  KSpan(const void* fn, const char* text, hipStream_t s);
  ~KSpan();
};
// the host function behind a launch expression: a kernel's name (decays to a pointer) or a variable holding such a pointer
template <class T>
static inline const void* kernel_address(T* f) { return reinterpret_cast<const void*>(f); }
}  // namespace pzn
#define PZN_LAUNCH(kernel, grid, block, shmem, stream, ...)                                    \
  do {                                                                                         \
    if (__builtin_expect(pzn_ktimer_is_on, 0)) {                                               \
      pzn::KSpan _span(pzn::kernel_address(kernel), #kernel, stream);                          \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                     \
    } else {                                                                                   \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                     \
    }                                                                                          \
  } while (0)

// Zero-fill as a KERNEL, not hipMemsetAsync: inside a captured HIP graph (ROCm 7.0 runtime shipped
// with torch) memset nodes were observed to run out of stream order relative to kernels that reuse the
// same allocation, so accumulators were zeroed too early.  A kernel node keeps the stream order.
static __global__ void pzn_zero_kernel(float* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = 0.f;
}
static inline int pzn_zero_async(float* p, size_t n, hipStream_t st) {
  if (n == 0) return PZN_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  PZN_LAUNCH(pzn_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, n);
  return hipGetLastError() == hipSuccess ? PZN_OK : PZN_ELAUNCH;
}

namespace pzn {

// ((dx*dx + dy*dy) + dz*dz) with every operation individually rounded: the
// *_rn intrinsics are never fused into an fma, whatever -ffp-contract says.
// Bit-identical to pointnet_util.square_distance (pointnet_util.py:36) on CPU.
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz) {
  float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// lane i <- lane i^J for one dword.  J = 1, 2, 4, 8 are DPP modifiers on a v_mov (no LDS round trip):
// quad_perm for 1 and 2, row_half_mirror (i^7) followed by quad_perm [3,2,1,0] (i^3) for 4,
// row_ror:8 for 8; J = 16 is a ds_swizzle (bit mode, no address VGPR); only J = 32 pays a ds_bpermute.
template <int J>
__device__ __forceinline__ uint32_t xor_lane(uint32_t v) {
  if (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
  if (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
  if (J == 4) {
    int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);
    return (uint32_t)__builtin_amdgcn_update_dpp(0, t, 0x1B, 0xF, 0xF, true);
  }
  if (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);
  if (J == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (16 << 10) | 0x1F);
  return (uint32_t)__shfl_xor((int)v, 32, PZN_WAVE);
}

template <int J>
__device__ __forceinline__ uint64_t xor_lane_u64(uint64_t v) {
  return ((uint64_t)xor_lane<J>((uint32_t)(v >> 32)) << 32) | xor_lane<J>((uint32_t)v);
}

// 64-lane max of a u64 key: five DPP / swizzle steps and one bpermute instead of six bpermute pairs
__device__ __forceinline__ uint64_t wave_max_u64_dpp(uint64_t v) {
  uint64_t o;
  o = xor_lane_u64<1>(v), v = o > v ? o : v;
  o = xor_lane_u64<2>(v), v = o > v ? o : v;
  o = xor_lane_u64<4>(v), v = o > v ? o : v;
  o = xor_lane_u64<8>(v), v = o > v ? o : v;
  o = xor_lane_u64<16>(v), v = o > v ? o : v;
  o = xor_lane_u64<32>(v), v = o > v ? o : v;
  return v;
}

// 64-lane max / min of a dword (the compiler folds the DPP move into v_max_u32_dpp / v_min_u32_dpp where it can)
__device__ __forceinline__ uint32_t wave_max_u32_dpp(uint32_t v) {
  uint32_t o;
  o = xor_lane<1>(v), v = o > v ? o : v;
  o = xor_lane<2>(v), v = o > v ? o : v;
  o = xor_lane<4>(v), v = o > v ? o : v;
  o = xor_lane<8>(v), v = o > v ? o : v;
  o = xor_lane<16>(v), v = o > v ? o : v;
  o = xor_lane<32>(v), v = o > v ? o : v;
  return v;
}
// 64-lane OR of a dword
__device__ __forceinline__ uint32_t wave_or_u32_dpp(uint32_t v) {
  v |= xor_lane<1>(v);
  v |= xor_lane<2>(v);
  v |= xor_lane<4>(v);
  v |= xor_lane<8>(v);
  v |= xor_lane<16>(v);
  v |= xor_lane<32>(v);
  return v;
}
__device__ __forceinline__ uint32_t wave_min_u32_dpp(uint32_t v) {
  uint32_t o;
  o = xor_lane<1>(v), v = o < v ? o : v;
  o = xor_lane<2>(v), v = o < v ? o : v;
  o = xor_lane<4>(v), v = o < v ? o : v;
  o = xor_lane<8>(v), v = o < v ? o : v;
  o = xor_lane<16>(v), v = o < v ? o : v;
  o = xor_lane<32>(v), v = o < v ? o : v;
  return v;
}

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int mask) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  lo = __shfl_xor(lo, mask, PZN_WAVE);
  hi = __shfl_xor(hi, mask, PZN_WAVE);
  return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    uint64_t o = shfl_xor_u64(v, m);
    v = o > v ? o : v;
  }
  return v;
}

__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    uint64_t o = shfl_xor_u64(v, m);
    v = o < v ? o : v;
  }
  return v;
}

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, PZN_WAVE);
  return v;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (PZN_WAVE - 1); }

// Lanes of ONE wavefront exchanging data through LDS: the hardware executes a wave's LDS instructions in order, so no
// s_barrier is needed, but the COMPILER must be told that other lanes wrote memory: llvm.amdgcn.wave.barrier alone is
// declared without memory effects, and a load hoisted or re-used across it reads a stale value (seen: lanes 32..63
// of a merge loop kept the previous iteration's keys).  The wavefront-scope fence is the IR-level memory barrier
// (it emits no instruction beyond the wait for outstanding LDS operations); the wave barrier pins the schedule.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

}  // namespace pzn
