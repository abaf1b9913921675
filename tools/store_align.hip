// store_align.hip — does the alignment of wave-wide 1 KB stores matter for a streaming write of [.., 32, 3 + D] blocks?
//   hipcc --offload-arch=gfx950 -O3 tools/store_align.hip -o scratch/store_align && scratch/store_align
// One wavefront per 16 768-byte (D = 128) or 8 576-byte (D = 64) block, 16 bytes per lane and store, non-temporal:
//   mode 0: the block as consecutive 1 KB spans from its (128-byte aligned) base + a tail
//   mode 1: the block in PIECES of KP rows (4 192 / 4 288 bytes), each as 1 KB spans from the piece's base + a tail — the
//           layout of knn_select_kernel's piece_finish (odd pieces start 32 bytes off a 64-byte boundary)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int BYTES, int PIECE, int MODE, bool NT>
__global__ __launch_bounds__(512) void k(float* out, int nblocks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int b = blockIdx.x * 8 + wave; b < nblocks; b += gridDim.x * 8) {
    char* base = reinterpret_cast<char*>(out) + (size_t)b * BYTES;
    if (MODE == 0) {
#pragma unroll
      for (int o = 0; o < BYTES; o += 1024)
        if (o + lane * 16 < BYTES) {
          if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(base + o + lane * 16));
          else *reinterpret_cast<f4*>(base + o + lane * 16) = v;
        }
    } else {
#pragma unroll
      for (int p = 0; p < BYTES; p += PIECE)
#pragma unroll
        for (int o = 0; o < PIECE; o += 1024)
          if (o + lane * 16 < PIECE) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(base + p + o + lane * 16));
            else *reinterpret_cast<f4*>(base + p + o + lane * 16) = v;
          }
    }
  }
}

template <int BYTES, int PIECE, int MODE, bool NT>
void run(const char* name, float* d, int nblocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  for (int grid : {512, 2048}) {
    hipLaunchKernelGGL((k<BYTES, PIECE, MODE, NT>), dim3(grid), dim3(512), 0, 0, d, nblocks);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<BYTES, PIECE, MODE, NT>), dim3(grid), dim3(512), 0, 0, d, nblocks);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    printf("%-34s grid %4d: %7.1f us  %6.0f GB/s\n", name, grid, ms * 1e3, (double)nblocks * BYTES / (ms * 1e-3) / 1e9);
  }
}

int main() {
  float* d;
  hipMalloc(&d, (size_t)300 << 20);
  run<16768, 4192, 0, true>("D=128 block as 1 KB spans, nt", d, 16384);
  run<16768, 4192, 1, true>("D=128 pieces of 8 rows, nt", d, 16384);
  run<16768, 4192, 0, false>("D=128 block as 1 KB spans", d, 16384);
  run<16768, 4192, 1, false>("D=128 pieces of 8 rows", d, 16384);
  run<8576, 4288, 0, true>("D=64 block as 1 KB spans, nt", d, 32768);
  run<8576, 4288, 1, true>("D=64 pieces of 16 rows, nt", d, 32768);
  run<8576, 4288, 0, false>("D=64 block as 1 KB spans", d, 32768);
  run<8576, 4288, 1, false>("D=64 pieces of 16 rows", d, 32768);
  return 0;
}
