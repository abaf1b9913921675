// pzn_internal.h — functions shared between translation units of libpzn.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// poolbwd.hip: sparse backward of linear + ReLU + max over 32 neighbours.
//   dh != NULL: dh[G*32, C1] = scatter(dout) W, ReLU-masked by h when h != NULL (overwritten)
//   dW != NULL: dW[C2, C1] += scatter(dout)^T h,  db[C2] += column sums (db may be NULL)
bool pzn_pool_bwd_supported(int C1, int C2, const float* W, const float* h, const float* dh);
int pzn_pool_bwd_sparse(const float* dout, const int32_t* argmax, const float* out, const float* W, const float* h,
                        float* dh, float* dW, float* db, int G, int C1, int C2, hipStream_t st);
