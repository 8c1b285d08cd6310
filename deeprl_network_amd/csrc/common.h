// Shared helpers for the gfx950 kernels of libnmarl_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nmarl.h"

#define NMARL_WAVE 64

static inline int nmarl_check_launch() {
    return hipGetLastError() == hipSuccess ? NMARL_OK : NMARL_EHIP;
}

// Philox4x32-10; contract shared with oracle/philox.py.
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u01_from_bits(uint32_t w) {
    return (float)(w >> 8) * 5.9604644775390625e-08f;  // 2^-24, exact
}

#define NMARL_STREAM_RESET 0u
#define NMARL_STREAM_ACTION 1u
#define NMARL_STREAM_GRID 2u
