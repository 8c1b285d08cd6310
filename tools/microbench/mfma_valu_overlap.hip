// Does vector-ALU work of one wave hide behind the fp32 matrix instructions of another wave on the same SIMD (gfx950)?
// One 512-thread block per CU = 2 waves per SIMD: waves 0-3 (one per SIMD) issue dependent-free chains of
// v_mfma_f32_16x16x4_f32, waves 4-7 chains of v_fma_f32 (mode bit 0: matrix waves active, bit 1: vector waves active).
// If the two pipes were independent, mode 3 would take max(mode 1, mode 2); if they share the fp32 FMA datapath, the sum.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

extern "C" __global__ __launch_bounds__(512, 1) void overlap_kernel(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const float seed = (float)(threadIdx.x & 63) * 1e-3f;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x4 acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = f32x4{seed, 0.f, 0.f, 0.f};
        float a = seed, b = seed + 1.0f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        if (!(mode & 2)) return;
        float v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = seed + t;
        const float m = 1.0001f, c = 1e-3f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r)          // 8 x 16 = 128 independent-enough FMAs per iteration = 512 issue cycles
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = __builtin_fmaf(v[t], m, c);
        }
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) s += v[t];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

extern "C" int overlap_launch(float* out, int blocks, int iters, int mode, void* stream) {
    hipLaunchKernelGGL(overlap_kernel, dim3(blocks), dim3(512), 0, static_cast<hipStream_t>(stream), out, iters, mode);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
