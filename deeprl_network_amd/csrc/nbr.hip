// Neighbourhood aggregation over the fixed agent adjacency, agent-major layout.
//
// Replaces the per-agent `tf.boolean_mask(X, masks[i])` + reshape/concat
// (NeurComm, agents/utils.py:192-195; neighbour actions for the critic,
// policies.py:305 + 66-68) and `tf.reduce_mean(tf.boolean_mask(out_m, masks[i]))`
// (CommNet, agents/utils.py:395) of the reference.  Neighbours are listed in
// ascending agent index (boolean_mask order), left-packed, -1 padded.
//
// Activations are agent-major [N, E, F]: the gather then is a block copy of
// contiguous [E,F] panels, done here with 16-byte accesses when F % 4 == 0.
// Backward passes are atomics-free (each dx element sums its own fan-in in a
// fixed order) so gradients are bit-reproducible run to run.
#include "common.h"

namespace {

constexpr int MAX_PAIRS = 1024;  // N * m_max upper bound held in LDS

template <typename V>   // V = float or float4
__global__ __launch_bounds__(256) void gather_fwd_kernel(
    const int64_t E, const int N, const int Fv, const int m_max, const int32_t* __restrict__ nbr,
    const V* __restrict__ x, V* __restrict__ y) {
    // grid.y = i (receiver), grid.z = k (slot); x dimension strides over E*Fv
    const int i = blockIdx.y, k = blockIdx.z;
    const int j = nbr[i * m_max + k];
    const int64_t n = E * Fv;
    const V* src = x + (int64_t)(j < 0 ? 0 : j) * n;
    V* dst = y + (int64_t)i * n * m_max;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = idx / Fv;
        const int f = (int)(idx - e * Fv);
        V val = V{};
        if (j >= 0) val = src[idx];
        dst[(e * m_max + k) * Fv + f] = val;
    }
}

__device__ __forceinline__ void acc(float& a, const float b) { a += b; }
__device__ __forceinline__ void acc(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ __forceinline__ float scale(const float a, const float s) { return a * s; }
__device__ __forceinline__ float4 scale(const float4 a, const float s) { return float4{a.x * s, a.y * s, a.z * s, a.w * s}; }

template <typename V>
__global__ __launch_bounds__(256) void gather_bwd_kernel(
    const int64_t E, const int N, const int Fv, const int m_max, const int32_t* __restrict__ nbr,
    const V* __restrict__ dy, const V* __restrict__ add, V* __restrict__ dx) {
    __shared__ int32_t s_nbr[MAX_PAIRS];
    for (int p = threadIdx.x; p < N * m_max; p += blockDim.x) s_nbr[p] = nbr[p];
    __syncthreads();
    const int j = blockIdx.y;   // source agent whose gradient is assembled
    const int64_t n = E * Fv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = idx / Fv;
        const int f = (int)(idx - e * Fv);
        V g = V{};
        for (int p = 0; p < N * m_max; ++p) {
            if (s_nbr[p] == j) {
                const int i = p / m_max, k = p - i * m_max;
                acc(g, dy[(int64_t)i * n * m_max + (e * m_max + k) * Fv + f]);
            }
        }
        if (add) acc(g, add[(int64_t)j * n + idx]);          // (sum over the fan-in) + add: the order of the separate pass
        dx[(int64_t)j * n + idx] = g;
    }
}

template <typename V>
__global__ __launch_bounds__(256) void mean_fwd_kernel(
    const int64_t E, const int N, const int Fv, const int m_max, const int32_t* __restrict__ nbr,
    const V* __restrict__ x, V* __restrict__ y) {
    const int i = blockIdx.y;
    int js[8], cnt = 0;
    for (int k = 0; k < m_max && k < 8; ++k) {
        const int j = nbr[i * m_max + k];
        if (j >= 0) js[cnt++] = j;
    }
    const float inv = cnt > 0 ? 1.0f / (float)cnt : 0.0f;
    const int64_t n = E * Fv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        V s = V{};
        for (int c = 0; c < cnt; ++c) acc(s, x[(int64_t)js[c] * n + idx]);
        y[(int64_t)i * n + idx] = scale(s, inv);
    }
}

template <typename V>
__global__ __launch_bounds__(256) void mean_bwd_kernel(
    const int64_t E, const int N, const int Fv, const int m_max, const int32_t* __restrict__ nbr,
    const V* __restrict__ dy, const V* __restrict__ add, V* __restrict__ dx) {
    __shared__ int32_t s_nbr[MAX_PAIRS];
    __shared__ float s_inv[MAX_PAIRS];
    for (int p = threadIdx.x; p < N * m_max; p += blockDim.x) s_nbr[p] = nbr[p];
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        int cnt = 0;
        for (int k = 0; k < m_max; ++k) cnt += s_nbr[i * m_max + k] >= 0;
        s_inv[i] = cnt > 0 ? 1.0f / (float)cnt : 0.0f;
    }
    __syncthreads();
    const int j = blockIdx.y;
    const int64_t n = E * Fv;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        V g = V{};
        for (int p = 0; p < N * m_max; ++p) {
            if (s_nbr[p] == j) {
                const int i = p / m_max;
                acc(g, scale(dy[(int64_t)i * n + idx], s_inv[i]));
            }
        }
        if (add) acc(g, add[(int64_t)j * n + idx]);
        dx[(int64_t)j * n + idx] = g;
    }
}

// one-hot of the neighbours' actions for the centralised critic (policies.py:66-68, 305)
__global__ __launch_bounds__(256) void nbr_onehot_kernel(
    const int64_t E, const int N, const int A, const int m_max, const int32_t* __restrict__ nbr,
    const uint8_t* __restrict__ action /*[E,N]*/, float* __restrict__ y /*[N,E,m_max*A]*/,
    const int64_t y_agent_stride) {
    const int i = blockIdx.y;
    const int W = m_max * A;
    const int64_t n = E * W;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = idx / W;
        const int r = (int)(idx - e * W);
        const int k = r / A, a = r - k * A;
        const int j = nbr[i * m_max + k];
        float val = 0.0f;
        if (j >= 0) val = action[e * N + j] == a ? 1.0f : 0.0f;
        y[(int64_t)i * y_agent_stride + idx] = val;
    }
}

inline int grid_x(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}

inline bool args_ok(int64_t E, int N, int F, int m_max, const void* a, const void* b, const void* c) {
    return E >= 0 && N > 0 && F > 0 && m_max > 0 && N * m_max <= MAX_PAIRS && m_max <= 8 && (E == 0 || (a && b && c));
}

inline bool vec4(int F, const void* a, const void* b) {
    return F % 4 == 0 && ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
}

}  // namespace

extern "C" int nmarl_nbr_gather_fwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                                    const float* x, float* y, void* stream) {
    if (!args_ok(E, N, F, m_max, nbr_idx, x, y)) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec4(F, x, y)) {
        hipLaunchKernelGGL(gather_fwd_kernel<float4>, dim3(grid_x(E * F / 4), N, m_max), dim3(256), 0, s, E, N, F / 4,
                           m_max, nbr_idx, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y));
    } else {
        hipLaunchKernelGGL(gather_fwd_kernel<float>, dim3(grid_x(E * F), N, m_max), dim3(256), 0, s, E, N, F, m_max,
                           nbr_idx, x, y);
    }
    return nmarl_check_launch();
}

static int launch_gather_bwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx, const float* dy, const float* add,
                             float* dx, void* stream) {
    if (!args_ok(E, N, F, m_max, nbr_idx, dy, dx)) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec4(F, dy, dx) && (!add || ((uintptr_t)add % 16) == 0)) {
        hipLaunchKernelGGL(gather_bwd_kernel<float4>, dim3(grid_x(E * F / 4), N), dim3(256), 0, s, E, N, F / 4, m_max,
                           nbr_idx, reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(add), reinterpret_cast<float4*>(dx));
    } else {
        hipLaunchKernelGGL(gather_bwd_kernel<float>, dim3(grid_x(E * F), N), dim3(256), 0, s, E, N, F, m_max, nbr_idx,
                           dy, add, dx);
    }
    return nmarl_check_launch();
}

extern "C" int nmarl_nbr_gather_bwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                                    const float* dy, float* dx, void* stream) {
    return launch_gather_bwd(E, N, F, m_max, nbr_idx, dy, nullptr, dx, stream);
}

extern "C" int nmarl_nbr_gather_bwd_add(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                                        const float* dy, const float* add, float* dx, void* stream) {
    if (!add) return NMARL_EINVAL;
    return launch_gather_bwd(E, N, F, m_max, nbr_idx, dy, add, dx, stream);
}

extern "C" int nmarl_nbr_mean_fwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                                  const float* x, float* y, void* stream) {
    if (!args_ok(E, N, F, m_max, nbr_idx, x, y)) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec4(F, x, y)) {
        hipLaunchKernelGGL(mean_fwd_kernel<float4>, dim3(grid_x(E * F / 4), N), dim3(256), 0, s, E, N, F / 4, m_max,
                           nbr_idx, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y));
    } else {
        hipLaunchKernelGGL(mean_fwd_kernel<float>, dim3(grid_x(E * F), N), dim3(256), 0, s, E, N, F, m_max, nbr_idx, x, y);
    }
    return nmarl_check_launch();
}

static int launch_mean_bwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx, const float* dy, const float* add,
                           float* dx, void* stream) {
    if (!args_ok(E, N, F, m_max, nbr_idx, dy, dx)) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec4(F, dy, dx) && (!add || ((uintptr_t)add % 16) == 0)) {
        hipLaunchKernelGGL(mean_bwd_kernel<float4>, dim3(grid_x(E * F / 4), N), dim3(256), 0, s, E, N, F / 4, m_max,
                           nbr_idx, reinterpret_cast<const float4*>(dy), reinterpret_cast<const float4*>(add), reinterpret_cast<float4*>(dx));
    } else {
        hipLaunchKernelGGL(mean_bwd_kernel<float>, dim3(grid_x(E * F), N), dim3(256), 0, s, E, N, F, m_max, nbr_idx, dy, add, dx);
    }
    return nmarl_check_launch();
}

extern "C" int nmarl_nbr_mean_bwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                                  const float* dy, float* dx, void* stream) {
    return launch_mean_bwd(E, N, F, m_max, nbr_idx, dy, nullptr, dx, stream);
}

extern "C" int nmarl_nbr_mean_bwd_add(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                                      const float* dy, const float* add, float* dx, void* stream) {
    if (!add) return NMARL_EINVAL;
    return launch_mean_bwd(E, N, F, m_max, nbr_idx, dy, add, dx, stream);
}

extern "C" int nmarl_nbr_onehot(int64_t E, int32_t N, int32_t A, int32_t m_max, const int32_t* nbr_idx,
                                const uint8_t* action, float* y, int64_t y_agent_stride, void* stream) {
    if (!args_ok(E, N, A, m_max, nbr_idx, action, y) || y_agent_stride < E * m_max * A) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipLaunchKernelGGL(nbr_onehot_kernel, dim3(grid_x(E * m_max * A), N), dim3(256), 0,
                       static_cast<hipStream_t>(stream), E, N, A, m_max, nbr_idx, action, y, y_agent_stride);
    return nmarl_check_launch();
}
