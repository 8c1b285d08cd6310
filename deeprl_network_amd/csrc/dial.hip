// lstm_dial's message adjoint of one reverse step of the update (the backward of agents/utils.py:560-580), all agents in one
// launch.  Forward (reference): msg_j = relu(h_j W_mfc + b) on the sender, hm_i = relu([msg_j : j in nbr(i)] W_msg + b_msg) on
// the receiver, s_i = enc_i + hm_i the LSTM input.  Backward, given ds_i = dL/ds_i of this step (nmarl_lstm_bptt_step's dx):
//   d1_i   = ds_i * (hm_i > 0)                                                     gradient at the receiver layer's pre-activation
//   dmsg_j = sum over (i, k) with nbr(i, k) = j of  d1_i @ W_msg_i[64 k : 64 k + 64, :]^T      (adjoint of the gather)
//   d2_j   = dmsg_j * (msg_j > 0)                                                  gradient at the sender layer's pre-activation
//   dh_j   = dhd_j + d2_j @ W_mfc_j^T                                              dL/dh_{t-1}: recurrent + message part
// which the step-wise loop did with a GEMM, a scatter kernel, four elementwise launches, a copy and a second GEMM per step.
// Every (i, k) term only needs rows of agent i's ds / hm of THIS step, complete before the launch: no exchange inside it.
// Both products run on the matrix cores like the step kernel's message pre-phase (lstm_mfma.hip, NMARL_MCHUNK): a wave owns
// 16 rows, A operands straight from global memory as float4 (k order permuted, same permutation in the LDS image), weights
// as nmarl_lstm_msg_wimage images of the TRANSPOSED blocks, staged in LDS once per block; d2 goes through the wave's LDS tile
// from the C/D layout into the A layout of the second product.  HBM-bound: per row (3 + 2 sources) reads and 3 writes of 256 B.
#include "common.h"

namespace {

constexpr int H = 64;
constexpr int ROWS_B = 128, WAVES = 8, R16 = 16;
constexpr int APITCH = H + 1;
constexpr int IMG = H * H;                       // floats of one 64 x 64 image
constexpr int CH_K = 32;
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AdjArgs {
    const float *ds, *hm, *msg, *dhd;
    int64_t ds_sn, hm_sn, msg_sn, dhd_sn;
    const float* img_msg; int64_t img_msg_sn;    // per agent: m_max blocks of 64 x 64 (block k = W_msg[64 k : 64 k + 64, :]^T)
    const float* img_mfc; int64_t img_mfc_sn;    // per agent: W_mfc^T
    const int32_t *rev_agent, *rev_col;          // [N, RS]: source agent, 64 * slot
    const float* rev_w;                          // [N, RS]: 1 for a source, 0 for padding
    float *d1, *d2, *dh;
    int64_t d1_sn, d2_sn, dh_sn;
    float *b1_part, *b2_part;                    // [N][parts][64] running column sums of d1 / d2 (the two bias gradients), or NULL
    int64_t b1_sn, b2_sn;
    int64_t E;
    int blocks_per_agent;
};

#define NMARL_DSTEP(ACC, av, kl)                                                               \
        {                                                                                      \
            const float4 b_ = *reinterpret_cast<const float4*>(mb + (kl) * 64);                \
            ACC[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b_.x, ACC[0], 0, 0, 0);          \
            ACC[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b_.y, ACC[1], 0, 0, 0);          \
            ACC[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b_.z, ACC[2], 0, 0, 0);          \
            ACC[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b_.w, ACC[3], 0, 0, 0);          \
        }
// 32 k rows of a 64-column product from an LDS image [k][c][4 t]; lane (c = lane & 15, grp = lane >> 4) holds the A values of row c
// at k = 32 kc + 16 j + 4 grp + {0..3} in m0 (j = 0) / m1 (j = 1); ACC[t][r] = row 4 grp + r, unit 4 c + t
#define NMARL_DCHUNK(ACC, IMGP, kc, m0, m1)                                                    \
        {                                                                                      \
            const float* mb = (IMGP) + (((kc) * CH_K + 4 * grp) * 16 + c) * 4;                 \
            NMARL_DSTEP(ACC, m0.x, 0) NMARL_DSTEP(ACC, m0.y, 1) NMARL_DSTEP(ACC, m0.z, 2) NMARL_DSTEP(ACC, m0.w, 3)  \
            NMARL_DSTEP(ACC, m1.x, 16) NMARL_DSTEP(ACC, m1.y, 17) NMARL_DSTEP(ACC, m1.z, 18) NMARL_DSTEP(ACC, m1.w, 19) \
        }

__device__ __forceinline__ float4 relu_grad(const float4 g, const float4 y, const float w) {
    return float4{y.x > 0.0f ? g.x * w : 0.0f, y.y > 0.0f ? g.y * w : 0.0f, y.z > 0.0f ? g.z * w : 0.0f, y.w > 0.0f ? g.w * w : 0.0f};
}

template <int RS>
__global__ __launch_bounds__(512, 1) void dial_msg_adjoint_kernel(const AdjArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int n, blk_u;
    nmarl_xcd_work(blockIdx.x, gridDim.x, a.blocks_per_agent, n, blk_u);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, grp = lane >> 4;
    const int64_t row0 = (int64_t)blk_u * ROWS_B + wave * R16;
    float* img_s = lds;                                              // RS source images
    float* img_f = lds + RS * IMG;                                   // W_mfc^T
    float* tile = img_f + IMG + wave * R16 * APITCH;

    int sa[RS], sc[RS];
    float sw[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        sa[s] = a.rev_agent[n * RS + s];
        sc[s] = a.rev_col[n * RS + s];
        sw[s] = a.rev_w[n * RS + s];
    }
    // weight images -> LDS: 1024 float4 per image, two per thread (all requested before the first store)
    float4 wv[RS + 1][2];
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        const float4* g = reinterpret_cast<const float4*>(a.img_msg + (int64_t)sa[s] * a.img_msg_sn + (int64_t)sc[s] * H);
        wv[s][0] = g[threadIdx.x]; wv[s][1] = g[threadIdx.x + 512];
    }
    {
        const float4* g = reinterpret_cast<const float4*>(a.img_mfc + (int64_t)n * a.img_mfc_sn);
        wv[RS][0] = g[threadIdx.x]; wv[RS][1] = g[threadIdx.x + 512];
    }
    // A operands of the first product: rows of the SOURCES' ds / hm (absent sources: the own rows with weight 0)
    const int64_t arow = row0 + c < a.E ? row0 + c : a.E - 1;
    float4 ug[RS][4], uy[RS][4];
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        const float* g_ = a.ds + (int64_t)sa[s] * a.ds_sn + arow * H + 4 * grp;
        const float* y_ = a.hm + (int64_t)sa[s] * a.hm_sn + arow * H + 4 * grp;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                // q = 2 kc + j: k = 32 kc + 16 j + 4 grp
            ug[s][q] = *reinterpret_cast<const float4*>(g_ + 16 * q);
            uy[s][q] = *reinterpret_cast<const float4*>(y_ + 16 * q);
        }
    }
    // the own rows in the C/D layout (row 4 grp + r, units 4 c .. 4 c + 3)
    int64_t rofs[4];
    float4 og[4], oy[4], om[4], od[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 4 * grp + r;
        rofs[r] = row < a.E ? row : a.E - 1;
        og[r] = *reinterpret_cast<const float4*>(a.ds + (int64_t)n * a.ds_sn + rofs[r] * H + 4 * c);
        oy[r] = *reinterpret_cast<const float4*>(a.hm + (int64_t)n * a.hm_sn + rofs[r] * H + 4 * c);
        om[r] = *reinterpret_cast<const float4*>(a.msg + (int64_t)n * a.msg_sn + rofs[r] * H + 4 * c);
        od[r] = *reinterpret_cast<const float4*>(a.dhd + (int64_t)n * a.dhd_sn + rofs[r] * H + 4 * c);
    }
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        float4* d = reinterpret_cast<float4*>(img_s + s * IMG);
        d[threadIdx.x] = wv[s][0]; d[threadIdx.x + 512] = wv[s][1];
    }
    {
        float4* d = reinterpret_cast<float4*>(img_f);
        d[threadIdx.x] = wv[RS][0]; d[threadIdx.x + 512] = wv[RS][1];
    }
    __syncthreads();

    f32x4 macc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) macc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < RS; ++s) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const float4 m0 = relu_grad(ug[s][2 * kc], uy[s][2 * kc], sw[s]);
            const float4 m1 = relu_grad(ug[s][2 * kc + 1], uy[s][2 * kc + 1], sw[s]);
            NMARL_DCHUNK(macc, img_s + s * IMG, kc, m0, m1)
        }
    }
    // d1 (own rows), d2 = dmsg * (msg > 0) -> global memory and the wave's tile
    float* d1n = a.d1 + (int64_t)n * a.d1_sn;
    float* d2n = a.d2 + (int64_t)n * a.d2_sn;
    float4 s1 = float4{0.f, 0.f, 0.f, 0.f}, s2 = float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool live = row0 + 4 * grp + r < a.E;
        const float4 v1 = relu_grad(og[r], oy[r], 1.0f);
        const float4 v2 = relu_grad(float4{macc[0][r], macc[1][r], macc[2][r], macc[3][r]}, om[r], 1.0f);
        if (live) {
            *reinterpret_cast<float4*>(d1n + rofs[r] * H + 4 * c) = v1;
            *reinterpret_cast<float4*>(d2n + rofs[r] * H + 4 * c) = v2;
            s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
            s2.x += v2.x; s2.y += v2.y; s2.z += v2.z; s2.w += v2.w;
        }
        float* t_ = tile + (4 * grp + r) * APITCH + 4 * c;
        t_[0] = v2.x; t_[1] = v2.y; t_[2] = v2.z; t_[3] = v2.w;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // dh = dhd + d2 @ W_mfc^T
    f32x4 acc2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc2[0][r] = od[r].x; acc2[1][r] = od[r].y; acc2[2][r] = od[r].z; acc2[3][r] = od[r].w; }
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        const float* t_ = tile + c * APITCH + kc * CH_K + 4 * grp;
        const float4 m0 = float4{t_[0], t_[1], t_[2], t_[3]};
        const float4 m1 = float4{t_[16], t_[17], t_[18], t_[19]};
        NMARL_DCHUNK(acc2, img_f, kc, m0, m1)
    }
    float* dhn = a.dh + (int64_t)n * a.dh_sn;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (row0 + 4 * grp + r < a.E)
            *reinterpret_cast<float4*>(dhn + rofs[r] * H + 4 * c) = float4{acc2[0][r], acc2[1][r], acc2[2][r], acc2[3][r]};
    // the two bias gradients on the way: this wave's 16-row column sums, added to ITS slot of the running partial sums (one
    // launch per reverse step on one stream: the read-modify-write needs no atomics; fixed order -> deterministic)
    if (a.b1_part) {
#define NMARL_QSUM(v) v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        NMARL_QSUM(s1.x) NMARL_QSUM(s1.y) NMARL_QSUM(s1.z) NMARL_QSUM(s1.w)
        NMARL_QSUM(s2.x) NMARL_QSUM(s2.y) NMARL_QSUM(s2.z) NMARL_QSUM(s2.w)
#undef NMARL_QSUM
        if (grp == 0) {
            const int64_t slot = ((int64_t)blk_u * WAVES + wave) * H + 4 * c;
            float4* p1 = reinterpret_cast<float4*>(a.b1_part + (int64_t)n * a.b1_sn + slot);
            float4* p2 = reinterpret_cast<float4*>(a.b2_part + (int64_t)n * a.b2_sn + slot);
            const float4 o1 = *p1, o2 = *p2;
            *p1 = float4{o1.x + s1.x, o1.y + s1.y, o1.z + s1.z, o1.w + s1.w};
            *p2 = float4{o2.x + s2.x, o2.y + s2.y, o2.z + s2.z, o2.w + s2.w};
        }
    }
}

inline bool panel_ok(const float* p, int64_t sn, int64_t E) { return p && sn >= E * H && (sn % 4) == 0 && ((uintptr_t)p % 16) == 0; }

template <int RS>
int launch_adj(const AdjArgs& a, unsigned grid, hipStream_t st) {
    const size_t lb = (size_t)((RS + 1) * IMG + WAVES * R16 * APITCH) * sizeof(float);
    static NmarlPerDeviceOnce once;
    if (const unsigned long long bit = once.pending(); bit != ~0ull) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(dial_msg_adjoint_kernel<RS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lb) != hipSuccess)
            return NMARL_EHIP;
        once.done(bit);
    }
    hipLaunchKernelGGL(dial_msg_adjoint_kernel<RS>, dim3(grid), dim3(512), lb, st, a);
    return nmarl_check_launch();
}

}  // namespace

extern "C" int nmarl_dial_msg_adjoint_parts(int64_t E) { return (int)((E + ROWS_B - 1) / ROWS_B * WAVES); }

extern "C" int nmarl_dial_msg_adjoint(int64_t E, int32_t N, int32_t m_max, const float* ds, int64_t ds_sn, const float* hm,
                                      int64_t hm_sn, const float* msg, int64_t msg_sn, const float* dhd, int64_t dhd_sn,
                                      const float* img_msg_t, int64_t img_msg_sn, const float* img_mfc_t, int64_t img_mfc_sn,
                                      const int32_t* rev_agent, const int32_t* rev_col, const float* rev_w, int32_t r_row,
                                      float* d1, int64_t d1_sn, float* d2, int64_t d2_sn, float* dh, int64_t dh_sn,
                                      float* b1_part, int64_t b1_sn, float* b2_part, int64_t b2_sn, void* stream) {
    if (E < 0 || N <= 0 || m_max <= 0 || m_max > 4 || (r_row != 2 && r_row != 4)) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    if (!panel_ok(ds, ds_sn, E) || !panel_ok(hm, hm_sn, E) || !panel_ok(msg, msg_sn, E) || !panel_ok(dhd, dhd_sn, E) ||
        !panel_ok(d1, d1_sn, E) || !panel_ok(d2, d2_sn, E) || !panel_ok(dh, dh_sn, E) || !img_msg_t || !img_mfc_t ||
        img_msg_sn < (int64_t)m_max * IMG || (img_msg_sn % 4) || ((uintptr_t)img_msg_t % 16) || img_mfc_sn < IMG || (img_mfc_sn % 4) ||
        ((uintptr_t)img_mfc_t % 16) || !rev_agent || !rev_col || !rev_w)
        return NMARL_EINVAL;
    const int64_t parts = (E + ROWS_B - 1) / ROWS_B * WAVES;
    if ((b1_part != nullptr) != (b2_part != nullptr) ||
        (b1_part && (b1_sn < parts * H || b2_sn < parts * H || (b1_sn % 4) || (b2_sn % 4) || ((uintptr_t)b1_part % 16) || ((uintptr_t)b2_part % 16))))
        return NMARL_EINVAL;
    AdjArgs a{};
    a.ds = ds; a.hm = hm; a.msg = msg; a.dhd = dhd; a.ds_sn = ds_sn; a.hm_sn = hm_sn; a.msg_sn = msg_sn; a.dhd_sn = dhd_sn;
    a.img_msg = img_msg_t; a.img_msg_sn = img_msg_sn; a.img_mfc = img_mfc_t; a.img_mfc_sn = img_mfc_sn;
    a.rev_agent = rev_agent; a.rev_col = rev_col; a.rev_w = rev_w;
    a.d1 = d1; a.d2 = d2; a.dh = dh; a.d1_sn = d1_sn; a.d2_sn = d2_sn; a.dh_sn = dh_sn;
    a.b1_part = b1_part; a.b2_part = b2_part; a.b1_sn = b1_sn; a.b2_sn = b2_sn;
    a.E = E;
    a.blocks_per_agent = (int)((E + ROWS_B - 1) / ROWS_B);
    const unsigned grid = (unsigned)(a.blocks_per_agent * N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return r_row == 2 ? launch_adj<2>(a, grid, st) : launch_adj<4>(a, grid, st);
}
