// Fused BPTT step of the agent-batched 64-unit LSTM on gfx950 matrix cores: the cell backward AND the dgrad product
// of one time step in ONE kernel,
//
//     dz_t      = d cell / d z   from the saved gates, c_{t-1}, c_t and dL/dh_t = dh + dh2, dL/dc_t      (elementwise)
//     [dx | dh] = dz_t @ [Wxm; Wh]^T     (K = 256 -> KM + 64 outputs, fp32 MFMA)
//     dh_rec    = dh * (1 - done_t)      (the recurrent part of dL/dh_{t-1});   dx optionally * (mask > 0)  (relu)
//
// i.e. one reverse step of the unrolled training graph of agents/utils.py:102-113 (lstm), 199-208 (lstm_comm: Wxm =
// the rows of wx_hid the message third hm meets, mask = hm), 401-408 (lstm_ic3: Wxm = wx_hid), 585-593 (lstm_dial).
// The separate pair -- nmarl_lstm_cell_bwd (writes dz) + a library GEMM (re-reads dz, writes dh) + the next cell_bwd
// (re-reads dh) -- moved 167 MB per step at E = 4096; this kernel moves 126 MB and is bound by that HBM traffic
// (the 1.07 / 2.15 GFLOP of the product hide behind it).
//
// Mapping.  512-thread block = 8 waves x 16 rows of ONE agent (blockIdx % N).  The A operand of the MFMA is dz itself,
// produced in registers in A layout: lane (row = lane & 15, q = lane >> 4) owns units {16 j + 4 q + i} of its row (j, i
// = 0..3) -- for every j one float4 per tensor (gates i/f/o/u, c_prev, c_new, dh, dh2, dc: 64-byte segments), from
// which it computes the 16 values dz[g][i] and stores them (float4 per gate: dz is needed again by the weight-gradient
// GEMMs over all T*E rows).  k-step s = (j, g, i) of the product takes the lane's dz[g][i]; the B operand is read from
// an LDS image of [Wxm; Wh]^T permuted the same way (nmarl_lstm_bptt_wimage, rebuilt once per update, 64 or 128 KB,
// resident for the whole block): image[(s, q)][c][slot(t)] = W[16 t + c][64 g + 16 j + 4 q + i].
// With 8 output tiles the 32-byte lane pitch is swizzled ((t >> 2) ^ (c >> 3)) so that every ds_read_b128 is
// conflict-free.  Loads of unit group j + 1 are issued before the MFMAs of group j.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int H = 64;
constexpr int G4 = 4 * H;
constexpr int R16 = 16;
constexpr int WAVES = 8;
constexpr int ROWS_B = R16 * WAVES;

__device__ __forceinline__ float sigm_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast_(float x) { return 2.0f * sigm_(2.0f * x) - 1.0f; }

struct BpttArgs {
    const float *gates, *c_prev, *c_new, *done, *dh, *dh2, *dc_in, *img, *mask;
    float *dz, *dc_prev, *dx, *dhd;
    int64_t gates_sn, c_prev_sn, c_new_sn, dh_sn, dh2_sn, dc_sn, img_sn, mask_sn, mask_row, dz_sn, dc_prev_sn, dx_sn, dhd_sn;
    int64_t E;
    int N, apply_keep;
};

struct UnitGroup {          // inputs of 4 consecutive units of one row
    float4 gi, gf, go, gu, cp, cn, gh, g2, gc;
};

template <int NT>           // NT = output column tiles: 4 (dh only) or 8 ([dx | dh])
__global__ __launch_bounds__(512, 1) void lstm_bptt_step_kernel(const BpttArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int PITCH = NT;                        // floats per (k, c) entry: 4 or 8
    constexpr int IMG_FLOATS = G4 * 16 * PITCH;
    const int n = blockIdx.x % a.N;
    const int64_t row_blk = (int64_t)(blockIdx.x / a.N) * ROWS_B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = row_blk + wave * R16;
    const int c = lane & 15, q = lane >> 4;
    const int64_t arow_raw = row0 + c;
    const bool arow_ok = arow_raw < a.E;
    const int64_t arow = arow_ok ? arow_raw : a.E - 1;

    // ---- stage the whole image (same order in global memory and LDS)
    {
        const float4* g = reinterpret_cast<const float4*>(a.img + (int64_t)n * a.img_sn);
        float4* d = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int i = 0; i < IMG_FLOATS / 4 / 512; ++i) d[i * 512 + threadIdx.x] = g[i * 512 + threadIdx.x];
    }

    const float keepA = 1.0f - a.done[arow];
    const float* gr = a.gates + (int64_t)n * a.gates_sn + arow * G4 + 4 * q;
    const float* cpr = a.c_prev + (int64_t)n * a.c_prev_sn + arow * H + 4 * q;
    const float* cnr = a.c_new + (int64_t)n * a.c_new_sn + arow * H + 4 * q;
    // absent gradient inputs read c_new instead and are multiplied by 0: the loads stay unconditional (a load inside
    // a branch makes the compiler's waitcnt pass drain everything in flight at the join)
    const float fh = a.dh ? 1.0f : 0.0f, f2 = a.dh2 ? 1.0f : 0.0f, fc = a.dc_in ? 1.0f : 0.0f;
    const float* dhr = a.dh ? a.dh + (int64_t)n * a.dh_sn + arow * H + 4 * q : cnr;
    const float* d2r = a.dh2 ? a.dh2 + (int64_t)n * a.dh2_sn + arow * H + 4 * q : cnr;
    const float* dcr = a.dc_in ? a.dc_in + (int64_t)n * a.dc_sn + arow * H + 4 * q : cnr;
    float* dzr = a.dz + (int64_t)n * a.dz_sn + arow * G4 + 4 * q;
    float* dcpr = a.dc_prev + (int64_t)n * a.dc_prev_sn + arow * H + 4 * q;

#define NMARL_LOADJ(U, j)                                                                  \
    {                                                                                      \
        U.gi = *reinterpret_cast<const float4*>(gr + 16 * (j));                            \
        U.gf = *reinterpret_cast<const float4*>(gr + H + 16 * (j));                        \
        U.go = *reinterpret_cast<const float4*>(gr + 2 * H + 16 * (j));                    \
        U.gu = *reinterpret_cast<const float4*>(gr + 3 * H + 16 * (j));                    \
        U.cp = *reinterpret_cast<const float4*>(cpr + 16 * (j));                           \
        U.cn = *reinterpret_cast<const float4*>(cnr + 16 * (j));                           \
        U.gh = *reinterpret_cast<const float4*>(dhr + 16 * (j));                           \
        U.g2 = *reinterpret_cast<const float4*>(d2r + 16 * (j));                           \
        U.gc = *reinterpret_cast<const float4*>(dcr + 16 * (j));                           \
    }
    UnitGroup u0, u1;
    NMARL_LOADJ(u0, 0)
    __syncthreads();                                 // image visible

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* bbase = lds + (q * 16 + c) * PITCH;             // + s * 64 * PITCH per k-step
    const int sw = NT == 8 ? (c >> 3) : 0;

    // one k-step: A value `av`, B operands of all tiles from image row (s, q)
#define NMARL_KSTEP(av, s)                                                                 \
    {                                                                                      \
        const float* p_ = bbase + (s) * 64 * PITCH;                                        \
        if (NT == 4) {                                                                     \
            const float4 b0 = *reinterpret_cast<const float4*>(p_);                        \
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.x, acc[0], 0, 0, 0);      \
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.y, acc[1], 0, 0, 0);      \
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.z, acc[2], 0, 0, 0);      \
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.w, acc[3], 0, 0, 0);      \
        } else {                                                                           \
            const float4 b0 = *reinterpret_cast<const float4*>(p_ + 4 * sw);               \
            const float4 b1 = *reinterpret_cast<const float4*>(p_ + 4 * (sw ^ 1));         \
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.x, acc[0], 0, 0, 0);      \
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.y, acc[1], 0, 0, 0);      \
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.z, acc[2], 0, 0, 0);      \
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.w, acc[3], 0, 0, 0);      \
            acc[NT - 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.x, acc[NT - 4], 0, 0, 0); \
            acc[NT - 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.y, acc[NT - 3], 0, 0, 0); \
            acc[NT - 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.z, acc[NT - 2], 0, 0, 0); \
            acc[NT - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.w, acc[NT - 1], 0, 0, 0); \
        }                                                                                  \
    }
    // cell backward of one unit (agents/utils.py:102-113 differentiated; same formulas as nmarl_lstm_cell_bwd)
#define NMARL_CELLB(U, k)                                                                  \
    {                                                                                      \
        const float tc = tanh_fast_(U.cn.k);                                               \
        const float gh_ = U.gh.k * fh + U.g2.k * f2;                                       \
        const float g_c = U.gc.k * fc + gh_ * U.go.k * (1.0f - tc * tc);                   \
        di.k = g_c * U.gu.k * U.gi.k * (1.0f - U.gi.k);                                    \
        df.k = g_c * (U.cp.k * keepA) * U.gf.k * (1.0f - U.gf.k);                          \
        dO.k = gh_ * tc * U.go.k * (1.0f - U.go.k);                                        \
        du.k = g_c * U.gi.k * (1.0f - U.gu.k * U.gu.k);                                    \
        dcp.k = g_c * U.gf.k * keepA;                                                      \
    }
#define NMARL_GROUP(U, j)                                                                  \
    {                                                                                      \
        float4 di, df, dO, du, dcp;                                                        \
        NMARL_CELLB(U, x) NMARL_CELLB(U, y) NMARL_CELLB(U, z) NMARL_CELLB(U, w)            \
        if (arow_ok) {                                                                     \
            *reinterpret_cast<float4*>(dzr + 16 * (j)) = di;                               \
            *reinterpret_cast<float4*>(dzr + H + 16 * (j)) = df;                           \
            *reinterpret_cast<float4*>(dzr + 2 * H + 16 * (j)) = dO;                       \
            *reinterpret_cast<float4*>(dzr + 3 * H + 16 * (j)) = du;                       \
            *reinterpret_cast<float4*>(dcpr + 16 * (j)) = dcp;                             \
        }                                                                                  \
        NMARL_KSTEP(di.x, (j) * 16 + 0) NMARL_KSTEP(di.y, (j) * 16 + 1)                    \
        NMARL_KSTEP(di.z, (j) * 16 + 2) NMARL_KSTEP(di.w, (j) * 16 + 3)                    \
        NMARL_KSTEP(df.x, (j) * 16 + 4) NMARL_KSTEP(df.y, (j) * 16 + 5)                    \
        NMARL_KSTEP(df.z, (j) * 16 + 6) NMARL_KSTEP(df.w, (j) * 16 + 7)                    \
        NMARL_KSTEP(dO.x, (j) * 16 + 8) NMARL_KSTEP(dO.y, (j) * 16 + 9)                    \
        NMARL_KSTEP(dO.z, (j) * 16 + 10) NMARL_KSTEP(dO.w, (j) * 16 + 11)                  \
        NMARL_KSTEP(du.x, (j) * 16 + 12) NMARL_KSTEP(du.y, (j) * 16 + 13)                  \
        NMARL_KSTEP(du.z, (j) * 16 + 14) NMARL_KSTEP(du.w, (j) * 16 + 15)                  \
    }
    NMARL_LOADJ(u1, 1)
    NMARL_GROUP(u0, 0)
    NMARL_LOADJ(u0, 2)
    NMARL_GROUP(u1, 1)
    NMARL_LOADJ(u1, 3)
    NMARL_GROUP(u0, 2)
    NMARL_GROUP(u1, 3)
#undef NMARL_LOADJ
#undef NMARL_KSTEP
#undef NMARL_CELLB
#undef NMARL_GROUP

    // ---- epilogue: C/D layout col = lane & 15, row = 4 (lane >> 4) + reg; tiles [0, NT-4) = dx, last 4 = dh
    float keepr[4];
    int64_t rows[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        rows[r] = row0 + 4 * q + r;
        keepr[r] = a.apply_keep ? 1.0f - a.done[rows[r] < a.E ? rows[r] : a.E - 1] : 1.0f;
    }
    float* dhn = a.dhd + (int64_t)n * a.dhd_sn;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (rows[r] < a.E) dhn[rows[r] * H + 16 * t + c] = acc[NT - 4 + t][r] * keepr[r];
    if (NT == 8) {
        float* dxn = a.dx + (int64_t)n * a.dx_sn;
        const float* mk = a.mask ? a.mask + (int64_t)n * a.mask_sn : nullptr;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (rows[r] < a.E) {
                    float v = acc[t][r];
                    if (mk && !(mk[rows[r] * a.mask_row + 16 * t + c] > 0.0f)) v = 0.0f;
                    dxn[rows[r] * H + 16 * t + c] = v;
                }
    }
}

// image[(s, q)][c][slot(t)] = Wb[16 t + c][64 g + 16 j + 4 q + i],  s = 16 j + 4 g + i,  Wb = [wxm (KM rows); wh (64 rows)]
__global__ void lstm_bptt_wimage_kernel(const int N, const int KM, const float* wxm, const int64_t wxm_sn, const float* wh,
                                        const int64_t wh_sn, float* img, const int64_t img_sn) {
    const int NT = (KM + H) / 16;
    const int per_agent = G4 * 16 * NT;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * per_agent) return;
    const int n = (int)(idx / per_agent), o = (int)(idx % per_agent);
    const int slot = o % NT, cc = (o / NT) % 16, kq = o / (NT * 16);
    const int qq = kq & 3, s = kq >> 2;
    const int j = s >> 4, g = (s >> 2) & 3, i = s & 3;
    const int col = 64 * g + 16 * j + 4 * qq + i;
    int t = slot;
    if (NT == 8) t = (((slot >> 2) ^ (cc >> 3)) << 2) | (slot & 3);
    const int rowb = 16 * t + cc;
    const float v = rowb < KM ? wxm[(int64_t)n * wxm_sn + (int64_t)rowb * G4 + col] : wh[(int64_t)n * wh_sn + (int64_t)(rowb - KM) * G4 + col];
    img[(int64_t)n * img_sn + o] = v;
}

inline bool sn_ok(int64_t s, int64_t need) { return s >= need && (s % 4) == 0; }

}  // namespace

extern "C" int nmarl_lstm_bptt_wimage_floats(int32_t KM) { return G4 * (KM + H); }

extern "C" int nmarl_lstm_bptt_wimage(int32_t N, int32_t KM, const float* wxm, int64_t wxm_sn, const float* wh, int64_t wh_sn,
                                      float* img, int64_t img_sn, void* stream) {
    if (N <= 0 || (KM != 0 && KM != H) || !wh || !img || (KM > 0 && !wxm) || img_sn < (int64_t)G4 * (KM + H) || (img_sn % 4) ||
        ((uintptr_t)img % 16) || wh_sn < H * G4 || (KM > 0 && wxm_sn < (int64_t)KM * G4))
        return NMARL_EINVAL;
    const int64_t total = (int64_t)N * G4 * (KM + H);
    hipLaunchKernelGGL(lstm_bptt_wimage_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       N, KM, wxm, wxm_sn, wh, wh_sn, img, img_sn);
    return nmarl_check_launch();
}

extern "C" int nmarl_lstm_bptt_step(int64_t E, int32_t N, int32_t Hh, int32_t KM, const float* gates, int64_t gates_sn,
                                    const float* c_prev, int64_t c_prev_sn, const float* c_new, int64_t c_new_sn,
                                    const float* done, const float* dh, int64_t dh_sn, const float* dh2, int64_t dh2_sn,
                                    const float* dc_in, int64_t dc_sn, const float* img, int64_t img_sn, float* dz,
                                    int64_t dz_sn, float* dc_prev, int64_t dc_prev_sn, float* dx, int64_t dx_sn,
                                    const float* mask, int64_t mask_sn, int64_t mask_row, float* dhd, int64_t dhd_sn,
                                    int32_t apply_keep, void* stream) {
    if (Hh != H || E < 0 || N <= 0 || (KM != 0 && KM != H) ||
        (E > 0 && (!gates || !c_prev || !c_new || !done || !img || !dz || !dc_prev || !dhd || (KM > 0 && !dx))))
        return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    if (!sn_ok(gates_sn, E * G4) || !sn_ok(dz_sn, E * G4) || !sn_ok(c_prev_sn, E * H) || !sn_ok(c_new_sn, E * H) ||
        !sn_ok(dc_prev_sn, E * H) || !sn_ok(dhd_sn, E * H) || (dh && !sn_ok(dh_sn, E * H)) || (dh2 && !sn_ok(dh2_sn, E * H)) ||
        (dc_in && !sn_ok(dc_sn, E * H)) || (KM > 0 && !sn_ok(dx_sn, E * H)) || (mask && (mask_row < H || mask_sn < E * mask_row)) ||
        img_sn < (int64_t)G4 * (KM + H) || (img_sn % 4) || ((uintptr_t)img % 16) || ((uintptr_t)gates % 16) || ((uintptr_t)dz % 16) ||
        ((uintptr_t)c_prev % 16) || ((uintptr_t)c_new % 16) || ((uintptr_t)dc_prev % 16) || (dh && ((uintptr_t)dh % 16)) ||
        (dh2 && ((uintptr_t)dh2 % 16)) || (dc_in && ((uintptr_t)dc_in % 16)))
        return NMARL_EINVAL;
    BpttArgs a{};
    a.gates = gates; a.c_prev = c_prev; a.c_new = c_new; a.done = done; a.dh = dh; a.dh2 = dh2; a.dc_in = dc_in; a.img = img;
    a.mask = KM > 0 ? mask : nullptr; a.dz = dz; a.dc_prev = dc_prev; a.dx = dx; a.dhd = dhd;
    a.gates_sn = gates_sn; a.c_prev_sn = c_prev_sn; a.c_new_sn = c_new_sn; a.dh_sn = dh_sn; a.dh2_sn = dh2_sn; a.dc_sn = dc_sn;
    a.img_sn = img_sn; a.mask_sn = mask_sn; a.mask_row = mask_row; a.dz_sn = dz_sn; a.dc_prev_sn = dc_prev_sn; a.dx_sn = dx_sn;
    a.dhd_sn = dhd_sn; a.E = E; a.N = N; a.apply_keep = apply_keep;
    static bool lds_set = false;
    if (!lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_step_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                G4 * 16 * 4 * 4) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_step_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                G4 * 16 * 8 * 4) != hipSuccess)
            return NMARL_EHIP;
        lds_set = true;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(((E + ROWS_B - 1) / ROWS_B) * N));
    if (KM == 0) hipLaunchKernelGGL(lstm_bptt_step_kernel<4>, grid, dim3(512), (size_t)G4 * 16 * 4 * 4, st, a);
    else hipLaunchKernelGGL(lstm_bptt_step_kernel<8>, grid, dim3(512), (size_t)G4 * 16 * 8 * 4, st, a);
    return nmarl_check_launch();
}
