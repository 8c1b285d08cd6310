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

// ---------------------------------------------------------------------------------------------------------------------
// The whole reverse recurrence in ONE launch (nets without a cross-agent term in the recurrence: IA2C, IA2C-FP, ConseNet).
// Nothing couples two rows of a batch, so a block keeps its 128 rows for all T steps: dL/dc and the recurrent dL/dh
// stay in registers, c_{t-1} read for step t is step t-1's c_t, the weight image is staged once instead of T times, and
// the loads of step t-1 are in flight while step t computes.  Per step and row the kernel reads gates (1 KB), c_{t-1}
// and the heads' dL/dh_t (256 B each) and writes dz (1 KB): 84 MB at E = 4096 x 8 agents instead of the 126 MB (+ a
// launch) of nmarl_lstm_bptt_step.
//
// The product runs TRANSPOSED: dh^T = Wh . dz^T.  A lane's dz registers (row = lane & 15, units 16 j + 4 q + i) are
// exactly the B-operand layout (k = lane >> 4, n = lane & 15) of dz^T, the image rows are the A operand (m = out unit,
// k) -- the same image, the same addresses as the step kernel -- and the C/D layout of the result (row of D = out unit
// 4 q + reg of tile j, column = lane & 15 = batch row) hands every lane dh of ITS row and ITS units: no transposition
// between two steps, no LDS beyond the image.
//
// The bias gradient (column sums of dz over all T x rows) is accumulated on the way: per group of 16 values a two-level
// DPP reduce-scatter over the lane quad (rows c, c^1, c^2, c^3) leaves 4 sums per lane (one gate each), so the running
// sums cost 16 registers instead of 64; they leave as one [256] partial per block (summed by the caller).
// Arithmetic per step is that of lstm_bptt_step_kernel<4> with apply_keep = 1, operation for operation.
struct BpttSeqArgs {
    const float *gates, *c_all, *done, *dh_ext, *img;
    float *dz, *db_part, *dh0, *dc0;
    int64_t gates_sn, gates_st, c_sn, c_st, dh_sn, dh_st, img_sn, dz_sn, dz_st, db_sn, dh0_sn, dc0_sn;
    int64_t E;
    int N, T;
};

struct SeqGroup {           // per-step inputs of 4 consecutive units of one row
    float4 gi, gf, go, gu, cp, gh;
};

constexpr int SEQ_IMG = G4 * 16 * 4;                 // image floats (NT = 4)

typedef unsigned int u32x4v __attribute__((__vector_size__(16)));
// raw buffer access (no stride): an offset at or past num_records reads 0 / is not written -- rows past E need no
// branch (a branch around memory operations makes the compiler's waitcnt pass drain every load in flight at the join,
// i.e. there would be no prefetch at all) and no clamped duplicate rows
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, const uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 bload4(const __amdgpu_buffer_rsrc_t r, const uint32_t off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ void bstore4(const __amdgpu_buffer_rsrc_t r, const uint32_t off, const float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, off, 0, 0);
}

__device__ __forceinline__ float dpp_xor1(float v) {  // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {  // value of lane ^ 2 (quad_perm [2,3,0,1])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

__global__ __launch_bounds__(512, 1) void lstm_bptt_seq_kernel(const BpttSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = blockIdx.x % a.N;
    const int blk = blockIdx.x / a.N;
    const int64_t row_blk = (int64_t)blk * ROWS_B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = row_blk + wave * R16;
    const int c = lane & 15, q = lane >> 4;
    const int64_t arow_raw = row0 + c;
    const bool arow_ok = arow_raw < a.E;
    const bool odd = (c & 1) != 0, hi = (c & 2) != 0;
    {
        const float4* g = reinterpret_cast<const float4*>(a.img + (int64_t)n * a.img_sn);
        float4* d = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int i = 0; i < SEQ_IMG / 4 / 512; ++i) d[i * 512 + threadIdx.x] = g[i * 512 + threadIdx.x];
    }
    const int T = a.T;
    // addresses = buffer resource of (agent, step) (scalar registers) + one 32-bit byte offset of the lane per row pitch
    const float* gA = a.gates + (int64_t)n * a.gates_sn;
    const float* cA = a.c_all + (int64_t)n * a.c_sn;
    const float* eA = a.dh_ext + (int64_t)n * a.dh_sn;
    float* zA = a.dz + (int64_t)n * a.dz_sn;
    const uint32_t lo4 = (uint32_t)(arow_raw * G4 + 4 * q) * 4u, lo1 = (uint32_t)(arow_raw * H + 4 * q) * 4u;
    const uint32_t nb4 = (uint32_t)(a.E * G4) * 4u, nb1 = (uint32_t)(a.E * H) * 4u;
    const uint32_t lor = (uint32_t)(arow_ok ? arow_raw : a.E - 1);

    // groups j, j + 1 of step t_ together: the two 64-byte halves of every 128-byte line are requested back to back
#define NMARL_SEQ_LOAD2(UA, UB, t_, j)                                                     \
    {                                                                                      \
        const int64_t ts_ = __builtin_amdgcn_readfirstlane(t_);   /* keeps the resources in scalar registers */ \
        const __amdgpu_buffer_rsrc_t rg_ = make_rsrc(gA + ts_ * a.gates_st, nb4);          \
        const __amdgpu_buffer_rsrc_t rc_ = make_rsrc(cA + ts_ * a.c_st, nb1);              \
        const __amdgpu_buffer_rsrc_t re_ = make_rsrc(eA + ts_ * a.dh_st, nb1);             \
        UA.gi = bload4(rg_, lo4 + 64 * (j));                                               \
        UB.gi = bload4(rg_, lo4 + 64 * (j) + 64);                                          \
        UA.gf = bload4(rg_, lo4 + 64 * (j) + 4 * H);                                       \
        UB.gf = bload4(rg_, lo4 + 64 * (j) + 4 * H + 64);                                  \
        UA.go = bload4(rg_, lo4 + 64 * (j) + 8 * H);                                       \
        UB.go = bload4(rg_, lo4 + 64 * (j) + 8 * H + 64);                                  \
        UA.gu = bload4(rg_, lo4 + 64 * (j) + 12 * H);                                      \
        UB.gu = bload4(rg_, lo4 + 64 * (j) + 12 * H + 64);                                 \
        UA.cp = bload4(rc_, lo1 + 64 * (j));                                               \
        UB.cp = bload4(rc_, lo1 + 64 * (j) + 64);                                          \
        UA.gh = bload4(re_, lo1 + 64 * (j));                                               \
        UB.gh = bload4(re_, lo1 + 64 * (j) + 64);                                          \
    }
    SeqGroup u0, u1, u2, u3;
    NMARL_SEQ_LOAD2(u0, u1, T - 1, 0)
    NMARL_SEQ_LOAD2(u2, u3, T - 1, 2)
    float4 cn[4], dc[4];
    f32x4 dhr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cn[j] = bload4(make_rsrc(cA + (int64_t)T * a.c_st, nb1), lo1 + 64 * j);
        dc[j] = float4{0.f, 0.f, 0.f, 0.f};
        dhr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float dbacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dbacc[i] = 0.0f;
    float keepA = 1.0f - (a.done + (int64_t)(T - 1) * a.E)[lor];
    __syncthreads();                                 // image visible

#define NMARL_SEQ_KSTEP(bv, s)                                                             \
    {                                                                                      \
        const float4 w_ = *reinterpret_cast<const float4*>(abase + (s) * 64 * 4);          \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.x, bv, acc[0], 0, 0, 0);          \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.y, bv, acc[1], 0, 0, 0);          \
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.z, bv, acc[2], 0, 0, 0);          \
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w_.w, bv, acc[3], 0, 0, 0);          \
    }
#define NMARL_SEQ_CELLB(U, j, k, i_)                                                       \
    {                                                                                      \
        const float tc = tanh_fast_(cn[j].k);                                              \
        const float gh_ = U.gh.k + dhr[j][i_];                                     \
        const float g_c = dc[j].k + gh_ * U.go.k * (1.0f - tc * tc);                       \
        di.k = g_c * U.gu.k * U.gi.k * (1.0f - U.gi.k);                                    \
        df.k = g_c * (U.cp.k * keepA) * U.gf.k * (1.0f - U.gf.k);                          \
        dO.k = gh_ * tc * U.go.k * (1.0f - U.go.k);                                        \
        du.k = g_c * U.gi.k * (1.0f - U.gu.k * U.gu.k);                                    \
        dc[j].k = g_c * U.gf.k * keepA;                                                    \
        cn[j].k = U.cp.k;                                                                  \
    }
    // bias sums of one group: 16 values -> (quad reduce-scatter) -> 4 per lane: gate 2 (c & 1) + ((c >> 1) & 1)
#define NMARL_SEQ_DB1(j, k, i_)                                                            \
    {                                                                                      \
        const float ka = odd ? dO.k : di.k, ga = odd ? di.k : dO.k;                        \
        const float kb = odd ? du.k : df.k, gb = odd ? df.k : du.k;                        \
        const float ra = ka + dpp_xor1(ga), rb = kb + dpp_xor1(gb);                        \
        const float k2 = hi ? rb : ra, g2 = hi ? ra : rb;                                  \
        dbacc[(j) * 4 + i_] += k2 + dpp_xor2(g2);                                          \
    }
#define NMARL_SEQ_CELL(U, j)                                                               \
    NMARL_SEQ_CELLB(U, j, x, 0) NMARL_SEQ_CELLB(U, j, y, 1) NMARL_SEQ_CELLB(U, j, z, 2) NMARL_SEQ_CELLB(U, j, w, 3)         \
    NMARL_SEQ_DB1(j, x, 0) NMARL_SEQ_DB1(j, y, 1) NMARL_SEQ_DB1(j, z, 2) NMARL_SEQ_DB1(j, w, 3)
#define NMARL_SEQ_PROD(j)                                                                  \
    {                                                                                      \
        bstore4(rz, lo4 + 64 * (j), di);                                                   \
        bstore4(rz, lo4 + 64 * (j) + 4 * H, df);                                           \
        bstore4(rz, lo4 + 64 * (j) + 8 * H, dO);                                           \
        bstore4(rz, lo4 + 64 * (j) + 12 * H, du);                                          \
        NMARL_SEQ_KSTEP(di.x, (j) * 16 + 0) NMARL_SEQ_KSTEP(di.y, (j) * 16 + 1)            \
        NMARL_SEQ_KSTEP(di.z, (j) * 16 + 2) NMARL_SEQ_KSTEP(di.w, (j) * 16 + 3)            \
        NMARL_SEQ_KSTEP(df.x, (j) * 16 + 4) NMARL_SEQ_KSTEP(df.y, (j) * 16 + 5)            \
        NMARL_SEQ_KSTEP(df.z, (j) * 16 + 6) NMARL_SEQ_KSTEP(df.w, (j) * 16 + 7)            \
        NMARL_SEQ_KSTEP(dO.x, (j) * 16 + 8) NMARL_SEQ_KSTEP(dO.y, (j) * 16 + 9)            \
        NMARL_SEQ_KSTEP(dO.z, (j) * 16 + 10) NMARL_SEQ_KSTEP(dO.w, (j) * 16 + 11)          \
        NMARL_SEQ_KSTEP(du.x, (j) * 16 + 12) NMARL_SEQ_KSTEP(du.y, (j) * 16 + 13)          \
        NMARL_SEQ_KSTEP(du.z, (j) * 16 + 14) NMARL_SEQ_KSTEP(du.w, (j) * 16 + 15)          \
    }
    for (int t = T - 1; t >= 0; --t) {
        const int tp = t > 0 ? t - 1 : 0;            // clamped: the last prefetch re-reads step 0 (unconditional loads)
        const float keep_next = 1.0f - (a.done + (int64_t)__builtin_amdgcn_readfirstlane(tp) * a.E)[lor];
        const __amdgpu_buffer_rsrc_t rz = make_rsrc(zA + (int64_t)__builtin_amdgcn_readfirstlane(t) * a.dz_st, nb4);
        // the image reads are the same every step: left visible as loop invariants the compiler hoists all 64 of them
        // out of the loop and spills them (1 KB of scratch per lane); an opaque copy of the address keeps them here
        int aoff = (q * 16 + c) * 4;                 // (an opaque POINTER would turn the LDS reads into flat loads)
        asm volatile("" : "+v"(aoff));
        const float* abase = lds + aoff;             // A operand: image row (s, q), out units 16 t + c of the 4 tiles
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 di, df, dO, du;
        // the scheduling fences keep the phases where they are written: left alone the scheduler bunches all 24 loads at
        // the end of the body (a quarter step before their use) and orders them against the waits' in-order counter
        NMARL_SEQ_CELL(u0, 0)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_SEQ_PROD(0)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_SEQ_CELL(u1, 1)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_SEQ_LOAD2(u0, u1, tp, 0)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_SEQ_PROD(1)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_SEQ_CELL(u2, 2)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_SEQ_PROD(2)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_SEQ_CELL(u3, 3)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_SEQ_LOAD2(u2, u3, tp, 2)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_SEQ_PROD(3)
        __builtin_amdgcn_sched_barrier(0);
        // dh_{t-1, rec} = (dz @ wh^T) keep_t: acc[j][i] = dh[row c][unit 16 j + 4 q + i], this lane's own units
#pragma unroll
        for (int j = 0; j < 4; ++j) dhr[j] = acc[j] * keepA;
        keepA = keep_next;
    }
#undef NMARL_SEQ_LOAD2
#undef NMARL_SEQ_KSTEP
#undef NMARL_SEQ_CELLB
#undef NMARL_SEQ_DB1
#undef NMARL_SEQ_CELL
#undef NMARL_SEQ_PROD

    if (arow_ok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (a.dh0) *reinterpret_cast<float4*>(a.dh0 + (int64_t)n * a.dh0_sn + arow_raw * H + 4 * q + 16 * j) =
                           float4{dhr[j][0], dhr[j][1], dhr[j][2], dhr[j][3]};
            if (a.dc0) *reinterpret_cast<float4*>(a.dc0 + (int64_t)n * a.dc0_sn + arow_raw * H + 4 * q + 16 * j) = dc[j];
        }
    }
    if (a.db_part) {                                 // the quads' sums: over the wave's 4 quads, then over the 8 waves
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = dbacc[i];
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            dbacc[i] = v;
        }
        __syncthreads();                             // every wave is done with the image: reuse its LDS
        if (c < 4) {
            const int g = 2 * (c & 1) + ((c >> 1) & 1);
#pragma unroll
            for (int i = 0; i < 16; ++i)             // i = 4 j + unit  ->  column 64 g + 16 j + 4 q + unit
                lds[wave * G4 + 64 * g + 16 * (i >> 2) + 4 * q + (i & 3)] = dbacc[i];
        }
        __syncthreads();
        if (threadIdx.x < G4) {
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) v += lds[w * G4 + threadIdx.x];
            a.db_part[(int64_t)n * a.db_sn + (int64_t)blk * G4 + threadIdx.x] = v;
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

extern "C" int nmarl_lstm_bptt_seq_blocks(int64_t E) { return (int)((E + ROWS_B - 1) / ROWS_B); }

extern "C" int nmarl_lstm_bptt_seq(int32_t T, int64_t E, int32_t N, int32_t Hh, const float* gates, int64_t gates_sn,
                                   int64_t gates_st, const float* c_all, int64_t c_sn, int64_t c_st, const float* done,
                                   const float* dh_ext, int64_t dh_sn, int64_t dh_st, const float* img, int64_t img_sn,
                                   float* dz, int64_t dz_sn, int64_t dz_st, float* db_part, int64_t db_sn, float* dh0,
                                   int64_t dh0_sn, float* dc0, int64_t dc0_sn, void* stream) {
    if (Hh != H || E < 0 || N <= 0 || T <= 0 || (E > 0 && (!gates || !c_all || !done || !dh_ext || !img || !dz)))
        return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    if (E > (1 << 21)) return NMARL_EINVAL;             // 32-bit byte offsets inside one (agent, step) panel
    const int64_t nblk = (E + ROWS_B - 1) / ROWS_B;
    if (gates_st < E * G4 || (gates_st % 4) || !sn_ok(gates_sn, (T - 1) * gates_st + E * G4) || dz_st < E * G4 || (dz_st % 4) ||
        !sn_ok(dz_sn, (T - 1) * dz_st + E * G4) || c_st < E * H || (c_st % 4) || !sn_ok(c_sn, T * c_st + E * H) ||
        dh_st < E * H || (dh_st % 4) || !sn_ok(dh_sn, (T - 1) * dh_st + E * H) || img_sn < (int64_t)G4 * H || (img_sn % 4) ||
        (db_part && db_sn < nblk * G4) || (dh0 && !sn_ok(dh0_sn, E * H)) || (dc0 && !sn_ok(dc0_sn, E * H)) ||
        ((uintptr_t)img % 16) || ((uintptr_t)gates % 16) || ((uintptr_t)dz % 16) || ((uintptr_t)c_all % 16) ||
        ((uintptr_t)dh_ext % 16) || (dh0 && ((uintptr_t)dh0 % 16)) || (dc0 && ((uintptr_t)dc0 % 16)))
        return NMARL_EINVAL;
    BpttSeqArgs a{};
    a.gates = gates; a.c_all = c_all; a.done = done; a.dh_ext = dh_ext; a.img = img; a.dz = dz; a.db_part = db_part;
    a.dh0 = dh0; a.dc0 = dc0; a.gates_sn = gates_sn; a.gates_st = gates_st; a.c_sn = c_sn; a.c_st = c_st; a.dh_sn = dh_sn;
    a.dh_st = dh_st; a.img_sn = img_sn; a.dz_sn = dz_sn; a.dz_st = dz_st; a.db_sn = db_sn; a.dh0_sn = dh0_sn; a.dc0_sn = dc0_sn;
    a.E = E; a.N = N; a.T = T;
    static bool lds_set = false;
    if (!lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_seq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SEQ_IMG * 4) != hipSuccess)
            return NMARL_EHIP;
        lds_set = true;
    }
    hipLaunchKernelGGL(lstm_bptt_seq_kernel, dim3((unsigned)(nblk * N)), dim3(512), (size_t)SEQ_IMG * 4,
                       static_cast<hipStream_t>(stream), a);
    return nmarl_check_launch();
}
