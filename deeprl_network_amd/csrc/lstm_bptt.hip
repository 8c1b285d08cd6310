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
#include <stdlib.h>

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
    float* db_part; int64_t db_sn;     // [N][blocks * 8][256] running column sums of dz (the bias gradient), or NULL
};

__device__ __forceinline__ float dpp_xor1(float v) {  // value of lane ^ 1 (quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {  // value of lane ^ 2 (quad_perm [2,3,0,1])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor8(float v) {  // value of lane ^ 8 (row_ror:8 -- a rotation by half a 16-lane row)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));
}

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
        if (a.db_part) {   /* uniform.  Column sums of this group's 16 dz values over the wave's 16 rows: a reduce-scatter over */ \
            /* the lanes of a row (c ^ 8, ^ 4, ^ 2, ^ 1) leaves lane c the sum of value c = 4 gate + unit */ \
            const float v_[16] = {di.x, di.y, di.z, di.w, df.x, df.y, df.z, df.w, dO.x, dO.y, dO.z, dO.w, du.x, du.y, du.z, du.w}; \
            float s8[8], s4[4], s2[2];                                                     \
            _Pragma("unroll") for (int k = 0; k < 8; ++k)                                  \
                s8[k] = (b3 ? v_[k + 8] : v_[k]) * okf + dpp_xor8((b3 ? v_[k] : v_[k + 8]) * okf); \
            _Pragma("unroll") for (int k = 0; k < 4; ++k)                                  \
                s4[k] = (b2 ? s8[k + 4] : s8[k]) + __shfl_xor(b2 ? s8[k] : s8[k + 4], 4);  \
            _Pragma("unroll") for (int k = 0; k < 2; ++k)                                  \
                s2[k] = (b1 ? s4[k + 2] : s4[k]) + dpp_xor2(b1 ? s4[k] : s4[k + 2]);       \
            dbs[j] = (b0 ? s2[1] : s2[0]) + dpp_xor1(b0 ? s2[0] : s2[1]);                  \
        }                                                                                  \
    }
    const bool b3 = (c & 8) != 0, b2 = (c & 4) != 0, b1 = (c & 2) != 0, b0 = (c & 1) != 0;
    const float okf = arow_ok ? 1.0f : 0.0f;         // rows past E repeat row E - 1: not in the sums
    float dbs[4] = {0.f, 0.f, 0.f, 0.f};
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

    if (a.db_part) {
        // this wave's slot of the running partial sums (one launch per reverse step on one stream: no atomics; fixed order)
        float* p_ = a.db_part + (int64_t)n * a.db_sn + ((int64_t)(blockIdx.x / a.N) * WAVES + wave) * G4 + (c >> 2) * H + 4 * q + (c & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) p_[16 * j] += dbs[j];
    }
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
// Arithmetic per step is that of lstm_bptt_step_kernel<4> with apply_keep = 1, operation for operation -- except that
// c_t (for tanh(c_t)) is not read: it is RECOMPUTED from the step's own gates and c_{t-1} exactly as every forward kernel
// forms it (lstm_mfma.hip / a2c.hip: cv = gf * (cp * keep) + gi * gu, four separately rounded operations under
// -ffp-contract=off), i.e. bit-identical to the stored c_all[t + 1] whenever the saved sequences come from a forward pass.
// That removes 16 persistent registers (the carried c_t) -- the kernel no longer spills -- and the read of c_all[T].
#ifdef NMARL_STEP_TIMELINE      // instrumentation build (tools/bptt_timeline.py): shader-clock stamps of block 0's waves, one mid step
__device__ unsigned long long* g_tl_b = nullptr;
__global__ void tl_b_set_kernel(unsigned long long* p) { g_tl_b = p; }
#define NMARL_BSTAMP(i) if (g_tl_b && blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_tl_b[(threadIdx.x >> 6) * 32 + (i)] = __builtin_amdgcn_s_memtime();
#define NMARL_BSTAMP_T(i) if (t == T / 2) { NMARL_BSTAMP(i) }
#else
#define NMARL_BSTAMP(i)
#define NMARL_BSTAMP_T(i)
#endif

struct BpttSeqArgs {
    const float *gates, *c_all, *done, *dh_ext, *img;
    float *dz, *db_part, *dh0, *dc0;
    int64_t gates_sn, gates_st, c_sn, c_st, dh_sn, dh_st, img_sn, dz_sn, dz_st, db_sn, dh0_sn, dc0_sn;
    int64_t E;
    int N, T;
    // DY (round 6): the heads' dL/dh is NOT read as a tensor: dy8 [N][T][E][8] = [d logits | d v | 0] (nmarl_heads_loss) and the
    // heads' weights hw [N][64][O] -- dL/dh_t(heads) = dy_t hw^T is two more k-steps of the step's transposed product
    const float *dy8, *hw;
    int64_t dy_sn, dy_st, hw_sn;
    int O;
};

struct SeqGroup {           // per-step inputs of 4 consecutive units of one row
    float4 gi, gf, go, gu, cp, gh;
};

struct GateGroup {          // gates and c_{t-1} of 4 consecutive units of one row (lstm_bptt_coupled_kernel's slots)
    float4 gi, gf, go, gu, cp;
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

template <bool DY>
__global__ __launch_bounds__(512, 1) void lstm_bptt_seq_kernel(const BpttSeqArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (plain round-robin over the agents: the XCD-aware grouping of common.h, which pays in the lock-step kernel, made this one
    // 14 % slower on the grid -- an XCD then hosts only interior or only border agents and the ring traffic is uneven)
    const int n = blockIdx.x % a.N;
    const int blk = blockIdx.x / a.N;
    const int64_t row_blk = (int64_t)blk * ROWS_B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = row_blk + wave * R16;
    const int c = lane & 15, q = lane >> 4;
    const int64_t arow_raw = row0 + c;
    const bool arow_ok = arow_raw < a.E;
    const bool odd = (c & 1) != 0, hi = (c & 2) != 0;
    NMARL_BSTAMP(0)
    {
        const float4* g = reinterpret_cast<const float4*>(a.img + (int64_t)n * a.img_sn);
        float4* d = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int i = 0; i < SEQ_IMG / 4 / 512; ++i) d[i * 512 + threadIdx.x] = g[i * 512 + threadIdx.x];
        if (DY) {
            // the heads' image behind it, same layout: [k-step s][q][c][tile t] = hw[unit 16 t + c][output 4 s + q] (0 past O): one float per thread
            const int s_ = threadIdx.x >> 8, q_ = (threadIdx.x >> 6) & 3, c_ = (threadIdx.x >> 2) & 15, t_ = threadIdx.x & 3;
            const int o_ = 4 * s_ + q_;
            lds[SEQ_IMG + threadIdx.x] = o_ < a.O ? a.hw[(int64_t)n * a.hw_sn + (16 * t_ + c_) * a.O + o_] : 0.0f;
        }
    }
    const int T = a.T;
    // addresses = buffer resource of (agent, step) (scalar registers) + one 32-bit byte offset of the lane per row pitch
    const float* gA = a.gates + (int64_t)n * a.gates_sn;
    const float* cA = a.c_all + (int64_t)n * a.c_sn;
    const float* eA = DY ? nullptr : a.dh_ext + (int64_t)n * a.dh_sn;
    const float* yA = DY ? a.dy8 + (int64_t)n * a.dy_sn : nullptr;
    float* zA = a.dz + (int64_t)n * a.dz_sn;
    const uint32_t lo4 = (uint32_t)(arow_raw * G4 + 4 * q) * 4u, lo1 = (uint32_t)(arow_raw * H + 4 * q) * 4u;
    const uint32_t nb4 = (uint32_t)(a.E * G4) * 4u, nb1 = (uint32_t)(a.E * H) * 4u;
    const uint32_t lo8 = (uint32_t)(arow_raw * 8 + q) * 4u, nb8 = (uint32_t)(a.E * 8) * 4u;
    const uint32_t lor = (uint32_t)(arow_ok ? arow_raw : a.E - 1);

    // groups j, j + 1 of step t_ together: the two 64-byte halves of every 128-byte line are requested back to back
#define NMARL_SEQ_LOAD2(UA, UB, t_, j)                                                     \
    {                                                                                      \
        const int64_t ts_ = __builtin_amdgcn_readfirstlane(t_);   /* keeps the resources in scalar registers */ \
        const __amdgpu_buffer_rsrc_t rg_ = make_rsrc(gA + ts_ * a.gates_st, nb4);          \
        const __amdgpu_buffer_rsrc_t rc_ = make_rsrc(cA + ts_ * a.c_st, nb1);              \
        const __amdgpu_buffer_rsrc_t re_ = make_rsrc(DY ? cA : eA + ts_ * a.dh_st, nb1);   \
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
        if (!DY) {                                                                         \
            UA.gh = bload4(re_, lo1 + 64 * (j));                                           \
            UB.gh = bload4(re_, lo1 + 64 * (j) + 64);                                      \
        }                                                                                  \
    }
    // DY: the lane's two values of dy8[t_][row c] (outputs q and 4 + q): the B operand of the heads' two k-steps
#define NMARL_SEQ_LOADY(d0_, d1_, t_)                                                      \
    {                                                                                      \
        const int64_t ts_ = __builtin_amdgcn_readfirstlane(t_);                            \
        const __amdgpu_buffer_rsrc_t ry_ = make_rsrc(yA + ts_ * a.dy_st, nb8);             \
        d0_ = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, lo8, 0, 0));       \
        d1_ = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, lo8 + 16, 0, 0));  \
    }
    // dhr += dy hw^T: two k-steps x four unit tiles (A = the heads' image rows, B = the lane's dy values)
#define NMARL_SEQ_HEADS(d0_, d1_)                                                          \
    {                                                                                      \
        const float* hb_ = lds + SEQ_IMG + (q * 16 + c) * 4;                               \
        const float4 p0_ = *reinterpret_cast<const float4*>(hb_), p1_ = *reinterpret_cast<const float4*>(hb_ + 256); \
        dhr[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p0_.x, d0_, dhr[0], 0, 0, 0);        \
        dhr[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p0_.y, d0_, dhr[1], 0, 0, 0);        \
        dhr[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p0_.z, d0_, dhr[2], 0, 0, 0);        \
        dhr[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p0_.w, d0_, dhr[3], 0, 0, 0);        \
        dhr[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1_.x, d1_, dhr[0], 0, 0, 0);        \
        dhr[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1_.y, d1_, dhr[1], 0, 0, 0);        \
        dhr[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1_.z, d1_, dhr[2], 0, 0, 0);        \
        dhr[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(p1_.w, d1_, dhr[3], 0, 0, 0);        \
    }
    SeqGroup u0, u1, u2, u3;
    float dya = 0.0f, dyb = 0.0f;                     // DY: dy8 of the step whose heads' dL/dh is added next
    if (DY) { NMARL_SEQ_LOADY(dya, dyb, T - 1) }
    NMARL_SEQ_LOAD2(u0, u1, T - 1, 0)
    NMARL_SEQ_LOAD2(u2, u3, T - 1, 2)
    float4 dc[4];
    f32x4 dhr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dc[j] = float4{0.f, 0.f, 0.f, 0.f};
        dhr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float dbacc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) dbacc[i] = 0.0f;
    float keepA = 1.0f - (a.done + (int64_t)(T - 1) * a.E)[lor];
    __syncthreads();                                 // image visible
    if (DY) {                                        // dL/dh_{T-1} = the heads' part alone
        NMARL_SEQ_HEADS(dya, dyb)
    }

    // one k-step = image row (s, q) of the 4 output tiles (one ds_read_b128) x the lane's dz value: 4 MFMAs.  Reads run two k-steps
    // ahead of their MFMAs through two register sets (left alone the compiler reads each row right before its use and every
    // k-step waits out the LDS latency; same scheme as lstm_bptt_coupled_kernel)
#define NMARL_SEQ_BL(P, s) P = *reinterpret_cast<const float4*>(abase + (s) * 64 * 4);
#define NMARL_SEQ_MF(bv, P)                                                                \
    {                                                                                      \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(P.x, bv, acc[0], 0, 0, 0);           \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(P.y, bv, acc[1], 0, 0, 0);           \
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(P.z, bv, acc[2], 0, 0, 0);           \
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(P.w, bv, acc[3], 0, 0, 0);           \
    }
#define NMARL_SEQ_KS(bv, P, s2)                                                            \
    NMARL_SEQ_MF(bv, P) __builtin_amdgcn_sched_barrier(0);                                 \
    NMARL_SEQ_BL(P, s2) __builtin_amdgcn_sched_barrier(0);
#define NMARL_SEQ_CELLB(U, j, k, i_)                                                       \
    {                                                                                      \
        const float cpk = U.cp.k * keepA;                                                  \
        const float tc = tanh_fast_(U.gf.k * cpk + U.gi.k * U.gu.k);   /* c_t, op for op the forward's (see above) */ \
        const float gh_ = DY ? dhr[j][i_] : U.gh.k + dhr[j][i_];                           \
        const float g_c = dc[j].k + gh_ * U.go.k * (1.0f - tc * tc);                       \
        di.k = g_c * U.gu.k * U.gi.k * (1.0f - U.gi.k);                                    \
        df.k = g_c * cpk * U.gf.k * (1.0f - U.gf.k);                                       \
        dO.k = gh_ * tc * U.go.k * (1.0f - U.go.k);                                        \
        du.k = g_c * U.gi.k * (1.0f - U.gu.k * U.gu.k);                                    \
        dc[j].k = g_c * U.gf.k * keepA;                                                    \
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
        float4 pa, pb;                                                                     \
        NMARL_SEQ_BL(pa, (j) * 16 + 0) NMARL_SEQ_BL(pb, (j) * 16 + 1) __builtin_amdgcn_sched_barrier(0); \
        NMARL_SEQ_KS(di.x, pa, (j) * 16 + 2) NMARL_SEQ_KS(di.y, pb, (j) * 16 + 3)          \
        NMARL_SEQ_KS(di.z, pa, (j) * 16 + 4) NMARL_SEQ_KS(di.w, pb, (j) * 16 + 5)          \
        NMARL_SEQ_KS(df.x, pa, (j) * 16 + 6) NMARL_SEQ_KS(df.y, pb, (j) * 16 + 7)          \
        NMARL_SEQ_KS(df.z, pa, (j) * 16 + 8) NMARL_SEQ_KS(df.w, pb, (j) * 16 + 9)          \
        NMARL_SEQ_KS(dO.x, pa, (j) * 16 + 10) NMARL_SEQ_KS(dO.y, pb, (j) * 16 + 11)        \
        NMARL_SEQ_KS(dO.z, pa, (j) * 16 + 12) NMARL_SEQ_KS(dO.w, pb, (j) * 16 + 13)        \
        NMARL_SEQ_KS(du.x, pa, (j) * 16 + 14) NMARL_SEQ_KS(du.y, pb, (j) * 16 + 15)        \
        NMARL_SEQ_MF(du.z, pa) __builtin_amdgcn_sched_barrier(0);                          \
        NMARL_SEQ_MF(du.w, pb)                                                             \
    }
    NMARL_BSTAMP(1)
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
        if (DY) { NMARL_SEQ_LOADY(dya, dyb, tp) }     // (step t's heads' part is inside dhr already)
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
        if (DY && t > 0) {                           // + the heads' dL/dh_{t-1} (not masked: the heads read h_{t-1} itself)
            NMARL_SEQ_HEADS(dya, dyb)
        }
        keepA = keep_next;
    }
    NMARL_BSTAMP(17)
#undef NMARL_SEQ_LOAD2
#undef NMARL_SEQ_LOADY
#undef NMARL_SEQ_HEADS
#undef NMARL_SEQ_BL
#undef NMARL_SEQ_MF
#undef NMARL_SEQ_KS
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


// ---------------------------------------------------------------------------------------------------------------------
// The reverse recurrence of a COUPLED net -- NeurComm (lstm_comm, agents/utils.py:182-208) and CommNet (lstm_ic3,
// agents/utils.py:395-408) -- in ONE launch.  Per reverse step t an agent's block does what lstm_bptt_seq_kernel does for
// its 128 rows (dL/dc, the recurrent dL/dh, c_t stay in registers; loads of step t-1 in flight while step t computes) and
// in addition the adjoint of the message term:
//
//     [dx | dh]^T = [Wxm; Wh] . dz_t^T                  (one transposed product, 8 output tiles)
//     D1_t        = dx (* (hm_t > 0): relu of lstm_comm's message layer)          -> d1 (weight / bias gradients of w_msg)
//     M_t^T       = W_msg . D1_t^T                      (second transposed product: D1 in registers IS its B operand)
//     dL/dh_{t-1}[i] = dh[i] keep_t  +  sum over {(a, k): agent i is neighbour k of a} w_a M_t[a][:, k-th slot]
//
// The last sum crosses agents, i.e. blocks: M_t of (agent a, rows r) is handed to the blocks of a's neighbours for the
// SAME rows through a message buffer in global memory ([slots][N][E][K]), wave to wave (a wave owns 16 rows for the whole launch, so the
// hand-off needs no block barrier): payload with write-through (sc1) 16-byte stores, every storing wave drains, one
// relaxed agent-scope flag per (agent, tile, wave) carrying the number of steps done; the consumer polls that one word,
// then reads the payload with sc1 loads (MI355X_MICROARCH.md "inter-workgroup visibility", recipe R1).  Results do not
// depend on dispatch order or placement; PROGRESS needs the blocks of a row tile co-resident, which the launcher
// guarantees (grid <= CUs, one 160-KB-LDS block per CU) -- otherwise it launches step by step (t_hi == t_lo: every
// flag a step polls was published by an earlier launch, state through dhr_io / dc_io), same kernel, same bits.  Spins are
// bounded: a wave that waits too long raises *err and stops waiting (the host checks it).
// The one-launch form writes every step's messages to its OWN slot (slots = T): a consumer must never re-read an address it
// read earlier in the launch -- the per-XCD L2s are not coherent, an sc1 load bypasses the reader's L1 only, and a line the
// reader's L2 still holds from two steps ago would be served stale (seen: rare wrong rows with a two-slot ring).  The
// step-wise form alternates two slots (kernel boundaries make them coherent); with a symmetric neighbour relation a
// producer's consumers are exactly the agents it waits for, so a slot is rewritten only after its readers are done.
struct CoupledArgs {
    const float *gates, *c_all, *done, *dh_ext, *img, *img_m, *mask;
    float *dz, *d1, *ring, *db_part, *dbm_part, *dhr_io, *dc_io;
    unsigned *flags, *err;
    const int32_t *rev_agent, *rev_col;     // [N][RMAX]: source agent (own index where absent), first column of my slot
    const float* rev_w;                     // [N][RMAX]: weight of that source (0 where absent)
    int64_t gates_sn, gates_st, c_sn, c_st, dh_sn, dh_st, img_sn, imgm_sn, mask_sn, mask_st, dz_sn, dz_st, d1_sn, d1_st,
        ring_sn, ring_slot, db_sn, dbm_sn, io_sn;
    int64_t E;
    int32_t N, T, t_hi, t_lo, mask_row, tiles, slots;
    int32_t* status;                        // may be NULL: hand-off status words ([0] <- 1 when a wave gives up, sticky)
    unsigned max_spins;
    int32_t fault;                          // test hook: block 0 never publishes (its neighbours time out)
    // DY (round 6): the heads' dL/dh arrives as dy8 [N][T][E][8] + the heads' weights hw [N][64][O] (see lstm_bptt_seq_kernel)
    const float *dy8, *hw;
    int64_t dy_sn, dy_st, hw_sn;
    int32_t O;
};

typedef __attribute__((address_space(1))) unsigned gu32;
// LOADS take (lane offset, compile-time byte offset): the constant travels in the instruction's scalar-offset field instead
// of one loop-invariant VGPR per distinct sum -- with ~40 distinct sums per step those VGPRs were spilled and every reload
// drained the loads in flight.  The range check is on the lane offset: rows past E stay out of range.  (Loads only: see the
// note on stores in lstm_bptt_coupled_kernel.)
template <int AUX = 0>
__device__ __forceinline__ float4 bload4i(const __amdgpu_buffer_rsrc_t r, const uint32_t off, const int imm) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, off, imm, AUX));
}
constexpr int SC1 = 16;                              // aux bit: write-through store / L1-bypassing load (agent scope)
__device__ __forceinline__ void bstore4_wt(const __amdgpu_buffer_rsrc_t r, const uint32_t off, const float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, off, 0, SC1);      // write-through (sc1)
}

template <int NTM, int RMAX, bool MASK, bool DY = false>     // NTM: 16-column tiles of a message row (64 m_max / 16 or 4); RMAX: max sources; DY: see CoupledArgs
__global__ __launch_bounds__(512, 1) void lstm_bptt_coupled_kernel(const CoupledArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int IMG8 = G4 * 16 * 8;                // [Wxm; Wh]^T image (NT = 8)
    constexpr int KMO = NTM * 16;                    // floats per message row
    // (plain round-robin over the agents: the XCD-aware grouping of common.h, which pays in the lock-step kernel, made this one
    // 14 % slower on the grid -- an XCD then hosts only interior or only border agents and the ring traffic is uneven)
    const int n = blockIdx.x % a.N;
    const int blk = blockIdx.x / a.N;
    const int64_t row_blk = (int64_t)blk * ROWS_B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = row_blk + wave * R16;
    const int c = lane & 15, q = lane >> 4;
    const int64_t arow_raw = row0 + c;
    const bool arow_ok = arow_raw < a.E;
    const bool odd = (c & 1) != 0, hi = (c & 2) != 0;
    NMARL_BSTAMP(0)
    {
        const float4* g = reinterpret_cast<const float4*>(a.img + (int64_t)n * a.img_sn);
        float4* d = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int i = 0; i < IMG8 / 4 / 512; ++i) d[i * 512 + threadIdx.x] = g[i * 512 + threadIdx.x];
        const float4* gm = reinterpret_cast<const float4*>(a.img_m + (int64_t)n * a.imgm_sn);
        float4* dm = reinterpret_cast<float4*>(lds + IMG8);
#pragma unroll
        for (int i = 0; i < KMO * 64 / 4 / 512; ++i) dm[i * 512 + threadIdx.x] = gm[i * 512 + threadIdx.x];
    }
    const int T = a.T, t_hi = a.t_hi, t_lo = a.t_lo;
    const float* gA = a.gates + (int64_t)n * a.gates_sn;
    const float* cA = a.c_all + (int64_t)n * a.c_sn;
    const float* eA = DY ? nullptr : a.dh_ext + (int64_t)n * a.dh_sn;
    const float* yA = DY ? a.dy8 + (int64_t)n * a.dy_sn : nullptr;
    float* zA = a.dz + (int64_t)n * a.dz_sn;
    float* d1A = a.d1 + (int64_t)n * a.d1_sn;
    const float* mkA = MASK ? a.mask + (int64_t)n * a.mask_sn : nullptr;
    const uint32_t lo4 = (uint32_t)(arow_raw * G4 + 4 * q) * 4u, lo1 = (uint32_t)(arow_raw * H + 4 * q) * 4u;
    const uint32_t nb4 = (uint32_t)(a.E * G4) * 4u, nb1 = (uint32_t)(a.E * H) * 4u;
    const uint32_t lom = MASK ? (uint32_t)(arow_raw * a.mask_row + 4 * q) * 4u : 0u;
    const uint32_t nbm = MASK ? (uint32_t)((a.E - 1) * a.mask_row + H) * 4u : 0u;
    const uint32_t lor = (uint32_t)(arow_ok ? arow_raw : a.E - 1);
    const uint32_t lo8 = (uint32_t)(arow_raw * 8 + q) * 4u, nb8 = (uint32_t)(a.E * 8) * 4u;
    // DY: the lane's A operands of the heads' two k-steps, kept in registers (the images fill the LDS): hw[unit 16 t + c][output 4 s + q]
    float4 hp0 = float4{0.f, 0.f, 0.f, 0.f}, hp1 = hp0;
    if (DY) {
        const float* hwn = a.hw + (int64_t)n * a.hw_sn;
        const int O = a.O, o0 = q, o1 = 4 + q;
        const bool k0 = o0 < O, k1 = o1 < O;
        const int i0 = k0 ? o0 : 0, i1 = k1 ? o1 : 0;
        hp0 = float4{hwn[c * O + i0], hwn[(16 + c) * O + i0], hwn[(32 + c) * O + i0], hwn[(48 + c) * O + i0]};
        hp1 = float4{hwn[c * O + i1], hwn[(16 + c) * O + i1], hwn[(32 + c) * O + i1], hwn[(48 + c) * O + i1]};
        if (!k0) hp0 = float4{0.f, 0.f, 0.f, 0.f};
        if (!k1) hp1 = float4{0.f, 0.f, 0.f, 0.f};
    }
    const uint32_t loR = (uint32_t)(arow_raw * KMO + 4 * q) * 4u;      // my row of a message tensor, + 64 tau (+ 4 col)
    const uint32_t nbR = (uint32_t)(a.E * KMO) * 4u;

    // the agents whose message adjoint reaches this agent (uniform per block)
    int src_n[RMAX];
    uint32_t src_c4[RMAX];
    float src_w[RMAX];
#pragma unroll
    for (int s = 0; s < RMAX; ++s) {
        src_n[s] = a.rev_agent[n * RMAX + s];
        src_c4[s] = (uint32_t)a.rev_col[n * RMAX + s] * 4u;
        src_w[s] = a.rev_w[n * RMAX + s];
    }
    const int my_flag_idx = (n * a.tiles + blk) * WAVES + wave;       // (the pointer is formed at the store: one live register, not two)
    bool give_up = false;
    // ONE poll for all sources: lane s < RMAX watches source s's flag (the other lanes source 0's), a ballot decides -- a
    // sequence of scalar polls costs one memory round trip per source on the critical path of every step
    gu32* poll_flag = (gu32*)(a.flags + ((int64_t)a.rev_agent[n * RMAX + (lane < RMAX ? lane : 0)] * a.tiles + blk) * WAVES + wave);

    // Register diet (round 4): the step's inputs are NOT all prefetched a step ahead any more.  Gates + c_{t-1} of one unit
    // group (5 float4 = 20 registers) cycle through TWO slots -- group j + 2 is requested as soon as group j's cell backward
    // has consumed its slot, i.e. one product + one cell phase (>= 4 k cycles) before its use; the two 64-byte halves of a
    // 128-byte line (groups 0 / 1 and 2 / 3) are requested one product apart, while the line still sits in the XCD's L2 --,
    // the heads' dL/dh of the whole step (16 registers) is requested a step ahead as before, and c_t is recomputed (see
    // lstm_bptt_seq_kernel).  96 + 16 persistent registers became 40 + 16: no spilled VGPR, no scratch traffic (the reloads
    // were VMEM loads that queued behind -- and drained -- the prefetches).
#define NMARL_CP_LOADG(S, t_, j)                                                           \
    {                                                                                      \
        const int64_t ts_ = __builtin_amdgcn_readfirstlane(t_);                            \
        const __amdgpu_buffer_rsrc_t rg_ = make_rsrc(gA + ts_ * a.gates_st, nb4);          \
        const __amdgpu_buffer_rsrc_t rc_ = make_rsrc(cA + ts_ * a.c_st, nb1);              \
        S.gi = bload4i(rg_, lo4, (64 * (j)) * 1);                                          \
        S.gf = bload4i(rg_, lo4, (64 * (j) + 4 * H) * 1);                                  \
        S.go = bload4i(rg_, lo4, (64 * (j) + 8 * H) * 1);                                  \
        S.gu = bload4i(rg_, lo4, (64 * (j) + 12 * H) * 1);                                 \
        S.cp = bload4i(rc_, lo1, (64 * (j)) * 1);                                          \
    }
#define NMARL_CP_LOADH(t_)                                                                 \
    if (DY) {       /* the lane's two values of dy8[t_][row c] (outputs q, 4 + q); gh starts from zero: NMARL_CP_HEADS adds the heads' part */ \
        const int64_t ts_ = __builtin_amdgcn_readfirstlane(t_);                            \
        const __amdgpu_buffer_rsrc_t ry_ = make_rsrc(yA + ts_ * a.dy_st, nb8);             \
        dya = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, lo8, 0, 0));       \
        dyb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ry_, lo8, 16, 0));      \
        gh[0] = gh[1] = gh[2] = gh[3] = float4{0.f, 0.f, 0.f, 0.f};                        \
    } else {                                                                               \
        const int64_t ts_ = __builtin_amdgcn_readfirstlane(t_);                            \
        const __amdgpu_buffer_rsrc_t re_ = make_rsrc(eA + ts_ * a.dh_st, nb1);             \
        gh[0] = bload4i(re_, lo1, 0);                                                      \
        gh[1] = bload4i(re_, lo1, 64);                                                     \
        gh[2] = bload4i(re_, lo1, 128);                                                    \
        gh[3] = bload4i(re_, lo1, 192);                                                    \
    }
    // gh += dy hw^T (two k-steps x four unit tiles; the heads read h_t itself: no done mask)
#define NMARL_CP_HEADS()                                                                   \
    {                                                                                      \
        f32x4 g0_ = f32x4{gh[0].x, gh[0].y, gh[0].z, gh[0].w}, g1_ = f32x4{gh[1].x, gh[1].y, gh[1].z, gh[1].w};   \
        f32x4 g2_ = f32x4{gh[2].x, gh[2].y, gh[2].z, gh[2].w}, g3_ = f32x4{gh[3].x, gh[3].y, gh[3].z, gh[3].w};   \
        g0_ = __builtin_amdgcn_mfma_f32_16x16x4f32(hp0.x, dya, g0_, 0, 0, 0);              \
        g1_ = __builtin_amdgcn_mfma_f32_16x16x4f32(hp0.y, dya, g1_, 0, 0, 0);              \
        g2_ = __builtin_amdgcn_mfma_f32_16x16x4f32(hp0.z, dya, g2_, 0, 0, 0);              \
        g3_ = __builtin_amdgcn_mfma_f32_16x16x4f32(hp0.w, dya, g3_, 0, 0, 0);              \
        g0_ = __builtin_amdgcn_mfma_f32_16x16x4f32(hp1.x, dyb, g0_, 0, 0, 0);              \
        g1_ = __builtin_amdgcn_mfma_f32_16x16x4f32(hp1.y, dyb, g1_, 0, 0, 0);              \
        g2_ = __builtin_amdgcn_mfma_f32_16x16x4f32(hp1.z, dyb, g2_, 0, 0, 0);              \
        g3_ = __builtin_amdgcn_mfma_f32_16x16x4f32(hp1.w, dyb, g3_, 0, 0, 0);              \
        gh[0] = float4{g0_[0], g0_[1], g0_[2], g0_[3]}; gh[1] = float4{g1_[0], g1_[1], g1_[2], g1_[3]};             \
        gh[2] = float4{g2_[0], g2_[1], g2_[2], g2_[3]}; gh[3] = float4{g3_[0], g3_[1], g3_[2], g3_[3]};             \
    }
    GateGroup sA, sB;
    float4 gh[4], dc[4];
    float dya = 0.0f, dyb = 0.0f;
    NMARL_CP_LOADG(sA, t_hi, 0)
    NMARL_CP_LOADG(sB, t_hi, 1)
    NMARL_CP_LOADH(t_hi)
    {
        // state the range starts from: zero at the end of the sequence, else what the previous launch left.  The recurrent
        // dL/dh lives INSIDE the prefetched inputs: it is added to the heads' dL/dh (gh) of the step it belongs to
        const bool first = t_hi == T - 1;
        const __amdgpu_buffer_rsrc_t rh = make_rsrc(a.dhr_io + (int64_t)n * a.io_sn, first ? 0u : nb1);
        const __amdgpu_buffer_rsrc_t rd = make_rsrc(a.dc_io + (int64_t)n * a.io_sn, first ? 0u : nb1);
#pragma unroll
        for (int j = 0; j < 4; ++j) dc[j] = bload4i(rd, lo1, 64 * j);                   // num_records 0: reads 0
        const float4 h0 = bload4i(rh, lo1, 0), h1 = bload4i(rh, lo1, 64), h2 = bload4i(rh, lo1, 128), h3 = bload4i(rh, lo1, 192);
        gh[0].x += h0.x; gh[0].y += h0.y; gh[0].z += h0.z; gh[0].w += h0.w;
        gh[1].x += h1.x; gh[1].y += h1.y; gh[1].z += h1.z; gh[1].w += h1.w;
        gh[2].x += h2.x; gh[2].y += h2.y; gh[2].z += h2.z; gh[2].w += h2.w;
        gh[3].x += h3.x; gh[3].y += h3.y; gh[3].z += h3.z; gh[3].w += h3.w;
    }
    if (DY) {                                        // + the heads' dL/dh of step t_hi
        NMARL_CP_HEADS()
    }
    // bias-gradient partial sums, fully reduced over the wave's 16 rows every step: lane (c, q) keeps, per unit group j,
    // the column of gate 2 (c & 1) + ((c >> 1) & 1), unit 16 j + 4 q + 2 ((c >> 2) & 1) + ((c >> 3) & 1)  (4 registers), and of
    // the message layer's bias the unit 16 jm + 4 q + 2 ((c >> 1) & 1) + (c & 1), jm = 2 ((c >> 2) & 1) + ((c >> 3) & 1)  (1)
    float dbacc[4], dbm = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) dbacc[i] = 0.0f;
    const bool b2 = (c & 4) != 0, b3 = (c & 8) != 0;
    float keepA = 1.0f - (a.done + (int64_t)t_hi * a.E)[lor];
    __syncthreads();                                 // images visible

    // one k-step = the image row (s, q) of all 8 output tiles (two ds_read_b128) x the lane's dz value: 8 MFMAs.  The reads of
    // k-step s + 2 are issued right after the MFMAs of step s released their register set (two sets, pinned by scheduling
    // fences): left alone the compiler reads each row immediately before its use and every k-step waits out the LDS latency
#define NMARL_CP_BL(P, s)                                                                  \
    {                                                                                      \
        const float* p_ = abase + (s) * 64 * 8;                                            \
        P##0 = *reinterpret_cast<const float4*>(p_ + 4 * sw);                              \
        P##1 = *reinterpret_cast<const float4*>(p_ + 4 * (sw ^ 1));                        \
    }
#define NMARL_CP_MF(bv, P)                                                                 \
    {                                                                                      \
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##0 .x, bv, acc[0], 0, 0, 0);        \
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##0 .y, bv, acc[1], 0, 0, 0);        \
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##0 .z, bv, acc[2], 0, 0, 0);        \
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##0 .w, bv, acc[3], 0, 0, 0);        \
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##1 .x, bv, acc[4], 0, 0, 0);        \
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##1 .y, bv, acc[5], 0, 0, 0);        \
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##1 .z, bv, acc[6], 0, 0, 0);        \
        acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##1 .w, bv, acc[7], 0, 0, 0);        \
    }
    // MFMAs of k-step s from set P, then the reads of k-step s + 2 into the same set
#define NMARL_CP_KS(bv, P, s2)                                                             \
    NMARL_CP_MF(bv, P) __builtin_amdgcn_sched_barrier(0);                                  \
    NMARL_CP_BL(P, s2) __builtin_amdgcn_sched_barrier(0);
#define NMARL_CP_CELLB(U, j, k, i_)                                                        \
    {                                                                                      \
        const float cpk = U.cp.k * keepA;                                                  \
        const float tc = tanh_fast_(U.gf.k * cpk + U.gi.k * U.gu.k);   /* c_t as the forward formed it */ \
        const float gh_ = gh[j].k;                                                         \
        const float g_c = dc[j].k + gh_ * U.go.k * (1.0f - tc * tc);                       \
        di.k = g_c * U.gu.k * U.gi.k * (1.0f - U.gi.k);                                    \
        df.k = g_c * cpk * U.gf.k * (1.0f - U.gf.k);                                       \
        dO.k = gh_ * tc * U.go.k * (1.0f - U.go.k);                                        \
        du.k = g_c * U.gi.k * (1.0f - U.gu.k * U.gu.k);                                    \
        dc[j].k = g_c * U.gf.k * keepA;                                                    \
    }
#define NMARL_CP_DB1(j, k, i_)                                                             \
    {                                                                                      \
        const float ka = odd ? dO.k : di.k, ga = odd ? di.k : dO.k;                        \
        const float kb = odd ? du.k : df.k, gb = odd ? df.k : du.k;                        \
        const float ra = ka + dpp_xor1(ga), rb = kb + dpp_xor1(gb);                        \
        const float k2 = hi ? rb : ra, g2 = hi ? ra : rb;                                  \
        qs[i_] = k2 + dpp_xor2(g2);                                                        \
    }
#define NMARL_CP_CELL(U, j)                                                                \
    NMARL_CP_CELLB(U, j, x, 0) NMARL_CP_CELLB(U, j, y, 1) NMARL_CP_CELLB(U, j, z, 2) NMARL_CP_CELLB(U, j, w, 3)         \
    {                                                                                      \
        float qs[4];                                                                       \
        NMARL_CP_DB1(j, x, 0) NMARL_CP_DB1(j, y, 1) NMARL_CP_DB1(j, z, 2) NMARL_CP_DB1(j, w, 3)                          \
        const float k0_ = b2 ? qs[2] : qs[0], g0_ = b2 ? qs[0] : qs[2];                    \
        const float k1_ = b2 ? qs[3] : qs[1], g1_ = b2 ? qs[1] : qs[3];                    \
        const float r0_ = k0_ + __shfl_xor(g0_, 4, 64), r1_ = k1_ + __shfl_xor(g1_, 4, 64); \
        const float k3_ = b3 ? r1_ : r0_, g3_ = b3 ? r0_ : r1_;                            \
        dbacc[j] += k3_ + __shfl_xor(g3_, 8, 64);                                          \
    }
#define NMARL_CP_PROD(j)                                                                   \
    {                                                                                      \
        bstore4(rz, so4 + (64 * (j)), di);                                                   \
        bstore4(rz, so4 + (64 * (j) + 4 * H), df);                                           \
        bstore4(rz, so4 + (64 * (j) + 8 * H), dO);                                           \
        bstore4(rz, so4 + (64 * (j) + 12 * H), du);                                          \
        float4 pa0, pa1, pb0, pb1;                                                         \
        NMARL_CP_BL(pa, (j) * 16 + 0) NMARL_CP_BL(pb, (j) * 16 + 1) __builtin_amdgcn_sched_barrier(0); \
        NMARL_CP_KS(di.x, pa, (j) * 16 + 2) NMARL_CP_KS(di.y, pb, (j) * 16 + 3)            \
        NMARL_CP_KS(di.z, pa, (j) * 16 + 4) NMARL_CP_KS(di.w, pb, (j) * 16 + 5)            \
        NMARL_CP_KS(df.x, pa, (j) * 16 + 6) NMARL_CP_KS(df.y, pb, (j) * 16 + 7)            \
        NMARL_CP_KS(df.z, pa, (j) * 16 + 8) NMARL_CP_KS(df.w, pb, (j) * 16 + 9)            \
        NMARL_CP_KS(dO.x, pa, (j) * 16 + 10) NMARL_CP_KS(dO.y, pb, (j) * 16 + 11)          \
        NMARL_CP_KS(dO.z, pa, (j) * 16 + 12) NMARL_CP_KS(dO.w, pb, (j) * 16 + 13)          \
        NMARL_CP_KS(du.x, pa, (j) * 16 + 14) NMARL_CP_KS(du.y, pb, (j) * 16 + 15)          \
        NMARL_CP_MF(du.z, pa) __builtin_amdgcn_sched_barrier(0);                           \
        NMARL_CP_MF(du.w, pb)                                                              \
    }
    // message product: k-step s = 4 t_ + r_ takes D1 unit 16 t_ + 4 q + r_ of the lane's row
#define NMARL_CP_MBL(P, s)                                                                 \
    {                                                                                      \
        const float* p_ = mbase + (s) * 64 * NTM;                                          \
        if (NTM == 4) {                                                                    \
            P##0 = *reinterpret_cast<const float4*>(p_);                                   \
        } else {                                                                           \
            P##0 = *reinterpret_cast<const float4*>(p_ + 4 * sw);                          \
            P##1 = *reinterpret_cast<const float4*>(p_ + 4 * (sw ^ 1));                    \
        }                                                                                  \
    }
#define NMARL_CP_MMF(bv, P)                                                                \
    {                                                                                      \
        am[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##0 .x, bv, am[0], 0, 0, 0);          \
        am[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##0 .y, bv, am[1], 0, 0, 0);          \
        am[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##0 .z, bv, am[2], 0, 0, 0);          \
        am[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##0 .w, bv, am[3], 0, 0, 0);          \
        if (NTM == 8) {                                                                    \
            am[NTM - 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##1 .x, bv, am[NTM - 4], 0, 0, 0); \
            am[NTM - 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##1 .y, bv, am[NTM - 3], 0, 0, 0); \
            am[NTM - 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##1 .z, bv, am[NTM - 2], 0, 0, 0); \
            am[NTM - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(P##1 .w, bv, am[NTM - 1], 0, 0, 0); \
        }                                                                                  \
    }
#define NMARL_CP_MKS(bv, P, s2)                                                            \
    NMARL_CP_MMF(bv, P) __builtin_amdgcn_sched_barrier(0);                                 \
    NMARL_CP_MBL(P, s2) __builtin_amdgcn_sched_barrier(0);
    const int sw = c >> 3;
    NMARL_BSTAMP(1)
    for (int t = t_hi; t >= t_lo; --t) {
        NMARL_BSTAMP_T(2)
        const int tp = t > t_lo ? t - 1 : t_lo;      // clamped: the last prefetch re-reads the range's last step
        const int64_t tu = __builtin_amdgcn_readfirstlane(t);
        const __amdgpu_buffer_rsrc_t rz = make_rsrc(zA + tu * a.dz_st, nb4);
        const __amdgpu_buffer_rsrc_t rd1 = make_rsrc(d1A + tu * a.d1_st, nb1);
        // ---- the neighbours' message adjoints of step t + 1: FIRST thing of the step (it is the hand-off's critical path):
        // one poll, then every source's four loads in flight together.  Nothing else is outstanding at this point (the step
        // ended on vmcnt(0)), so neither the poll's value nor the payload queue behind other loads.
        float4 mm[RMAX][4];
        {
            const unsigned need = (unsigned)(T - 1 - t);
            const float* rbase = a.ring + (int64_t)((t + 1) % a.slots) * a.ring_slot;
            if (!give_up) {
                for (unsigned spins = 0;; ++spins) {
                    const unsigned v = __hip_atomic_load(poll_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__builtin_amdgcn_ballot_w64(v < need) == 0) break;
                    if (spins > a.max_spins) {
                        if (lane == 0) {                 // sticky: the optimiser step refuses this batch
                            __hip_atomic_store((gu32*)a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (a.status) __hip_atomic_store((gu32*)a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        give_up = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            asm volatile("" ::: "memory");               // payload loads stay below the poll
            NMARL_BSTAMP_T(3)
#pragma unroll
            for (int s = 0; s < RMAX; ++s) {
                // nothing to read at the end of the sequence (nothing was handed over yet) and from an absent source (weight 0:
                // padding of the table): a zero-record resource returns 0.0f without touching memory -- no stale, possibly
                // non-finite ring contents times zero, no traffic for the padding
                const uint32_t nb_ = (t == T - 1 || src_w[s] == 0.0f) ? 0u : nbR;
                const __amdgpu_buffer_rsrc_t rr = make_rsrc(rbase + (int64_t)src_n[s] * a.ring_sn, nb_);
                const uint32_t ro_ = loR + src_c4[s];
                mm[s][0] = bload4i<SC1>(rr, ro_, 0);
                mm[s][1] = bload4i<SC1>(rr, ro_, 64);
                mm[s][2] = bload4i<SC1>(rr, ro_, 128);
                mm[s][3] = bload4i<SC1>(rr, ro_, 192);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- relu mask of the message layer at step t and the next step's done flag: needed late, requested behind the payload
        const float keep_next = 1.0f - (a.done + (int64_t)__builtin_amdgcn_readfirstlane(tp) * a.E)[lor];
        float4 mk0, mk1, mk2, mk3;
        if (MASK) {
            const __amdgpu_buffer_rsrc_t rm = make_rsrc(mkA + tu * a.mask_st, nbm);
            mk0 = bload4i(rm, lom, 0);
            mk1 = bload4i(rm, lom, 64);
            mk2 = bload4i(rm, lom, 128);
            mk3 = bload4i(rm, lom, 192);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < RMAX; ++s) {                 // summed in source order (the restatement's order)
            const float w_ = src_w[s];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gh[j].x += w_ * mm[s][j].x; gh[j].y += w_ * mm[s][j].y; gh[j].z += w_ * mm[s][j].z; gh[j].w += w_ * mm[s][j].w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        unsigned mbits = 0xFFFFu;                     // bit 4 j + i: hm > 0 for unit 16 j + 4 q + i of the lane's row
        if (MASK) {
            mbits = (mk0.x > 0.0f ? 1u : 0u) | (mk0.y > 0.0f ? 2u : 0u) | (mk0.z > 0.0f ? 4u : 0u) | (mk0.w > 0.0f ? 8u : 0u) |
                    (mk1.x > 0.0f ? 16u : 0u) | (mk1.y > 0.0f ? 32u : 0u) | (mk1.z > 0.0f ? 64u : 0u) | (mk1.w > 0.0f ? 128u : 0u) |
                    (mk2.x > 0.0f ? 256u : 0u) | (mk2.y > 0.0f ? 512u : 0u) | (mk2.z > 0.0f ? 1024u : 0u) | (mk2.w > 0.0f ? 2048u : 0u) |
                    (mk3.x > 0.0f ? 4096u : 0u) | (mk3.y > 0.0f ? 8192u : 0u) | (mk3.z > 0.0f ? 16384u : 0u) | (mk3.w > 0.0f ? 32768u : 0u);
        }
        __builtin_amdgcn_sched_barrier(0);
        int aoff = (q * 16 + c) * 8;                 // opaque copies: see lstm_bptt_seq_kernel
        asm volatile("" : "+v"(aoff));
        // STORES take (lane offset + constant) in a VGPR, never a scalar offset: after a 16-byte buffer store with an SGPR
        // soffset hipcc (ROCm 7.2) lets the next VALU instruction overwrite the store's data registers without the wait
        // state it inserts for the soffset-less form -- on gfx950 the store then picks up the NEXT value of its first data
        // register (seen: the x component of a gate's dz replaced by the next gate's, intermittently).  Opaque copies keep
        // the 28 sums from being hoisted out of the loop into registers.
        uint32_t so4 = lo4, so1 = lo1, soR = loR;
        asm volatile("" : "+v"(so4), "+v"(so1), "+v"(soR));
        const float* abase = lds + aoff;
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 di, df, dO, du;
        NMARL_BSTAMP_T(4)
        NMARL_CP_CELL(sA, 0)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_CP_LOADG(sA, t, 2)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_BSTAMP_T(5)
        NMARL_CP_PROD(0)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_BSTAMP_T(6)
        NMARL_CP_CELL(sB, 1)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_CP_LOADG(sB, t, 3)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_BSTAMP_T(7)
        NMARL_CP_PROD(1)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_BSTAMP_T(8)
        NMARL_CP_CELL(sA, 2)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_CP_LOADG(sA, tp, 0)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_BSTAMP_T(9)
        NMARL_CP_PROD(2)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_BSTAMP_T(10)
        NMARL_CP_CELL(sB, 3)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_CP_LOADG(sB, tp, 1)
        NMARL_CP_LOADH(tp)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_BSTAMP_T(11)
        NMARL_CP_PROD(3)
        __builtin_amdgcn_sched_barrier(0);
        NMARL_BSTAMP_T(12)
        // ---- D1 = dx (relu-masked) in the lane's own units; bias gradient of the message layer
        f32x4 d1v[4];
        {
            float qm[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d1v[j] = acc[j];
                if (MASK) {
                    if (!((mbits >> (4 * j)) & 1u)) d1v[j][0] = 0.0f;
                    if (!((mbits >> (4 * j + 1)) & 1u)) d1v[j][1] = 0.0f;
                    if (!((mbits >> (4 * j + 2)) & 1u)) d1v[j][2] = 0.0f;
                    if (!((mbits >> (4 * j + 3)) & 1u)) d1v[j][3] = 0.0f;
                }
                bstore4(rd1, so1 + 64 * j, float4{d1v[j][0], d1v[j][1], d1v[j][2], d1v[j][3]});
                const float a0 = odd ? d1v[j][1] : d1v[j][0], g0 = odd ? d1v[j][0] : d1v[j][1];
                const float a1 = odd ? d1v[j][3] : d1v[j][2], g1 = odd ? d1v[j][2] : d1v[j][3];
                const float r0 = a0 + dpp_xor1(g0), r1 = a1 + dpp_xor1(g1);
                const float k2 = hi ? r1 : r0, g2 = hi ? r0 : r1;
                qm[j] = k2 + dpp_xor2(g2);            // unit 2 hi + odd of group j, summed over the quad's rows
            }
            const float k0_ = b2 ? qm[2] : qm[0], g0_ = b2 ? qm[0] : qm[2];
            const float k1_ = b2 ? qm[3] : qm[1], g1_ = b2 ? qm[1] : qm[3];
            const float r0_ = k0_ + __shfl_xor(g0_, 4, 64), r1_ = k1_ + __shfl_xor(g1_, 4, 64);
            const float k3_ = b3 ? r1_ : r0_, g3_ = b3 ? r0_ : r1_;
            dbm += k3_ + __shfl_xor(g3_, 8, 64);
        }
        NMARL_BSTAMP_T(13)
        // ---- M_t^T = W_msg . D1^T, handed to the neighbours' blocks
        {
            int moff = (q * 16 + c) * NTM;
            asm volatile("" : "+v"(moff));
            const float* mbase = lds + IMG8 + moff;
            f32x4 am[NTM];
#pragma unroll
            for (int i = 0; i < NTM; ++i) am[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                float4 pa0, pa1, pb0, pb1;
                pa1 = pb1 = float4{0.f, 0.f, 0.f, 0.f};
                NMARL_CP_MBL(pa, 0) NMARL_CP_MBL(pb, 1) __builtin_amdgcn_sched_barrier(0);
                NMARL_CP_MKS(d1v[0][0], pa, 2) NMARL_CP_MKS(d1v[0][1], pb, 3) NMARL_CP_MKS(d1v[0][2], pa, 4) NMARL_CP_MKS(d1v[0][3], pb, 5)
                NMARL_CP_MKS(d1v[1][0], pa, 6) NMARL_CP_MKS(d1v[1][1], pb, 7) NMARL_CP_MKS(d1v[1][2], pa, 8) NMARL_CP_MKS(d1v[1][3], pb, 9)
                NMARL_CP_MKS(d1v[2][0], pa, 10) NMARL_CP_MKS(d1v[2][1], pb, 11) NMARL_CP_MKS(d1v[2][2], pa, 12) NMARL_CP_MKS(d1v[2][3], pb, 13)
                NMARL_CP_MKS(d1v[3][0], pa, 14) NMARL_CP_MKS(d1v[3][1], pb, 15)
                NMARL_CP_MMF(d1v[3][2], pa) __builtin_amdgcn_sched_barrier(0);
                NMARL_CP_MMF(d1v[3][3], pb)
            }
            const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.ring + (int64_t)(t % a.slots) * a.ring_slot + (int64_t)n * a.ring_sn, nbR);
#pragma unroll
            for (int i = 0; i < NTM; ++i) bstore4_wt(rw, soR + 64 * i, float4{am[i][0], am[i][1], am[i][2], am[i][3]});
        }
        NMARL_BSTAMP_T(14)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains its write-through stores
        NMARL_BSTAMP_T(15)
        if (lane == 0 && !(a.fault && blockIdx.x == 0))
            __hip_atomic_store((gu32*)(a.flags + my_flag_idx), (unsigned)(T - t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // own recurrent part of dL/dh_{t-1} = (dz @ wh^T) keep_t: into the prefetched inputs of step t - 1 (landed: drained)
        if (t == t_lo && arow_ok) {                  // end of the range: state for the next launch of a step-wise run
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 v_ = acc[4 + j] * keepA;
                *reinterpret_cast<float4*>(a.dhr_io + (int64_t)n * a.io_sn + arow_raw * H + 4 * q + 16 * j) =
                    float4{v_[0], v_[1], v_[2], v_[3]};
                *reinterpret_cast<float4*>(a.dc_io + (int64_t)n * a.io_sn + arow_raw * H + 4 * q + 16 * j) = dc[j];
            }
        }
        gh[0].x += acc[4][0] * keepA; gh[0].y += acc[4][1] * keepA; gh[0].z += acc[4][2] * keepA; gh[0].w += acc[4][3] * keepA;
        gh[1].x += acc[5][0] * keepA; gh[1].y += acc[5][1] * keepA; gh[1].z += acc[5][2] * keepA; gh[1].w += acc[5][3] * keepA;
        gh[2].x += acc[6][0] * keepA; gh[2].y += acc[6][1] * keepA; gh[2].z += acc[6][2] * keepA; gh[2].w += acc[6][3] * keepA;
        gh[3].x += acc[7][0] * keepA; gh[3].y += acc[7][1] * keepA; gh[3].z += acc[7][2] * keepA; gh[3].w += acc[7][3] * keepA;
        if (DY) {                                    // + the heads' dL/dh of step t - 1 (its dy8 was requested mid-step)
            NMARL_CP_HEADS()
        }
        keepA = keep_next;
        NMARL_BSTAMP_T(16)
    }
    NMARL_BSTAMP(17)
#undef NMARL_CP_LOADG
#undef NMARL_CP_LOADH
#undef NMARL_CP_HEADS
#undef NMARL_CP_BL
#undef NMARL_CP_MF
#undef NMARL_CP_KS
#undef NMARL_CP_CELLB
#undef NMARL_CP_DB1
#undef NMARL_CP_CELL
#undef NMARL_CP_PROD
#undef NMARL_CP_MBL
#undef NMARL_CP_MMF
#undef NMARL_CP_MKS

    // ---- bias gradients: every lane holds 4 + 1 finished column sums of its wave; sum over the 8 waves, a step-wise run
    // accumulates over its launches
    __syncthreads();                                 // every wave is done with the images: reuse their LDS
    {
        const int g = 2 * (c & 1) + ((c >> 1) & 1), iu = 2 * ((c >> 2) & 1) + ((c >> 3) & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) lds[wave * G4 + 64 * g + 16 * j + 4 * q + iu] = dbacc[j];
        lds[WAVES * G4 + wave * H + 16 * iu + 4 * q + 2 * ((c >> 1) & 1) + (c & 1)] = dbm;
    }
    __syncthreads();
    const bool accum = t_hi != T - 1;
    if (threadIdx.x < G4) {
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += lds[w * G4 + threadIdx.x];
        float* o = a.db_part + (int64_t)n * a.db_sn + (int64_t)blk * G4 + threadIdx.x;
        *o = accum ? *o + v : v;
    } else if (threadIdx.x < G4 + H) {
        const int u = threadIdx.x - G4;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) v += lds[WAVES * G4 + w * H + u];
        float* o = a.dbm_part + (int64_t)n * a.dbm_sn + (int64_t)blk * H + u;
        *o = accum ? *o + v : v;
    }
}

// image_m[(s, q)][c][slot(tau)] = W_msg[16 tau + c][16 t_ + 4 q + r_],  s = 4 t_ + r_;  NTM = K / 16 tiles (8: swizzled like the
// main image)
__global__ void lstm_bptt_msg_wimage_kernel(const int N, const int K, const float* w, const int64_t w_sn, float* img,
                                            const int64_t img_sn) {
    const int NTM = K / 16;
    const int per_agent = K * H;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * per_agent) return;
    const int n = (int)(idx / per_agent), o = (int)(idx % per_agent);
    const int slot = o % NTM, cc = (o / NTM) % 16, sq = o / (NTM * 16);
    const int qq = sq & 3, s = sq >> 2;
    const int t_ = s >> 2, r_ = s & 3;
    int tau = slot;
    if (NTM == 8) tau = (((slot >> 2) ^ (cc >> 3)) << 2) | (slot & 3);
    img[(int64_t)n * img_sn + o] = w[(int64_t)n * w_sn + (int64_t)(16 * tau + cc) * H + 16 * t_ + 4 * qq + r_];
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

extern "C" int nmarl_lstm_bptt_step_parts(int64_t E) { return (int)((E + ROWS_B - 1) / ROWS_B * WAVES); }

extern "C" int nmarl_lstm_bptt_step_db(int64_t E, int32_t N, int32_t Hh, int32_t KM, const float* gates, int64_t gates_sn,
                                       const float* c_prev, int64_t c_prev_sn, const float* c_new, int64_t c_new_sn,
                                       const float* done, const float* dh, int64_t dh_sn, const float* dh2, int64_t dh2_sn,
                                       const float* dc_in, int64_t dc_sn, const float* img, int64_t img_sn, float* dz,
                                       int64_t dz_sn, float* dc_prev, int64_t dc_prev_sn, float* dx, int64_t dx_sn,
                                       const float* mask, int64_t mask_sn, int64_t mask_row, float* dhd, int64_t dhd_sn,
                                       int32_t apply_keep, float* db_part, int64_t db_sn, void* stream) {
    if (db_part && E > 0 && (db_sn < (E + ROWS_B - 1) / ROWS_B * WAVES * (int64_t)G4 || ((uintptr_t)db_part % 4))) return NMARL_EINVAL;
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
    a.db_part = db_part; a.db_sn = db_sn;
    static NmarlPerDeviceOnce lds_once;
    if (const unsigned long long lds_bit = lds_once.pending(); lds_bit != ~0ull) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_step_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                G4 * 16 * 4 * 4) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_step_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                G4 * 16 * 8 * 4) != hipSuccess)
            return NMARL_EHIP;
        lds_once.done(lds_bit);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(((E + ROWS_B - 1) / ROWS_B) * N));
    if (KM == 0) hipLaunchKernelGGL(lstm_bptt_step_kernel<4>, grid, dim3(512), (size_t)G4 * 16 * 4 * 4, st, a);
    else hipLaunchKernelGGL(lstm_bptt_step_kernel<8>, grid, dim3(512), (size_t)G4 * 16 * 8 * 4, st, a);
    return nmarl_check_launch();
}

extern "C" int nmarl_lstm_bptt_step(int64_t E, int32_t N, int32_t Hh, int32_t KM, const float* gates, int64_t gates_sn,
                                    const float* c_prev, int64_t c_prev_sn, const float* c_new, int64_t c_new_sn,
                                    const float* done, const float* dh, int64_t dh_sn, const float* dh2, int64_t dh2_sn,
                                    const float* dc_in, int64_t dc_sn, const float* img, int64_t img_sn, float* dz,
                                    int64_t dz_sn, float* dc_prev, int64_t dc_prev_sn, float* dx, int64_t dx_sn,
                                    const float* mask, int64_t mask_sn, int64_t mask_row, float* dhd, int64_t dhd_sn,
                                    int32_t apply_keep, void* stream) {
    return nmarl_lstm_bptt_step_db(E, N, Hh, KM, gates, gates_sn, c_prev, c_prev_sn, c_new, c_new_sn, done, dh, dh_sn, dh2, dh2_sn, dc_in,
                                   dc_sn, img, img_sn, dz, dz_sn, dc_prev, dc_prev_sn, dx, dx_sn, mask, mask_sn, mask_row, dhd, dhd_sn,
                                   apply_keep, nullptr, 0, stream);
}

extern "C" int nmarl_lstm_bptt_seq_blocks(int64_t E) { return (int)((E + ROWS_B - 1) / ROWS_B); }

static int launch_bptt_seq(int32_t T, int64_t E, int32_t N, int32_t Hh, const float* gates, int64_t gates_sn,
                           int64_t gates_st, const float* c_all, int64_t c_sn, int64_t c_st, const float* done,
                           const float* dh_ext, int64_t dh_sn, int64_t dh_st, const float* dy8, int64_t dy_sn, int64_t dy_st,
                           const float* hw, int64_t hw_sn, int32_t O, const float* img, int64_t img_sn,
                           float* dz, int64_t dz_sn, int64_t dz_st, float* db_part, int64_t db_sn, float* dh0,
                           int64_t dh0_sn, float* dc0, int64_t dc0_sn, void* stream) {
    if (Hh != H || E < 0 || N <= 0 || T <= 0 || (E > 0 && (!gates || !c_all || !done || (!dh_ext && !dy8) || (dh_ext && dy8) || !img || !dz)))
        return NMARL_EINVAL;
    if (dy8 && (!hw || O <= 0 || O > 8 || hw_sn < (int64_t)H * O || dy_st < E * 8 || (dy_st % 4) || !sn_ok(dy_sn, (T - 1) * dy_st + E * 8) ||
                ((uintptr_t)dy8 % 16)))
        return NMARL_EINVAL;
    if (!dh_ext) { dh_ext = c_all; dh_sn = c_sn; dh_st = c_st; }       // (not read: any valid panel keeps the checks below uniform)
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
    a.dy8 = dy8; a.dy_sn = dy_sn; a.dy_st = dy_st; a.hw = hw; a.hw_sn = hw_sn; a.O = O;
    static NmarlPerDeviceOnce lds_once;
    if (const unsigned long long lds_bit = lds_once.pending(); lds_bit != ~0ull) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_seq_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                SEQ_IMG * 4) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_seq_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (SEQ_IMG + 512) * 4) != hipSuccess)
            return NMARL_EHIP;
        lds_once.done(lds_bit);
    }
    if (dy8)
        hipLaunchKernelGGL(lstm_bptt_seq_kernel<true>, dim3((unsigned)(nblk * N)), dim3(512), (size_t)(SEQ_IMG + 512) * 4,
                           static_cast<hipStream_t>(stream), a);
    else
        hipLaunchKernelGGL(lstm_bptt_seq_kernel<false>, dim3((unsigned)(nblk * N)), dim3(512), (size_t)SEQ_IMG * 4,
                           static_cast<hipStream_t>(stream), a);
    return nmarl_check_launch();
}

extern "C" int nmarl_lstm_bptt_seq(int32_t T, int64_t E, int32_t N, int32_t Hh, const float* gates, int64_t gates_sn,
                                   int64_t gates_st, const float* c_all, int64_t c_sn, int64_t c_st, const float* done,
                                   const float* dh_ext, int64_t dh_sn, int64_t dh_st, const float* img, int64_t img_sn,
                                   float* dz, int64_t dz_sn, int64_t dz_st, float* db_part, int64_t db_sn, float* dh0,
                                   int64_t dh0_sn, float* dc0, int64_t dc0_sn, void* stream) {
    if (!dh_ext) return NMARL_EINVAL;
    return launch_bptt_seq(T, E, N, Hh, gates, gates_sn, gates_st, c_all, c_sn, c_st, done, dh_ext, dh_sn, dh_st, nullptr, 0, 0, nullptr, 0, 0,
                           img, img_sn, dz, dz_sn, dz_st, db_part, db_sn, dh0, dh0_sn, dc0, dc0_sn, stream);
}

extern "C" int nmarl_lstm_bptt_seq_dy(int32_t T, int64_t E, int32_t N, int32_t Hh, const float* gates, int64_t gates_sn,
                                      int64_t gates_st, const float* c_all, int64_t c_sn, int64_t c_st, const float* done,
                                      const float* dy8, int64_t dy_sn, int64_t dy_st, const float* hw, int64_t hw_sn, int32_t O,
                                      const float* img, int64_t img_sn, float* dz, int64_t dz_sn, int64_t dz_st, float* db_part,
                                      int64_t db_sn, float* dh0, int64_t dh0_sn, float* dc0, int64_t dc0_sn, void* stream) {
    if (!dy8) return NMARL_EINVAL;
    return launch_bptt_seq(T, E, N, Hh, gates, gates_sn, gates_st, c_all, c_sn, c_st, done, nullptr, 0, 0, dy8, dy_sn, dy_st, hw, hw_sn, O,
                           img, img_sn, dz, dz_sn, dz_st, db_part, db_sn, dh0, dh0_sn, dc0, dc0_sn, stream);
}

extern "C" int nmarl_lstm_bptt_msg_wimage(int32_t N, int32_t K, const float* w_msg, int64_t w_sn, float* img, int64_t img_sn,
                                          void* stream) {
    if (N <= 0 || (K != 64 && K != 128) || !w_msg || !img || w_sn < (int64_t)K * H || img_sn < (int64_t)K * H || (img_sn % 4) ||
        ((uintptr_t)img % 16))
        return NMARL_EINVAL;
    const int64_t total = (int64_t)N * K * H;
    hipLaunchKernelGGL(lstm_bptt_msg_wimage_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), N, K, w_msg, w_sn, img, img_sn);
    return nmarl_check_launch();
}

extern "C" int nmarl_lstm_bptt_coupled_ws_words(int64_t E, int32_t N) {
    // flags: one word per (agent, 128-row tile, wave) + the error word, rounded up to 64 words
    const int64_t tiles = (E + ROWS_B - 1) / ROWS_B;
    return (int)(((int64_t)N * tiles * WAVES + 1 + 63) / 64 * 64);
}

namespace {
template <int NTM, int RMAX, bool MASK>
int launch_coupled(const CoupledArgs& a, unsigned grid, size_t lds_bytes, hipStream_t st) {
    static NmarlPerDeviceOnce once;
    if (const unsigned long long bit = once.pending(); bit != ~0ull) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_coupled_kernel<NTM, RMAX, MASK, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(lstm_bptt_coupled_kernel<NTM, RMAX, MASK, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
            return NMARL_EHIP;
        once.done(bit);
    }
    if (a.dy8) hipLaunchKernelGGL((lstm_bptt_coupled_kernel<NTM, RMAX, MASK, true>), dim3(grid), dim3(512), lds_bytes, st, a);
    else hipLaunchKernelGGL((lstm_bptt_coupled_kernel<NTM, RMAX, MASK, false>), dim3(grid), dim3(512), lds_bytes, st, a);
    return nmarl_check_launch();
}
}  // namespace

// blocks per CU of the coupled kernel with message rows of K floats (the occupancy API's answer for its 512 threads, registers
// and 128 + K / 4 KB of LDS: 1 on gfx950) -- nmarl_handoff_capacity
NMARL_INTERNAL int nmarl_bptt_coupled_occupancy(int K) {
    if (K != 64 && K != 128) return -1;
    const size_t lds_bytes = ((size_t)G4 * 16 * 8 + (size_t)K * H) * 4;
    // the smallest answer over the instantiations a message row of K floats can launch (lstm_comm with two neighbour slots: <8,2,true>;
    // K = 64: lstm_comm with one slot <4,2,true>, lstm_ic3 with <= 2 / <= 4 sources per agent <4,2,false> / <4,4,false>): the
    // residency check of the one-launch form must hold for the kernel that is launched, whichever it is.  Asked once per device.
    static std::atomic<int> cache[64][2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    const int kd = K == 128 ? 1 : 0;
    if (dev >= 0 && dev < 64) {
        const int hit = cache[dev][kd].load();
        if (hit > 0) return hit - 1;
    }
    int best = -1;
    auto ask = [&](auto kernel) {
        int per_cu = 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 512, lds_bytes) != hipSuccess)
            return false;
        best = best < 0 || per_cu < best ? per_cu : best;
        return true;
    };
    const bool ok = K == 128 ? ask(lstm_bptt_coupled_kernel<8, 2, true>)
                             : (ask(lstm_bptt_coupled_kernel<4, 2, true>) && ask(lstm_bptt_coupled_kernel<4, 2, false>) &&
                                ask(lstm_bptt_coupled_kernel<4, 4, false>));
    if (!ok) return -1;
    if (dev >= 0 && dev < 64) cache[dev][kd].store(best + 1);
    return best;
}

namespace {
__global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ w, const int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) w[i] = 0u;
}
}  // namespace

extern "C" int nmarl_lstm_bptt_coupled(const nmarl_bptt_coupled_t* p, void* stream) {
    if (!p || p->H != H || p->E < 0 || p->N <= 0 || p->T <= 0 || (p->kind != 1 && p->kind != 2) || p->r_max <= 0 || p->r_max > 4)
        return NMARL_EINVAL;
    const int64_t E = p->E;
    const int N = p->N, T = p->T;
    const int K = p->kind == 1 ? H * p->m_max : H;                        // floats per message row
    if (K != 64 && K != 128) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    if (E > (1 << 21) || ((uintptr_t)p->status % 4)) return NMARL_EINVAL;
    if ((!p->dh_ext) == (!p->dy8)) return NMARL_EINVAL;            // the heads' dL/dh: as a tensor, or as dy8 + the heads' weights
    if (p->dy8 && (!p->hw || p->O <= 0 || p->O > 8 || p->hw_sn < (int64_t)H * p->O || p->dy_st < E * 8 || (p->dy_st % 4) ||
                   !sn_ok(p->dy_sn, (T - 1) * p->dy_st + E * 8) || ((uintptr_t)p->dy8 % 16)))
        return NMARL_EINVAL;
    if (!p->gates || !p->c_all || !p->done || !p->img || !p->img_m || !p->dz || !p->d1 || !p->ring || !p->db_part ||
        !p->dbm_part || !p->dhr_io || !p->dc_io || !p->ws || !p->rev_agent || !p->rev_col || !p->rev_w || (p->kind == 1 && !p->mask))
        return NMARL_EINVAL;
    const int64_t tiles = (E + ROWS_B - 1) / ROWS_B;
    if (p->gates_st < E * G4 || (p->gates_st % 4) || !sn_ok(p->gates_sn, (T - 1) * p->gates_st + E * G4) || p->dz_st < E * G4 ||
        (p->dz_st % 4) || !sn_ok(p->dz_sn, (T - 1) * p->dz_st + E * G4) || p->c_st < E * H || (p->c_st % 4) ||
        !sn_ok(p->c_sn, T * p->c_st + E * H) || (p->dh_ext && (p->dh_st < E * H || (p->dh_st % 4) || !sn_ok(p->dh_sn, (T - 1) * p->dh_st + E * H))) ||
        p->d1_st < E * H || (p->d1_st % 4) || !sn_ok(p->d1_sn, (T - 1) * p->d1_st + E * H) || p->img_sn < (int64_t)G4 * 2 * H ||
        (p->img_sn % 4) || p->imgm_sn < (int64_t)K * H || (p->imgm_sn % 4) || !sn_ok(p->ring_sn, E * K) ||
        !sn_ok(p->ring_slot, (N - 1) * p->ring_sn + E * K) || p->db_sn < tiles * G4 || p->dbm_sn < tiles * H || !sn_ok(p->io_sn, E * H) ||
        (p->kind == 1 && (p->mask_row < H || (p->mask_row % 4) || p->mask_st < (E - 1) * p->mask_row + H || (p->mask_st % 4) ||
                          !sn_ok(p->mask_sn, (T - 1) * p->mask_st + (E - 1) * p->mask_row + H) || ((uintptr_t)p->mask % 16))) ||
        ((uintptr_t)p->img % 16) || ((uintptr_t)p->img_m % 16) || ((uintptr_t)p->gates % 16) || ((uintptr_t)p->dz % 16) ||
        ((uintptr_t)p->c_all % 16) || (p->dh_ext && ((uintptr_t)p->dh_ext % 16)) || ((uintptr_t)p->d1 % 16) || ((uintptr_t)p->ring % 16) ||
        ((uintptr_t)p->dhr_io % 16) || ((uintptr_t)p->dc_io % 16) || ((uintptr_t)p->ws % 4))
        return NMARL_EINVAL;
    CoupledArgs a{};
    a.gates = p->gates; a.c_all = p->c_all; a.done = p->done; a.dh_ext = p->dh_ext; a.img = p->img; a.img_m = p->img_m;
    a.mask = p->kind == 1 ? p->mask : nullptr;
    a.dz = p->dz; a.d1 = p->d1; a.ring = p->ring; a.db_part = p->db_part; a.dbm_part = p->dbm_part; a.dhr_io = p->dhr_io;
    a.dc_io = p->dc_io;
    a.flags = reinterpret_cast<unsigned*>(p->ws);
    a.err = a.flags + (int64_t)N * tiles * WAVES;
    a.rev_agent = p->rev_agent; a.rev_col = p->rev_col; a.rev_w = p->rev_w;
    a.gates_sn = p->gates_sn; a.gates_st = p->gates_st; a.c_sn = p->c_sn; a.c_st = p->c_st; a.dh_sn = p->dh_sn; a.dh_st = p->dh_st;
    a.img_sn = p->img_sn; a.imgm_sn = p->imgm_sn; a.mask_sn = p->mask_sn; a.mask_st = p->mask_st; a.dz_sn = p->dz_sn;
    a.dz_st = p->dz_st; a.d1_sn = p->d1_sn; a.d1_st = p->d1_st; a.ring_sn = p->ring_sn; a.ring_slot = p->ring_slot;
    a.db_sn = p->db_sn; a.dbm_sn = p->dbm_sn; a.io_sn = p->io_sn;
    a.E = E; a.N = N; a.T = T; a.mask_row = (int32_t)p->mask_row; a.tiles = (int32_t)tiles;
    a.dy8 = p->dy8; a.hw = p->hw; a.dy_sn = p->dy_sn; a.dy_st = p->dy_st; a.hw_sn = p->hw_sn; a.O = p->O;
    if (p->ring_slots < 2) return NMARL_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // every polled word starts at zero for every call (flags count the steps done WITHIN the call); the error word behind
    // them is NOT cleared: a time-out stays visible until the host has dealt with it
    // (a kernel, not hipMemsetAsync: inside a captured hipGraph the memset node was not reliably ordered in front of the kernel node
    // behind it -- after some hundred replays of the update graph the blocks saw the previous replay's step counts, read message
    // adjoints of the previous batch and the gradient came out slightly, and not reproducibly, wrong: tools/determinism.py)
    {
        const int64_t nw = (int64_t)N * tiles * WAVES;
        hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, reinterpret_cast<unsigned*>(p->ws), nw);
    }
    const int64_t grid = tiles * N;
    // one launch for all T steps needs every block resident (the waves wait for their neighbours' blocks): one 512-thread
    // block with 160 KB of LDS per CU, so grid <= CUs; and a symmetric neighbour relation (two ring slots).  Otherwise
    // step by step: T launches of the same kernel, state through dhr_io / dc_io.
    const int cap = nmarl_handoff_capacity(2, K);
    if (cap < 0) return NMARL_EHIP;
    const bool one_launch = p->mode != 2 && p->ring_slots >= T && (p->mode == 1 || (p->symmetric && grid <= cap));
    a.slots = one_launch ? T : 2;
    a.status = p->status;
    a.fault = one_launch && nmarl_handoff_take_fault() ? 1 : 0;
    a.max_spins = a.fault ? NMARL_HANDOFF_FAULT_SPINS : NMARL_HANDOFF_MAX_SPINS;
    const size_t lds_bytes = ((size_t)G4 * 16 * 8 + (size_t)K * H) * 4;
    const int ntm = K / 16;
    const int rmax = p->r_max <= 2 ? 2 : 4;
    if (p->r_row != rmax) return NMARL_EINVAL;            // tables come padded to 2 or 4 entries per agent
    int rc = NMARL_OK;
    for (int t_hi = T - 1; t_hi >= 0 && rc == NMARL_OK; t_hi = one_launch ? -1 : t_hi - 1) {
        a.t_hi = t_hi;
        a.t_lo = one_launch ? 0 : t_hi;
        if (p->kind == 1 && ntm == 8 && rmax == 2) rc = launch_coupled<8, 2, true>(a, (unsigned)grid, lds_bytes, st);
        else if (p->kind == 1 && ntm == 4 && rmax == 2) rc = launch_coupled<4, 2, true>(a, (unsigned)grid, lds_bytes, st);
        else if (p->kind == 2 && rmax == 2) rc = launch_coupled<4, 2, false>(a, (unsigned)grid, lds_bytes, st);
        else if (p->kind == 2 && rmax == 4) rc = launch_coupled<4, 4, false>(a, (unsigned)grid, lds_bytes, st);
        else rc = NMARL_EINVAL;
    }
    return rc;
}

#ifdef NMARL_STEP_TIMELINE
extern "C" int nmarl_timeline_set_bptt(unsigned long long* p, void* stream) {
    hipLaunchKernelGGL(tl_b_set_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), p);
    return nmarl_check_launch();
}
#endif
