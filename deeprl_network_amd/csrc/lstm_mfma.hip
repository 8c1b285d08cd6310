// Fused LSTM step for agent-batched 64-unit cells on gfx950 matrix cores:
//
//     z = zadd1 (+ zadd2) + (h * (1 - done)) @ Wh      [rows x 256], fp32 MFMA
//     i,f,o = sigmoid(z + b), u = tanh(z + b);  c' = f * (c * (1 - done)) + i * u;  h' = o * tanh(c')
//
// i.e. the recurrent half of agents/utils.py:102-113 (lstm), 199-208 (lstm_comm), 401-408 (lstm_ic3)
// with the cell fused into the GEMM epilogue: the [rows x 256] pre-activation never goes to HBM
// (the separate GEMM + cell pair writes and re-reads it: 2 x 33.5 MB per step at E = 4096).
//
// Mapping (H = 64 -> 256 gate columns).  One 256-thread block = 4 waves = 128 rows of ONE agent and ALL
// 256 columns, because unit j needs columns j, 64+j, 128+j, 192+j together.  Wh (64 x 256 fp32 = 64 KB) is
// staged once per block in LDS, each wave's 32 x 64 tile of h in a padded LDS tile (row pitch 65: the 32
// rows a wave reads per MFMA operand fall in 32 different banks).  A wave owns a 32 x 256 strip =
// 8 accumulator tiles of v_mfma_f32_32x32x2_f32 (128 accumulator registers), K = 64 -> 32 MFMAs per tile.
// The accumulators are INITIALISED with zadd1 (+ zadd2) + bias (C operand), so the epilogue needs no
// further loads except c; in the C/D layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
// a lane holds all four gates of its (row, unit) pairs in registers: the cell is lane-local.
// fp32 MFMA = the fp32 vector rate (157 TFLOP/s chip peak, MI355X_MICROARCH.md): 256 MFMAs x 64 cycles
// per 32 rows and wave -> 6.8 us of matrix time for 32768 rows on 256 CUs.
#include "common.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int H = 64;
constexpr int G4 = 4 * H;          // 256 gate columns
constexpr int ROWS_W = 32;         // rows per wave
constexpr int WAVES = 4;
constexpr int ROWS_B = ROWS_W * WAVES;
constexpr int APITCH = H + 1;      // 65

struct FusedArgs {
    const float *h_in, *wh, *bias, *zadd1, *zadd2, *c_prev, *done;
    float *gates, *c_new, *h_new;
    int64_t h_sn, wh_sn, bias_sn, zadd1_sn, zadd2_sn, c_prev_sn, gates_sn, c_new_sn, h_new_sn;
    int64_t E;
    int blocks_per_agent;
    nmarl_head_t hd;               // actor / critic head of the epilogue (kind 0: none)
};

constexpr int MAXA = 8;            // widest action set the head epilogue keeps in registers

// Epilogue transcendentals: 160 per lane with ONE wave per SIMD, so their latency is fully exposed.
// sigmoid(x) = rcp(1 + 2^(-x log2 e)) on the hardware exp2 / rcp units (about 1 ulp each, relative error of the
// result <= ~1e-6 for |x| < 20; both saturate correctly), tanh(x) = 2 sigmoid(2x) - 1 (absolute error <= 2.4e-7).
// The separate cell kernel (other H, backward checks) keeps libm's expf / tanhf; both agree within 1e-6.
__device__ __forceinline__ float sigm(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) { return 2.0f * sigm(2.0f * x) - 1.0f; }

template <bool HAS_Z2>
__global__ __launch_bounds__(256, 1) void lstm_step_mfma_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w_lds = lds;                                   // [64][256]
    float* a_lds = lds + H * G4;                          // [WAVES][32][65]
    const int n = blockIdx.x / a.blocks_per_agent;
    const int64_t row_blk = (int64_t)(blockIdx.x - n * a.blocks_per_agent) * ROWS_B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = row_blk + wave * ROWS_W;         // first row of this wave's strip
    const int col = lane & 31, half = lane >> 5;
    float* a_tile = a_lds + wave * ROWS_W * APITCH;

    // ---- accumulators <- zadd1: 128 independent loads per lane, nothing consumes them before the K loop, so
    // they all stay in flight behind the LDS staging (an add right after each load serialised them: 29 us)
    f32x16 acc[8];
    const float* z1 = a.zadd1 + (int64_t)n * a.zadd1_sn;
    int64_t rofs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        rofs[r] = (row < a.E ? row : a.E - 1);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = z1[rofs[r] * G4 + t * 32 + col];
    // ---- c_prev of this lane's (row, unit) pairs
    float cp[2][16];
    const float* cpn = a.c_prev + (int64_t)n * a.c_prev_sn;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            cp[jj][r] = cpn[rofs[r] * H + jj * 32 + col];
        }

    // ---- stage Wh (64 KB, whole block) and this wave's h tile (masked by 1 - done) in LDS
    {
        const float4* wg = reinterpret_cast<const float4*>(a.wh + (int64_t)n * a.wh_sn);
        float4* wl = reinterpret_cast<float4*>(w_lds);
#pragma unroll
        for (int i = 0; i < (H * G4 / 4) / 256; ++i) wl[i * 256 + threadIdx.x] = wg[i * 256 + threadIdx.x];
        const float* hn = a.h_in + (int64_t)n * a.h_sn;
#pragma unroll
        for (int i = 0; i < (ROWS_W * H / 4) / 64; ++i) {       // 8 float4 per lane, coalesced
            const int v = i * 64 + lane;
            const int r = v >> 4, k4 = (v & 15) * 4;
            int64_t row = row0 + r;
            row = row < a.E ? row : a.E - 1;
            const float keep = 1.0f - a.done[row];
            const float4 x = *reinterpret_cast<const float4*>(hn + row * H + k4);
            float* d = a_tile + r * APITCH + k4;
            d[0] = x.x * keep; d[1] = x.y * keep; d[2] = x.z * keep; d[3] = x.w * keep;
        }
    }
    __syncthreads();

    // ---- + bias (+ zadd2): C operand of the first MFMA of every tile
    {
        const float* bn = a.bias + (int64_t)n * a.bias_sn;
        const float* z2 = HAS_Z2 ? a.zadd2 + (int64_t)n * a.zadd2_sn : nullptr;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float b = bn[t * 32 + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][r] + b;
                if (HAS_Z2) v += z2[rofs[r] * G4 + t * 32 + col];
                acc[t][r] = v;
            }
        }
    }

    // ---- K loop: 32 steps of K = 2, 8 column tiles each
#pragma unroll 4
    for (int kk = 0; kk < H / 2; ++kk) {
        const float av = a_tile[col * APITCH + 2 * kk + half];          // A[i = lane & 31][k = lane >> 5]
        const float* wrow = w_lds + (2 * kk + half) * G4 + col;          // B[k = lane >> 5][j = lane & 31]
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wrow[t * 32], acc[t], 0, 0, 0);
    }

    // ---- lane-local cell epilogue
    float* gn = a.gates ? a.gates + (int64_t)n * a.gates_sn : nullptr;
    float* cn = a.c_new + (int64_t)n * a.c_new_sn;
    float* hn_out = a.h_new + (int64_t)n * a.h_new_sn;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t row = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const bool ok = row < a.E;
            const float keep = 1.0f - a.done[rofs[r]];
            const float gi = sigm(acc[0 + jj][r]), gf = sigm(acc[2 + jj][r]);
            const float go = sigm(acc[4 + jj][r]), gu = tanh_fast(acc[6 + jj][r]);
            const float c = gf * (cp[jj][r] * keep) + gi * gu;
            const float hv = go * tanh_fast(c);
            if (ok) {
                const int j = jj * 32 + col;
                cn[row * H + j] = c;
                hn_out[row * H + j] = hv;
                if (gn) {
                    float* g = gn + row * G4 + j;
                    g[0] = gi; g[H] = gf; g[2 * H] = go; g[3 * H] = gu;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// v2: 16-row wave strips on v_mfma_f32_16x16x4_f32, two waves per SIMD, staggered halves.
//
// v1 above runs ONE wave per SIMD, so its phases (HBM loads ~11 us chip-wide, MFMA 8 us, epilogue 5.6 us) add up
// (26-28 us per call at E = 4096).  Here a 512-thread block = 8 waves x 16 rows (same 128 rows per block, same
// grid); waves 0-3 ("A") issue their input loads at once while waves 4-7 ("B") stage Wh into LDS; after the one
// block barrier B issues its loads.  Each SIMD hosts one A and one B wave: B's loads overlap A's MFMAs, and B's
// MFMAs overlap A's (VALU) epilogue.  Wh sits in LDS as [k][lane c][tile t] with pitch 20 floats per lane, so a
// lane fetches the B operands of all 16 column tiles of one k with four conflict-free ds_read_b128.
// C/D layout of 16x16x4: col = lane & 15, row = 4 (lane >> 4) + reg -> the four gates of unit j = 16 jj + col
// are acc[jj], acc[4 + jj], acc[8 + jj], acc[12 + jj] at the same reg: the cell stays lane-local.
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int R16 = 16;                 // rows per wave
constexpr int WAVES2 = 8;
constexpr int WPITCH = 16 * 20;         // floats per k row of the permuted W image
constexpr int LDS2_FLOATS = H * WPITCH + WAVES2 * R16 * APITCH;

// Head epilogue of one wave: its 16 fresh rows of h' sit in the wave's (now idle) LDS tile.  Lane
// (row = lane & 15, quarter = lane >> 4) accumulates the quarter's 16 k of every output, two xor-shuffles
// finish the 64-long dots; lanes 0..15 then own one row each: softmax + action draw (kind 1) or the
// critic's one-hot rows gathered from the neighbours' action bytes (kind 2).
template <int KIND>
__device__ __forceinline__ void head_epilogue(const FusedArgs& a, const int n, const int N, const int64_t row0,
                                              const int lane, const float* a_tile) {
    const nmarl_head_t& hd = a.hd;
    const int A = hd.A;
    const int nout = KIND == 1 ? A : 1;
    const float* w = KIND == 3 ? hd.w2 + (int64_t)n * hd.w2_sn : hd.w + (int64_t)n * hd.w_sn;
    const int rl = lane & 15, q = lane >> 4;
    float acc[MAXA];
#pragma unroll
    for (int o = 0; o < MAXA; ++o) acc[o] = 0.0f;
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
        const int k = q * 16 + kk;
        const float hk = a_tile[rl * APITCH + k];
        if (KIND == 1) {
#pragma unroll
            for (int o = 0; o < MAXA; ++o)
                if (o < nout) acc[o] += hk * w[k * nout + o];
        } else {
            acc[0] += hk * w[k];
        }
    }
#pragma unroll
    for (int o = 0; o < (KIND == 1 ? MAXA : 1); ++o) {
        acc[o] += __shfl_xor(acc[o], 16, 64);
        acc[o] += __shfl_xor(acc[o], 32, 64);
    }
    const int64_t row = row0 + rl;
    if (q != 0 || row >= a.E) return;
    const float* b = KIND == 3 ? hd.b2 + (int64_t)n * hd.b2_sn : hd.b + (int64_t)n * hd.b_sn;
    if (KIND == 1) {
        float p[MAXA];
        float m = -INFINITY;
#pragma unroll
        for (int o = 0; o < MAXA; ++o) {
            p[o] = o < A ? acc[o] + b[o] : -INFINITY;
            m = fmaxf(m, p[o]);
        }
        float ssum = 0.0f;
#pragma unroll
        for (int o = 0; o < MAXA; ++o) {
            p[o] = o < A ? expf(p[o] - m) : 0.0f;
            ssum += p[o];
        }
        float* po = hd.pi_out + (int64_t)n * hd.pi_sn + row * A;
#pragma unroll
        for (int o = 0; o < MAXA; ++o) {
            p[o] = p[o] / ssum;
            if (o < A) po[o] = p[o];
        }
        const int64_t step = hd.step + (hd.step_dev ? *hd.step_dev : 0);
        const float uh = hd.mode == 0 ? hd.u[row * N + n] : 0.0f;
        hd.act_out[row * N + n] = (uint8_t)nmarl_draw_action<MAXA>(p, A, hd.mode, uh, hd.seed, hd.env_id_base + row, n, step);
    } else {
        float v = acc[0] + b[0];
        if (KIND == 2)                      // kind 3 leaves the neighbour-action term to nmarl_nbr_action_value_fwd (the
            for (int k = 0; k < hd.m_max; ++k) {       // other agents' draws of this lock-step are made by other blocks)
                const int j = hd.nbr_idx[n * hd.m_max + k];
                if (j >= 0) v += w[H + k * A + (int)hd.act_in[row * N + j]];
            }
        hd.v_out[(int64_t)n * hd.v_sn + row] = v;
    }
}

template <bool HAS_Z2, int HEAD>
__global__ __launch_bounds__(512, 1) void lstm_step_mfma16_kernel(const FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* w_lds = lds;                                   // [64][16][20]
    const int n = blockIdx.x / a.blocks_per_agent;
    const int64_t row_blk = (int64_t)(blockIdx.x - n * a.blocks_per_agent) * ROWS_B;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = row_blk + wave * R16;
    const int c = lane & 15, grp = lane >> 4;
    float* a_tile = lds + H * WPITCH + wave * R16 * APITCH;
    const bool groupA = wave < 4;

    int64_t rofs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = row0 + 4 * grp + r;
        rofs[r] = row < a.E ? row : a.E - 1;
    }
    f32x4 acc[16];
    float cp[4][4];
    float4 hreg[4];
    float hkeep[4];
    const float* z1 = a.zadd1 + (int64_t)n * a.zadd1_sn;
    const float* cpn = a.c_prev + (int64_t)n * a.c_prev_sn;
    const float* hn = a.h_in + (int64_t)n * a.h_sn;

    auto issue_loads = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                      // this wave's 16 x 64 tile of h: 4 float4 per lane, coalesced
            const int v = i * 64 + lane;
            int64_t row = row0 + (v >> 4);
            row = row < a.E ? row : a.E - 1;
            hreg[i] = *reinterpret_cast<const float4*>(hn + row * H + (v & 15) * 4);
            hkeep[i] = 1.0f - a.done[row];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = z1[rofs[r] * G4 + t * 16 + c];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) cp[jj][r] = cpn[rofs[r] * H + jj * 16 + c];
    };

    if (groupA) {
        issue_loads();
    } else {
        // stage Wh (64 x 256) as [k][c][t]: thread -> 16 float4 of 4 consecutive columns (same tile, lanes c..c+3)
        const float* wg = a.wh + (int64_t)n * a.wh_sn;
        const int tid = threadIdx.x - 256;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int v = i * 256 + tid;                   // float4 index in the 64 x 64 float4 image
            const int k = v >> 6, col = (v & 63) * 4;
            const float4 x = *reinterpret_cast<const float4*>(wg + k * G4 + col);
            float* d = w_lds + k * WPITCH + (col & 15) * 20 + (col >> 4);
            d[0] = x.x; d[20] = x.y; d[40] = x.z; d[60] = x.w;
        }
    }
    __syncthreads();
    if (!groupA) issue_loads();

    // own h tile -> LDS (wave-private region), masked by 1 - done
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = i * 64 + lane;
        float* d = a_tile + (v >> 4) * APITCH + (v & 15) * 4;
        d[0] = hreg[i].x * hkeep[i]; d[1] = hreg[i].y * hkeep[i]; d[2] = hreg[i].z * hkeep[i]; d[3] = hreg[i].w * hkeep[i];
    }
    __builtin_amdgcn_wave_barrier();

    {   // + bias (+ zadd2)
        const float* bn = a.bias + (int64_t)n * a.bias_sn;
        const float* z2 = HAS_Z2 ? a.zadd2 + (int64_t)n * a.zadd2_sn : nullptr;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float b = bn[t * 16 + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[t][r] + b;
                if (HAS_Z2) v += z2[rofs[r] * G4 + t * 16 + c];
                acc[t][r] = v;
            }
        }
    }

    // HEAD 3 (forward 'p' + forward 'v' of one lock-step in one launch, uncoupled nets): the value re-step (quirk Q1)
    // starts from the state this step produces and adds the SAME x-side addend, so keep a copy of it
    f32x4 zs[HEAD == 3 ? 16 : 1];
    if (HEAD == 3) {
#pragma unroll
        for (int t = 0; t < 16; ++t) zs[t] = acc[t];
    }

    // K loop: 16 steps of K = 4;  A[i = lane & 15][k = lane >> 4],  B[k = lane >> 4][j = lane & 15]
#pragma unroll 2
    for (int kk = 0; kk < H / 4; ++kk) {
        const float av = a_tile[c * APITCH + 4 * kk + grp];
        const float4* wq = reinterpret_cast<const float4*>(w_lds + (4 * kk + grp) * WPITCH + c * 20);
        const float4 b0 = wq[0], b1 = wq[1], b2 = wq[2], b3 = wq[3];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.w, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.x, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.y, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.z, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.w, acc[7], 0, 0, 0);
        acc[8] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.x, acc[8], 0, 0, 0);
        acc[9] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.y, acc[9], 0, 0, 0);
        acc[10] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.z, acc[10], 0, 0, 0);
        acc[11] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.w, acc[11], 0, 0, 0);
        acc[12] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3.x, acc[12], 0, 0, 0);
        acc[13] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3.y, acc[13], 0, 0, 0);
        acc[14] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3.z, acc[14], 0, 0, 0);
        acc[15] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3.w, acc[15], 0, 0, 0);
    }

    float* gn = a.gates ? a.gates + (int64_t)n * a.gates_sn : nullptr;
    float* cn = a.c_new + (int64_t)n * a.c_new_sn;
    float* hn_out = a.h_new + (int64_t)n * a.h_new_sn;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = row0 + 4 * grp + r;
            const bool ok = row < a.E;
            const float keep = 1.0f - a.done[rofs[r]];
            const float gi = sigm(acc[0 + jj][r]), gf = sigm(acc[4 + jj][r]);
            const float go = sigm(acc[8 + jj][r]), gu = tanh_fast(acc[12 + jj][r]);
            const float cv = gf * (cp[jj][r] * keep) + gi * gu;
            const float hv = go * tanh_fast(cv);
            if (HEAD != 0) a_tile[(4 * grp + r) * APITCH + jj * 16 + c] = hv;      // K loop done: the tile is free
            if (HEAD == 3) cp[jj][r] = cv;                                         // c' = the re-step's previous cell
            if (ok) {
                const int j = jj * 16 + c;
                cn[row * H + j] = cv;
                hn_out[row * H + j] = hv;
                if (gn) {
                    float* g = gn + row * G4 + j;
                    g[0] = gi; g[H] = gf; g[2 * H] = go; g[3 * H] = gu;
                }
            }
        }
    }
    if (HEAD != 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        head_epilogue<(HEAD == 3 ? 1 : HEAD)>(a, n, (int)(gridDim.x / a.blocks_per_agent), row0, lane, a_tile);
    }
    if (HEAD == 3) {
        // ---- the value re-step: z = addend + (h' * keep) @ Wh, cell from c' * keep, critic on h'' (nothing stored but v)
        const float keepA = 1.0f - a.done[row0 + c < a.E ? row0 + c : a.E - 1];      // row of this lane's A operand
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = zs[t];
#pragma unroll 2
        for (int kk = 0; kk < H / 4; ++kk) {
            const float av = a_tile[c * APITCH + 4 * kk + grp] * keepA;
            const float4* wq = reinterpret_cast<const float4*>(w_lds + (4 * kk + grp) * WPITCH + c * 20);
            const float4 b0 = wq[0], b1 = wq[1], b2 = wq[2], b3 = wq[3];
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b0.w, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.x, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.y, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.z, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b1.w, acc[7], 0, 0, 0);
            acc[8] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.x, acc[8], 0, 0, 0);
            acc[9] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.y, acc[9], 0, 0, 0);
            acc[10] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.z, acc[10], 0, 0, 0);
            acc[11] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b2.w, acc[11], 0, 0, 0);
            acc[12] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3.x, acc[12], 0, 0, 0);
            acc[13] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3.y, acc[13], 0, 0, 0);
            acc[14] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3.z, acc[14], 0, 0, 0);
            acc[15] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b3.w, acc[15], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();                 // every lane has read its A operands: the tile may be overwritten
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float keep = 1.0f - a.done[rofs[r]];
                const float gi = sigm(acc[0 + jj][r]), gf = sigm(acc[4 + jj][r]);
                const float go = sigm(acc[8 + jj][r]), gu = tanh_fast(acc[12 + jj][r]);
                const float cv = gf * (cp[jj][r] * keep) + gi * gu;
                a_tile[(4 * grp + r) * APITCH + jj * 16 + c] = go * tanh_fast(cv);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        head_epilogue<3>(a, n, (int)(gridDim.x / a.blocks_per_agent), row0, lane, a_tile);
    }
}

inline bool stride_ok(int64_t s, int64_t need) { return s >= need && (s % 4) == 0; }

}  // namespace

static int launch_fused(int64_t E, int32_t N, int32_t Hh, const float* h_in, int64_t h_sn, const float* wh,
                        int64_t wh_sn, const float* bias, int64_t bias_sn, const float* zadd1, int64_t zadd1_sn,
                        const float* zadd2, int64_t zadd2_sn, const float* c_prev, int64_t c_prev_sn,
                        const float* done, float* gates, int64_t gates_sn, float* c_new, int64_t c_new_sn,
                        float* h_new, int64_t h_new_sn, const nmarl_head_t* head, void* stream) {
    if (Hh != H || E < 0 || N <= 0 || (E > 0 && (!h_in || !wh || !bias || !zadd1 || !c_prev || !done || !c_new || !h_new)))
        return NMARL_EINVAL;
    const int kind = head ? head->kind : 0;
    if (kind < 0 || kind > 3) return NMARL_EINVAL;
    if (kind != 0 && E > 0) {
        if (head->A <= 0 || head->A > MAXA || !head->w || !head->b || head->b_sn < (kind == 2 ? 1 : head->A)) return NMARL_EINVAL;
        if ((kind == 1 || kind == 3) && (head->w_sn < (int64_t)H * head->A || !head->pi_out || head->pi_sn < E * head->A || !head->act_out ||
                          head->mode < 0 || head->mode > 2 || (head->mode == 0 && !head->u)))
            return NMARL_EINVAL;
        if (kind == 3 && (!head->w2 || !head->b2 || head->w2_sn < H || head->b2_sn < 1 || !head->v_out || head->v_sn < E))
            return NMARL_EINVAL;
        if (kind == 2 && (head->m_max < 0 || head->w_sn < H + (int64_t)head->m_max * head->A || !head->v_out || head->v_sn < E ||
                          (head->m_max > 0 && (!head->act_in || !head->nbr_idx))))
            return NMARL_EINVAL;
    }
    if (E == 0) return NMARL_OK;
    if (!stride_ok(h_sn, E * H) || !stride_ok(wh_sn, H * G4) || !stride_ok(bias_sn, G4) || !stride_ok(zadd1_sn, E * G4) ||
        (zadd2 && !stride_ok(zadd2_sn, E * G4)) || !stride_ok(c_prev_sn, E * H) || !stride_ok(c_new_sn, E * H) ||
        !stride_ok(h_new_sn, E * H) || (gates && !stride_ok(gates_sn, E * G4)) || ((uintptr_t)wh % 16) || ((uintptr_t)h_in % 16))
        return NMARL_EINVAL;
    FusedArgs a{};
    a.h_in = h_in; a.wh = wh; a.bias = bias; a.zadd1 = zadd1; a.zadd2 = zadd2; a.c_prev = c_prev; a.done = done;
    a.gates = gates; a.c_new = c_new; a.h_new = h_new;
    a.h_sn = h_sn; a.wh_sn = wh_sn; a.bias_sn = bias_sn; a.zadd1_sn = zadd1_sn; a.zadd2_sn = zadd2_sn;
    a.c_prev_sn = c_prev_sn; a.gates_sn = gates_sn; a.c_new_sn = c_new_sn; a.h_new_sn = h_new_sn;
    a.E = E;
    a.blocks_per_agent = (int)((E + ROWS_B - 1) / ROWS_B);
    if (kind != 0) a.hd = *head;
    static int variant = -1;        // NMARL_FUSED_VARIANT=1 selects the one-wave-per-SIMD 32x32x2 kernel (A/B comparisons)
    if (variant < 0) {
        const char* ev = getenv("NMARL_FUSED_VARIANT");
        const int l1 = (int)((H * G4 + WAVES * ROWS_W * APITCH) * sizeof(float)), l2 = (int)(LDS2_FLOATS * sizeof(float));
#define NMARL_SET_LDS(k, bytes) \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return NMARL_EHIP;
        NMARL_SET_LDS(lstm_step_mfma_kernel<false>, l1) NMARL_SET_LDS(lstm_step_mfma_kernel<true>, l1)
        NMARL_SET_LDS((lstm_step_mfma16_kernel<false, 0>), l2) NMARL_SET_LDS((lstm_step_mfma16_kernel<true, 0>), l2)
        NMARL_SET_LDS((lstm_step_mfma16_kernel<false, 1>), l2) NMARL_SET_LDS((lstm_step_mfma16_kernel<true, 1>), l2)
        NMARL_SET_LDS((lstm_step_mfma16_kernel<false, 2>), l2) NMARL_SET_LDS((lstm_step_mfma16_kernel<true, 2>), l2)
        NMARL_SET_LDS((lstm_step_mfma16_kernel<false, 3>), l2) NMARL_SET_LDS((lstm_step_mfma16_kernel<true, 3>), l2)
#undef NMARL_SET_LDS
        variant = (ev && ev[0] == '1') ? 1 : 2;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(a.blocks_per_agent * N);
    if (variant == 1 && kind == 0) {
        const size_t lds_bytes = (size_t)(H * G4 + WAVES * ROWS_W * APITCH) * sizeof(float);
        if (zadd2) hipLaunchKernelGGL(lstm_step_mfma_kernel<true>, grid, dim3(256), lds_bytes, st, a);
        else hipLaunchKernelGGL(lstm_step_mfma_kernel<false>, grid, dim3(256), lds_bytes, st, a);
    } else {
        const size_t lds_bytes = (size_t)LDS2_FLOATS * sizeof(float);
#define NMARL_LAUNCH16(Z2, HD) hipLaunchKernelGGL((lstm_step_mfma16_kernel<Z2, HD>), grid, dim3(512), lds_bytes, st, a)
        if (zadd2) { if (kind == 0) NMARL_LAUNCH16(true, 0); else if (kind == 1) NMARL_LAUNCH16(true, 1); else if (kind == 2) NMARL_LAUNCH16(true, 2); else NMARL_LAUNCH16(true, 3); }
        else       { if (kind == 0) NMARL_LAUNCH16(false, 0); else if (kind == 1) NMARL_LAUNCH16(false, 1); else if (kind == 2) NMARL_LAUNCH16(false, 2); else NMARL_LAUNCH16(false, 3); }
#undef NMARL_LAUNCH16
    }
    return nmarl_check_launch();
}

extern "C" int nmarl_lstm_step_fused(int64_t E, int32_t N, int32_t Hh, const float* h_in, int64_t h_sn,
                                     const float* wh, int64_t wh_sn, const float* bias, int64_t bias_sn,
                                     const float* zadd1, int64_t zadd1_sn, const float* zadd2, int64_t zadd2_sn,
                                     const float* c_prev, int64_t c_prev_sn, const float* done, float* gates,
                                     int64_t gates_sn, float* c_new, int64_t c_new_sn, float* h_new,
                                     int64_t h_new_sn, void* stream) {
    return launch_fused(E, N, Hh, h_in, h_sn, wh, wh_sn, bias, bias_sn, zadd1, zadd1_sn, zadd2, zadd2_sn, c_prev, c_prev_sn,
                        done, gates, gates_sn, c_new, c_new_sn, h_new, h_new_sn, nullptr, stream);
}

extern "C" int nmarl_lstm_step_fused_head(int64_t E, int32_t N, int32_t Hh, const float* h_in, int64_t h_sn,
                                          const float* wh, int64_t wh_sn, const float* bias, int64_t bias_sn,
                                          const float* zadd1, int64_t zadd1_sn, const float* zadd2, int64_t zadd2_sn,
                                          const float* c_prev, int64_t c_prev_sn, const float* done, float* gates,
                                          int64_t gates_sn, float* c_new, int64_t c_new_sn, float* h_new,
                                          int64_t h_new_sn, const nmarl_head_t* head, void* stream) {
    return launch_fused(E, N, Hh, h_in, h_sn, wh, wh_sn, bias, bias_sn, zadd1, zadd1_sn, zadd2, zadd2_sn, c_prev, c_prev_sn,
                        done, gates, gates_sn, c_new, c_new_sn, h_new, h_new_sn, head, stream);
}
