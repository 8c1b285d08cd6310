// Per-agent fully connected encoder layers with a SMALL input width (F <= 64) and 64 outputs:
//
//     y[n, r, :] = act(x[n, r, :F] @ w[n] + b[n])            fc of agents/utils.py:65-73
//
// i.e. `fc(ob, 'fcs' / 'fcp' / 'fc', n_fc, relu)` of LstmPolicy / FPPolicy (policies.py:145, 177-180) and the
// w_ob / w_fp encoders of lstm_comm / lstm_ic3 / lstm_dial (agents/utils.py:186-198, 395-400, 566-575) on the
// observation slab (F = 15 CACC, 60 grid) and the gathered neighbour fingerprints (F = 8 / 20), for all
// agents and rows (rows = E in the rollout, T*E in the update) in one launch.
//
// K = F is far too small for the matrix cores to matter (the GEMM library picks 16x16 tiles and runs the
// wgrad x^T dy at a few % of anything), and the layer is HBM-bound on its 64-wide output anyway: these
// kernels stream y (and dy) once at full width and keep everything else on chip.
//   fwd: thread = (output column j, row lane); w[:, j] lives in registers, a 64-row tile of x is staged in LDS and
//        read back as wave-wide broadcasts (float4); bias + activation fused; y may be a column block of a
//        wider buffer (row pitch y_row), which is how tf.concat([hx, hp]) (policies.py:181) costs no copy.
//   bwd: same mapping; g = dy * act'(y) is formed in registers (never stored), dW[f, j] += x[r, f] * g and
//        db[j] += g accumulate in registers over the block's rows; blocks write partial sums that a second
//        kernel adds in a fixed order (deterministic: no atomics).  dx is not produced: x is data.
// HBM bytes per row: fwd 4 (F + 64), bwd 4 (F + 128).
#include "common.h"

namespace {

constexpr int J = 64;            // outputs per layer (n_fc / n_h of every shipped config)
constexpr int TILE = 64;         // rows staged per LDS tile
// row-loop unroll by input width: the compiler hoists the LDS reads of all unrolled rows (registers ~ unroll * FMAX)
template <int FM> struct RowUnroll { static constexpr int value = FM <= 16 ? 4 : FM <= 32 ? 2 : 1; };

__device__ __forceinline__ float act_fwd(const float v, const int act) {
    return act == 1 ? fmaxf(v, 0.0f) : act == 2 ? tanhf(v) : v;
}

// d act / d pre-activation expressed through the OUTPUT y (relu: y > 0; tanh: 1 - y^2)
__device__ __forceinline__ float act_bwd(const float g, const float y, const int act) {
    return act == 1 ? (y > 0.0f ? g : 0.0f) : act == 2 ? g * (1.0f - y * y) : g;
}

// optional neighbour gather of a layer's input: feature f of agent n = x[nbr_idx[n, f / A]][row][f % A] (zero for a
// -1 slot), i.e. tf.boolean_mask + concat of policies.py:171-174 / agents/utils.py:186-195 without materialising it
struct XGather { const int32_t* nbr_idx; int A, m_max; };

__device__ __forceinline__ float x_elem(const float* __restrict__ x, const int64_t x_sn, const int64_t x_row, const int n, const int64_t row,
                                        const int f, const bool ok, const XGather g) {
    // unconditional load: clamped indices, the caller zeroes the value when !ok
    if (g.nbr_idx == nullptr) return x[(int64_t)n * x_sn + (ok ? row : 0) * x_row + (ok ? f : 0)];
    const int k = (ok ? f : 0) / g.A;
    const int src = g.nbr_idx[n * g.m_max + k];
    const bool ok2 = ok && src >= 0;
    const float v = x[(int64_t)(ok2 ? src : 0) * x_sn + (ok2 ? row : 0) * x_row + (ok2 ? f - k * g.A : 0)];
    return ok2 ? v : 0.0f;
}

template <int FMAX>
__device__ __forceinline__ void stage_tile(float* xs, const float* __restrict__ xn, const int64_t x_row, const int64_t row0,
                                           const int64_t rows, const int F) {
    constexpr int FP = FMAX + 4;
    for (int idx = threadIdx.x; idx < TILE * FMAX; idx += 256) {
        const int r = idx / FMAX, f = idx - r * FMAX;
        const int64_t row = row0 + r;
        xs[r * FP + f] = (f < F && row < rows) ? xn[row * x_row + f] : 0.0f;
    }
}

// Gathered input x~[n,r,k*A+a] = x[nbr(n,k),r,a] with A % 4 == 0 and 16-byte aligned rows (the grid's compact observation:
// A = 12): staged as float4 units through a per-block table of source offsets -- the element-wise form divides by A and
// re-reads the neighbour table for every one of the tile's 4096 elements.
template <int FMAX>
__device__ __forceinline__ void gather_table(int64_t* gsrc, const int32_t* __restrict__ nbr_idx, const int n, const int m_max,
                                             const int A, const int F, const int64_t x_sn) {
    if (threadIdx.x < FMAX / 4) {
        const int f = 4 * threadIdx.x;
        int64_t o = -1;
        if (f < F) {
            const int k = f / A;
            const int src = nbr_idx[n * m_max + k];
            if (src >= 0) o = (int64_t)src * x_sn + (f - k * A);
        }
        gsrc[threadIdx.x] = o;
    }
    __syncthreads();
}
template <int FMAX>
__device__ __forceinline__ void stage_tile_gather4(float* xs, const float* __restrict__ x, const int64_t x_row, const int64_t row0,
                                                   const int64_t rows, const int64_t* gsrc) {
    constexpr int FP = FMAX + 4, Q = FMAX / 4;
#pragma unroll
    for (int m = 0; m < TILE * Q / 256; ++m) {             // unconditional loads: clamped source, value selected afterwards
        const int idx = threadIdx.x + 256 * m;
        const int r = idx / Q, f4 = idx - r * Q;
        const int64_t row = row0 + r, o = gsrc[f4];
        const bool ok = o >= 0 && row < rows;
        const float4 v = *reinterpret_cast<const float4*>(x + (ok ? o : 0) + (ok ? row : 0) * x_row);
        *reinterpret_cast<float4*>(xs + r * FP + 4 * f4) = ok ? v : float4{0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ bool gather4_ok(const float* x, const int64_t x_sn, const int64_t x_row, const int A) {
    return (A & 3) == 0 && (x_sn & 3) == 0 && (x_row & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
}

template <int FMAX>
__global__ __launch_bounds__(256) void fc_fwd_kernel(const int64_t rows, const int F, const int tiles_per_block,
                                                     const float* __restrict__ x, const int64_t x_sn, const int64_t x_row,
                                                     const float* __restrict__ w, const int64_t w_sn,
                                                     const float* __restrict__ b, const int64_t b_sn, const int act,
                                                     float* __restrict__ y, const int64_t y_sn, const int64_t y_row) {
    constexpr int FP = FMAX + 4;
    constexpr int RU = RowUnroll<FMAX>::value;
    __shared__ __attribute__((aligned(16))) float xs[TILE * FP];
    const int n = blockIdx.y, j = threadIdx.x & 63, rl = threadIdx.x >> 6;
    float wr[FMAX];
#pragma unroll
    for (int f = 0; f < FMAX; ++f) wr[f] = f < F ? w[(int64_t)n * w_sn + f * J + j] : 0.0f;
    const float bj = b[(int64_t)n * b_sn + j];
    const float* xn = x + (int64_t)n * x_sn;
    float* yn = y + (int64_t)n * y_sn;
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * TILE;
        if (row0 >= rows) break;
        stage_tile<FMAX>(xs, xn, x_row, row0, rows, F);
        __syncthreads();
#pragma unroll RU
        for (int rr = rl; rr < TILE; rr += 4) {
            float acc = 0.0f;
#pragma unroll
            for (int f4 = 0; f4 < FMAX / 4; ++f4) {
                const float4 xv = *reinterpret_cast<const float4*>(xs + rr * FP + 4 * f4);
                acc = fmaf(xv.x, wr[4 * f4 + 0], acc);
                acc = fmaf(xv.y, wr[4 * f4 + 1], acc);
                acc = fmaf(xv.z, wr[4 * f4 + 2], acc);
                acc = fmaf(xv.w, wr[4 * f4 + 3], acc);
            }
            const int64_t row = row0 + rr;
            if (row < rows) yn[row * y_row + j] = act_fwd(acc + bj, act);
        }
        __syncthreads();
    }
}

// Several encoder layers of one lock-step in ONE launch (blockIdx.z = layer): layer p writes columns
// [64 p, 64 p + 64) of the concatenated encoding.  A layer may take its input through the neighbour table:
// x~[n, r, k*A + a] = x[nbr[n,k], r, a] (0 where padded) -- the fingerprint gather of FPPolicy / lstm_comm
// (policies.py:171-179, agents/utils.py:188-193) folded into the tile staging, so that the whole h-independent
// encoding of the rollout is one kernel before the s @ Wx GEMM.
struct FcParts { nmarl_fc_part_t p[NMARL_FC_MAX_PARTS]; };

template <int FMAX, bool VEC = false>
__global__ __launch_bounds__(256) void fc_fwd_multi_kernel(const int64_t rows, const int tiles_per_block, const FcParts parts,
                                                           const int act, float* __restrict__ y, const int64_t y_sn,
                                                           const int64_t y_row) {
    constexpr int FP = FMAX + 4;
    constexpr int RU = RowUnroll<FMAX>::value;
    __shared__ __attribute__((aligned(16))) float xs[TILE * FP];
    const nmarl_fc_part_t& pt = parts.p[blockIdx.z];
    const int F = pt.F;
    const int n = blockIdx.y, j = threadIdx.x & 63, rl = threadIdx.x >> 6;
    float* yn = y + (int64_t)n * y_sn + blockIdx.z * J;
    if (FMAX <= 16 && VEC) {
        // narrow inputs (the rollout's encoders): thread = (4 consecutive columns, row lane of 16), outputs leave as
        // float4; per output the same ascending fmaf chain over the inputs as below (identical results)
        const int j4 = (threadIdx.x & 15) * 4, r16 = threadIdx.x >> 4;
        float4 w4[FMAX];
#pragma unroll
        for (int f = 0; f < FMAX; ++f) {
            const float4 v = *reinterpret_cast<const float4*>(pt.w + (int64_t)n * pt.w_sn + (f < F ? f : 0) * J + j4);
            const float m = f < F ? 1.0f : 0.0f;
            w4[f] = float4{v.x * m, v.y * m, v.z * m, v.w * m};
        }
        const float4 b4 = *reinterpret_cast<const float4*>(pt.b + (int64_t)n * pt.b_sn + j4);
        for (int tile = 0; tile < tiles_per_block; ++tile) {
            const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * TILE;
            if (row0 >= rows) break;
            if (pt.nbr_idx == nullptr) {
                stage_tile<FMAX>(xs, pt.x + (int64_t)n * pt.x_sn, pt.x_row, row0, rows, F);
            } else {
                const int A = pt.gather_A;
#pragma unroll
                for (int m = 0; m < TILE * FMAX / 256; ++m) {      // unconditional loads: clamped source, zero weight
                    const int idx = threadIdx.x + 256 * m;
                    const int r = idx / FMAX, f = idx - r * FMAX;
                    const int64_t row = row0 + r;
                    const int k = (f < F ? f : 0) / A;
                    const int src = pt.nbr_idx[n * pt.m_max + k];
                    const bool ok = f < F && row < rows && src >= 0;
                    const float v = pt.x[(int64_t)(ok ? src : 0) * pt.x_sn + (ok ? row : 0) * pt.x_row + (ok ? f - k * A : 0)];
                    xs[r * FP + f] = ok ? v : 0.0f;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rr = r16 + 16 * i;
                float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
                for (int f4 = 0; f4 < FMAX / 4; ++f4) {
                    const float4 xv = *reinterpret_cast<const float4*>(xs + rr * FP + 4 * f4);
                    const float xq[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a0 = fmaf(xq[u], w4[4 * f4 + u].x, a0);
                        a1 = fmaf(xq[u], w4[4 * f4 + u].y, a1);
                        a2 = fmaf(xq[u], w4[4 * f4 + u].z, a2);
                        a3 = fmaf(xq[u], w4[4 * f4 + u].w, a3);
                    }
                }
                const int64_t row = row0 + rr;
                if (row < rows)
                    *reinterpret_cast<float4*>(yn + row * y_row + j4) =
                        float4{act_fwd(a0 + b4.x, act), act_fwd(a1 + b4.y, act), act_fwd(a2 + b4.z, act), act_fwd(a3 + b4.w, act)};
            }
            __syncthreads();
        }
        return;
    }
    float wr[FMAX];
#pragma unroll
    for (int f = 0; f < FMAX; ++f) wr[f] = f < F ? pt.w[(int64_t)n * pt.w_sn + f * J + j] : 0.0f;
    const float bj = pt.b[(int64_t)n * pt.b_sn + j];
    __shared__ int64_t gsrc[FMAX / 4];
    const bool g4 = pt.nbr_idx != nullptr && gather4_ok(pt.x, pt.x_sn, pt.x_row, pt.gather_A);
    if (g4) gather_table<FMAX>(gsrc, pt.nbr_idx, n, pt.m_max, pt.gather_A, F, pt.x_sn);
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * TILE;
        if (row0 >= rows) break;
        if (pt.nbr_idx == nullptr) {
            stage_tile<FMAX>(xs, pt.x + (int64_t)n * pt.x_sn, pt.x_row, row0, rows, F);
        } else if (g4) {
            stage_tile_gather4<FMAX>(xs, pt.x, pt.x_row, row0, rows, gsrc);
        } else {
            const int A = pt.gather_A;
            for (int idx = threadIdx.x; idx < TILE * FMAX; idx += 256) {
                const int r = idx / FMAX, f = idx - r * FMAX;
                const int64_t row = row0 + r;
                float v = 0.0f;
                if (f < F && row < rows) {
                    const int k = f / A;
                    const int src = pt.nbr_idx[n * pt.m_max + k];
                    if (src >= 0) v = pt.x[(int64_t)src * pt.x_sn + row * pt.x_row + (f - k * A)];
                }
                xs[r * FP + f] = v;
            }
        }
        __syncthreads();
#pragma unroll RU
        for (int rr = rl; rr < TILE; rr += 4) {
            float acc = 0.0f;
#pragma unroll
            for (int f4 = 0; f4 < FMAX / 4; ++f4) {
                const float4 xv = *reinterpret_cast<const float4*>(xs + rr * FP + 4 * f4);
                acc = fmaf(xv.x, wr[4 * f4 + 0], acc);
                acc = fmaf(xv.y, wr[4 * f4 + 1], acc);
                acc = fmaf(xv.z, wr[4 * f4 + 2], acc);
                acc = fmaf(xv.w, wr[4 * f4 + 3], acc);
            }
            const int64_t row = row0 + rr;
            if (row < rows) yn[row * y_row + j] = act_fwd(acc + bj, act);
        }
        __syncthreads();
    }
}

// partial: [N, gridDim.x, F + 1, 64] (rows 0..F-1: dW, row F: db)
// Thread = (4 consecutive output columns j4, row lane rl of 16): dy / y are read as float4 (16 threads cover a row's
// 256 bytes), all four row passes of a 64-row tile are in flight together, rows past the end are clamped loads with a
// zero weight (no branch around a load).  Bound by the two streams dy and y (2 x rows x 256 B per agent).
template <int FMAX>
__global__ __launch_bounds__(256) void fc_bwd_kernel(const int64_t rows, const int F, const int tiles_per_block,
                                                     const float* __restrict__ x, const int64_t x_sn, const int64_t x_row,
                                                     const float* __restrict__ y, const int64_t y_sn, const int64_t y_row,
                                                     const float* __restrict__ dy, const int64_t dy_sn, const int64_t dy_row,
                                                     const int act, float* __restrict__ partial, const XGather xg) {
    constexpr int FP = FMAX + 4;
    __shared__ __attribute__((aligned(16))) float xs[(FMAX + 1) * J > TILE * FP ? (FMAX + 1) * J : TILE * FP];   // x tile, later the reduction pad
    const int n = blockIdx.y, j4 = (threadIdx.x & 15) * 4, rl = threadIdx.x >> 4;
    float acc[FMAX][4];
#pragma unroll
    for (int f = 0; f < FMAX; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[f][q] = 0.0f;
    float db[4] = {0.f, 0.f, 0.f, 0.f};
    const float* yn = y + (int64_t)n * y_sn + j4;
    const float* dyn = dy + (int64_t)n * dy_sn + j4;
    float4 gy[4], gd[4];
    float okf[4];
    // loads of a tile (clamped rows, zero weight past the end / past this block's tiles): issued one tile ahead
#define NMARL_FCB_LOAD(tile_)                                                              \
    {                                                                                      \
        const int64_t r0_ = ((int64_t)blockIdx.x * tiles_per_block + (tile_)) * TILE;      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
            const int64_t row = r0_ + rl + 16 * i;                                         \
            const bool ok = row < rows && (tile_) < tiles_per_block;                       \
            const int64_t rc = ok ? row : rows - 1;                                        \
            okf[i] = ok ? 1.0f : 0.0f;                                                     \
            gd[i] = *reinterpret_cast<const float4*>(dyn + rc * dy_row);                   \
            gy[i] = *reinterpret_cast<const float4*>(yn + rc * y_row);                     \
        }                                                                                  \
    }
    // the x tile of the NEXT tile waits in registers as well (TILE * FMAX / 256 values per thread), so that no global
    // latency is exposed between two tiles
    constexpr int XR = TILE * FMAX / 256;
    float xr[XR];
    // a thread stages the same features of every tile: their source (agent, offset) through the neighbour table is
    // looked up once, so that the per-tile loads are independent (no table load -> input load chain per tile)
    int64_t xoff[XR];
    bool xok[XR];
#pragma unroll
    for (int m = 0; m < XR; ++m) {
        const int f = (threadIdx.x + 256 * m) % FMAX;
        xok[m] = f < F;
        xoff[m] = (int64_t)n * x_sn + (xok[m] ? f : 0);
        if (xg.nbr_idx != nullptr) {
            const int k = (xok[m] ? f : 0) / xg.A;
            const int src = xg.nbr_idx[n * xg.m_max + k];
            xok[m] = xok[m] && src >= 0;
            xoff[m] = (int64_t)(src >= 0 ? src : 0) * x_sn + (xok[m] ? f - k * xg.A : 0);
        }
    }
#define NMARL_FCB_XLOAD(tile_)                                                             \
    {                                                                                      \
        const int64_t r0_ = ((int64_t)blockIdx.x * tiles_per_block + (tile_)) * TILE;      \
        _Pragma("unroll") for (int m = 0; m < XR; ++m) {                                   \
            const int64_t row = r0_ + (threadIdx.x + 256 * m) / FMAX;                      \
            const bool ok = xok[m] && row < rows;                                          \
            xr[m] = x[xoff[m] + (ok ? row : 0) * x_row] * (ok ? 1.0f : 0.0f);              \
        }                                                                                  \
    }
    NMARL_FCB_LOAD(0)
    NMARL_FCB_XLOAD(0)
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * TILE;
        if (row0 >= rows) break;
#pragma unroll
        for (int m = 0; m < XR; ++m) {
            const int idx = threadIdx.x + 256 * m;
            const int r = idx / FMAX;
            xs[r * FP + (idx - r * FMAX)] = xr[m];
        }
        NMARL_FCB_XLOAD(tile + 1)
        float g[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g[i][0] = act_bwd(gd[i].x, gy[i].x, act) * okf[i];
            g[i][1] = act_bwd(gd[i].y, gy[i].y, act) * okf[i];
            g[i][2] = act_bwd(gd[i].z, gy[i].z, act) * okf[i];
            g[i][3] = act_bwd(gd[i].w, gy[i].w, act) * okf[i];
        }
        NMARL_FCB_LOAD(tile + 1)                 // in flight during this tile's FMAs
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = rl + 16 * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) db[q] += g[i][q];
#pragma unroll
            for (int f4 = 0; f4 < FMAX / 4; ++f4) {
                const float4 xv = *reinterpret_cast<const float4*>(xs + rr * FP + 4 * f4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[4 * f4 + 0][q] = fmaf(xv.x, g[i][q], acc[4 * f4 + 0][q]);
                    acc[4 * f4 + 1][q] = fmaf(xv.y, g[i][q], acc[4 * f4 + 1][q]);
                    acc[4 * f4 + 2][q] = fmaf(xv.z, g[i][q], acc[4 * f4 + 2][q]);
                    acc[4 * f4 + 3][q] = fmaf(xv.w, g[i][q], acc[4 * f4 + 3][q]);
                }
            }
        }
        __syncthreads();
    }
#undef NMARL_FCB_LOAD
#undef NMARL_FCB_XLOAD
    // the 16 row lanes add up in a fixed order: 4 inside the wave (lanes 16 i + j), then the 4 waves through LDS
#pragma unroll
    for (int f = 0; f < FMAX; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = acc[f][q];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[f][q] = v;
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v = db[q];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        db[q] = v;
    }
    float* red = xs;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int k = 0; k < 4; ++k) {
        if (wave == k && lane < 16) {
#pragma unroll
            for (int f = 0; f < FMAX; ++f)
                if (f < F) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) red[f * J + j4 + q] = (k == 0 ? 0.0f : red[f * J + j4 + q]) + acc[f][q];
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) red[FMAX * J + j4 + q] = (k == 0 ? 0.0f : red[FMAX * J + j4 + q]) + db[q];
        }
        __syncthreads();
    }
    float* out = partial + ((int64_t)n * gridDim.x + blockIdx.x) * (int64_t)(F + 1) * J;
    for (int idx = threadIdx.x; idx < (F + 1) * J; idx += 256) {
        const int f = idx >> 6;
        out[idx] = red[(f < F ? f : FMAX) * J + (idx & 63)];
    }
}

// BOTH layers of a two-part encoding [act(x_0 w_0 + b_0) | act(x_1 w_1 + b_1)] (policies.py:176-181) in ONE pass over the
// 128-column dy: 512 threads = (4 consecutive columns of 128, row lane of 16), so that a thread owns exactly the rows and
// columns a thread of fc_bwd_kernel<16> owns in the two-launch form and adds them in the same order; the block's reduction
// pairs the waves the same way (below) -- the results are bit-identical to two fc_bwd launches.  BITS: the relu derivative
// comes from a bit image the producer of y wrote (16 bytes per row instead of the 512-byte y row): word q of a row, bit
// 4 t + i  <=>  y[row, 16 t + 4 q + i] > 0  (the C/D register layout of the lock-step kernel's encoder pre-phase).
// partial: [N, gridDim.x, 2, 17, 64] (rows 0..15 dW of the part, zero past its F; row 16 db).
struct FcPair { nmarl_fc_part_t p[2]; };

template <bool BITS>
__global__ __launch_bounds__(512) void fc_bwd_pair_kernel(const int64_t rows, const int tiles_per_block, const FcPair parts,
                                                          const float* __restrict__ y, const int64_t y_sn, const int64_t y_row,
                                                          const uint32_t* __restrict__ bits, const int64_t bits_sn,
                                                          const float* __restrict__ dy, const int64_t dy_sn, const int64_t dy_row,
                                                          const int act, float* __restrict__ partial) {
    constexpr int FMAX = 16, FP = FMAX + 4;
    __shared__ __attribute__((aligned(16))) float xs[4 * (FMAX + 1) * J];      // x tiles [2][64][FP]; later the reduction pads: result [2][17][64], scratch [2][17][64]
    static_assert(2 * TILE * FP <= 4 * (FMAX + 1) * J, "x tiles fit");
    const int n = blockIdx.y, ct = threadIdx.x & 31, j4 = ct * 4, rl = threadIdx.x >> 5, half = ct >> 4;
    float acc[FMAX][4];
#pragma unroll
    for (int f = 0; f < FMAX; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[f][q] = 0.0f;
    float db[4] = {0.f, 0.f, 0.f, 0.f};
    const float* yn = BITS ? nullptr : y + (int64_t)n * y_sn + j4;
    const uint32_t* bn = BITS ? bits + (int64_t)n * bits_sn + (ct & 3) : nullptr;
    const int bshift = 4 * (ct >> 2);
    const float* dyn = dy + (int64_t)n * dy_sn + j4;
    float4 gy[4], gd[4];
    uint32_t gb[4];
    float okf[4];
#define NMARL_FCP_LOAD(tile_)                                                             \
    {                                                                                      \
        const int64_t r0_ = ((int64_t)blockIdx.x * tiles_per_block + (tile_)) * TILE;      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
            const int64_t row = r0_ + rl + 16 * i;                                         \
            const bool ok = row < rows && (tile_) < tiles_per_block;                       \
            const int64_t rc = ok ? row : rows - 1;                                        \
            okf[i] = ok ? 1.0f : 0.0f;                                                     \
            gd[i] = *reinterpret_cast<const float4*>(dyn + rc * dy_row);                   \
            if (BITS) gb[i] = bn[rc * 4];                                                  \
            else gy[i] = *reinterpret_cast<const float4*>(yn + rc * y_row);                \
        }                                                                                  \
    }
    // x tiles of both parts: 2 x 64 x 16 values, four per thread (m = 0, 1: part 0; m = 2, 3: part 1), the next tile's in registers
    constexpr int XR = 2 * TILE * FMAX / 512;
    float xr[XR];
    int64_t xoff[XR], xrow[XR];
    bool xok[XR];
#pragma unroll
    for (int m = 0; m < XR; ++m) {
        const nmarl_fc_part_t& pt = parts.p[m >> 1];
        const int f = threadIdx.x % FMAX;
        xok[m] = f < pt.F;
        xrow[m] = pt.x_row;
        xoff[m] = (int64_t)n * pt.x_sn + (xok[m] ? f : 0);
        if (pt.nbr_idx != nullptr) {
            const int k = (xok[m] ? f : 0) / pt.gather_A;
            const int src = pt.nbr_idx[n * pt.m_max + k];
            xok[m] = xok[m] && src >= 0;
            xoff[m] = (int64_t)(src >= 0 ? src : 0) * pt.x_sn + (xok[m] ? f - k * pt.gather_A : 0);
        }
    }
#define NMARL_FCP_XLOAD(tile_)                                                             \
    {                                                                                      \
        const int64_t r0_ = ((int64_t)blockIdx.x * tiles_per_block + (tile_)) * TILE;      \
        _Pragma("unroll") for (int m = 0; m < XR; ++m) {                                   \
            const int64_t row = r0_ + (threadIdx.x + 512 * (m & 1)) / FMAX;                \
            const bool ok = xok[m] && row < rows;                                          \
            xr[m] = parts.p[m >> 1].x[xoff[m] + (ok ? row : 0) * xrow[m]] * (ok ? 1.0f : 0.0f); \
        }                                                                                  \
    }
    NMARL_FCP_LOAD(0)
    NMARL_FCP_XLOAD(0)
    const float* xh = xs + half * TILE * FP;
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * TILE;
        if (row0 >= rows) break;
#pragma unroll
        for (int m = 0; m < XR; ++m) {
            const int r = (threadIdx.x + 512 * (m & 1)) / FMAX;
            xs[(m >> 1) * TILE * FP + r * FP + threadIdx.x % FMAX] = xr[m];
        }
        NMARL_FCP_XLOAD(tile + 1)
        float g[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (BITS) {
                const uint32_t nb = gb[i] >> bshift;
                g[i][0] = ((nb & 1u) ? gd[i].x : 0.0f) * okf[i];
                g[i][1] = ((nb & 2u) ? gd[i].y : 0.0f) * okf[i];
                g[i][2] = ((nb & 4u) ? gd[i].z : 0.0f) * okf[i];
                g[i][3] = ((nb & 8u) ? gd[i].w : 0.0f) * okf[i];
            } else {
                g[i][0] = act_bwd(gd[i].x, gy[i].x, act) * okf[i];
                g[i][1] = act_bwd(gd[i].y, gy[i].y, act) * okf[i];
                g[i][2] = act_bwd(gd[i].z, gy[i].z, act) * okf[i];
                g[i][3] = act_bwd(gd[i].w, gy[i].w, act) * okf[i];
            }
        }
        NMARL_FCP_LOAD(tile + 1)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = rl + 16 * i;
#pragma unroll
            for (int q = 0; q < 4; ++q) db[q] += g[i][q];
#pragma unroll
            for (int f4 = 0; f4 < FMAX / 4; ++f4) {
                const float4 xv = *reinterpret_cast<const float4*>(xh + rr * FP + 4 * f4);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[4 * f4 + 0][q] = fmaf(xv.x, g[i][q], acc[4 * f4 + 0][q]);
                    acc[4 * f4 + 1][q] = fmaf(xv.y, g[i][q], acc[4 * f4 + 1][q]);
                    acc[4 * f4 + 2][q] = fmaf(xv.z, g[i][q], acc[4 * f4 + 2][q]);
                    acc[4 * f4 + 3][q] = fmaf(xv.w, g[i][q], acc[4 * f4 + 3][q]);
                }
            }
        }
        __syncthreads();
    }
#undef NMARL_FCP_LOAD
#undef NMARL_FCP_XLOAD
    // fc_bwd_kernel's order: row lanes (4 k, 4 k + 1) and (4 k + 2, 4 k + 3) pairwise, the two pairs, then k = 0..3 in turn.
    // Here a wave holds two row lanes (lanes 32 apart): wave 2 k = the first pair, wave 2 k + 1 = the second.
#pragma unroll
    for (int f = 0; f < FMAX; ++f)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[f][q] += __shfl_xor(acc[f][q], 32, 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) db[q] += __shfl_xor(db[q], 32, 64);
    float* red = xs + half * (FMAX + 1) * J;
    float* tmp = xs + (2 + half) * (FMAX + 1) * J;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, jj = j4 & 63;
    for (int k = 0; k < 4; ++k) {
        // second pair first into LDS, then (first + second) added to the running sum
        if (wave == 2 * k + 1 && lane < 32) {
#pragma unroll
            for (int f = 0; f < FMAX; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) tmp[f * J + jj + q] = acc[f][q];
#pragma unroll
            for (int q = 0; q < 4; ++q) tmp[FMAX * J + jj + q] = db[q];
        }
        __syncthreads();
        if (wave == 2 * k && lane < 32) {
#pragma unroll
            for (int f = 0; f < FMAX; ++f)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float pr = acc[f][q] + tmp[f * J + jj + q];
                    red[f * J + jj + q] = (k == 0 ? 0.0f : red[f * J + jj + q]) + pr;
                }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float pr = db[q] + tmp[FMAX * J + jj + q];
                red[FMAX * J + jj + q] = (k == 0 ? 0.0f : red[FMAX * J + jj + q]) + pr;
            }
        }
        __syncthreads();
    }
    float* out = partial + ((int64_t)n * gridDim.x + blockIdx.x) * (int64_t)(2 * (FMAX + 1) * J);
    for (int idx = threadIdx.x; idx < 2 * (FMAX + 1) * J; idx += 512) out[idx] = xs[idx];
}

__global__ __launch_bounds__(256) void fc_bwd_pair_reduce_kernel(const int C, const float* __restrict__ partial, float* __restrict__ dwb) {
    constexpr int per = 2 * 17 * J;
    const int n = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per) return;
    dwb[(int64_t)n * per + idx] = nmarl_ordered_sum(partial + (int64_t)n * C * per + idx, per, C);
}

// Wider inputs (F > 16; or unaligned column blocks): thread = (column j, row lane of 4), scalar loads.
template <int FMAX>
__global__ __launch_bounds__(256) void fc_bwd_rows_kernel(const int64_t rows, const int F, const int tiles_per_block,
                                                     const float* __restrict__ x, const int64_t x_sn, const int64_t x_row,
                                                     const float* __restrict__ y, const int64_t y_sn, const int64_t y_row,
                                                     const float* __restrict__ dy, const int64_t dy_sn, const int64_t dy_row,
                                                     const int act, float* __restrict__ partial, const XGather xg) {
    constexpr int FP = FMAX + 4;
    constexpr int RU = RowUnroll<FMAX>::value;
    __shared__ __attribute__((aligned(16))) float xs[TILE * FP];        // later reused as the [FMAX + 1, 64] reduction pad
    const int n = blockIdx.y, j = threadIdx.x & 63, rl = threadIdx.x >> 6;
    float acc[FMAX];
#pragma unroll
    for (int f = 0; f < FMAX; ++f) acc[f] = 0.0f;
    float db = 0.0f;
    const float* yn = y + (int64_t)n * y_sn;
    const float* dyn = dy + (int64_t)n * dy_sn;
    __shared__ int64_t gsrc[FMAX / 4];
    const bool g4 = xg.nbr_idx != nullptr && gather4_ok(x, x_sn, x_row, xg.A);
    if (g4) gather_table<FMAX>(gsrc, xg.nbr_idx, n, xg.m_max, xg.A, F, x_sn);
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * TILE;
        if (row0 >= rows) break;
        if (g4) {
            stage_tile_gather4<FMAX>(xs, x, x_row, row0, rows, gsrc);
        } else {
            for (int idx = threadIdx.x; idx < TILE * FMAX; idx += 256) {
                const int r = idx / FMAX, f = idx - r * FMAX;
                const int64_t row = row0 + r;
                const bool ok = f < F && row < rows;
                const float v = x_elem(x, x_sn, x_row, n, row, f, ok, xg);
                xs[r * FP + f] = ok ? v : 0.0f;
            }
        }
        __syncthreads();
#pragma unroll RU
        for (int rr = rl; rr < TILE; rr += 4) {
            const int64_t row = row0 + rr;
            float g = 0.0f;
            if (row < rows) g = act_bwd(dyn[row * dy_row + j], yn[row * y_row + j], act);
            db += g;
#pragma unroll
            for (int f4 = 0; f4 < FMAX / 4; ++f4) {
                const float4 xv = *reinterpret_cast<const float4*>(xs + rr * FP + 4 * f4);
                acc[4 * f4 + 0] = fmaf(xv.x, g, acc[4 * f4 + 0]);
                acc[4 * f4 + 1] = fmaf(xv.y, g, acc[4 * f4 + 1]);
                acc[4 * f4 + 2] = fmaf(xv.z, g, acc[4 * f4 + 2]);
                acc[4 * f4 + 3] = fmaf(xv.w, g, acc[4 * f4 + 3]);
            }
        }
        __syncthreads();
    }
    // the four row lanes add up in a fixed order through LDS
    float* red = xs;
    for (int k = 0; k < 4; ++k) {
        if (rl == k) {
#pragma unroll
            for (int f = 0; f < FMAX; ++f)
                if (f < F) red[f * J + j] = (k == 0 ? 0.0f : red[f * J + j]) + acc[f];
            red[FMAX * J + j] = (k == 0 ? 0.0f : red[FMAX * J + j]) + db;
        }
        __syncthreads();
    }
    float* out = partial + ((int64_t)n * gridDim.x + blockIdx.x) * (int64_t)(F + 1) * J;
    for (int idx = threadIdx.x; idx < (F + 1) * J; idx += 256) {
        const int f = idx >> 6;
        out[idx] = red[(f < F ? f : FMAX) * J + (idx & 63)];
    }
}

// Wider inputs on the matrix cores (F > 16, 16-byte aligned y / dy rows): dW[f][j] = sum_r x~[r][f] g[r][j] with
// g = dy * act'(y) as v_mfma_f32_16x16x4_f32 products that contract over FOUR ROWS at a time (M = input feature, N = output
// column, K = row).  Wave w owns output columns 16 w .. 16 w + 15 and all F input features (FMAX / 16 accumulator tiles); x~ and g
// tiles of 64 rows wait in LDS (x~ gathered as in fc_bwd_rows_kernel).  The VALU form needs one LDS read per 4 FMAs and is bound
// by LDS bandwidth (1.0 ms for the grid's 60-wide observation encoder over 25 x 122 880 rows); this one by HBM (y and dy).
// Same partial layout / fixed-order reduction as the other backward kernels.
typedef float fc_f32x4 __attribute__((ext_vector_type(4)));
template <int FMAX>
__global__ __launch_bounds__(256) void fc_bwd_mfma_kernel(const int64_t rows, const int F, const int tiles_per_block,
                                                     const float* __restrict__ x, const int64_t x_sn, const int64_t x_row,
                                                     const float* __restrict__ y, const int64_t y_sn, const int64_t y_row,
                                                     const float* __restrict__ dy, const int64_t dy_sn, const int64_t dy_row,
                                                     const int act, float* __restrict__ partial, const XGather xg) {
    constexpr int FP = FMAX + 4;          // x~ tile pitch (stage_tile_gather4's)
    constexpr int GP = J + 16;            // g tile pitch: the four row groups of a k-step fall into disjoint banks
    constexpr int NT = FMAX / 16;
    __shared__ __attribute__((aligned(16))) float xs[TILE * FP];
    __shared__ __attribute__((aligned(16))) float gs[TILE * GP];
    __shared__ int64_t gsrc[FMAX / 4];
    const int n = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = lane & 15, kq = lane >> 4;
    fc_f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = fc_f32x4{0.f, 0.f, 0.f, 0.f};
    float dbl = 0.0f;
    const float* yn = y + (int64_t)n * y_sn;
    const float* dyn = dy + (int64_t)n * dy_sn;
    const bool g4 = xg.nbr_idx != nullptr && gather4_ok(x, x_sn, x_row, xg.A);
    if (g4) gather_table<FMAX>(gsrc, xg.nbr_idx, n, xg.m_max, xg.A, F, x_sn);
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * TILE;
        if (row0 >= rows) break;
        if (g4) {
            stage_tile_gather4<FMAX>(xs, x, x_row, row0, rows, gsrc);
        } else {
            for (int idx = threadIdx.x; idx < TILE * FMAX; idx += 256) {
                const int r = idx / FMAX, f = idx - r * FMAX;
                const int64_t row = row0 + r;
                const bool ok = f < F && row < rows;
                const float v = x_elem(x, x_sn, x_row, n, row, f, ok, xg);
                xs[r * FP + f] = ok ? v : 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < TILE * (J / 4) / 256; ++i) {            // g tile: 64 rows x 16 float4, unconditional clamped loads
            const int idx = threadIdx.x + 256 * i;
            const int r = idx >> 4, j4 = (idx & 15) * 4;
            const int64_t row = row0 + r;
            const bool ok = row < rows;
            const float4 d4 = *reinterpret_cast<const float4*>(dyn + (ok ? row : 0) * dy_row + j4);
            const float4 y4 = *reinterpret_cast<const float4*>(yn + (ok ? row : 0) * y_row + j4);
            const float w = ok ? 1.0f : 0.0f;
            *reinterpret_cast<float4*>(gs + r * GP + j4) = float4{w * act_bwd(d4.x, y4.x, act), w * act_bwd(d4.y, y4.y, act),
                                                                  w * act_bwd(d4.z, y4.z, act), w * act_bwd(d4.w, y4.w, act)};
        }
        __syncthreads();
#pragma unroll 4
        for (int rs = 0; rs < TILE; rs += 4) {
            const float b = gs[(rs + kq) * GP + 16 * wave + m];
            dbl += b;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[(rs + kq) * FP + 16 * t + m], b, acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
    // D layout: column (lane & 15) = output j of this wave's block, row 4 (lane >> 4) + reg = feature within the tile
    float* out = partial + ((int64_t)n * gridDim.x + blockIdx.x) * (int64_t)(F + 1) * J;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * t + 4 * kq + r;
            if (f < F) out[f * J + 16 * wave + m] = acc[t][r];
        }
    dbl += __shfl_xor(dbl, 16, 64);
    dbl += __shfl_xor(dbl, 32, 64);
    if (kq == 0) out[F * J + 16 * wave + m] = dbl;
}

__global__ __launch_bounds__(256) void fc_bwd_reduce_kernel(const int C, const int F, const float* __restrict__ partial,
                                                            float* __restrict__ dw, const int64_t dw_sn,
                                                            float* __restrict__ db, const int64_t db_sn) {
    const int n = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = (F + 1) * J;
    if (idx >= per) return;
    const float s = nmarl_ordered_sum(partial + (int64_t)n * C * per + idx, per, C);
    if (idx < F * J) dw[(int64_t)n * dw_sn + idx] = s;
    else db[(int64_t)n * db_sn + idx - F * J] = s;
}

// ---------------------------------------------------------------------------------------------------------
// Backward of a THIN linear layer y[n,r,:O] = h[n,r,:64] @ w[n] + b[n] with O <= 8 outputs: the actor / critic
// heads (policies.py:50-77: fc(h, 'pi', n_a) and fc([h, na], 'v', 1)) over all T*E rows of the update.
// The GEMM library runs the dgrad (K = O) and the wgrad (64 x O, K = rows) at ~1 ms each; both are one
// streaming pass over h and dy here: thread = (hidden unit k, row lane) keeps w[k, :] and the dW[k, :] / db
// accumulators in registers, reads h[r, k] and writes dh[r, k] coalesced, and gets dy[r, :] as an LDS broadcast.
// partial: [N, gridDim.x, 65, O] (rows 0..63 dW, row 64 db), summed in fixed order by thin_bwd_reduce_kernel.
constexpr int MAXO = 8;

// thread = (4 consecutive hidden units k4, row lane rl of 16): h is read and dh written as float4 (16 threads cover a
// row's 256 bytes), the four row passes of a 64-row tile are in flight together; rows past the end: clamped loads, zero
// gradient (their dy tile entries are zero), predicated store.
__global__ __launch_bounds__(256) void thin_bwd_kernel(const int64_t rows, const int O, const int tiles_per_block,
                                                       const float* __restrict__ h, const int64_t h_sn,
                                                       const float* __restrict__ dy, const int64_t dy_sn,
                                                       const float* __restrict__ dy2, const int64_t dy2_sn,
                                                       const float* __restrict__ w, const int64_t w_sn,
                                                       float* __restrict__ dh, const int64_t dh_sn,
                                                       float* __restrict__ partial) {
    __shared__ float ds[TILE * MAXO];
    __shared__ float red[4 * 65 * MAXO];
    const int n = blockIdx.y, k4 = (threadIdx.x & 15) * 4, rl = threadIdx.x >> 4;
    float wk[4][MAXO], acc[4][MAXO], dbacc[MAXO];
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wk[q][o] = o < O ? w[(int64_t)n * w_sn + (k4 + q) * O + o] : 0.0f;
            acc[q][o] = 0.0f;
        }
        dbacc[o] = 0.0f;
    }
    const float* hn = h + (int64_t)n * h_sn + k4;
    const float* dyn = dy + (int64_t)n * dy_sn;
    const float* dy2n = dy2 ? dy2 + (int64_t)n * dy2_sn : nullptr;
    const int O1 = dy2 ? O - 1 : O;          // dy holds the first O1 columns, dy2 (optional, [rows]) the last one
    float* dhn = dh + (int64_t)n * dh_sn + k4;
    // the dy tile of the NEXT tile waits in registers (TILE * O1 <= 512 values: two per thread, + one of dy2), the h rows
    // of the next tile are requested before this tile's FMAs: no global latency exposed between two tiles
    float dr0, dr1, dr2;
    float4 hv[4];
#define NMARL_THIN_LOAD(tile_)                                                             \
    {                                                                                      \
        const int64_t r0_ = ((int64_t)blockIdx.x * tiles_per_block + (tile_)) * TILE;      \
        const int64_t g0_ = r0_ * O1 + threadIdx.x, g1_ = g0_ + 256, lim_ = rows * O1;     \
        dr0 = dyn[g0_ < lim_ ? g0_ : 0] * (g0_ < lim_ ? 1.0f : 0.0f);                      \
        dr1 = dyn[g1_ < lim_ ? g1_ : 0] * (g1_ < lim_ ? 1.0f : 0.0f);                      \
        const int64_t r2_ = r0_ + (threadIdx.x & (TILE - 1));                              \
        dr2 = dy2n ? dy2n[r2_ < rows ? r2_ : 0] * (r2_ < rows ? 1.0f : 0.0f) : 0.0f;       \
    }
#define NMARL_THIN_HLOAD(tile_)                                                            \
    {                                                                                      \
        const int64_t r0_ = ((int64_t)blockIdx.x * tiles_per_block + (tile_)) * TILE;      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
            const int64_t row = r0_ + rl + 16 * i;                                         \
            hv[i] = *reinterpret_cast<const float4*>(hn + (row < rows ? row : rows - 1) * J); \
        }                                                                                  \
    }
    NMARL_THIN_LOAD(0)
    NMARL_THIN_HLOAD(0)
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * TILE;
        if (row0 >= rows) break;
        {
            const int i0 = threadIdx.x, i1 = threadIdx.x + 256;             // the tile's dy rows are contiguous
            if (i0 < TILE * O1) ds[(i0 / O1) * O + i0 % O1] = dr0;
            if (i1 < TILE * O1) ds[(i1 / O1) * O + i1 % O1] = dr1;
            if (dy2n && threadIdx.x < TILE) ds[threadIdx.x * O + O1] = dr2;
        }
        NMARL_THIN_LOAD(tile + 1)
        float4 hc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hc[i] = hv[i];
        NMARL_THIN_HLOAD(tile + 1)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = rl + 16 * i;
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            const float hq[4] = {hc[i].x, hc[i].y, hc[i].z, hc[i].w};
#pragma unroll
            for (int o = 0; o < MAXO; ++o)
                if (o < O) {
                    const float d = ds[rr * O + o];          // zero for rows past the end
                    dbacc[o] += d;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        g[q] = fmaf(d, wk[q][o], g[q]);
                        acc[q][o] = fmaf(hq[q], d, acc[q][o]);
                    }
                }
            if (row0 + rr < rows) *reinterpret_cast<float4*>(dhn + (row0 + rr) * J) = float4{g[0], g[1], g[2], g[3]};
        }
        __syncthreads();
    }
#undef NMARL_THIN_LOAD
#undef NMARL_THIN_HLOAD
    // 16 row lanes: 4 inside the wave (lanes 16 i + j) by shuffles, then the 4 waves through LDS in a fixed order
#pragma unroll
    for (int o = 0; o < MAXO; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = acc[q][o];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[q][o] = v;
        }
        float v = dbacc[o];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        dbacc[o] = v;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 16) {
#pragma unroll
        for (int o = 0; o < MAXO; ++o)
            if (o < O) {
#pragma unroll
                for (int q = 0; q < 4; ++q) red[wave * 65 * MAXO + (k4 + q) * O + o] = acc[q][o];
                if (lane == 0) red[wave * 65 * MAXO + 64 * O + o] = dbacc[o];
            }
    }
    __syncthreads();
    float* out = partial + ((int64_t)n * gridDim.x + blockIdx.x) * (int64_t)65 * O;
    for (int idx = threadIdx.x; idx < 65 * O; idx += 256)
        out[idx] = ((red[idx] + red[65 * MAXO + idx]) + red[2 * 65 * MAXO + idx]) + red[3 * 65 * MAXO + idx];
}

// ---------------------------------------------------------------------------------------------------------
// The update's heads, the A2C loss and the heads' backward in ONE streaming pass over h (round 6; policies.py:20-30, 50-77):
//   logits = h w[:, :A] + b[:A];  v = h w[:, A] + b[A] + va      (va: the critic's neighbour-action term, nbr_action_value_fwd_kernel)
//   loss terms / d logits / d v: the formulas of a2c_loss_kernel (csrc/a2c.hip) with an upstream gradient of 1
//   dW / db of [pi_w | v_w[:64]] as thin_bwd_kernel; dh (DH: written for the recurrences that take dL/dh as a tensor)
//   dy8 [N,rows,8] = [d logits | d v | 0]: what the one-launch BPTT kernels expand to dL/dh themselves (csrc/lstm_bptt.hip)
// instead of the skinny GEMM (h read), the loss forward and backward kernels (logits read twice) and thin_bwd_kernel (h read
// again, dh written): h is read ONCE.  Thread = (4 consecutive hidden units k4, row lane rl of 16) as in thin_bwd_kernel; the 16
// threads of a row finish the A + 1 dots by DPP row operations over their 16 lanes; four of them compute one row's loss gradient
// each and broadcast it back (no LDS, no barrier inside the tile loop).
// partial: [N, gridDim.x, 65 * O + 3] (dW rows 0..63, db row 64, then the three loss sums), summed in fixed order by
// heads_loss_reduce_kernel.
template <int CTRL>
__device__ __forceinline__ float hl_dpp(const float v) {     // 0x110 + n: row_shr:n (zeros shifted in); 0x150 + n: row_newbcast:n
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// OT: compile-time bound of O = A + 1 (5: CACC, 6: the grids, 8: anything else) -- the register arrays are that long
template <bool DH, int OT>
__global__ __launch_bounds__(256) void heads_loss_kernel(const int64_t rows, const int N, const int A_, const int tiles_per_block,
                                                         const float* __restrict__ h, const int64_t h_sn,
                                                         const float* __restrict__ w, const int64_t w_sn,
                                                         const float* __restrict__ b, const int64_t b_sn,
                                                         const float* __restrict__ va, const uint8_t* __restrict__ action,
                                                         const float* __restrict__ adv, const float* __restrict__ R,
                                                         const float v_coef, const float e_coef, float* __restrict__ dy8,
                                                         float* __restrict__ dv_out, float* __restrict__ dh, const int64_t dh_sn,
                                                         float* __restrict__ partial) {
    __shared__ float red[4 * (65 * OT + 3)];
    // OT 5 / 6: A is exactly OT - 1 (the launcher's choice): every `k < A` / `o < O` below is decided at compile time
    const int A = OT < 8 ? OT - 1 : A_;
    const int O = A + 1;
    const int n = blockIdx.y, ki = threadIdx.x & 15, k4 = ki * 4, rl = threadIdx.x >> 4;
    float wk[4][OT], acc[4][OT], dbacc[OT], bo[OT];
#pragma unroll
    for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wk[q][o] = o < O ? w[(int64_t)n * w_sn + (k4 + q) * O + o] : 0.0f;
            acc[q][o] = 0.0f;
        }
        dbacc[o] = 0.0f;
        bo[o] = o < O ? b[(int64_t)n * b_sn + o] : 0.0f;
    }
    const float inv_m = 1.0f / (float)rows;
    float s_pol = 0.0f, s_val = 0.0f, s_ent = 0.0f;
    const float* hn = h + (int64_t)n * h_sn + k4;
    float* dhn = DH ? dh + (int64_t)n * dh_sn + k4 : nullptr;
    // the NEXT tile's h rows and per-row scalars are requested before this tile's arithmetic (no global latency between two tiles)
    // (the per-row scalars of ONE row per lane: lane ki computes the loss of row pass ki & 3 only)
    float4 hv[4];
    float r_va, r_adv, r_R;
    int r_a;
    const int ip = ki & 3;
#define NMARL_HL_LOAD(tile_)                                                               \
    {                                                                                      \
        const int64_t r0_ = ((int64_t)blockIdx.x * tiles_per_block + (tile_)) * TILE;      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                    \
            const int64_t row = r0_ + rl + 16 * i, rc = row < rows ? row : rows - 1;       \
            hv[i] = *reinterpret_cast<const float4*>(hn + rc * J);                         \
        }                                                                                  \
        const int64_t rw_ = r0_ + rl + 16 * ip, rs_ = rw_ < rows ? rw_ : rows - 1;         \
        const int64_t ix_ = (int64_t)n * rows + rs_;                                       \
        r_va = va[ix_]; r_adv = adv[ix_]; r_R = R[ix_];                                    \
        r_a = action[rs_ * N + n];                                                         \
    }
    NMARL_HL_LOAD(0)
    for (int tile = 0; tile < tiles_per_block; ++tile) {
        const int64_t row0 = ((int64_t)blockIdx.x * tiles_per_block + tile) * TILE;
        if (row0 >= rows) break;
        float4 hc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) hc[i] = hv[i];
        const float s_va = r_va, s_ad = r_adv, s_R = r_R;
        const int a = r_a;
        NMARL_HL_LOAD(tile + 1)
        // ---- the A + 1 dots of the tile's four row passes: 4 FMAs per lane, then the row's 16 lanes (one DPP row) add up by four
        // shifted adds -- lane 15 holds the sum -- and take it back by a row broadcast (full-rate VALU, no LDS crossbar)
        float zz[4][OT];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float hq[4] = {hc[i].x, hc[i].y, hc[i].z, hc[i].w};
#pragma unroll
            for (int o = 0; o < OT; ++o) {
                float v = 0.0f;
                if (o < O) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v = fmaf(hq[q], wk[q][o], v);
                    v += hl_dpp<0x111>(v); v += hl_dpp<0x112>(v); v += hl_dpp<0x114>(v); v += hl_dpp<0x118>(v);
                    v = hl_dpp<0x15F>(v) + bo[o];
                }
                zz[i][o] = v;
            }
        }
        // ---- the loss of ONE row per lane and its gradient (a2c_loss_kernel's arithmetic, upstream gradient 1): lane ki takes row
        // pass ki & 3 (lanes 0..3 of the row are the ones whose result is used) -- all four passes in every lane would be 4 x the
        // transcendental work for the same 64 rows per wave
        float z[OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) z[o] = ip == 0 ? zz[0][o] : ip == 1 ? zz[1][o] : ip == 2 ? zz[2][o] : zz[3][o];
        const int64_t row = row0 + rl + 16 * ip;
        const bool live = row < rows;
        float d[OT];
        {
            float pr[OT], lp[OT];
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < OT; ++k) {
                pr[k] = k < A ? z[k] : -INFINITY;
                m = fmaxf(m, pr[k]);
            }
            float zs = 0.0f;
#pragma unroll
            for (int k = 0; k < OT; ++k) {
                pr[k] = k < A ? expf(pr[k] - m) : 0.0f;
                zs += pr[k];
            }
            float Hent = 0.0f, lpa = 0.0f, pa = 0.0f;
#pragma unroll
            for (int k = 0; k < OT; ++k) {
                pr[k] = pr[k] / zs;
                lp[k] = logf(fminf(fmaxf(pr[k], 1e-10f), 1.0f));
                if (k < A) Hent -= pr[k] * lp[k];
                if (k == a) { lpa = lp[k]; pa = pr[k]; }
            }
            float vrow = 0.0f;
#pragma unroll
            for (int k = 0; k < OT; ++k) vrow = k == A ? z[k] : vrow;
            vrow += s_va;
            const float ad = s_ad, dR = s_R - vrow;
            if (live && ki < 4) {
                s_pol -= lpa * ad;
                s_val += dR * dR;
                s_ent += Hent;
            }
            float gbar = 0.0f;
#pragma unroll
            for (int k = 0; k < OT; ++k)
                if (k < A) gbar += pr[k] * -(lp[k] + (pr[k] >= 1e-10f ? 1.0f : 0.0f));
            const float ca = pa >= 1e-10f ? 1.0f : 0.0f;
            const float gn = live ? inv_m : 0.0f;            // rows past the end: zero gradient
#pragma unroll
            for (int k = 0; k < OT; ++k) {
                float dk = 0.0f;
                if (k < A) {
                    const float gk = -(lp[k] + (pr[k] >= 1e-10f ? 1.0f : 0.0f));
                    const float d_pol = -ad * ca * ((k == a ? 1.0f : 0.0f) - pr[k]);
                    const float d_ent = -e_coef * pr[k] * (gk - gbar);
                    dk = gn * (d_pol + d_ent);
                } else if (k == A) {
                    dk = -gn * v_coef * dR;
                }
                d[k] = dk;
            }
            if (live && ki < 4) {
                const int64_t ix = (int64_t)n * rows + row;
                *reinterpret_cast<float4*>(dy8 + ix * 8) = float4{d[0], d[1], d[2], d[3]};
                *reinterpret_cast<float4*>(dy8 + ix * 8 + 4) = float4{d[4], OT > 5 ? d[OT > 5 ? 5 : 0] : 0.0f, OT > 6 ? d[OT > 6 ? 6 : 0] : 0.0f,
                                                                  OT > 7 ? d[OT > 7 ? 7 : 0] : 0.0f};
                float dvv = 0.0f;
#pragma unroll
                for (int k = 0; k < OT; ++k) dvv = k == A ? d[k] : dvv;
                dv_out[ix] = dvv;
            }
        }
        // ---- the heads' backward per row pass: the pass's gradient comes back from lane i of the row (row broadcast); dW / db
        // accumulators, dL/dh of the lane's four units
#define NMARL_HL_BWD(i)                                                                    \
        {                                                                                  \
            const float hq[4] = {hc[i].x, hc[i].y, hc[i].z, hc[i].w};                      \
            float g[4] = {0.f, 0.f, 0.f, 0.f};                                             \
            _Pragma("unroll") for (int o = 0; o < OT; ++o)                                 \
                if (o < O) {                                                               \
                    float di = hl_dpp<0x150 + i>(d[o]);                                    \
                    asm volatile("" : "+v"(di));      /* its own register: hipcc 7.2 folded the last pass's broadcast of d[A] into the bias \
                                                         sum and fed the weight gradient a stale register (DH = false, O = 5) */ \
                    dbacc[o] += di;                                                        \
                    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                        \
                        g[q] = fmaf(di, wk[q][o], g[q]);                                   \
                        acc[q][o] = fmaf(hq[q], di, acc[q][o]);                            \
                    }                                                                      \
                }                                                                          \
            if (DH && row0 + rl + 16 * i < rows)                                           \
                *reinterpret_cast<float4*>(dhn + (row0 + rl + 16 * i) * J) = float4{g[0], g[1], g[2], g[3]}; \
        }
        NMARL_HL_BWD(0) NMARL_HL_BWD(1) NMARL_HL_BWD(2) NMARL_HL_BWD(3)
#undef NMARL_HL_BWD
    }
#undef NMARL_HL_LOAD
    // 16 row lanes: 4 inside the wave (lanes 16 i + j) by shuffles, then the 4 waves through LDS in a fixed order
#pragma unroll
    for (int o = 0; o < OT; ++o) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = acc[q][o];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[q][o] = v;
        }
        float v = dbacc[o];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        dbacc[o] = v;
    }
    // (lanes 0..3 of every row hold loss sums: first those four, then the row lanes like the accumulators above)
    s_pol += __shfl_xor(s_pol, 1, 64); s_pol += __shfl_xor(s_pol, 2, 64);
    s_val += __shfl_xor(s_val, 1, 64); s_val += __shfl_xor(s_val, 2, 64);
    s_ent += __shfl_xor(s_ent, 1, 64); s_ent += __shfl_xor(s_ent, 2, 64);
    s_pol += __shfl_xor(s_pol, 16, 64); s_pol += __shfl_xor(s_pol, 32, 64);
    s_val += __shfl_xor(s_val, 16, 64); s_val += __shfl_xor(s_val, 32, 64);
    s_ent += __shfl_xor(s_ent, 16, 64); s_ent += __shfl_xor(s_ent, 32, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int PW = 65 * OT + 3;
    if (lane < 16) {
#pragma unroll
        for (int o = 0; o < OT; ++o)
            if (o < O) {
#pragma unroll
                for (int q = 0; q < 4; ++q) red[wave * PW + (k4 + q) * O + o] = acc[q][o];
                if (lane == 0) red[wave * PW + 64 * O + o] = dbacc[o];
            }
        if (lane == 0) { red[wave * PW + 65 * O] = s_pol; red[wave * PW + 65 * O + 1] = s_val; red[wave * PW + 65 * O + 2] = s_ent; }
    }
    __syncthreads();
    const int per = 65 * O + 3;
    float* out = partial + ((int64_t)n * gridDim.x + blockIdx.x) * (int64_t)per;
    for (int idx = threadIdx.x; idx < per; idx += 256)
        out[idx] = ((red[idx] + red[PW + idx]) + red[2 * PW + idx]) + red[3 * PW + idx];
}

__global__ __launch_bounds__(256) void heads_loss_reduce_kernel(const int C, const int O, const int64_t rows, const float v_coef,
                                                                const float e_coef, const float* __restrict__ partial,
                                                                float* __restrict__ dw, const int64_t dw_sn, float* __restrict__ db,
                                                                const int64_t db_sn, float* __restrict__ loss_out /*[N,3]*/) {
    const int n = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 65 * O + 3;
    if (idx >= per) return;
    float s = nmarl_ordered_sum(partial + (int64_t)n * C * per + idx, per, C);
    if (idx < 64 * O) dw[(int64_t)n * dw_sn + idx] = s;
    else if (idx < 65 * O) db[(int64_t)n * db_sn + idx - 64 * O] = s;
    else {
        const int k = idx - 65 * O;
        s /= (float)rows;
        loss_out[n * 3 + k] = k == 0 ? s : k == 1 ? s * 0.5f * v_coef : -s * e_coef;
    }
}

__global__ __launch_bounds__(256) void thin_bwd_reduce_kernel(const int C, const int O, const float* __restrict__ partial,
                                                              float* __restrict__ dw, const int64_t dw_sn,
                                                              float* __restrict__ db, const int64_t db_sn) {
    const int n = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = 65 * O;
    if (idx >= per) return;
    const float s = nmarl_ordered_sum(partial + (int64_t)n * C * per + idx, per, C);
    if (idx < 64 * O) dw[(int64_t)n * dw_sn + idx] = s;
    else db[(int64_t)n * db_sn + idx - 64 * O] = s;
}

// ---------------------------------------------------------------------------------------------------------
// Neighbour-action term of the centralised critic (policies.py:59-77): v += onehot(neighbours' actions) @ w_a,
// with the one-hot never materialised: a gather of w_a rows by the action bytes (fwd) and a histogram of dv
// (bwd: dw_a[k*A + a] = sum of dv over the rows whose k-th neighbour played a).  act [rows,N] u8,
// nbr [N,m_max] (-1 padded), w_a [N, m_max*A], va / dv [N,rows].
constexpr int MAXW = 32;

__global__ __launch_bounds__(256) void nbr_action_value_kernel(const int64_t rows, const int N, const int A, const int m_max,
                                                               const int32_t* __restrict__ nbr, const uint8_t* __restrict__ act,
                                                               const float* __restrict__ w, const int64_t w_sn,
                                                               float* __restrict__ va, const int accumulate) {
    const int64_t total = (int64_t)N * rows;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / rows);
        const int64_t r = idx - (int64_t)n * rows;
        float v = 0.0f;
        for (int k = 0; k < m_max; ++k) {
            const int j = nbr[n * m_max + k];
            if (j >= 0) v += w[(int64_t)n * w_sn + k * A + act[r * N + j]];
        }
        va[idx] = accumulate ? va[idx] + v : v;
    }
}

// partial: [N, gridDim.x, m_max*A]
__global__ __launch_bounds__(256) void nbr_action_value_bwd_kernel(const int64_t rows, const int N, const int A, const int m_max,
                                                                   const int rows_per_block, const int32_t* __restrict__ nbr,
                                                                   const uint8_t* __restrict__ act, const float* __restrict__ dv,
                                                                   float* __restrict__ partial) {
    __shared__ float red[4 * MAXW];
    const int n = blockIdx.y, W = m_max * A;
    float acc[MAXW];
#pragma unroll
    for (int i = 0; i < MAXW; ++i) acc[i] = 0.0f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const float d = dv[(int64_t)n * rows + r];
        for (int k = 0; k < m_max; ++k) {
            const int j = nbr[n * m_max + k];
            if (j < 0) continue;
            const int hit = k * A + act[r * N + j];
#pragma unroll
            for (int i = 0; i < MAXW; ++i) acc[i] += (i == hit) ? d : 0.0f;
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
        float s = acc[i];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) red[wv * MAXW + i] = s;
    }
    __syncthreads();
    if (threadIdx.x < W)
        partial[((int64_t)n * gridDim.x + blockIdx.x) * W + threadIdx.x] =
            ((red[threadIdx.x] + red[MAXW + threadIdx.x]) + red[2 * MAXW + threadIdx.x]) + red[3 * MAXW + threadIdx.x];
}

__global__ __launch_bounds__(64) void nbr_action_value_reduce_kernel(const int C, const int W, const float* __restrict__ partial,
                                                                     float* __restrict__ dw, const int64_t dw_sn) {
    const int n = blockIdx.x, i = threadIdx.x;
    if (i >= W) return;
    const float s = nmarl_ordered_sum(partial + (int64_t)n * C * W + i, W, C);
    dw[(int64_t)n * dw_sn + i] = s;
}

inline bool view_ok(const void* p, int64_t sn, int64_t row, int64_t rows, int W) {
    return p != nullptr && row >= W && sn >= (rows > 0 ? (rows - 1) * row + W : 0);
}

}  // namespace

extern "C" int nmarl_fc_bwd_chunks(int64_t rows, int32_t N) {
    if (rows <= 0 || N <= 0) return 0;
    const int64_t tiles = (rows + TILE - 1) / TILE;
    int64_t tpb = tiles * N / 1024;                 // about 1024 blocks; at most 32 tiles (2048 rows) per block
    tpb = tpb < 1 ? 1 : (tpb > 32 ? 32 : tpb);
    return (int)((tiles + tpb - 1) / tpb);
}

extern "C" int nmarl_fc_fwd(int64_t rows, int32_t N, int32_t F, int32_t Jw, const float* x, int64_t x_sn, int64_t x_row,
                            const float* w, int64_t w_sn, const float* b, int64_t b_sn, int32_t act, float* y,
                            int64_t y_sn, int64_t y_row, void* stream) {
    if (rows < 0 || N <= 0 || F <= 0 || F > 64 || Jw != J || act < 0 || act > 2 || w_sn < (int64_t)F * J || b_sn < J)
        return NMARL_EINVAL;
    if (rows == 0) return NMARL_OK;
    if (!view_ok(x, x_sn, x_row, 1, F) || !view_ok(y, y_sn, y_row, rows, J) || !w || !b) return NMARL_EINVAL;
    const int64_t tiles = (rows + TILE - 1) / TILE;
    int64_t tpb = tiles * N / 2048;
    tpb = tpb < 1 ? 1 : (tpb > 16 ? 16 : tpb);
    const dim3 grid((unsigned)((tiles + tpb - 1) / tpb), N);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NMARL_FC_FWD(FM) hipLaunchKernelGGL(fc_fwd_kernel<FM>, grid, dim3(256), 0, st, rows, F, (int)tpb, x, x_sn, x_row, w, w_sn, \
                                            b, b_sn, act, y, y_sn, y_row)
    if (F <= 16) NMARL_FC_FWD(16); else if (F <= 32) NMARL_FC_FWD(32); else NMARL_FC_FWD(64);
#undef NMARL_FC_FWD
    return nmarl_check_launch();
}

extern "C" int nmarl_fc_fwd_multi(int64_t rows, int32_t N, int32_t n_parts, const nmarl_fc_part_t* parts, int32_t act,
                                  float* y, int64_t y_sn, int64_t y_row, void* stream) {
    if (rows < 0 || N <= 0 || n_parts <= 0 || n_parts > NMARL_FC_MAX_PARTS || !parts || act < 0 || act > 2) return NMARL_EINVAL;
    if (rows == 0) return NMARL_OK;
    if (!view_ok(y, y_sn, y_row, rows, J * n_parts)) return NMARL_EINVAL;
    FcParts ps{};
    int fmax = 0;
    for (int i = 0; i < n_parts; ++i) {
        const nmarl_fc_part_t& p = parts[i];
        if (p.F <= 0 || p.F > 64 || !p.x || !p.w || !p.b || p.w_sn < (int64_t)p.F * J || p.b_sn < J) return NMARL_EINVAL;
        if (p.nbr_idx ? (p.gather_A <= 0 || p.m_max <= 0 || p.F != p.gather_A * p.m_max || p.x_row < p.gather_A)
                      : p.x_row < p.F)
            return NMARL_EINVAL;
        ps.p[i] = p;
        fmax = p.F > fmax ? p.F : fmax;
    }
    const int64_t tiles = (rows + TILE - 1) / TILE;
    int64_t tpb = tiles * N * n_parts / 2048;
    tpb = tpb < 1 ? 1 : (tpb > 16 ? 16 : tpb);
    const dim3 grid((unsigned)((tiles + tpb - 1) / tpb), N, n_parts);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NMARL_FC_MULTI(FM) hipLaunchKernelGGL(fc_fwd_multi_kernel<FM>, grid, dim3(256), 0, st, rows, (int)tpb, ps, act, y, y_sn, y_row)
    bool vec = ((uintptr_t)y % 16) == 0 && (y_sn % 4) == 0 && (y_row % 4) == 0;           // float4 outputs / weights / biases
    for (int i = 0; i < n_parts; ++i)
        vec = vec && ((uintptr_t)ps.p[i].w % 16) == 0 && ((uintptr_t)ps.p[i].b % 16) == 0 && (ps.p[i].w_sn % 4) == 0 && (ps.p[i].b_sn % 4) == 0;
    if (fmax <= 16 && vec) hipLaunchKernelGGL((fc_fwd_multi_kernel<16, true>), grid, dim3(256), 0, st, rows, (int)tpb, ps, act, y, y_sn, y_row);
    else if (fmax <= 16) NMARL_FC_MULTI(16); else if (fmax <= 32) NMARL_FC_MULTI(32); else NMARL_FC_MULTI(64);
#undef NMARL_FC_MULTI
    return nmarl_check_launch();
}

// lstm_dial's own-action term `ai` (agents/utils.py:577: one_hot(argmax(p_i), n_h) added to the encoded observation):
// y[n,r,argmax_a p[n,r,a]] += scale[n] -- the first maximum, like tf.argmax; one thread per row (A <= 8 floats in, one
// read-modify-write out) instead of argmax / one_hot / cast / scale / add launches over [N,rows,64]
__global__ __launch_bounds__(256) void onehot_argmax_add_kernel(const int64_t rows, const int A, const float* __restrict__ p,
                                                                const int64_t p_sn, const float* __restrict__ scale,
                                                                float* __restrict__ y, const int64_t y_sn, const int64_t y_row) {
    const int n = blockIdx.y;
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float* pr = p + (int64_t)n * p_sn + r * A;
    float best = pr[0];
    int bi = 0;
    for (int a = 1; a < A; ++a) {
        const float v = pr[a];
        if (v > best) { best = v; bi = a; }
    }
    y[(int64_t)n * y_sn + r * y_row + bi] += scale ? scale[n] : 1.0f;
}

extern "C" int nmarl_onehot_argmax_add(int64_t rows, int32_t N, int32_t A, int32_t W, const float* p, int64_t p_sn,
                                       const float* scale, float* y, int64_t y_sn, int64_t y_row, void* stream) {
    if (rows < 0 || N <= 0 || A <= 0 || A > W || (rows > 0 && (!p || !y || p_sn < rows * (int64_t)A || y_row < W))) return NMARL_EINVAL;
    if (rows == 0) return NMARL_OK;
    hipLaunchKernelGGL(onehot_argmax_add_kernel, dim3((unsigned)((rows + 255) / 256), N), dim3(256), 0, static_cast<hipStream_t>(stream),
                       rows, A, p, p_sn, scale, y, y_sn, y_row);
    return nmarl_check_launch();
}

static int launch_fc_bwd(int64_t rows, int32_t N, int32_t F, int32_t Jw, const float* x, int64_t x_sn, int64_t x_row,
                         const int32_t* nbr_idx, int32_t gather_A, int32_t m_max, const float* y, int64_t y_sn, int64_t y_row,
                         const float* dy, int64_t dy_sn, int64_t dy_row, int32_t act, float* partial, float* dw, int64_t dw_sn,
                         float* db, int64_t db_sn, void* stream) {
    if (rows <= 0 || N <= 0 || F <= 0 || F > 64 || Jw != J || act < 0 || act > 2 || dw_sn < (int64_t)F * J || db_sn < J ||
        !partial || !dw || !db)
        return NMARL_EINVAL;
    if (nbr_idx ? (gather_A <= 0 || m_max <= 0 || F != gather_A * m_max || !x || x_row < gather_A) : !view_ok(x, x_sn, x_row, 1, F))
        return NMARL_EINVAL;
    if (!view_ok(y, y_sn, y_row, rows, J) || !view_ok(dy, dy_sn, dy_row, rows, J)) return NMARL_EINVAL;
    const int C = nmarl_fc_bwd_chunks(rows, N);
    const int64_t tiles = (rows + TILE - 1) / TILE;
    const int tpb = (int)((tiles + C - 1) / C);
    const dim3 grid(C, N);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const XGather xg{nbr_idx, gather_A, m_max};
#define NMARL_FC_BWD(K, FM) hipLaunchKernelGGL(K<FM>, grid, dim3(256), 0, st, rows, F, tpb, x, x_sn, x_row, y, y_sn, y_row, \
                                               dy, dy_sn, dy_row, act, partial, xg)
    const bool vec = ((uintptr_t)y % 16) == 0 && ((uintptr_t)dy % 16) == 0 && (y_sn % 4) == 0 && (y_row % 4) == 0 &&
                     (dy_sn % 4) == 0 && (dy_row % 4) == 0;
    if (F <= 16 && vec) NMARL_FC_BWD(fc_bwd_kernel, 16);
    else if (F > 32 && vec) NMARL_FC_BWD(fc_bwd_mfma_kernel, 64);
    else if (F > 16 && vec) NMARL_FC_BWD(fc_bwd_mfma_kernel, 32);
    else if (F <= 16) NMARL_FC_BWD(fc_bwd_rows_kernel, 16);
    else if (F <= 32) NMARL_FC_BWD(fc_bwd_rows_kernel, 32);
    else NMARL_FC_BWD(fc_bwd_rows_kernel, 64);
#undef NMARL_FC_BWD
    hipLaunchKernelGGL(fc_bwd_reduce_kernel, dim3(((F + 1) * J + 255) / 256, N), dim3(256), 0, st, C, F, partial, dw, dw_sn, db, db_sn);
    return nmarl_check_launch();
}

extern "C" int nmarl_fc_bwd(int64_t rows, int32_t N, int32_t F, int32_t Jw, const float* x, int64_t x_sn, int64_t x_row,
                            const float* y, int64_t y_sn, int64_t y_row, const float* dy, int64_t dy_sn, int64_t dy_row,
                            int32_t act, float* partial, float* dw, int64_t dw_sn, float* db, int64_t db_sn, void* stream) {
    return launch_fc_bwd(rows, N, F, Jw, x, x_sn, x_row, nullptr, 0, 0, y, y_sn, y_row, dy, dy_sn, dy_row, act, partial, dw, dw_sn, db,
                         db_sn, stream);
}

extern "C" int nmarl_fc_bwd_gather(int64_t rows, int32_t N, int32_t gather_A, int32_t m_max, const int32_t* nbr_idx, int32_t Jw,
                                   const float* x, int64_t x_sn, int64_t x_row, const float* y, int64_t y_sn, int64_t y_row,
                                   const float* dy, int64_t dy_sn, int64_t dy_row, int32_t act, float* partial, float* dw,
                                   int64_t dw_sn, float* db, int64_t db_sn, void* stream) {
    if (!nbr_idx || gather_A <= 0 || m_max <= 0) return NMARL_EINVAL;
    return launch_fc_bwd(rows, N, gather_A * m_max, Jw, x, x_sn, x_row, nbr_idx, gather_A, m_max, y, y_sn, y_row, dy, dy_sn, dy_row, act,
                         partial, dw, dw_sn, db, db_sn, stream);
}

extern "C" int nmarl_fc_bwd_pair(int64_t rows, int32_t N, const nmarl_fc_part_t* parts, const float* y, int64_t y_sn, int64_t y_row,
                                 const uint32_t* relu_bits, int64_t bits_sn, const float* dy, int64_t dy_sn, int64_t dy_row,
                                 int32_t act, float* partial, float* dwb, void* stream) {
    if (rows <= 0 || N <= 0 || !parts || act < 0 || act > 2 || !partial || !dwb) return NMARL_EINVAL;
    FcPair ps{};
    for (int i = 0; i < 2; ++i) {
        const nmarl_fc_part_t& p = parts[i];
        if (p.F <= 0 || p.F > 16 || !p.x) return NMARL_EINVAL;
        if (p.nbr_idx ? (p.gather_A <= 0 || p.m_max <= 0 || p.F != p.gather_A * p.m_max || p.x_row < p.gather_A) : p.x_row < p.F)
            return NMARL_EINVAL;
        ps.p[i] = p;
    }
    if (!view_ok(dy, dy_sn, dy_row, rows, 2 * J) || ((uintptr_t)dy % 16) || (dy_sn % 4) || (dy_row % 4)) return NMARL_EINVAL;
    if (relu_bits ? (act != 1 || bits_sn < rows * 4) : (!view_ok(y, y_sn, y_row, rows, 2 * J) || ((uintptr_t)y % 16) || (y_sn % 4) || (y_row % 4)))
        return NMARL_EINVAL;
    const int C = nmarl_fc_bwd_chunks(rows, N);          // the two-launch form's partition of the rows (same partial sums)
    const int64_t tiles = (rows + TILE - 1) / TILE;
    const int tpb = (int)((tiles + C - 1) / C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (relu_bits)
        hipLaunchKernelGGL(fc_bwd_pair_kernel<true>, dim3(C, N), dim3(512), 0, st, rows, tpb, ps, y, y_sn, y_row, relu_bits, bits_sn, dy, dy_sn,
                           dy_row, act, partial);
    else
        hipLaunchKernelGGL(fc_bwd_pair_kernel<false>, dim3(C, N), dim3(512), 0, st, rows, tpb, ps, y, y_sn, y_row, relu_bits, bits_sn, dy, dy_sn,
                           dy_row, act, partial);
    hipLaunchKernelGGL(fc_bwd_pair_reduce_kernel, dim3((2 * 17 * J + 255) / 256, N), dim3(256), 0, st, C, partial, dwb);
    return nmarl_check_launch();
}

extern "C" int nmarl_thin_linear_bwd(int64_t rows, int32_t N, int32_t H, int32_t O, const float* h, int64_t h_sn,
                                     const float* dy, int64_t dy_sn, const float* dy2, int64_t dy2_sn, const float* w,
                                     int64_t w_sn, float* partial, float* dh, int64_t dh_sn, float* dw, int64_t dw_sn,
                                     float* db, int64_t db_sn, void* stream) {
    const int O1 = dy2 ? O - 1 : O;
    if (rows <= 0 || N <= 0 || H != J || O <= 0 || O > MAXO || O1 <= 0 || !h || !dy || !w || !partial || !dh || !dw || !db ||
        h_sn < rows * J || dh_sn < rows * J || (h_sn % 4) || (dh_sn % 4) || ((uintptr_t)h % 16) || ((uintptr_t)dh % 16) ||
        dy_sn < rows * O1 || (dy2 && dy2_sn < rows) || w_sn < (int64_t)J * O ||
        dw_sn < (int64_t)J * O || db_sn < O)
        return NMARL_EINVAL;
    const int C = nmarl_fc_bwd_chunks(rows, N);
    const int64_t tiles = (rows + TILE - 1) / TILE;
    const int tpb = (int)((tiles + C - 1) / C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(thin_bwd_kernel, dim3(C, N), dim3(256), 0, st, rows, O, tpb, h, h_sn, dy, dy_sn, dy2, dy2_sn, w, w_sn, dh,
                       dh_sn, partial);
    hipLaunchKernelGGL(thin_bwd_reduce_kernel, dim3((65 * O + 255) / 256, N), dim3(256), 0, st, C, O, partial, dw, dw_sn, db, db_sn);
    return nmarl_check_launch();
}

extern "C" int nmarl_heads_loss(int64_t rows, int32_t N, int32_t H, int32_t A, const float* h, int64_t h_sn, const float* w, int64_t w_sn,
                                const float* b, int64_t b_sn, const float* va, const uint8_t* action, const float* adv, const float* R,
                                float v_coef, float e_coef, float* partial, float* loss_out, float* dy8, float* dv, float* dh,
                                int64_t dh_sn, float* dw, int64_t dw_sn, float* db, int64_t db_sn, void* stream) {
    const int O = A + 1;
    if (rows <= 0 || N <= 0 || H != J || A <= 0 || O > MAXO || !h || !w || !b || !va || !action || !adv || !R || !partial || !loss_out ||
        !dy8 || !dv || !dw || !db || h_sn < rows * J || (h_sn % 4) || ((uintptr_t)h % 16) || ((uintptr_t)dy8 % 16) ||
        (dh && (dh_sn < rows * J || (dh_sn % 4) || ((uintptr_t)dh % 16))) || w_sn < (int64_t)J * O || b_sn < O ||
        dw_sn < (int64_t)J * O || db_sn < O)
        return NMARL_EINVAL;
    const int C = nmarl_fc_bwd_chunks(rows, N);
    const int64_t tiles = (rows + TILE - 1) / TILE;
    const int tpb = (int)((tiles + C - 1) / C);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define NMARL_HL(DHF, OTV) hipLaunchKernelGGL((heads_loss_kernel<DHF, OTV>), dim3(C, N), dim3(256), 0, st, rows, N, A, tpb, h, h_sn, w, w_sn, \
                                              b, b_sn, va, action, adv, R, v_coef, e_coef, dy8, dv, dh, dh_sn, partial)
    if (dh) { if (O == 5) NMARL_HL(true, 5); else if (O == 6) NMARL_HL(true, 6); else NMARL_HL(true, 8); }
    else { if (O == 5) NMARL_HL(false, 5); else if (O == 6) NMARL_HL(false, 6); else NMARL_HL(false, 8); }
#undef NMARL_HL
    hipLaunchKernelGGL(heads_loss_reduce_kernel, dim3((65 * O + 3 + 255) / 256, N), dim3(256), 0, st, C, O, rows, v_coef, e_coef, partial,
                       dw, dw_sn, db, db_sn, loss_out);
    return nmarl_check_launch();
}

extern "C" int nmarl_nbr_action_value_fwd(int64_t rows, int32_t N, int32_t A, int32_t m_max, const int32_t* nbr_idx,
                                          const uint8_t* action, const float* w, int64_t w_sn, float* va,
                                          int32_t accumulate, void* stream) {
    if (rows < 0 || N <= 0 || A <= 0 || m_max <= 0 || w_sn < (int64_t)m_max * A || (rows > 0 && (!nbr_idx || !action || !w || !va)))
        return NMARL_EINVAL;
    if (rows == 0) return NMARL_OK;
    int64_t blocks = ((int64_t)N * rows + 255) / 256;
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(nbr_action_value_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), rows, N, A,
                       m_max, nbr_idx, action, w, w_sn, va, accumulate);
    return nmarl_check_launch();
}

extern "C" int nmarl_nbr_action_value_bwd(int64_t rows, int32_t N, int32_t A, int32_t m_max, const int32_t* nbr_idx,
                                          const uint8_t* action, const float* dv, float* partial, float* dw, int64_t dw_sn,
                                          void* stream) {
    const int W = m_max * A;
    if (rows <= 0 || N <= 0 || A <= 0 || m_max <= 0 || W > MAXW || dw_sn < W || !nbr_idx || !action || !dv || !partial || !dw)
        return NMARL_EINVAL;
    const int C = nmarl_fc_bwd_chunks(rows, N);
    const int rpb = (int)((rows + C - 1) / C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(nbr_action_value_bwd_kernel, dim3(C, N), dim3(256), 0, st, rows, N, A, m_max, rpb, nbr_idx, action, dv, partial);
    hipLaunchKernelGGL(nbr_action_value_reduce_kernel, dim3(N), dim3(64), 0, st, C, W, partial, dw, dw_sn);
    return nmarl_check_launch();
}
