// CACC platoon environment for E lock-stepped replicas on gfx950 (MI355X).
//
// Replaces envs/cacc_env.py of the reference: step (191-242), _get_accel
// (31-38), OVMCarFollowing (346-385), _constrain_speed (24-29), _get_reward
// (40-52), _get_state/_get_veh_state (54-79), reset/_init_* (166-189, 285-318).
//
// Mapping: one lane per (replica, vehicle); a replica is an aligned 8-lane
// group, so a wave64 steps 8 replicas.  Per-vehicle state h,v,u is [E,8] fp32
// SoA: a wave reads/writes 256 contiguous bytes per array (fully coalesced).
// The line-graph neighbourhood (vehicle i-1 / i+1) is exchanged with wave
// shuffles of width 8; min(h) (collision) and sum(reward) are 3-stage xor
// butterflies over the 8-lane group -- the same tree numpy's pairwise sum
// uses for 8 elements, so the fp32 sum order matches the oracle.  The
// gathered observation [E,8,15] (own + 2 neighbour slots) is staged through
// LDS so the wave writes its 3840 contiguous bytes as 16-byte stores.
//
// HBM-bound, no contraction: no MFMA.  Algorithmic bytes per replica-step are
// listed in DESIGN.md (B_alg).  Compiled with -ffp-contract=off so the fp32
// arithmetic follows the oracle operation by operation.
#include "common.h"
#include "cacc_tile.h"
#include <stdlib.h>

// ---- tuning knobs (defaults = shipped configuration; tools/ab_env.py builds variants with -D...)
#ifndef NMARL_CACC_GRIDCAP
#define NMARL_CACC_GRIDCAP (256 * 32)      // resident blocks, grid-stride beyond (A/B: 16 -> 32: +3 %)
#endif
#ifndef NMARL_CACC_BLOCK_LARGE
#define NMARL_CACC_BLOCK_LARGE 256
#endif
#ifndef NMARL_CACC_NT_LARGE
#define NMARL_CACC_NT_LARGE 2              // non-temporal stores when the working set exceeds the caches:
#endif                                     //   1 = observation slab, 2 = also h, v, u.  A/B at E = 2^21
#ifndef NMARL_CACC_NT_SMALL                //   (tools/ab_env.py): NT 0 / 1 / 2 = 363 / 303 / 209 us per launch
#define NMARL_CACC_NT_SMALL 0              // small E lives in L2 / Infinity Cache between steps: plain stores
#endif
#ifndef NMARL_CACC_BLOCK_SMALL
#define NMARL_CACC_BLOCK_SMALL 64
#endif
#ifndef NMARL_CACC_NOOBS
#define NMARL_CACC_NOOBS 0                 // diagnostic only: skip the slab store
#endif

namespace {

using namespace nmarl_cacc;

template <int BLOCK, int NT, bool COMPACT>
__global__ __launch_bounds__(BLOCK) void cacc_step_kernel(
    const nmarl_cacc_params_t p, const int64_t E, const uint8_t* __restrict__ action,
    float* __restrict__ hs, float* __restrict__ vs, float* __restrict__ us,
    int32_t* __restrict__ ts, uint8_t* __restrict__ coll, float* __restrict__ v0_init,
    float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
    float* __restrict__ greward, const int auto_reset, const uint64_t seed,
    const int64_t env_id_base, int32_t* __restrict__ episode) {
    constexpr int W = COMPACT ? NF : NOBS;
    __shared__ __attribute__((aligned(16))) float lds[BLOCK * W];
    const int lane = threadIdx.x & (NMARL_WAVE - 1);
    const int wave = threadIdx.x / NMARL_WAVE;
    float* lds_wave = lds + wave * NMARL_WAVE * W;
    const int64_t n_lanes = E * N;
    const int64_t waves_total = (n_lanes + NMARL_WAVE - 1) / NMARL_WAVE;
    const int64_t wave_stride = (int64_t)gridDim.x * (BLOCK / NMARL_WAVE);

    // The inputs of the NEXT tile of this wave are loaded before the current one is computed (one tile = 64 lanes = 8
    // replicas): the step is a dependent load -> compute -> store chain, and at large E its HBM rate is set by the
    // bytes in flight.  The prefetch index is clamped, so the loads stay unconditional.
    int64_t w = (int64_t)blockIdx.x * (BLOCK / NMARL_WAVE) + wave;
    float h_n = 0.f, v_n = 0.f, v0i_n = 0.f, h_m = 0.f, v_m = 0.f, v0i_m = 0.f;
    int act_n = 0, t_n = 0, coll_n = 0, act_m = 0, t_m = 0, coll_m = 0;
#define NMARL_CACC_LOAD(wt, S)                                                             \
    {                                                                                      \
        const int64_t wc_ = (wt) < waves_total ? (wt) : waves_total - 1;                   \
        const int64_t gid_ = wc_ * NMARL_WAVE + lane;                                      \
        const int64_t g_ = gid_ < n_lanes ? gid_ : n_lanes - 1;                            \
        h_##S = hs[g_]; v_##S = vs[g_]; act_##S = action[g_];                              \
        t_##S = ts[g_ >> 3]; coll_##S = coll[g_ >> 3]; v0i_##S = v0_init[g_ >> 3];         \
    }
    // two tiles ahead: (n) = tile w, (m) = tile w + stride; a clamped index re-reads the wave's last tile, whose
    // values are then never used
    if (w < waves_total) {
        NMARL_CACC_LOAD(w, n)
        NMARL_CACC_LOAD(w + wave_stride, m)
    }
    for (; w < waves_total; w += wave_stride) {
        float h = h_n, v = v_n;
        const int act = act_n;
        const int t = t_n;
        const bool collided = coll_n != 0;
        const float v0i = v0i_n;
        h_n = h_m; v_n = v_m; act_n = act_m; t_n = t_m; coll_n = coll_m; v0i_n = v0i_m;
        NMARL_CACC_LOAD(w + 2 * wave_stride, m)
        cacc_tile<NT, COMPACT>(p, n_lanes, w, lane, h, v, act, t, collided, v0i, hs, vs, us, ts, coll, v0_init, obs, reward, done,
                               greward, auto_reset, seed, env_id_base, episode, lds_wave);
    }
}

// The HBM-regime form (round 5): four vehicles per lane, 16-byte accesses (cacc_quad, csrc/cacc_tile.h).  A wave steps 32 replicas per
// tile; the inputs of its next two tiles are in flight while it computes (unconditional, clamped loads).  Compact observation only.
template <int BLOCK, int NT>
__global__ __launch_bounds__(BLOCK) void cacc_step4_kernel(
    const nmarl_cacc_params_t p, const int64_t E, const uint8_t* __restrict__ action,
    float* __restrict__ hs, float* __restrict__ vs, float* __restrict__ us,
    int32_t* __restrict__ ts, uint8_t* __restrict__ coll, float* __restrict__ v0_init,
    float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
    float* __restrict__ greward, const int auto_reset, const uint64_t seed,
    const int64_t env_id_base, int32_t* __restrict__ episode) {
    __shared__ __attribute__((aligned(16))) float lds[BLOCK * 4 * NF];           // the wave's observation slab: 64 lanes x 20 floats
    const int lane = threadIdx.x & (NMARL_WAVE - 1);
    const int wave = threadIdx.x / NMARL_WAVE;
    float* lds_wave = lds + wave * NMARL_WAVE * 4 * NF;
    constexpr int REPS = NMARL_WAVE / 2;                     // replicas per wave tile
    const int64_t tiles_total = (E + REPS - 1) / REPS;
    const int64_t stride = (int64_t)gridDim.x * (BLOCK / NMARL_WAVE);
    const int half = lane & 1;
    int64_t w = (int64_t)blockIdx.x * (BLOCK / NMARL_WAVE) + wave;
    float4 h_n, v_n, h_m, v_m;
    uint32_t a_n = 0, a_m = 0;
    int t_n = 0, t_m = 0, c_n = 0, c_m = 0;
    float z_n = 0.f, z_m = 0.f;
#define NMARL_CACC4_LOAD(wt, S)                                                            \
    {                                                                                      \
        const int64_t wc_ = (wt) < tiles_total ? (wt) : tiles_total - 1;                   \
        const int64_t er_ = wc_ * REPS + (lane >> 1);                                      \
        const int64_t e_ = er_ < E ? er_ : E - 1;                                          \
        const int64_t g_ = e_ * N + 4 * half;                                              \
        h_##S = *reinterpret_cast<const float4*>(hs + g_); v_##S = *reinterpret_cast<const float4*>(vs + g_); \
        a_##S = *reinterpret_cast<const uint32_t*>(action + g_);                           \
        t_##S = ts[e_]; c_##S = coll[e_]; z_##S = v0_init[e_];                             \
    }
    if (w < tiles_total) {
        NMARL_CACC4_LOAD(w, n)
        NMARL_CACC4_LOAD(w + stride, m)
    }
    for (; w < tiles_total; w += stride) {
        const float4 h4 = h_n, v4 = v_n;
        const uint32_t a4 = a_n;
        const int t = t_n;
        const bool collided = c_n != 0;
        const float v0i = z_n;
        h_n = h_m; v_n = v_m; a_n = a_m; t_n = t_m; c_n = c_m; z_n = z_m;
        NMARL_CACC4_LOAD(w + 2 * stride, m)
        cacc_quad<NT>(p, E, w * REPS + (lane >> 1), half, h4, v4, a4, t, collided, v0i, hs, vs, us, ts, coll, v0_init, obs, reward, done,
                      greward, auto_reset, seed, env_id_base, episode, lds_wave, lane, w * REPS);
    }
#undef NMARL_CACC4_LOAD
}

template <int BLOCK, bool COMPACT>
__global__ __launch_bounds__(BLOCK) void cacc_reset_kernel(
    const nmarl_cacc_params_t p, const int64_t E, const uint8_t* __restrict__ mask,
    const float* __restrict__ u0, const uint64_t seed, const int64_t env_id_base,
    int32_t* __restrict__ episode, float* __restrict__ hs, float* __restrict__ vs,
    float* __restrict__ us, int32_t* __restrict__ ts, uint8_t* __restrict__ coll,
    float* __restrict__ v0_init, float* __restrict__ obs, float* __restrict__ fp, const int A) {
    constexpr int W = COMPACT ? NF : NOBS;
    __shared__ __attribute__((aligned(16))) float lds[BLOCK * W];
    const int lane = threadIdx.x & (NMARL_WAVE - 1);
    const int wave = threadIdx.x / NMARL_WAVE;
    float* lds_wave = lds + wave * NMARL_WAVE * W;
    const int64_t n_lanes = E * N;
    const int64_t waves_total = (n_lanes + NMARL_WAVE - 1) / NMARL_WAVE;
    const int64_t wave_stride = (int64_t)gridDim.x * (BLOCK / NMARL_WAVE);

    for (int64_t w = (int64_t)blockIdx.x * (BLOCK / NMARL_WAVE) + wave; w < waves_total; w += wave_stride) {
        const int64_t gid = w * NMARL_WAVE + lane;
        const bool valid = gid < n_lanes;
        const int64_t g = valid ? gid : n_lanes - 1;
        const int64_t e = g >> 3;
        const int a = (int)(g & 7);
        const bool sel = mask == nullptr || mask[e] != 0;

        float h = hs[g], v = vs[g], u = us[g], v0i = v0_init[e];
        int t = ts[e];
        if (sel) {
            float U;
            if (u0 != nullptr) {
                U = u0[e];
            } else {
                const int ep = episode[e];
                U = reset_uniform(seed, env_id_base + e, ep);
                if (valid && a == 0) episode[e] = ep + 1;
            }
            init_state(p, U, a, h, v, v0i);
            u = 0.0f; t = 0;
            if (valid) {
                hs[g] = h; vs[g] = v; us[g] = 0.0f;
                if (a == 0) { ts[e] = 0; coll[e] = 0; v0_init[e] = v0i; }
                if (fp != nullptr) {
                    const float q = 1.0f / (float)A;                    // :184
                    for (int k = 0; k < A; ++k) fp[g * A + k] = q;
                }
            }
        }
        const float up_v = __shfl_up(v, 1, N);
        const float v_lead = a == 0 ? lead_speed(p, v0i, t) : up_v;
        const int64_t lanes_here = n_lanes - w * NMARL_WAVE;
        const int n_valid = lanes_here >= NMARL_WAVE ? NMARL_WAVE : (int)lanes_here;
        // the slab of a wave is rewritten as a whole; unselected replicas re-emit
        // their current observation (same values), so no read-modify-write is needed
        __builtin_amdgcn_wave_barrier();
        emit_obs<0, COMPACT>(p, h, v, u, v_lead, a, valid, lane, lds_wave, obs + w * NMARL_WAVE * W, n_valid);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Lock-step tail of the batched rollout in ONE launch: the env step AND the next lock-step's input encoders
// (fc of policies.py:145 / 176-181, w_ob / w_fp of agents/utils.py:186-199) -- the observation the step produces never
// leaves the CU before it is encoded (it is also written once, compact, for the update).  Saves one launch and the
// ~9 us latency chain of the separate encoder kernel per lock-step; same arithmetic as nmarl_fc_fwd_multi, bit for bit.
//
// Block = 256 threads = 8 replicas.  Phase 1: wave 0 steps them (cacc_tile) while wave 2 fetches the replicas' previous-step
// policies (the fingerprints) into LDS; all waves load their encoder weights.  Phase 2: thread =
// (layer 0: observation / 1: fingerprints, agent, 4 output columns): its 15 (8) x 4 weights sit in registers, the inputs
// are LDS broadcasts ([own | neighbours] gathered through the neighbour table), 8 rows per thread, float4 stores into
// columns [64 layer, 64 layer + 64) of the LSTM input.
struct EncodeArgs {
    const float *w_ob, *b_ob, *w_fp, *b_fp, *fp;
    const int32_t* nbr_idx;
    float* out;
    int64_t w_ob_sn, b_ob_sn, w_fp_sn, b_fp_sn, fp_sn, out_sn, out_row;
    int32_t act, n_parts;
};
constexpr int ENC_REPS = 8;           // replicas per block (one wave steps them; 2 blocks per CU overlap their phases)
constexpr int ENC_J = 64;             // outputs per layer
constexpr int ENC_A = 4;              // actions (fingerprint width per neighbour)

__device__ __forceinline__ float enc_act(float x, int act) { return act == 1 ? fmaxf(x, 0.0f) : (act == 2 ? tanhf(x) : x); }

__global__ __launch_bounds__(256) void cacc_step_encode_kernel(
    const nmarl_cacc_params_t p, const int64_t E, const uint8_t* __restrict__ action,
    float* __restrict__ hs, float* __restrict__ vs, float* __restrict__ us,
    int32_t* __restrict__ ts, uint8_t* __restrict__ coll, float* __restrict__ v0_init,
    float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
    float* __restrict__ greward, const int auto_reset, const uint64_t seed,
    const int64_t env_id_base, int32_t* __restrict__ episode, const EncodeArgs en) {
    __shared__ __attribute__((aligned(16))) float lds_obs[ENC_REPS * N * NF];        // [16 replicas][8 vehicles][5]
    __shared__ __attribute__((aligned(16))) float lds_fp[ENC_REPS * N * ENC_A];      // [16][8 agents][4]
    const int lane = threadIdx.x & (NMARL_WAVE - 1);
    const int wave = threadIdx.x / NMARL_WAVE;
    const int64_t n_lanes = E * N;
    const int64_t e0 = (int64_t)blockIdx.x * ENC_REPS;
    // ---- encoder role of this thread and its weights (registers; issued before the env phase, landed after it)
    const int part = threadIdx.x >> 7, ag = (threadIdx.x >> 4) & 7, j4 = (threadIdx.x & 15) * 4;
    const bool has_part = part < en.n_parts;
    const int n0 = en.nbr_idx[ag * 2], n1 = en.nbr_idx[ag * 2 + 1];
    float4 wq[3 * NF];
    float4 bq;
    if (part == 0) {
        const float* wp = en.w_ob + (int64_t)ag * en.w_ob_sn + j4;
#pragma unroll
        for (int f = 0; f < 3 * NF; ++f) wq[f] = *reinterpret_cast<const float4*>(wp + f * ENC_J);
        bq = *reinterpret_cast<const float4*>(en.b_ob + (int64_t)ag * en.b_ob_sn + j4);
    } else {
        const float* wp = has_part ? en.w_fp + (int64_t)ag * en.w_fp_sn + j4 : en.w_ob + j4;
#pragma unroll
        for (int f = 0; f < 2 * ENC_A; ++f) wq[f] = *reinterpret_cast<const float4*>(wp + f * ENC_J);
#pragma unroll
        for (int f = 2 * ENC_A; f < 3 * NF; ++f) wq[f] = float4{0.f, 0.f, 0.f, 0.f};
        bq = *reinterpret_cast<const float4*>((has_part ? en.b_fp + (int64_t)ag * en.b_fp_sn : en.b_ob) + j4);
    }
    // ---- phase 1: env step (wave 0) | fingerprint tile (wave 2)
    if (wave < ENC_REPS / 8) {
        const int64_t w = (int64_t)blockIdx.x * (ENC_REPS / 8) + wave;   // wave tile index: 8 replicas
        const int64_t gid = w * NMARL_WAVE + lane;
        const int64_t g = gid < n_lanes ? gid : n_lanes - 1;
        if (w * NMARL_WAVE < n_lanes) {
            const float h = hs[g], v = vs[g];
            const int act = action[g], t = ts[g >> 3];
            const bool collided = coll[g >> 3] != 0;
            const float v0i = v0_init[g >> 3];
            cacc_tile<0, true>(p, n_lanes, w, lane, h, v, act, t, collided, v0i, hs, vs, us, ts, coll, v0_init, obs, reward, done,
                               greward, auto_reset, seed, env_id_base, episode, lds_obs + wave * NMARL_WAVE * NF);
        }
    } else if (en.n_parts > 1 && threadIdx.x >= 128 && threadIdx.x < 128 + N * ENC_REPS) {
        const int i = threadIdx.x - 128;                                 // (agent, replica of the block): one float4 each
        const int a2 = i / ENC_REPS, r = i % ENC_REPS;
        const int64_t e = e0 + r;
        const float4 v4 = *reinterpret_cast<const float4*>(en.fp + (int64_t)a2 * en.fp_sn + (e < E ? e : E - 1) * ENC_A);
        *reinterpret_cast<float4*>(lds_fp + (r * N + a2) * ENC_A) = v4;
    }
    __syncthreads();
    // ---- phase 2: encoders of the block's replicas
    if (!has_part) return;
    float* outp = en.out + (int64_t)ag * en.out_sn + part * ENC_J + j4;
#pragma unroll 4
    for (int r = 0; r < ENC_REPS; ++r) {
        const int64_t e = e0 + r;
        float x[3 * NF];
        if (part == 0) {
            const float* o = lds_obs + r * N * NF;
#pragma unroll
            for (int k = 0; k < NF; ++k) {
                x[k] = o[ag * NF + k];
                x[NF + k] = n0 >= 0 ? o[n0 * NF + k] : 0.0f;
                x[2 * NF + k] = n1 >= 0 ? o[n1 * NF + k] : 0.0f;
            }
        } else {
            const float* q = lds_fp + r * N * ENC_A;
#pragma unroll
            for (int k = 0; k < ENC_A; ++k) {
                x[k] = n0 >= 0 ? q[n0 * ENC_A + k] : 0.0f;
                x[ENC_A + k] = n1 >= 0 ? q[n1 * ENC_A + k] : 0.0f;
            }
#pragma unroll
            for (int k = 2 * ENC_A; k < 3 * NF; ++k) x[k] = 0.0f;
        }
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;        // the ascending fmaf chain of nmarl_fc_fwd_multi
        if (part == 0) {
#pragma unroll
            for (int f = 0; f < 3 * NF; ++f) {
                a0 = fmaf(x[f], wq[f].x, a0); a1 = fmaf(x[f], wq[f].y, a1); a2 = fmaf(x[f], wq[f].z, a2); a3 = fmaf(x[f], wq[f].w, a3);
            }
        } else {
#pragma unroll
            for (int f = 0; f < 2 * ENC_A; ++f) {
                a0 = fmaf(x[f], wq[f].x, a0); a1 = fmaf(x[f], wq[f].y, a1); a2 = fmaf(x[f], wq[f].z, a2); a3 = fmaf(x[f], wq[f].w, a3);
            }
        }
        if (e < E)
            *reinterpret_cast<float4*>(outp + e * en.out_row) =
                float4{enc_act(a0 + bq.x, en.act), enc_act(a1 + bq.y, en.act), enc_act(a2 + bq.z, en.act), enc_act(a3 + bq.w, en.act)};
    }
}

inline int pick_grid(int64_t E, int block) {
    const int64_t waves = (E * N + NMARL_WAVE - 1) / NMARL_WAVE;
    const int64_t blocks = (waves + block / NMARL_WAVE - 1) / (block / NMARL_WAVE);
    const int64_t cap = NMARL_CACC_GRIDCAP;
    return (int)(blocks < cap ? blocks : cap);
}

bool params_ok(const nmarl_cacc_params_t* p) {
    return p != nullptr && p->T > 0 && p->batch_size > 0 && (p->scenario == 0 || p->scenario == 1) &&
           (p->compact_obs == 0 || p->compact_obs == 1) &&
           p->dt > 0.f && p->h_g > p->h_s && p->u_max != 0.f && p->v_star != 0.f && p->h_star != 0.f;
}

}  // namespace

extern "C" int nmarl_abi_version(void) { return 1; }

#ifndef NMARL_SRC_HASH_STR
#define NMARL_SRC_HASH_STR "NMARL_SRC_HASH=unknown"
#endif
// "NMARL_SRC_HASH=<sha256 prefix of csrc/* + include/nmarl.h>", set by deeprl_network_amd/build.py
extern "C" const char* nmarl_source_hash(void) { return NMARL_SRC_HASH_STR; }

extern "C" int nmarl_cacc_step(const nmarl_cacc_params_t* p, int64_t E, const uint8_t* action,
                               float* h, float* v, float* u, int32_t* t, uint8_t* collided,
                               float* v0_init, float* obs, float* reward, uint8_t* done,
                               float* global_reward, int32_t auto_reset, uint64_t seed,
                               int64_t env_id_base, int32_t* episode, void* stream) {
    if (!params_ok(p) || E < 0 || (E > 0 && (!action || !h || !v || !u || !t || !collided || !v0_init ||
                                             !obs || !reward || !done || !global_reward)))
        return NMARL_EINVAL;
    if (auto_reset && !episode) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // small E is latency bound and cache resident: 1-wave blocks spread the replicas over more CUs and
    // plain stores keep the state in L2 for the next step; large E streams (non-temporal stores)
#define NMARL_CACC_LAUNCH(BLK, NTV, CMP)                                                                              \
    hipLaunchKernelGGL((cacc_step_kernel<BLK, NTV, CMP>), dim3(pick_grid(E, BLK)), dim3(BLK), 0, s, *p, E, action, h, v, u, t, \
                       collided, v0_init, obs, reward, done, global_reward, auto_reset, seed, env_id_base, episode)
    if (E * N <= 256 * 4 * NMARL_WAVE) {
        if (p->compact_obs) NMARL_CACC_LAUNCH(NMARL_CACC_BLOCK_SMALL, NMARL_CACC_NT_SMALL, true);
        else NMARL_CACC_LAUNCH(NMARL_CACC_BLOCK_SMALL, NMARL_CACC_NT_SMALL, false);
    } else {
        // HBM regime: the compact layout takes the four-vehicles-per-lane form (16-byte accesses; NMARL_CACC_QUAD=0: the lane-per-
        // vehicle form, for A/B and for the test that compares the two bit for bit)
        const char* qe = getenv("NMARL_CACC_QUAD");
        const bool quad = !(qe && qe[0] == '0');
        if (p->compact_obs && quad && ((uintptr_t)h % 16) == 0 && ((uintptr_t)v % 16) == 0 && ((uintptr_t)u % 16) == 0 &&
            ((uintptr_t)obs % 16) == 0 && ((uintptr_t)action % 4) == 0 && (!p->per_agent_reward || ((uintptr_t)reward % 16) == 0)) {
            const int64_t tiles = (E + 31) / 32;
            const int64_t blocks = (tiles + 3) / 4;
            hipLaunchKernelGGL((cacc_step4_kernel<256, NMARL_CACC_NT_LARGE>), dim3((unsigned)(blocks < NMARL_CACC_GRIDCAP ? blocks : NMARL_CACC_GRIDCAP)),
                               dim3(256), 0, s, *p, E, action, h, v, u, t, collided, v0_init, obs, reward, done, global_reward, auto_reset, seed,
                               env_id_base, episode);
        } else if (p->compact_obs) NMARL_CACC_LAUNCH(NMARL_CACC_BLOCK_LARGE, NMARL_CACC_NT_LARGE, true);
        else NMARL_CACC_LAUNCH(NMARL_CACC_BLOCK_LARGE, NMARL_CACC_NT_LARGE, false);
    }
#undef NMARL_CACC_LAUNCH
    return nmarl_check_launch();
}

extern "C" int nmarl_cacc_step_encode(const nmarl_cacc_params_t* p, int64_t E, const uint8_t* action,
                                      float* h, float* v, float* u, int32_t* t, uint8_t* collided,
                                      float* v0_init, float* obs, float* reward, uint8_t* done,
                                      float* global_reward, int32_t auto_reset, uint64_t seed,
                                      int64_t env_id_base, int32_t* episode, const nmarl_cacc_encode_t* enc, void* stream) {
    if (!params_ok(p) || !p->compact_obs || E < 0 || !enc ||
        (E > 0 && (!action || !h || !v || !u || !t || !collided || !v0_init || !obs || !reward || !done || !global_reward)))
        return NMARL_EINVAL;
    if (auto_reset && !episode) return NMARL_EINVAL;
    if (enc->n_parts < 1 || enc->n_parts > 2 || enc->act < 0 || enc->act > 2 || !enc->w_ob || !enc->b_ob || !enc->nbr_idx || !enc->out ||
        enc->w_ob_sn < 3 * NF * 64 || (enc->w_ob_sn % 4) || enc->b_ob_sn < 64 || (enc->b_ob_sn % 4) || enc->out_row < 64 * enc->n_parts ||
        (enc->out_row % 4) || (enc->out_sn % 4) || ((uintptr_t)enc->w_ob % 16) || ((uintptr_t)enc->b_ob % 16) || ((uintptr_t)enc->out % 16))
        return NMARL_EINVAL;
    if (enc->n_parts == 2 && (!enc->w_fp || !enc->b_fp || !enc->fp || enc->w_fp_sn < 2 * 4 * 64 || (enc->w_fp_sn % 4) || enc->b_fp_sn < 64 ||
                              (enc->b_fp_sn % 4) || enc->fp_sn < E * 4 || (enc->fp_sn % 4) || ((uintptr_t)enc->w_fp % 16) ||
                              ((uintptr_t)enc->b_fp % 16) || ((uintptr_t)enc->fp % 16)))
        return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    EncodeArgs en{};
    en.w_ob = enc->w_ob; en.b_ob = enc->b_ob; en.w_fp = enc->w_fp; en.b_fp = enc->b_fp; en.fp = enc->fp; en.nbr_idx = enc->nbr_idx;
    en.out = enc->out; en.w_ob_sn = enc->w_ob_sn; en.b_ob_sn = enc->b_ob_sn; en.w_fp_sn = enc->w_fp_sn; en.b_fp_sn = enc->b_fp_sn;
    en.fp_sn = enc->fp_sn; en.out_sn = enc->out_sn; en.out_row = enc->out_row; en.act = enc->act; en.n_parts = enc->n_parts;
    hipLaunchKernelGGL(cacc_step_encode_kernel, dim3((unsigned)((E + ENC_REPS - 1) / ENC_REPS)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *p, E, action, h, v, u, t, collided, v0_init, obs, reward, done, global_reward,
                       auto_reset, seed, env_id_base, episode, en);
    return nmarl_check_launch();
}

extern "C" int nmarl_cacc_reset(const nmarl_cacc_params_t* p, int64_t E, const uint8_t* mask,
                                const float* u0, uint64_t seed, int64_t env_id_base, int32_t* episode,
                                float* h, float* v, float* u, int32_t* t, uint8_t* collided,
                                float* v0_init, float* obs, float* fp, int32_t A, void* stream) {
    if (!params_ok(p) || E < 0 || (E > 0 && (!h || !v || !u || !t || !collided || !v0_init || !obs)))
        return NMARL_EINVAL;
    if (!u0 && !episode) return NMARL_EINVAL;
    if (fp && A <= 0) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (p->compact_obs)
        hipLaunchKernelGGL((cacc_reset_kernel<256, true>), dim3(pick_grid(E, 256)), dim3(256), 0, s, *p, E, mask, u0, seed,
                           env_id_base, episode, h, v, u, t, collided, v0_init, obs, fp, A);
    else
        hipLaunchKernelGGL((cacc_reset_kernel<256, false>), dim3(pick_grid(E, 256)), dim3(256), 0, s, *p, E, mask, u0, seed,
                           env_id_base, episode, h, v, u, t, collided, v0_init, obs, fp, A);
    return nmarl_check_launch();
}
