// The CACC platoon step of one wave tile (8 replicas x 8 vehicles = 64 lanes) as DEVICE functions shared by the env kernels
// (csrc/cacc.hip) and by the lock-step kernel that steps the env behind its action draw (csrc/lstm_mfma.hip, ENC 2).
// envs/cacc_env.py of the reference: step (191-242), _get_accel (31-38), OVMCarFollowing (346-385), _constrain_speed
// (24-29), _get_reward (40-52), _get_state / _get_veh_state (54-79), _init_* (285-318).
#pragma once
#include "common.h"

#ifndef NMARL_CACC_NOOBS
#define NMARL_CACC_NOOBS 0                 // diagnostic only: skip the slab store
#endif

namespace nmarl_cacc {

constexpr int N = NMARL_CACC_N;      // 8
constexpr int NF = NMARL_CACC_NF;    // 5
constexpr int NOBS = NMARL_CACC_OBS; // 15
constexpr int DECEL_STEPS = 300;     // cacc_env.py:317
constexpr float PI_F = 3.14159265358979323846f;

// v0s[t], cacc_env.py:299 / 316-318 (np.linspace(v_init, v*, 300) then v*)
__device__ __forceinline__ float lead_speed(const nmarl_cacc_params_t& p, float v0_init, int t) {
    if (p.scenario == 0) return p.v_star;
    const float step = (p.v_star - v0_init) / (float)(DECEL_STEPS - 1);
    const float ramp = (float)t * step + v0_init;
    return t >= DECEL_STEPS - 1 ? p.v_star : ramp;
}

// fp32 cos on [0, pi] (the only range the OVM ramp produces): cos(t) = -sin(t - pi/2) with a
// two-piece pi/2 and an odd polynomial to r^13 (truncation 7e-10 at |r| = pi/2, i.e. < 1 ulp).
// libm's cosf carries a large-argument reduction path that costs ~25 VGPRs this kernel never needs.
__device__ __forceinline__ float cosf_0_pi(float t) {
    const float r = (t - 1.57079637050628662109375f) + 4.37113900018624283e-8f;
    const float r2 = r * r;
    float q = 1.6059044e-10f;                 //  1/13!
    q = fmaf(q, r2, -2.5052108e-08f);         // -1/11!
    q = fmaf(q, r2, 2.7557319e-06f);          //  1/9!
    q = fmaf(q, r2, -1.9841270e-04f);         // -1/7!
    q = fmaf(q, r2, 8.3333333e-03f);          //  1/5!
    q = fmaf(q, r2, -1.6666667e-01f);         // -1/3!
    return -(r + r * r2 * q);                 // -sin r
}

// OVMCarFollowing.get_vh, cacc_env.py:360-369
__device__ __forceinline__ float ovm_vh(const nmarl_cacc_params_t& p, float h) {
    const float mid = p.v_max / 2.0f * (1.0f - cosf_0_pi(PI_F * (h - p.h_s) / (p.h_g - p.h_s)));
    return h <= p.h_s ? 0.0f : (h < p.h_g ? mid : p.v_max);
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// float64 cos on [0, pi] (the OVM ramp argument) without libm's register-hungry general path:
// cos(t) = -sin(t - pi/2), pi/2 subtracted in two pieces so that cos(double(pi/2)) = 6.123e-17 like
// libm / NumPy; odd Taylor polynomial to r^21 (|r| <= pi/2: truncation < 3e-16).
static __constant__ double c_sin_taylor[10] = {
    -1.9572941063391263e-20, 8.22063524662433e-18, -2.8114572543455206e-15, 7.647163731819816e-13,
    -1.6059043836821613e-10, 2.505210838544172e-08, -2.7557319223985893e-06, 1.984126984126984e-04,
    -8.333333333333333e-03, 1.6666666666666666e-01};     // -1/21!, 1/19!, ..., -1/5!, 1/3!

// Rarely executed (equilibrium states only): a rolled loop over scalar-loaded coefficients costs ~6
// VGPRs instead of ~26 (an unrolled fp64 Horner chain keeps every literal in a register pair, which
// cost the whole kernel two waves per SIMD of occupancy).
__device__ __noinline__ double cos_0_pi(double t) {
    const double r = (t - 1.5707963267948966) - 6.123233995736766e-17;
    const double r2 = r * r;
    double q = c_sin_taylor[0];
#pragma unroll 1
    for (int k = 1; k < 10; ++k) q = fma(q, r2, c_sin_taylor[k]);
    return -(r - r * r2 * q);                             // -sin r
}

// _get_veh_state (cacc_env.py:54-65): the five features of this lane's vehicle.
__device__ __forceinline__ void obs_features(const nmarl_cacc_params_t& p, float h, float v, float u, float v_lead, float (&x)[NF]) {
    x[0] = (v - p.v_star) / p.v_star;
    x[1] = clampf((v_lead - v) / 5.0f, -2.0f, 2.0f);
    // The platoon's equilibrium (h = h*, v = v*) makes vh(h) - v vanish, and the reference's float64
    // cos(pi/2) = 6.1e-17 leaves this feature a NEGATIVE 3.6e-16; an fp32 cosf gives +1.3e-7.  The
    // feature multiplies O(1) weights into a relu whose mask (hence the bias gradient) depends on that
    // sign, so where the fp32 difference is within its own rounding noise (sign not trustworthy) the term is
    // re-evaluated in float64 like the reference (cacc_env.py:58-59, 365-366).  Only near-exact equilibria.
    {
        const float d32 = ovm_vh(p, h) - v;
        x[2] = clampf(d32 / 5.0f, -2.0f, 2.0f);
        if (fabsf(d32) < 2e-5f) {      // ~10x the fp32 error of vh - v (cos poly 1.2e-7 * 15, ulp(15) = 9.5e-7)
            const double hd = (double)h;
            const double th = 3.141592653589793 * (hd - (double)p.h_s) / ((double)p.h_g - (double)p.h_s);
            const double mid = (double)p.v_max / 2.0 * (1.0 - cos_0_pi(th));
            const double vh64 = hd <= (double)p.h_s ? 0.0 : (hd < (double)p.h_g ? mid : (double)p.v_max);
            x[2] = (float)((vh64 - (double)v) / 5.0);
        }
    }
    x[3] = (h + (v_lead - v) * p.dt - p.h_star) / p.h_star;
    x[4] = u / p.u_max;
}

// _get_veh_state (cacc_env.py:54-65) for this lane's vehicle (obs_features), then the
// neighbour gather and the LDS-staged coalesced store of the wave's slab.
// COMPACT: only the vehicle's own 5 features are written ([E,8,5]: SURVEY.md 8d's 41 N + 19 B layout); the policy's
// encoder gathers the neighbours' features itself (nmarl_fc_fwd_multi with a neighbour table whose slot 0 is the
// agent).  Otherwise the 'ia2c' pre-gathered observation [E,8,15] of cacc_env.py:70-73.
template <int NT, bool COMPACT>
__device__ __forceinline__ void emit_obs(const nmarl_cacc_params_t& p, float h, float v, float u,
                                         float v_lead, int a, bool valid, int lane, float* lds_wave,
                                         float* __restrict__ obs_wave, int n_valid_lanes) {
    constexpr int W = COMPACT ? NF : NOBS;
    float x[NF];
    obs_features(p, h, v, u, v_lead, x);
    float* row = lds_wave + lane * W;
#pragma unroll
    for (int k = 0; k < NF; ++k) {
        row[k] = x[k];
        if (!COMPACT) {
            const float lo = __shfl_up(x[k], 1, N);     // vehicle a-1
            const float hi = __shfl_down(x[k], 1, N);   // vehicle a+1
            // slots hold the neighbours in ascending index, left-packed (cacc_env.py:72)
            const float s1 = a == 0 ? hi : lo;
            const float s2 = (a == 0 || a == N - 1) ? 0.0f : hi;
            row[NF + k] = s1;
            row[2 * NF + k] = s2;
        }
    }
    __builtin_amdgcn_wave_barrier();
    // 64 lanes x W floats = 240 (80) float4, contiguous in HBM
    const float4* src = reinterpret_cast<const float4*>(lds_wave);
    float4* dst = reinterpret_cast<float4*>(obs_wave);
    const int n_vec = n_valid_lanes * W / 4;     // n_valid_lanes is a multiple of 8 -> exact
#pragma unroll
    for (int i = 0; i < (COMPACT ? 2 : 4); ++i) {
        const int idx = i * NMARL_WAVE + lane;
#if NMARL_CACC_NOOBS
        (void)dst;
#else
        if (idx < n_vec) {
            if (NT >= 1) {
                const float4 val = src[idx];
                __builtin_nontemporal_store(val.x, &dst[idx].x); __builtin_nontemporal_store(val.y, &dst[idx].y);
                __builtin_nontemporal_store(val.z, &dst[idx].z); __builtin_nontemporal_store(val.w, &dst[idx].w);
            } else {
                dst[idx] = src[idx];
            }
        }
#endif
    }
    (void)valid;
}

// initial condition, cacc_env.py:285-318
__device__ __forceinline__ void init_state(const nmarl_cacc_params_t& p, float U, int a,
                                           float& h, float& v, float& v0i) {
    h = p.h_star; v = p.v_star; v0i = p.v_star;
    if (p.scenario == 0) {
        if (a == 0) h = p.h_star * (1.5f + U);          // :294
    } else {
        v = p.v_star * (1.5f + U);                      // :314
        v0i = v;                                        // :317
    }
}

__device__ __forceinline__ float reset_uniform(uint64_t seed, int64_t env_id, int episode) {
    const Philox4 r = philox4x32_10((uint32_t)env_id, 0u, (uint32_t)episode, NMARL_STREAM_RESET,
                                    (uint32_t)seed, (uint32_t)(seed >> 32));
    return u01_from_bits(r.x);
}

// CACCEnv.step (cacc_env.py:191-242) for this lane's vehicle a of the platoon held by its aligned 8-lane group: new (h, v, u), the
// sticky collision flag, the per-vehicle reward r and the platoon sum rsum, t + 1, done.  Pure arithmetic + width-8 shuffles; the
// previous acceleration is only needed by frozen (collided) platoons: `load_u_old` is called for those alone.
template <class ULoad>
__device__ __forceinline__ void cacc_advance(const nmarl_cacc_params_t& p, const int a, float& h, float& v, const int act, int& t,
                                             bool& collided, const float v0i, float& u_new, float& r, float& rsum_out, bool& is_done_out,
                                             ULoad load_u_old) {
    const bool frozen = collided;                                   // :193

    const float alpha = (act & 1) ? 0.5f : 0.0f;                    // a_map, :275
    const float beta = (act & 2) ? 0.5f : 0.0f;
    const float up_v = __shfl_up(v, 1, N);
    const float v_lead = a == 0 ? lead_speed(p, v0i, t) : up_v;     // :33-37
    const float u_raw = alpha * (ovm_vh(p, h) - v) + beta * (v_lead - v);   // :385
    float v_next = v + clampf(u_raw, p.u_min, p.u_max) * p.dt;      // :26
    v_next = clampf(v_next, 0.0f, p.v_max);                         // :27
    const float u_c = (v_next - v) / p.dt;                          // :28
    const float up_vn = __shfl_up(v_next, 1, N);
    const float v_lead_next = a == 0 ? lead_speed(p, v0i, t + 1) : up_vn;
    const float h_next = h + (0.5f * p.dt) * (v_lead + v_lead_next - v - v_next);  // :220

    if (!frozen) { h = h_next; v = v_next; u_new = u_c; }
    else { u_new = load_u_old(); }

    // collision test: min over the platoon (:42)
    float hmin = h;
    hmin = fminf(hmin, __shfl_xor(hmin, 1, N));
    hmin = fminf(hmin, __shfl_xor(hmin, 2, N));
    hmin = fminf(hmin, __shfl_xor(hmin, 4, N));
    if (!frozen && hmin < p.h_min) collided = true;

    if (collided) {
        r = -p.G;                                                   // :44, :194
    } else {
        const float dh = h - p.h_star, dv = v - p.v_star;
        r = -(dh * dh);
        r = r + (-p.reward_a * (dv * dv));
        r = r + (-p.reward_b * (u_new * u_new));
        if (p.train_mode) {
            const float c = fminf(h - 10.0f, 0.0f);                 // COLLISION_HEADWAY, :10
            r = r + (-5.0f * (c * c));                              // COLLISION_WT, :9
        }
    }
    float rsum = r;                                                 // np.sum(reward), :229
    rsum = rsum + __shfl_xor(rsum, 1, N);
    rsum = rsum + __shfl_xor(rsum, 2, N);
    rsum = rsum + __shfl_xor(rsum, 4, N);

    t += 1;
    is_done_out = (collided && (t % p.batch_size == 0)) || (t == p.T);   // :231-235
    rsum_out = rsum;
}

// One tile = 64 lanes = 8 replicas of one wave: the step of cacc_env.py:191-242 from the tile's loaded inputs, all stores, and
// the observation of the new state staged in `lds_wave` ([64 lanes][W]) and written out.
template <int NT, bool COMPACT>
__device__ __forceinline__ void cacc_tile(const nmarl_cacc_params_t& p, const int64_t n_lanes, const int64_t w, const int lane,
                                          float h, float v, const int act, int t, bool collided, float v0i,
                                          float* __restrict__ hs, float* __restrict__ vs, float* __restrict__ us,
                                          int32_t* __restrict__ ts, uint8_t* __restrict__ coll, float* __restrict__ v0_init,
                                          float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
                                          float* __restrict__ greward, const int auto_reset, const uint64_t seed,
                                          const int64_t env_id_base, int32_t* __restrict__ episode, float* lds_wave) {
    constexpr int W = COMPACT ? NF : NOBS;
    const int64_t gid = w * NMARL_WAVE + lane;      // = e*8 + a
    const bool valid = gid < n_lanes;
    const int64_t g = valid ? gid : n_lanes - 1;    // clamp: tail lanes mirror the last vehicle
    const int64_t e = g >> 3;
    const int a = (int)(g & 7);

    float u_new, r, rsum;
    bool is_done;
    cacc_advance(p, a, h, v, act, t, collided, v0i, u_new, r, rsum, is_done, [&]() { return us[g]; });

    if (valid) {
        if (p.per_agent_reward) reward[g] = r;
        if (a == 0) {
            if (!p.per_agent_reward) reward[e] = rsum;
            greward[e] = rsum;
            done[e] = is_done ? 1 : 0;
        }
    }

    if (auto_reset && is_done) {
        const int ep = episode[e];
        const float U = reset_uniform(seed, env_id_base + e, ep);
        init_state(p, U, a, h, v, v0i);
        u_new = 0.0f; t = 0; collided = false;
        if (valid && a == 0) episode[e] = ep + 1;
    }

    if (valid) {
        if (NT >= 2) {
            __builtin_nontemporal_store(h, &hs[g]); __builtin_nontemporal_store(v, &vs[g]);
            __builtin_nontemporal_store(u_new, &us[g]);
        } else {
            hs[g] = h; vs[g] = v; us[g] = u_new;
        }
        if (a == 0) {
            ts[e] = t;
            coll[e] = collided ? 1 : 0;
            if (auto_reset && is_done) v0_init[e] = v0i;
        }
    }

    const float up_v2 = __shfl_up(v, 1, N);
    const float v_lead_obs = a == 0 ? lead_speed(p, v0i, t) : up_v2;   // :55, with the new t
    const int64_t lanes_here = n_lanes - w * NMARL_WAVE;
    const int n_valid = lanes_here >= NMARL_WAVE ? NMARL_WAVE : (int)lanes_here;
    __builtin_amdgcn_wave_barrier();
    emit_obs<NT, COMPACT>(p, h, v, u_new, v_lead_obs, a, valid, lane, lds_wave,
                          obs + w * NMARL_WAVE * W, n_valid);
    __builtin_amdgcn_wave_barrier();
}

// The same step for ONE ARBITRARY replica per aligned 8-lane group (lane = 8 k + vehicle): e = the group's replica, valid = the group
// holds one (idle groups pass any in-range e and store nothing).  Inputs arrive in registers, u_old is fetched only by frozen
// platoons; every output is a per-lane store (compact observation [E][8][5] only).  Used by the lock-step kernel, whose waves
// step the replicas for which they were the last agent to draw (csrc/lstm_mfma.hip, ENV) -- operation for operation cacc_tile.
__device__ __forceinline__ void cacc_step_group(const nmarl_cacc_params_t& p, const int64_t e, const int a, const bool valid, float h, float v,
                                                const int act, int t, bool collided, float v0i, float* __restrict__ hs,
                                                float* __restrict__ vs, float* __restrict__ us, int32_t* __restrict__ ts,
                                                uint8_t* __restrict__ coll, float* __restrict__ v0_init, float* __restrict__ obs,
                                                float* __restrict__ reward, uint8_t* __restrict__ done, float* __restrict__ greward,
                                                const int auto_reset, const uint64_t seed, const int64_t env_id_base,
                                                int32_t* __restrict__ episode) {
    const int64_t g = e * N + a;
    float u_new, r, rsum;
    bool is_done;
    cacc_advance(p, a, h, v, act, t, collided, v0i, u_new, r, rsum, is_done, [&]() { return us[g]; });
    if (valid) {
        if (p.per_agent_reward) reward[g] = r;
        if (a == 0) {
            if (!p.per_agent_reward) reward[e] = rsum;
            greward[e] = rsum;
            done[e] = is_done ? 1 : 0;
        }
    }
    if (auto_reset && is_done) {
        const int ep = episode[e];
        const float U = reset_uniform(seed, env_id_base + e, ep);
        init_state(p, U, a, h, v, v0i);
        u_new = 0.0f; t = 0; collided = false;
        if (valid && a == 0) episode[e] = ep + 1;
    }
    if (valid) {
        hs[g] = h; vs[g] = v; us[g] = u_new;
        if (a == 0) {
            ts[e] = t;
            coll[e] = collided ? 1 : 0;
            if (auto_reset && is_done) v0_init[e] = v0i;
        }
    }
    const float up_v2 = __shfl_up(v, 1, N);
    const float v_lead_obs = a == 0 ? lead_speed(p, v0i, t) : up_v2;   // :55, with the new t
    float x[NF];
    obs_features(p, h, v, u_new, v_lead_obs, x);
    if (valid) {
#pragma unroll
        for (int k = 0; k < NF; ++k) obs[g * NF + k] = x[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The step in the FOUR-VEHICLES-PER-LANE mapping of the HBM regime (round 5): lane = (replica, half platoon), a wave64 steps
// 32 replicas, every per-vehicle array is read and written as ONE 16-byte access per lane, the compact observation leaves as
// five (4 vehicles x 5 features = 80 contiguous bytes per lane) -- 19 vector-memory instructions per 11.2 KB instead of 19 per
// 2.8 KB in the lane-per-vehicle mapping, whose launch was bound by the instruction rate of the memory pipeline, not by HBM.
// Arithmetic: cacc_advance / obs_features operation for operation -- the two-stage in-lane sums and the one exchange with the
// partner lane form the same trees as the three xor butterflies over 8 lanes ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)) (fp32 add and
// min commute exactly), so the two mappings agree bit for bit (tests/test_gpu_cacc.py).  Compact observation only.
template <int NT>
__device__ __forceinline__ void cacc_quad(const nmarl_cacc_params_t& p, const int64_t E, const int64_t e_raw, const int half, float4 h4,
                                          float4 v4, const uint32_t act4, int t, bool collided, float v0i, float* __restrict__ hs,
                                          float* __restrict__ vs, float* __restrict__ us, int32_t* __restrict__ ts, uint8_t* __restrict__ coll,
                                          float* __restrict__ v0_init, float* __restrict__ obs, float* __restrict__ reward,
                                          uint8_t* __restrict__ done, float* __restrict__ greward, const int auto_reset, const uint64_t seed,
                                          const int64_t env_id_base, int32_t* __restrict__ episode, float* lds_wave, const int lane,
                                          const int64_t e_tile0) {
    const bool valid = e_raw < E;
    const int64_t e = valid ? e_raw : E - 1;
    const int64_t g0 = e * N + 4 * half;                 // first of this lane's four vehicles
    float h[4] = {h4.x, h4.y, h4.z, h4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w};
    const bool frozen = collided;                                   // :193
    // the speed of the vehicle in front of this lane's first one: the leader's profile (half 0) or the partner lane's last vehicle
    const float pv3 = __shfl_xor(v[3], 1, NMARL_WAVE);
    float vl[4], vn[4], uc[4], hn[4];
    vl[0] = half == 0 ? lead_speed(p, v0i, t) : pv3;
    vl[1] = v[0]; vl[2] = v[1]; vl[3] = v[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int act = (int)((act4 >> (8 * j)) & 0xffu);
        const float alpha = (act & 1) ? 0.5f : 0.0f;                // a_map, :275
        const float beta = (act & 2) ? 0.5f : 0.0f;
        const float u_raw = alpha * (ovm_vh(p, h[j]) - v[j]) + beta * (vl[j] - v[j]);   // :385
        float v_next = v[j] + clampf(u_raw, p.u_min, p.u_max) * p.dt;      // :26
        v_next = clampf(v_next, 0.0f, p.v_max);                     // :27
        uc[j] = (v_next - v[j]) / p.dt;                             // :28
        vn[j] = v_next;
    }
    const float pvn3 = __shfl_xor(vn[3], 1, NMARL_WAVE);
    float u_new[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v_lead_next = j == 0 ? (half == 0 ? lead_speed(p, v0i, t + 1) : pvn3) : vn[j - 1];
        hn[j] = h[j] + (0.5f * p.dt) * (vl[j] + v_lead_next - v[j] - vn[j]);  // :220
    }
    if (!frozen) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { h[j] = hn[j]; v[j] = vn[j]; u_new[j] = uc[j]; }
    } else {
        const float4 uo = *reinterpret_cast<const float4*>(us + g0);
        u_new[0] = uo.x; u_new[1] = uo.y; u_new[2] = uo.z; u_new[3] = uo.w;
    }
    // collision test: min over the platoon (:42) -- pairs, quads, then the partner's quad
    float hmin = fminf(fminf(h[0], h[1]), fminf(h[2], h[3]));
    hmin = fminf(hmin, __shfl_xor(hmin, 1, NMARL_WAVE));
    if (!frozen && hmin < p.h_min) collided = true;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (collided) {
            r[j] = -p.G;                                            // :44, :194
        } else {
            const float dh = h[j] - p.h_star, dv = v[j] - p.v_star;
            float rr = -(dh * dh);
            rr = rr + (-p.reward_a * (dv * dv));
            rr = rr + (-p.reward_b * (u_new[j] * u_new[j]));
            if (p.train_mode) {
                const float c = fminf(h[j] - 10.0f, 0.0f);          // COLLISION_HEADWAY, :10
                rr = rr + (-5.0f * (c * c));                        // COLLISION_WT, :9
            }
            r[j] = rr;
        }
    }
    float rsum = (r[0] + r[1]) + (r[2] + r[3]);                     // np.sum(reward), :229: the pairwise tree of 8
    rsum = rsum + __shfl_xor(rsum, 1, NMARL_WAVE);
    t += 1;
    const bool is_done = (collided && (t % p.batch_size == 0)) || (t == p.T);   // :231-235
    if (valid) {
        if (p.per_agent_reward) *reinterpret_cast<float4*>(reward + g0) = float4{r[0], r[1], r[2], r[3]};
        if (half == 0) {
            if (!p.per_agent_reward) reward[e] = rsum;
            greward[e] = rsum;
            done[e] = is_done ? 1 : 0;
        }
    }
    if (auto_reset && is_done) {
        const int ep = episode[e];
        const float U = reset_uniform(seed, env_id_base + e, ep);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v0n;
            init_state(p, U, 4 * half + j, h[j], v[j], v0n);
            v0i = v0n;
            u_new[j] = 0.0f;
        }
        t = 0; collided = false;
        if (valid && half == 0) episode[e] = ep + 1;
    }
    if (valid) {
        typedef float f32x4s __attribute__((ext_vector_type(4)));
        const f32x4s H4 = {h[0], h[1], h[2], h[3]}, V4 = {v[0], v[1], v[2], v[3]}, U4 = {u_new[0], u_new[1], u_new[2], u_new[3]};
        if (NT >= 2) {
            __builtin_nontemporal_store(H4, reinterpret_cast<f32x4s*>(hs + g0));
            __builtin_nontemporal_store(V4, reinterpret_cast<f32x4s*>(vs + g0));
            __builtin_nontemporal_store(U4, reinterpret_cast<f32x4s*>(us + g0));
        } else {
            *reinterpret_cast<f32x4s*>(hs + g0) = H4; *reinterpret_cast<f32x4s*>(vs + g0) = V4; *reinterpret_cast<f32x4s*>(us + g0) = U4;
        }
        if (half == 0) {
            ts[e] = t;
            coll[e] = collided ? 1 : 0;
            if (auto_reset && is_done) v0_init[e] = v0i;
        }
    }
    // observation of the new state (:55, with the new t): 4 vehicles x 5 features = 20 consecutive floats of obs [E][8][5]
    const float pv3o = __shfl_xor(v[3], 1, NMARL_WAVE);
    float xo[4 * NF];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v_lead_obs = j == 0 ? (half == 0 ? lead_speed(p, v0i, t) : pv3o) : v[j - 1];
        float x[NF];
        obs_features(p, h[j], v[j], u_new[j], v_lead_obs, x);
#pragma unroll
        for (int k = 0; k < NF; ++k) xo[j * NF + k] = x[k];
    }
    // a lane's 80 bytes are contiguous, but 16-byte stores at an 80-byte lane stride are 64 separate segments per instruction (the
    // first version: 505 us instead of 166 at E = 2^21): the wave's 32 x 40 floats go through LDS and leave as 5 coalesced stores
    typedef float f32x4s __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NF; ++i)
        *reinterpret_cast<float4*>(lds_wave + lane * (4 * NF) + 4 * i) = float4{xo[4 * i], xo[4 * i + 1], xo[4 * i + 2], xo[4 * i + 3]};
    __builtin_amdgcn_wave_barrier();
    const int64_t reps_here = E - e_tile0 < (NMARL_WAVE / 2) ? E - e_tile0 : (NMARL_WAVE / 2);     // replicas of this tile inside the batch
    const int n_vec = (int)reps_here * (N * NF / 4);                                             // float4 of the tile's slab
    f32x4s* dst = reinterpret_cast<f32x4s*>(obs + e_tile0 * (N * NF));
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int idx = i * NMARL_WAVE + lane;
        if (idx < n_vec) {
            const float4 q = reinterpret_cast<const float4*>(lds_wave)[idx];
            const f32x4s val = {q.x, q.y, q.z, q.w};
            if (NT >= 1) __builtin_nontemporal_store(val, dst + idx);
            else dst[idx] = val;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

}  // namespace nmarl_cacc
