// The synthetic 5x5 ATSC grid step as DEVICE code shared by the env kernel (csrc/grid.hip) and by the env role of the CommNet
// lock-step launch (csrc/lstm_mfma.hip, round 6).  Contract: envs/atsc_env.py:181-207 (step), 216-240 (yellow), 383-462 (reward /
// state); envs/large_grid_env.py:23-27 (phases), 58-105 (topology); build_file.py:268-326 (demand); dynamics: oracle/grid_ref.py.
#pragma once
#include "common.h"
#include <cstddef>

namespace nmarl_grid {

constexpr int NN = NMARL_GRID_N;        // 25
constexpr int SIDE = 5;
constexpr int NL = 12;                  // signal links per node
constexpr int NLANE = 6;
constexpr int NSLOT = 5;                // own + 4 neighbour slots
constexpr int OBSW = NSLOT * NL;        // 60
constexpr float DT = 5.0f, YELLOW = 2.0f, SAT = 0.5f, Q_MAX = 26.0f, DET_CAP = 7.0f, YELLOW_EFF = 1.0f;
constexpr float WAIT_EPS = 1e-3f;       // vehicles: below this a lane holds no standing queue / discharged nothing (oracle/grid_ref.py step 6)

// All static tables in ONE __constant__ object: one base address in scalar registers instead of twelve (the twelve separate
// arrays cost 24 SGPRs of addresses and the kernel spilled 23).
struct GridTables {
    uint8_t green[5][NL];        // large_grid_env.py:25-26   0 = r, 1 = G, 2 = g
    int8_t link_lane[NL];
    int8_t lane_approach[NLANE];
    int8_t dest[NL][3];          // link -> (drow, dcol, receiving approach)
    int8_t from[4][2];
    int8_t feed[4][3];
    int8_t entry[NN][4];         // entry group (+1) per (node, approach); 0 = no external entry   build_file.py:285-295
    float link_share[NL];
    float split[NLANE];
    float ratio1[7];
    float ratio2[7];
};
static __constant__ GridTables c_tab = {
    {{1, 1, 2, 0, 0, 0, 1, 1, 2, 0, 0, 0}, {0, 0, 0, 1, 0, 1, 0, 0, 0, 1, 0, 1}, {0, 0, 0, 1, 1, 0, 0, 0, 0, 1, 1, 0},
     {0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1}},
    {0, 0, 0, 1, 1, 2, 3, 3, 3, 4, 4, 5},
    {0, 1, 1, 2, 3, 3},
    {{0, -1, 1}, {-1, 0, 0}, {0, 1, 3}, {1, 0, 2}, {0, -1, 1}, {-1, 0, 0},
     {0, 1, 3}, {1, 0, 2}, {0, -1, 1}, {-1, 0, 0}, {0, 1, 3}, {1, 0, 2}},
    {{1, 0}, {0, 1}, {-1, 0}, {0, -1}},
    {{1, 5, 9}, {0, 4, 8}, {3, 7, 11}, {2, 6, 10}},
    {{0, 0, 0, 2}, {0, 0, 3, 0}, {0, 0, 3, 0}, {0, 0, 3, 0}, {0, 4, 0, 0},
     {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},
     {0, 0, 0, 2}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 4, 0, 0},
     {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0},
     {0, 0, 0, 2}, {1, 0, 0, 0}, {1, 0, 0, 0}, {1, 0, 0, 0}, {0, 4, 0, 0}},
    {.2f, .6f, .2f, .15f / .85f, .7f / .85f, 1.0f, .2f, .6f, .2f, .15f / .85f, .7f / .85f, 1.0f},
    {1.0f, 0.85f, 0.15f, 1.0f, 0.85f, 0.15f},
    {0.4f, 0.7f, 0.9f, 1.0f, 0.75f, 0.5f, 0.25f},
    {0.3f, 0.8f, 0.9f, 1.0f, 0.8f, 0.6f, 0.2f}};
#define c_green c_tab.green
#define c_link_lane c_tab.link_lane
#define c_link_share c_tab.link_share
#define c_lane_approach c_tab.lane_approach
#define c_split c_tab.split
#define c_dest c_tab.dest
#define c_from c_tab.from
#define c_feed c_tab.feed
#define c_ratio1 c_tab.ratio1
#define c_ratio2 c_tab.ratio2
#define c_entry c_tab.entry

__device__ __forceinline__ float demand_rate(int group, int sec, float peak1, float peak2) {
    const int piece = sec / 300;
    if (group < 2) {
        if (piece >= 7) return 0.0f;
        return peak1 * (group == 0 ? 0.6f : 1.0f) * c_ratio1[piece];
    }
    if (piece < 3 || piece >= 10) return 0.0f;
    return peak2 * (group == 2 ? 0.6f : 1.0f) * c_ratio2[piece - 3];
}

constexpr int NQ = NN * NLANE;      // 150 floats of q (and of transit) per replica
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Lds {           // per replica (q / transit live in the block-wide staging arrays: 8 replicas = 4800 contiguous bytes)
    union {
        float D[NN * NL];         // phases B, C: the desired link flows (read by the neighbouring nodes' lanes)
        float wave[NN * NL];      // phases D, E: the wave vectors (D is dead behind phase C's barrier); 16-byte aligned for the emit
    };
    union {
        float space[NN * 4];      // phase B -> C: read by the node's own lane, which then writes
        float inflow[NN * 4];     // phase C -> D: ... the scaled inflow of the same approach over it
    };
    float scale[NN * 4];
};
static_assert(offsetof(Lds, wave) % 16 == 0 && sizeof(Lds) % 16 == 0, "Lds::wave must be 16-byte aligned in every array slot");

__device__ __forceinline__ void half_barrier() { __builtin_amdgcn_wave_barrier(); }

// ascending-index neighbour k of node n (or -1): order S(n-5), W(n-1), E(n+1), N(n+5)
__device__ __forceinline__ int nbr_of(int n, int k) {
    const int r = n / SIDE, c = n - r * SIDE;
    int cand[4] = {r > 0 ? n - SIDE : -1, c > 0 ? n - 1 : -1, c < SIDE - 1 ? n + 1 : -1, r < SIDE - 1 ? n + SIDE : -1};
    int cnt = 0;
    for (int i = 0; i < 4; ++i) {
        if (cand[i] >= 0) {
            if (cnt == k) return cand[i];
            ++cnt;
        }
    }
    return -1;
}

template <int NT, bool COMPACT>
__device__ __forceinline__ void emit_obs_slab(const Lds& s, float* __restrict__ obs_env, int l32) {
    float4* dst = reinterpret_cast<float4*>(obs_env);
    if (COMPACT) {
        // 25 x 12 floats = 75 float4: the wave vectors as they sit in LDS
        const float4* src = reinterpret_cast<const float4*>(s.wave);
        for (int v = l32; v < NN * NL / 4; v += 32) {
            const float4 val = src[v];
            if (NT) __builtin_nontemporal_store(f32x4{val.x, val.y, val.z, val.w}, reinterpret_cast<f32x4*>(dst) + v);
            else dst[v] = val;
        }
        return;
    }
    // 25 x 60 floats = 375 float4, coalesced
    for (int v = l32; v < NN * OBSW / 4; v += 32) {
        const int node = v / (OBSW / 4);
        const int w = (v - node * (OBSW / 4)) * 4;          // first float within the 60-wide row
        const int slot = w / NL, f = w - slot * NL;          // 12 % 4 == 0: a float4 never straddles slots
        const int src = slot == 0 ? node : nbr_of(node, slot - 1);
        float4 val = float4{0.f, 0.f, 0.f, 0.f};
        if (src >= 0) {
            const float* p = s.wave + src * NL + f;
            val = float4{p[0], p[1], p[2], p[3]};
        }
        if (NT) __builtin_nontemporal_store(f32x4{val.x, val.y, val.z, val.w}, reinterpret_cast<f32x4*>(dst) + v);
        else dst[v] = val;
    }
}

// The step of groups of NREP replicas (one per 32-lane half wave; NREP x 32 threads of the calling block), groups first_group,
// first_group + group_stride, ...: the body of grid_step_kernel (csrc/grid.hip) and of the env role of the one-launch lock-step
// (csrc/lstm_mfma.hip) -- ONE definition, so the two are bit-identical by construction.  lds / blk_*: the caller's LDS
// (Lds[NREP], NREP x 150 floats each of q, transit[, head wait]).
template <int NT, bool COMPACT, bool WAIT, int NREP, bool WORDS>
__device__ __forceinline__ void grid_step_groups(
    const nmarl_grid_params_t& p, const int64_t E, const int64_t first_group, const int64_t group_stride,
    const uint8_t* __restrict__ action, unsigned long long* __restrict__ words, int32_t* status, const unsigned max_spins,
    float* __restrict__ qs, float* __restrict__ trs, uint8_t* __restrict__ prev, int32_t* __restrict__ ts,
    float* __restrict__ xi, float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
    float* __restrict__ greward, const int auto_reset, const uint64_t seed, const int64_t env_id_base,
    int32_t* __restrict__ episode, Lds* lds, float* blk_q, float* blk_tr, float* blk_w) {
    float* __restrict__ const hws = p.head_wait;
    const int l32 = threadIdx.x & 31;
    const int sub = threadIdx.x >> 5;                       // replica slot in the block (0 .. NREP - 1)
    Lds& s = lds[sub];
    float* const sq = blk_q + sub * NQ;
    float* const str = blk_tr + sub * NQ;
    float* const sw = blk_w + (WAIT ? sub * NQ : 0);
    for (int64_t e0 = first_group * NREP; e0 < E; e0 += group_stride * NREP) {
        const int64_t e = e0 + sub;
        const bool live = e < E;
        const int64_t ec = live ? e : E - 1;
        // ---- A. coalesced load of the block's state into LDS
        const bool full = e0 + NREP <= E;                      // (uniform) the last, partial group of replicas: 4-byte accesses
        if (full) {
            const float4* qg4 = reinterpret_cast<const float4*>(qs + e0 * NQ);
            const float4* tg4 = reinterpret_cast<const float4*>(trs + e0 * NQ);
            for (int i = threadIdx.x; i < NREP * NQ / 4; i += NREP * 32) {
                reinterpret_cast<float4*>(blk_q)[i] = qg4[i];
                reinterpret_cast<float4*>(blk_tr)[i] = tg4[i];
                if (WAIT) reinterpret_cast<float4*>(blk_w)[i] = reinterpret_cast<const float4*>(hws + e0 * NQ)[i];
            }
        } else {
            const float* qg = qs + ec * NQ;
            const float* tg = trs + ec * NQ;
            for (int i = l32; i < NQ; i += 32) {
                sq[i] = qg[i]; str[i] = tg[i];
                if (WAIT) sw[i] = hws[ec * NQ + i];
            }
        }
        const int t = ts[ec];
        // WORDS (the env step as a role of the lock-step launch, csrc/lstm_mfma.hip): the 25 actions of replica e arrive in two
        // hand-off words -- every agent's wave adds (action << 3 (agent % 13)) | 1 << 59 to word (agent >= 13) of its rows; a half wave
        // waits (bounded) until the counts read 13 and 12, then every node lane takes its three bits.  The words are left zero.
        unsigned long long wa_ = 0ull, wb_ = 0ull;
        if (WORDS) {
            typedef __attribute__((address_space(1))) unsigned long long gu64;
            gu64* wp = (gu64*)(words + 2 * ec);
            bool ok_ = false;
            for (unsigned spins = 0; spins <= max_spins; ++spins) {
                unsigned long long v_ = 0ull;
                if (l32 < 2) v_ = __hip_atomic_load(wp + l32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wa_ = __shfl(v_, 0, 32); wb_ = __shfl(v_, 1, 32);
                ok_ = (wa_ >> 59) == 13ull && (wb_ >> 59) == 12ull;
                if (ok_ || !live) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (live && !ok_ && l32 == 0 && status) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (live && l32 < 2) words[2 * e + l32] = 0ull;
        }
        __syncthreads();
        const int n = l32;
        const bool node = n < NN;
        const int row = n / SIDE, col = n - row * SIDE;
        int a = 0, pa = 0;
        float q[NLANE], tr[NLANE], D[NL];
        if (node) {
            a = WORDS ? (int)(((n < 13 ? wa_ : wb_) >> (3 * (n < 13 ? n : n - 13))) & 7ull) : (int)action[ec * NN + n];
            pa = prev[ec * NN + n];
            a = a > 4 ? 4 : a;
#pragma unroll
            for (int l = 0; l < NLANE; ++l) { q[l] = sq[n * NLANE + l]; tr[l] = str[n * NLANE + l]; }
            // ---- B. desired link flows and receiving space
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int gc = c_green[a][k], gp = c_green[pa][k];
                float g = gc ? (gp ? DT : DT - YELLOW) : (gp ? YELLOW_EFF : 0.0f);
                if (a == pa) g = gc ? DT : 0.0f;
                if (gc == 2) g *= 0.5f;
                const float sh = c_link_share[k];
                D[k] = fminf(q[c_link_lane[k]] * sh, SAT * g * sh);
                s.D[n * NL + k] = D[k];
            }
            float sp[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int l = 0; l < NLANE; ++l) sp[c_lane_approach[l]] += fmaxf(Q_MAX - q[l] - tr[l], 0.0f);
#pragma unroll
            for (int r = 0; r < 4; ++r) s.space[n * 4 + r] = sp[r];
        }
        half_barrier();
        // ---- C. spill-back scale of every receiving approach
        if (node) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = row + c_from[r][0], cc = col + c_from[r][1];
                float in = 0.0f;
                if (rr >= 0 && rr < SIDE && cc >= 0 && cc < SIDE) {
                    const float* Dm = s.D + (rr * SIDE + cc) * NL;
                    in = Dm[c_feed[r][0]] + Dm[c_feed[r][1]] + Dm[c_feed[r][2]];
                }
                const float sc = fminf(1.0f, s.space[n * 4 + r] / fmaxf(in, 1e-6f));
                s.scale[n * 4 + r] = sc;
                s.inflow[n * 4 + r] = in * sc;
            }
        }
        half_barrier();
        // ---- D. served flows, queue update, arrivals, reward, wave
        float r_node = 0.0f;
        if (node) {
            float served[NLANE] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int rr = row + c_dest[k][0], cc = col + c_dest[k][1];
                float fl = D[k];
                if (rr >= 0 && rr < SIDE && cc >= 0 && cc < SIDE) fl = D[k] * s.scale[(rr * SIDE + cc) * 4 + c_dest[k][2]];
                served[c_link_lane[k]] += fl;
            }
            float inflow[4];
            const int sec = t * 5;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                inflow[r] = s.inflow[n * 4 + r];
                const int grp = c_entry[n][r];
                if (grp) inflow[r] += demand_rate(grp - 1, sec, p.peak1, p.peak2) / 3600.0f * DT * xi[ec * 4 + grp - 1];
            }
            float hw[NLANE];
#pragma unroll
            for (int l = 0; l < NLANE; ++l) {
                if (WAIT) {
                    // the lane's head vehicle keeps waiting while a standing queue discharges nothing (oracle/grid_ref.py step 6)
                    const bool moved = served[l] > WAIT_EPS || q[l] <= WAIT_EPS;
                    hw[l] = moved ? 0.0f : sw[n * NLANE + l] + DT;
                }
                q[l] = q[l] - served[l] + tr[l];
                tr[l] = inflow[c_lane_approach[l]] * c_split[l];
            }
            float r_wait = 0.0f;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const float c = fminf(q[c_link_lane[k]], DET_CAP);
                r_node -= c;
                if (WAIT) r_wait -= hw[c_link_lane[k]];
                float w = c / p.norm_wave;
                if (p.clip_wave >= 0.0f) w = fminf(fmaxf(w, 0.0f), p.clip_wave);
                s.wave[n * NL + k] = w;
            }
            if (WAIT) {
                r_node = p.objective == 1 ? r_wait : r_node + p.coef_wait * r_wait;      // atsc_env.py:411-416
#pragma unroll
                for (int l = 0; l < NLANE; ++l) sw[n * NLANE + l] = hw[l];
            }
        }
        float gsum = r_node;                     // sum over the 25 nodes of the half wave
        for (int off = 16; off > 0; off >>= 1) gsum += __shfl_xor(gsum, off, 32);
        const int t_new = t + 1;
        const bool is_done = t_new >= p.T;       // atsc_env.py:189-191
        const bool rst = auto_reset && is_done;
        if (node) {
            if (rst) {
#pragma unroll
                for (int l = 0; l < NLANE; ++l) { q[l] = 0.0f; tr[l] = 0.0f; if (WAIT) sw[n * NLANE + l] = 0.0f; }
#pragma unroll
                for (int k = 0; k < NL; ++k) s.wave[n * NL + k] = 0.0f;
                a = 0;                            // _reset_state: prev_action = 0 (atsc_env.py:509-513)
            }
#pragma unroll
            for (int l = 0; l < NLANE; ++l) { sq[n * NLANE + l] = q[l]; str[n * NLANE + l] = tr[l]; }
        }
        __syncthreads();
        // ---- E. coalesced write-back
        if (full) {
            f32x4* qo4 = reinterpret_cast<f32x4*>(qs + e0 * NQ);
            f32x4* to4 = reinterpret_cast<f32x4*>(trs + e0 * NQ);
            for (int i = threadIdx.x; i < NREP * NQ / 4; i += NREP * 32) {
                const f32x4 a4 = reinterpret_cast<const f32x4*>(blk_q)[i], b4 = reinterpret_cast<const f32x4*>(blk_tr)[i];
                if (NT) { __builtin_nontemporal_store(a4, qo4 + i); __builtin_nontemporal_store(b4, to4 + i); }
                else { qo4[i] = a4; to4[i] = b4; }
                if (WAIT) reinterpret_cast<f32x4*>(hws + e0 * NQ)[i] = reinterpret_cast<const f32x4*>(blk_w)[i];
            }
        } else if (live) {
            float* qo = qs + e * NQ;
            float* to = trs + e * NQ;
            for (int i = l32; i < NQ; i += 32) {
                qo[i] = sq[i]; to[i] = str[i];
                if (WAIT) hws[e * NQ + i] = sw[i];
            }
        }
        if (live) {
            if (node) {
                prev[e * NN + n] = (uint8_t)a;
                if (p.per_agent_reward) reward[e * NN + n] = r_node;
            }
            if (l32 == 0) {
                if (!p.per_agent_reward) reward[e] = gsum;
                greward[e] = gsum;
                done[e] = is_done ? 1 : 0;
                ts[e] = rst ? 0 : t_new;
            }
            if (rst && l32 < 4) {
                const int ep = episode[e];
                const Philox4 r4 = philox4x32_10((uint32_t)(env_id_base + e), 0u, (uint32_t)ep, NMARL_STREAM_RESET,
                                                 (uint32_t)seed, (uint32_t)(seed >> 32));
                const uint32_t w = l32 == 0 ? r4.x : l32 == 1 ? r4.y : l32 == 2 ? r4.z : r4.w;
                xi[e * 4 + l32] = 0.8f + 0.4f * u01_from_bits(w);
            }
            if (rst && l32 == 4) episode[e] = episode[e] + 1;
            emit_obs_slab<NT, COMPACT>(s, obs + e * NN * (COMPACT ? NL : OBSW), l32);
        }
        __syncthreads();                          // the staging arrays are refilled by the next group of replicas
    }
}

}  // namespace nmarl_grid
