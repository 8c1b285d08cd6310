// Synthetic (SUMO-free) 5x5 ATSC grid for E lock-stepped replicas on gfx950.
//
// Contract taken from the reference (envs/atsc_env.py:181-207 step, 216-240 yellow,
// 383-462 reward/state; envs/large_grid_env.py:23-27 phases, 58-105 topology;
// envs/large_grid_data/build_file.py:268-326 demand).  The dynamics are the
// store-and-forward model SPECIFIED in oracle/grid_ref.py (SUMO is not available,
// so the dynamics are "parity unpinned" w.r.t. the reference; the oracle is the spec).
//
// Mapping: one replica per 32-lane half wave, lane = intersection (25 of 32 lanes
// active), two replicas per wave64, 8 per 256-thread block.  A replica's state
// (q, transit: 2 x 25 x 6 fp32 = 1200 B, contiguous) is loaded with coalesced
// accesses into LDS; the neighbour exchange (desired link flows D, receiving space,
// spill-back scale) goes through LDS with wave barriers -- the 4-neighbourhood of the
// lattice never leaves the half wave.  Observation: COMPACT [25, 12] -- every node's OWN wave vector, what the reference
// hands an agent (atsc_env.py:253-262; 1200 contiguous bytes per replica, the batched engine's layout: the policy's
// encoder kernel gathers the neighbours itself) -- or the gathered slab [25, 5*12] (own + up to 4 neighbours in ascending
// node index, 6000 bytes) for the reference duck-type; assembled in LDS, 16-byte coalesced stores.
// HBM-bound (3.7 KB per replica-step with the compact observation, 8.5 KB with the slab); no MFMA.
#include "common.h"
#include <cstddef>

namespace {

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
__constant__ GridTables c_tab = {
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

// NT = 1: non-temporal stores for state and slab when the working set exceeds the caches (same effect as
// in csrc/cacc.hip: streaming writes at the fill ceiling instead of ~60 % of it).
// WAIT: the `wait` / `hybrid` objectives (atsc_env.py:383-418) -- one more state array, the head vehicle's waiting time per
// lane (oracle/grid_ref.py step 6); the shipped `queue` configs run the instantiation without it (same bytes as before).
template <int NT, bool COMPACT, bool WAIT>
__global__ __launch_bounds__(256) void grid_step_kernel(
    const nmarl_grid_params_t p, const int64_t E, const uint8_t* __restrict__ action,
    float* __restrict__ qs, float* __restrict__ trs, uint8_t* __restrict__ prev, int32_t* __restrict__ ts,
    float* __restrict__ xi, float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
    float* __restrict__ greward, const int auto_reset, const uint64_t seed, const int64_t env_id_base,
    int32_t* __restrict__ episode) {
    __shared__ __attribute__((aligned(16))) Lds lds[8];
    // state of the block's 8 replicas: 8 x 150 floats of q and of transit are CONTIGUOUS in memory (replica-major) and 16-byte
    // aligned (8 x 600 B per block step), so the whole block moves them with 16-byte accesses (300 float4 each way per array;
    // the 4-byte per-replica loops of rounds 1-3 were the kernel's issue limit in the HBM regime)
    __shared__ __attribute__((aligned(16))) float blk_q[8 * NQ], blk_tr[8 * NQ];
    __shared__ __attribute__((aligned(16))) float blk_w[WAIT ? 8 * NQ : 4];
    float* __restrict__ const hws = p.head_wait;
    const int l32 = threadIdx.x & 31;
    const int sub = threadIdx.x >> 5;                       // replica slot in the block (0..7)
    Lds& s = lds[sub];
    float* const sq = blk_q + sub * NQ;
    float* const str = blk_tr + sub * NQ;
    float* const sw = blk_w + (WAIT ? sub * NQ : 0);
    const int64_t stride = (int64_t)gridDim.x * 8;
    for (int64_t e0 = (int64_t)blockIdx.x * 8; e0 < E; e0 += stride) {
        const int64_t e = e0 + sub;
        const bool live = e < E;
        const int64_t ec = live ? e : E - 1;
        // ---- A. coalesced load of the block's state into LDS
        const bool full = e0 + 8 <= E;                      // (uniform) the last, partial group of replicas: 4-byte accesses
        if (full) {
            const float4* qg4 = reinterpret_cast<const float4*>(qs + e0 * NQ);
            const float4* tg4 = reinterpret_cast<const float4*>(trs + e0 * NQ);
            for (int i = threadIdx.x; i < 8 * NQ / 4; i += 256) {
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
        __syncthreads();
        const int n = l32;
        const bool node = n < NN;
        const int row = n / SIDE, col = n - row * SIDE;
        int a = 0, pa = 0;
        float q[NLANE], tr[NLANE], D[NL];
        if (node) {
            a = action[ec * NN + n];
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
            for (int i = threadIdx.x; i < 8 * NQ / 4; i += 256) {
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

__global__ __launch_bounds__(256) void grid_reset_kernel(
    float* __restrict__ hws, const int64_t E, const uint8_t* __restrict__ mask, const float* __restrict__ u0, float* __restrict__ qs,
    float* __restrict__ trs, uint8_t* __restrict__ prev, int32_t* __restrict__ ts, float* __restrict__ xi,
    float* __restrict__ obs, const int obs_w, const uint64_t seed, const int64_t env_id_base, int32_t* __restrict__ episode) {
    const int l32 = threadIdx.x & 31;
    const int sub = threadIdx.x >> 5;
    for (int64_t e = (int64_t)blockIdx.x * 8 + sub; e < E; e += (int64_t)gridDim.x * 8) {
        if (mask != nullptr && mask[e] == 0) continue;
        for (int i = l32; i < NN * NLANE; i += 32) {
            qs[e * NN * NLANE + i] = 0.0f; trs[e * NN * NLANE + i] = 0.0f;
            if (hws) hws[e * NN * NLANE + i] = 0.0f;
        }
        for (int i = l32; i < NN * obs_w; i += 32) obs[e * NN * obs_w + i] = 0.0f;
        if (l32 < NN) prev[e * NN + l32] = 0;
        if (l32 == 0) ts[e] = 0;
        if (l32 < 4) {
            float U;
            if (u0 != nullptr) {
                U = u0[e * 4 + l32];
            } else {
                const Philox4 r4 = philox4x32_10((uint32_t)(env_id_base + e), 0u, (uint32_t)episode[e], NMARL_STREAM_RESET,
                                                 (uint32_t)seed, (uint32_t)(seed >> 32));
                const uint32_t w = l32 == 0 ? r4.x : l32 == 1 ? r4.y : l32 == 2 ? r4.z : r4.w;
                U = u01_from_bits(w);
            }
            xi[e * 4 + l32] = 0.8f + 0.4f * U;
        }
        __builtin_amdgcn_wave_barrier();
        if (u0 == nullptr && l32 == 4) episode[e] = episode[e] + 1;
    }
}

inline int grid_blocks(int64_t E) {
    const int64_t b = (E + 7) / 8;
    return (int)(b < 4096 ? b : 4096);
}

}  // namespace

extern "C" int nmarl_grid_step(const nmarl_grid_params_t* p, int64_t E, const uint8_t* action, float* q,
                               float* transit, uint8_t* prev_action, int32_t* t, float* xi, float* obs,
                               float* reward, uint8_t* done, float* global_reward, int32_t auto_reset,
                               uint64_t seed, int64_t env_id_base, int32_t* episode, void* stream) {
    if (!p || p->T <= 0 || p->norm_wave <= 0.f || E < 0 ||
        (E > 0 && (!action || !q || !transit || !prev_action || !t || !xi || !obs || !reward || !done || !global_reward)))
        return NMARL_EINVAL;
    if (auto_reset && !episode) return NMARL_EINVAL;
    if (p->objective < 0 || p->objective > 2 || (p->objective != 0 && E > 0 && !p->head_wait)) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    const bool nt = E * (p->compact_obs ? 4000 : 8800) > (int64_t)256 << 20;     // beyond the 256 MB Infinity Cache: stream the writes
    const bool wait = p->objective != 0;
#define NMARL_GRID_LAUNCH(NT_, C_, W_)                                                                                      \
    hipLaunchKernelGGL((grid_step_kernel<NT_, C_, W_>), dim3(grid_blocks(E)), dim3(256), 0, static_cast<hipStream_t>(stream), \
                       *p, E, action, q, transit, prev_action, t, xi, obs, reward, done, global_reward, auto_reset, seed,  \
                       env_id_base, episode)
#define NMARL_GRID_LAUNCH2(NT_, C_) { if (wait) NMARL_GRID_LAUNCH(NT_, C_, true); else NMARL_GRID_LAUNCH(NT_, C_, false); }
    if (p->compact_obs) { if (nt) NMARL_GRID_LAUNCH2(1, true) else NMARL_GRID_LAUNCH2(0, true) }
    else { if (nt) NMARL_GRID_LAUNCH2(1, false) else NMARL_GRID_LAUNCH2(0, false) }
#undef NMARL_GRID_LAUNCH2
#undef NMARL_GRID_LAUNCH
    return nmarl_check_launch();
}

extern "C" int nmarl_grid_reset(const nmarl_grid_params_t* p, int64_t E, const uint8_t* mask, const float* u0,
                                uint64_t seed, int64_t env_id_base, int32_t* episode, float* q, float* transit,
                                uint8_t* prev_action, int32_t* t, float* xi, float* obs, void* stream) {
    if (!p || E < 0 || (E > 0 && (!q || !transit || !prev_action || !t || !xi || !obs))) return NMARL_EINVAL;
    if (!u0 && !episode) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipLaunchKernelGGL(grid_reset_kernel, dim3(grid_blocks(E)), dim3(256), 0, static_cast<hipStream_t>(stream), p->head_wait, E, mask,
                       u0, q, transit, prev_action, t, xi, obs, p->compact_obs ? NL : OBSW, seed, env_id_base, episode);
    return nmarl_check_launch();
}
