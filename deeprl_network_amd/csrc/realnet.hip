// Synthetic signalised NETWORK environment with heterogeneous intersections (the Monaco scenario of
// envs/real_net_env.py: 28 nodes, 2..6 phases over 2..22 signal links, directed neighbour lists), batched over
// replicas.  Contract (actions, yellow logic, `wave` observation, queue reward, episode timing) from
// envs/atsc_env.py:181-240, 383-462; dynamics specified in oracle/realnet_ref.py (SUMO is not available).
//
// The topology is DATA (nmarl_net_topo_t, built by the host from the reference's NODES / PHASES tables), so the
// same kernel serves any network with N <= 32 nodes and L <= 24 links per node.
//
// Mapping (as csrc/grid.hip): one replica per 32-lane half wave, lane = intersection; a lane keeps the queues of
// its <= 24 links in registers.  Cross-node quantities go through LDS between half-wave barriers:
//   out[j]      what node j offers to each link it feeds (its discharge / its fan-out)
//   acc[i][k]   what link k of node i accepts from its feeder (spill-back: limited by the free space)
//   cnt[i][k]   detector counts, from which the neighbour-gathered observation slab is assembled and written
//               with coalesced stores.
// All sums run in a fixed order (the feeder adds up its fan-out list in ascending (node, link)): bit-reproducible.
// The static tables are copied into LDS once per block from a host-packed image; the replica's state rows are staged
// through LDS with coalesced accesses.  Algorithmic bytes per replica-step: q, transit 2*4*sum(n_s) in and out +
// action/prev N + 20 B + the slab 4*L*(1+m_max)*N (12.3 KB for Monaco, of which 4*sum_i(n_s_i + sum_nbr n_s_j) =
// 5.0 KB are non-padding).  The intended bound is HBM; today the kernel is instruction-bound.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int NMAX = 32;          // nodes (lanes of a half wave)
constexpr int LMAX = 24;          // links per node kept in registers
constexpr int AMAX = 8;           // phases per node
constexpr float DT = 5.0f, YELLOW = 2.0f, YELLOW_EFF = 1.0f, SAT = 0.5f, Q_MAX = 26.0f, DET_CAP = 7.0f;

struct RepShared {
    float out[NMAX];
    float acc[NMAX * LMAX];          // also the coalesced staging area of q (load) / the new q (store)
    float cnt[NMAX * LMAX];          // also the staging area of transit
};

__device__ __forceinline__ void half_barrier() { __syncthreads(); }   // the replicas of a block run in lock step

// real_net_data/build_file.py:70-72: number of active flows per 5-minute piece, groups 0,1 (a) and 2,3 (b); pure
// arithmetic (a table lookup per link put one constant-memory round trip per link on the critical path)
__device__ __forceinline__ float activity_a(const int piece) {
    return piece > 7 ? 0.0f : (piece == 0 || piece == 7) ? 1.0f : (piece == 1 || piece == 6) ? 2.0f : 4.0f;
}
__device__ __forceinline__ float activity_b(const int piece) {
    return (piece < 3 || piece > 10) ? 0.0f : (piece == 3 || piece == 10) ? 1.0f : (piece == 4 || piece == 9) ? 2.0f : 4.0f;
}

template <int REPS>
__global__ __launch_bounds__(32 * REPS) void net_step_kernel(
    const nmarl_net_params_t p, const nmarl_net_topo_t tp, const int64_t E, const uint8_t* __restrict__ action,
    float* __restrict__ qs, float* __restrict__ trs, uint8_t* __restrict__ prev, int32_t* __restrict__ ts,
    float* __restrict__ xi, float* __restrict__ obs, float* __restrict__ reward, uint8_t* __restrict__ done,
    float* __restrict__ greward, const int auto_reset, const uint64_t seed, const int64_t env_id_base,
    int32_t* __restrict__ episode) {
    __shared__ RepShared sh[REPS];
    // the static network, staged once per block from the host-packed image (nmarl_net_topo_t.image: already in the LDS
    // layout below, so the copy is a fixed number of 16-byte loads all in flight at once; element-wise staging from the
    // individual tables cost 21 us per launch): every later table access is an LDS read.
    __shared__ __attribute__((aligned(16))) uint8_t tbl[NMARL_NET_IMAGE_BYTES];
    const uint8_t* t_green = tbl + NMARL_NET_OFF_GREEN;                                   // [N][A][LMAX] u8
    const int16_t* t_src = reinterpret_cast<const int16_t*>(tbl + NMARL_NET_OFF_SRC);     // [N][LMAX]
    const int8_t* t_group = reinterpret_cast<const int8_t*>(tbl + NMARL_NET_OFF_GROUP);   // [N][LMAX]
    const float* t_share = reinterpret_cast<const float*>(tbl + NMARL_NET_OFF_SHARE);     // [N][LMAX]
    const float* t_fan = reinterpret_cast<const float*>(tbl + NMARL_NET_OFF_FAN);         // [N]
    const int16_t* t_dnptr = reinterpret_cast<const int16_t*>(tbl + NMARL_NET_OFF_DNPTR); // [N+1]
    const int16_t* t_dnpair = reinterpret_cast<const int16_t*>(tbl + NMARL_NET_OFF_DNPAIR);   // node*LMAX + link
    const int8_t* t_nbr = reinterpret_cast<const int8_t*>(tbl + NMARL_NET_OFF_NBR);       // [N][8]
    const int l32 = threadIdx.x & 31, sub = threadIdx.x >> 5;
    RepShared& s = sh[sub];
    const int N = tp.N, L = tp.L, W = L * (1 + tp.m_max);
    {
        constexpr int NV = NMARL_NET_IMAGE_BYTES / 16, PER = (NV + 32 * REPS - 1) / (32 * REPS);
        const uint4* src4 = reinterpret_cast<const uint4*>(tp.image);
        uint4 v[PER];
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int i = it * 32 * REPS + threadIdx.x;
            if (i < NV) v[it] = src4[i];
        }
#pragma unroll
        for (int it = 0; it < PER; ++it) {
            const int i = it * 32 * REPS + threadIdx.x;
            if (i < NV) reinterpret_cast<uint4*>(tbl)[i] = v[it];
        }
    }
    __syncthreads();
    const int Lmagic = (65536 + L - 1) / L;     // i / L == (i * Lmagic) >> 16 for i < 768, L <= 24 (tests/test_abi.py)
    const int n = l32;
    const bool node = n < N;
    const int ns = node ? tp.n_s[n] : 0;
    const int64_t rounds = (E + (int64_t)gridDim.x * REPS - 1) / ((int64_t)gridDim.x * REPS);
    for (int64_t rd = 0; rd < rounds; ++rd) {
        const int64_t e = (rd * gridDim.x + blockIdx.x) * REPS + sub;
        const bool live = e < E;
        const int64_t ec = live ? e : E - 1;
        float q[LMAX], tr[LMAX], D[LMAX];
        int a = 0, pa = 0;
        const int t = ts[ec];
        if (node) {
            a = action[ec * N + n];
            pa = prev[ec * N + n];
        }
        // the replica's [N, L] state rows are contiguous: load them coalesced through LDS
        {   // fixed trip count: all loads of the tile are in flight together (a runtime-bounded loop waits per iteration)
            float lq[NMAX * LMAX / 32], lt[NMAX * LMAX / 32];
#pragma unroll
            for (int it = 0; it < NMAX * LMAX / 32; ++it) {
                const int i = it * 32 + l32;
                lq[it] = i < N * L ? qs[ec * N * L + i] : 0.0f;
                lt[it] = i < N * L ? trs[ec * N * L + i] : 0.0f;
            }
#pragma unroll
            for (int it = 0; it < NMAX * LMAX / 32; ++it) {
                const int i = it * 32 + l32;
                if (i < N * L) {
                    const int r = (i * Lmagic) >> 16;
                    s.acc[r * LMAX + (i - r * L)] = lq[it];
                    s.cnt[r * LMAX + (i - r * L)] = lt[it];
                }
            }
        }
        half_barrier();
        // ---- A. effective green, desired discharge
        float out = 0.0f;
#pragma unroll
        for (int k = 0; k < LMAX; ++k) {
            q[k] = 0.0f; tr[k] = 0.0f; D[k] = 0.0f;
            if (k < ns) {
                q[k] = s.acc[n * LMAX + k];
                tr[k] = s.cnt[n * LMAX + k];
                const int gc = t_green[(n * AMAX + a) * LMAX + k], gp = t_green[(n * AMAX + pa) * LMAX + k];
                float g;
                if (a == pa) g = gc ? DT : 0.0f;
                else g = gc ? (gp ? DT : DT - YELLOW) : (gp ? YELLOW_EFF : 0.0f);
                if (gc == 2) g *= 0.5f;
                D[k] = fminf(q[k], SAT * g);
                out += D[k];
            }
        }
        half_barrier();                                      // everybody has read its staged rows
        if (node) s.out[n] = t_fan[n] > 0.0f ? out / t_fan[n] : 0.0f;      // what each fed link is offered
        half_barrier();
        // ---- B. what every fed link accepts
#pragma unroll
        for (int k = 0; k < LMAX; ++k)
            if (k < ns) {
                const int src = t_src[n * LMAX + k];
                float acc = 0.0f;
                if (src >= 0) acc = fminf(s.out[src], fmaxf(Q_MAX - q[k] - tr[k], 0.0f));
                s.acc[n * LMAX + k] = acc;
            }
        half_barrier();
        // ---- C. delivered per feeder (fixed order), served flows, queue update, arrivals, counts, reward
        float r_node = 0.0f;
        // external arrivals of this step per flow group: flow_rate * activity_g(t) / 3600 * DT * xi_g  (x the link's share)
        const int piece = (t * 5) / 300;
        const float ra = p.flow_rate * activity_a(piece) / 3600.0f * DT, rb = p.flow_rate * activity_b(piece) / 3600.0f * DT;
        const float xg0 = ra * xi[ec * 4], xg1 = ra * xi[ec * 4 + 1], xg2 = rb * xi[ec * 4 + 2], xg3 = rb * xi[ec * 4 + 3];
        if (node) {
            float delivered = out;                           // a node feeding nothing discharges out of the network
            const int f0 = t_dnptr[n], f1 = t_dnptr[n + 1];
            if (f1 > f0) {
                delivered = 0.0f;
                for (int f = f0; f < f1; ++f) delivered += s.acc[t_dnpair[f]];
            }
            const float scale = out > 1e-6f ? delivered / fmaxf(out, 1e-6f) : 0.0f;
#pragma unroll
            for (int k = 0; k < LMAX; ++k)
                if (k < ns) {
                    const float served = D[k] * scale;
                    q[k] = q[k] - served + tr[k];
                    const int grp = t_group[n * LMAX + k];
                    float in = s.acc[n * LMAX + k];
                    if (grp >= 0) in += (grp == 0 ? xg0 : grp == 1 ? xg1 : grp == 2 ? xg2 : xg3) * t_share[n * LMAX + k];
                    tr[k] = in;
                    const float c = fminf(q[k], DET_CAP);
                    r_node -= c;
                }
        }
        float gsum = r_node;
        for (int off = 16; off > 0; off >>= 1) gsum += __shfl_xor(gsum, off, 32);
        const int t_new = t + 1;
        const bool is_done = t_new >= p.T;                   // atsc_env.py:189-191
        const bool rst = auto_reset && is_done;
        float wv[LMAX];
        if (node) {
#pragma unroll
            for (int k = 0; k < LMAX; ++k) {
                wv[k] = 0.0f;
                if (k < ns) {
                    if (rst) { q[k] = 0.0f; tr[k] = 0.0f; }
                    float w = fminf(q[k], DET_CAP);
                    if (p.norm_wave != 1.0f) w = w / p.norm_wave;
                    if (p.clip_wave >= 0.0f) w = fminf(fmaxf(w, 0.0f), p.clip_wave);
                    wv[k] = w;
                }
                if (k < L) s.acc[n * LMAX + k] = q[k];         // staging for the coalesced write-back (acc is dead now)
            }
            if (live) {
                prev[e * N + n] = (uint8_t)(rst ? 0 : a);
                if (p.per_agent_reward) reward[e * N + n] = r_node;
            }
        }
        half_barrier();
        if (live) {
#pragma unroll
            for (int it = 0; it < NMAX * LMAX / 32; ++it) {
                const int i = it * 32 + l32;
                if (i < N * L) {
                    const int r = (i * Lmagic) >> 16;
                    qs[e * N * L + i] = s.acc[r * LMAX + (i - r * L)];
                }
            }
        }
        half_barrier();
        if (node) {
#pragma unroll
            for (int k = 0; k < LMAX; ++k)
                if (k < L) { s.acc[n * LMAX + k] = tr[k]; s.cnt[n * LMAX + k] = wv[k]; }
        }
        half_barrier();
        if (live) {
#pragma unroll
            for (int it = 0; it < NMAX * LMAX / 32; ++it) {
                const int i = it * 32 + l32;
                if (i < N * L) {
                    const int r = (i * Lmagic) >> 16;
                    trs[e * N * L + i] = s.acc[r * LMAX + (i - r * L)];
                }
            }
        }
        if (live && l32 == 0) {
            if (!p.per_agent_reward) reward[e] = gsum;
            greward[e] = gsum;
            done[e] = is_done ? 1 : 0;
            ts[e] = rst ? 0 : t_new;
        }
        if (live && rst && l32 < 4) {
            const int ep = episode[e];
            const Philox4 r4 = philox4x32_10((uint32_t)(env_id_base + e), 0u, (uint32_t)ep, NMARL_STREAM_RESET,
                                             (uint32_t)seed, (uint32_t)(seed >> 32));
            const uint32_t w = l32 == 0 ? r4.x : l32 == 1 ? r4.y : l32 == 2 ? r4.z : r4.w;
            xi[e * 4 + l32] = 0.8f + 0.4f * u01_from_bits(w);
        }
        half_barrier();
        if (live && rst && l32 == 4) episode[e] = episode[e] + 1;
        // ---- D. the neighbour-gathered observation slab [N, (1 + m_max) * L]: one 4 L-byte piece per store, rows contiguous
        // (assembling the slab in LDS and streaming it out as float4 was slower: 2-way conflicts on the 110-float rows)
        if (live) {
            float* o = obs + e * N * W;
#pragma unroll 2
            for (int i = 0; i < N; ++i) {
#pragma unroll
                for (int slot = 0; slot < 9; ++slot)
                    if (slot <= tp.m_max) {
                        const int j = slot == 0 ? i : t_nbr[i * 8 + slot - 1];
                        if (l32 < L) o[i * W + slot * L + l32] = j >= 0 ? s.cnt[j * LMAX + l32] : 0.0f;
                    }
            }
        }
        half_barrier();
    }
}

__global__ __launch_bounds__(256) void net_reset_kernel(
    const nmarl_net_topo_t tp, const int64_t E, const uint8_t* __restrict__ mask, const float* __restrict__ u0,
    float* __restrict__ qs, float* __restrict__ trs, uint8_t* __restrict__ prev, int32_t* __restrict__ ts,
    float* __restrict__ xi, float* __restrict__ obs, const uint64_t seed, const int64_t env_id_base,
    int32_t* __restrict__ episode) {
    const int l32 = threadIdx.x & 31, sub = threadIdx.x >> 5;
    const int NL = tp.N * tp.L, NW = tp.N * tp.L * (1 + tp.m_max);
    constexpr int REPS = 8;
    for (int64_t e = (int64_t)blockIdx.x * REPS + sub; e < E; e += (int64_t)gridDim.x * REPS) {
        if (mask != nullptr && mask[e] == 0) continue;
        for (int i = l32; i < NL; i += 32) { qs[e * NL + i] = 0.0f; trs[e * NL + i] = 0.0f; }
        for (int i = l32; i < NW; i += 32) obs[e * NW + i] = 0.0f;
        if (l32 < tp.N) prev[e * tp.N + l32] = 0;
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

inline int net_blocks(int64_t E, int reps) {
    const int64_t b = (E + reps - 1) / reps;
    return (int)(b < 4096 ? b : 4096);
}

inline bool topo_ok(const nmarl_net_topo_t* tp) {
    return tp && tp->N > 0 && tp->N <= NMAX && tp->L > 0 && tp->L <= LMAX && tp->A > 0 && tp->A <= AMAX && tp->m_max > 0 &&
           tp->m_max <= 8 && tp->n_s && tp->image && ((uintptr_t)tp->image % 16) == 0;
}

}  // namespace

extern "C" int nmarl_net_step(const nmarl_net_params_t* p, const nmarl_net_topo_t* tp, int64_t E, const uint8_t* action,
                              float* q, float* transit, uint8_t* prev_action, int32_t* t, float* xi, float* obs,
                              float* reward, uint8_t* done, float* global_reward, int32_t auto_reset, uint64_t seed,
                              int64_t env_id_base, int32_t* episode, void* stream) {
    if (!p || !topo_ok(tp) || p->T <= 0 || p->norm_wave <= 0.f || E < 0 ||
        (E > 0 && (!action || !q || !transit || !prev_action || !t || !xi || !obs || !reward || !done || !global_reward)))
        return NMARL_EINVAL;
    if (auto_reset && !episode) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // replicas per block: enough blocks to cover the 256 CUs a few times at small E, 8 per block at large E
#define NMARL_NET_LAUNCH(R) hipLaunchKernelGGL(net_step_kernel<R>, dim3(net_blocks(E, R)), dim3(32 * R), 0, st, *p, *tp, E, action, q, \
                                               transit, prev_action, t, xi, obs, reward, done, global_reward, auto_reset, seed,   \
                                               env_id_base, episode)
    static int reps_env = -1;
    if (reps_env < 0) { const char* ev = getenv("NMARL_NET_REPS"); reps_env = ev ? atoi(ev) : 0; }
    const int reps = reps_env ? reps_env : (E <= 2048 ? 4 : 8);
    if (reps == 2) NMARL_NET_LAUNCH(2); else if (reps == 4) NMARL_NET_LAUNCH(4); else NMARL_NET_LAUNCH(8);
#undef NMARL_NET_LAUNCH
    return nmarl_check_launch();
}

extern "C" int nmarl_net_reset(const nmarl_net_topo_t* tp, int64_t E, const uint8_t* mask, const float* u0, uint64_t seed,
                               int64_t env_id_base, int32_t* episode, float* q, float* transit, uint8_t* prev_action,
                               int32_t* t, float* xi, float* obs, void* stream) {
    if (!topo_ok(tp) || E < 0 || (E > 0 && (!q || !transit || !prev_action || !t || !xi || !obs))) return NMARL_EINVAL;
    if (!u0 && !episode) return NMARL_EINVAL;
    if (E == 0) return NMARL_OK;
    hipLaunchKernelGGL(net_reset_kernel, dim3(net_blocks(E, 8)), dim3(256), 0, static_cast<hipStream_t>(stream), *tp, E, mask,
                       u0, q, transit, prev_action, t, xi, obs, seed, env_id_base, episode);
    return nmarl_check_launch();
}
