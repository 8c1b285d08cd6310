// Shared helpers for the gfx950 kernels of libnmarl_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/nmarl.h"

#define NMARL_WAVE 64

static inline int nmarl_check_launch() {
    return hipGetLastError() == hipSuccess ? NMARL_OK : NMARL_EHIP;
}

// Function attributes (the dynamic-LDS limit) are per DEVICE: a call site keeps one bit per device ordinal and sets the
// attributes of its kernels the first time it launches on each (a process may drive several GPUs, from several threads).
struct NmarlPerDeviceOnce {
    std::atomic<unsigned long long> seen{0};
    // -> ~0: this device is set up; otherwise the bit to hand to done() after setting the attributes (0: ordinal unknown,
    //    the attributes are then simply set on every launch -- hipFuncSetAttribute is cheap)
    unsigned long long pending() const {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0ull;
        const unsigned long long bit = 1ull << dev;
        return (seen.load(std::memory_order_acquire) & bit) ? ~0ull : bit;
    }
    void done(unsigned long long bit) { seen.fetch_or(bit, std::memory_order_release); }
};

// Sum of C partial results p[0], p[stride], ... in index order (s = (((0 + p0) + p1) + ...): the deterministic second stage of
// every two-stage reduction here).  The loads of U terms are issued together and only the adds are serial: the plain loop waits
// out one memory latency per term (128 chunks: ~50 us for a kernel that moves a few MB).
template <int U = 32>
__device__ __forceinline__ float nmarl_ordered_sum(const float* __restrict__ p, const int64_t stride, const int C) {
    float s = 0.0f;
    int c = 0;
    for (; c + U <= C; c += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(int64_t)(c + u) * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u];
    }
    if (c < C) {                      // the tail: clamped loads, terms past the end skipped (same order)
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[(int64_t)(c + u < C ? c + u : C - 1) * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) if (c + u < C) s += v[u];
    }
    return s;
}

// Philox4x32-10; contract shared with oracle/philox.py.
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u01_from_bits(uint32_t w) {
    return (float)(w >> 8) * 5.9604644775390625e-08f;  // 2^-24, exact
}

#define NMARL_STREAM_RESET 0u
#define NMARL_STREAM_ACTION 1u
#define NMARL_STREAM_GRID 2u

// The draw of nmarl_sample_actions (a2c.hip: sample_kernel) on a register-resident probability row of at
// most MAXA entries: argmax (mode 2) or numpy's choice = searchsorted(cumsum(p)/sum, u, 'right') with the
// caller's uniform (mode 0) or the Philox word of (env_id, agent, step) (mode 1).
template <int MAXA>
__device__ __forceinline__ int nmarl_draw_action(const float (&p)[MAXA], const int A, const int mode, const float u_host,
                                                 const uint64_t seed, const int64_t env_id, const int n,
                                                 const int64_t step) {
    int a = 0;
    if (mode == 2) {
        float best = p[0];
#pragma unroll
        for (int k = 1; k < MAXA; ++k)
            if (k < A && p[k] > best) { best = p[k]; a = k; }
        return a;
    }
    float uu = u_host;
    if (mode == 1) {
        const Philox4 r = philox4x32_10((uint32_t)env_id, (uint32_t)(n >> 2), (uint32_t)step, NMARL_STREAM_ACTION,
                                        (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint32_t w = (n & 3) == 0 ? r.x : (n & 3) == 1 ? r.y : (n & 3) == 2 ? r.z : r.w;
        uu = u01_from_bits(w);
    }
    double tot = 0.0;
#pragma unroll
    for (int k = 0; k < MAXA; ++k)
        if (k < A) tot += (double)p[k];
    double cum = 0.0;
#pragma unroll
    for (int k = 0; k < MAXA; ++k)
        if (k < A) {
            cum += (double)p[k];
            if (cum / tot <= (double)uu) a = k + 1;
        }
    return a > A - 1 ? A - 1 : a;
}



// ---- in-launch hand-off: residency and the test hook (defined in lstm_mfma.hip, used by lstm_bptt.hip too)
// compute units the residency decisions count with: the device's, or NMARL_TEST_FAKE_CUS from the environment (tests)
#define NMARL_INTERNAL __attribute__((visibility("hidden")))      // library-internal C++ helpers: not part of the C-ABI
NMARL_INTERNAL int nmarl_handoff_cus();
// consume one armed fault (nmarl_test_handoff_fault): true exactly for the nth hand-off launch after arming
NMARL_INTERNAL bool nmarl_handoff_take_fault();
constexpr unsigned NMARL_HANDOFF_MAX_SPINS = 1u << 20;      // ~0.1 s of s_sleep polling before a wave gives up
constexpr unsigned NMARL_HANDOFF_FAULT_SPINS = 1u << 12;    // the injected fault: give up quickly

// XCD-aware block -> (agent, row block).  Blocks are dispatched round-robin over the 8 XCDs (block b runs on XCD b % 8:
// observed, used for speed only -- MI355X_MICROARCH.md), each XCD has its own L2, and every block of an agent streams that
// agent's weight image(s).  Work item w = agent * blocks_per_agent + row block; XCD x is handed a CONTIGUOUS range of work
// items, so an agent's blocks sit on one XCD (two at a range boundary) and its image is fetched into one L2 instead of eight.
// N = 8 agents: identical to the plain b % N mapping.  Bijection for every grid size.
__device__ __forceinline__ void nmarl_xcd_work(const unsigned b, const unsigned grid, const int blocks_per_agent, int& agent,
                                               int& row_block) {
    const unsigned x = b & 7u, j = b >> 3, q = grid >> 3, r = grid & 7u;
    const unsigned w = x * q + (x < r ? x : r) + j;
    agent = (int)(w / (unsigned)blocks_per_agent);
    row_block = (int)(w - (unsigned)agent * (unsigned)blocks_per_agent);
}
