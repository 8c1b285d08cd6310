"""Philox4x32-10 counter-based RNG -- the RNG *contract* shared by the HIP
kernels (deeprl_network_amd/csrc/philox.h) and this oracle.  TEST INFRASTRUCTURE.

The reference draws from the global MT19937 stream (cacc_env.py:175,294,314;
utils.py:138), which cannot be split over E >> 1 replicas.  The batched path
therefore defines one counter-based stream:

    key     = (seed_lo, seed_hi)
    counter = (env_id, block, step_or_episode, stream)

    stream 0 (RESET):  block = 0, c2 = per-env episode index; word 0 -> U for
                       the initial-condition draw (cacc_env.py:294 / :314).
    stream 1 (ACTION): block = agent >> 2, c2 = global lock-step index;
                       word (agent & 3) -> U for the inverse-CDF action draw
                       (utils.py:138).
    stream 2 (GRID):   block = node, c2 = lock-step, arrival noise of the
                       synthetic ATSC grid.

    U = (word >> 8) * 2**-24   (exact in fp32, in [0, 1)).
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)

STREAM_RESET = 0
STREAM_ACTION = 1
STREAM_GRID = 2


def philox4x32(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32; all args broadcastable uint32-valued arrays.
    Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & MASK for c in
                      np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint64(int(k0) & 0xFFFFFFFF)
    k1 = np.uint64(int(k1) & 0xFFFFFFFF)
    for _ in range(rounds):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & MASK, lo1, (hi0 ^ c3 ^ k1) & MASK, lo0
        k0 = np.uint64((int(k0) + W0) & 0xFFFFFFFF)
        k1 = np.uint64((int(k1) + W1) & 0xFFFFFFFF)
    return [c.astype(np.uint32) for c in (c0, c1, c2, c3)]


def u01(word):
    """uint32 -> float32 uniform in [0,1) with 24 random bits."""
    return ((np.asarray(word, dtype=np.uint32) >> np.uint32(8)).astype(np.float32)
            * np.float32(2.0 ** -24))


def reset_uniform(seed, env_ids, episode):
    """U for the initial-condition draw of env `env_ids` in its `episode`-th episode."""
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    episode = np.asarray(episode, dtype=np.uint64)
    w = philox4x32(env_ids, 0, episode, STREAM_RESET, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return u01(w[0])


def action_uniform(seed, env_ids, n_agent, step):
    """[len(env_ids), n_agent] uniforms for the action draw at global lock-step `step`."""
    env_ids = np.asarray(env_ids, dtype=np.uint64)[:, None]
    agents = np.arange(n_agent, dtype=np.uint64)[None, :]
    w = philox4x32(env_ids, agents >> np.uint64(2), step, STREAM_ACTION,
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    w = np.stack(w, axis=-1)  # [E, N, 4]
    sel = (agents & np.uint64(3)).astype(np.int64)
    sel = np.broadcast_to(sel, w.shape[:2])
    return u01(np.take_along_axis(w, sel[..., None], axis=-1)[..., 0])
