/*
 * nmarl.h -- C-ABI of libnmarl_hip.so, the MI355X (gfx950) native hot path of
 * cts198859/deeprl_network: batched multi-agent environment rollout + A2C update.
 *
 * The reference is pure Python and has NO FFI / plugin API; its
 * seam is two Python duck-types.  This header is the boundary a maintainer of
 * the reference would bind (ctypes stubs in INTEGRATION.md): every entry point
 * names the reference function (file:line under the reference repo) whose
 * arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in `_host`;
 *   - the caller owns every buffer; nothing is allocated, no internal threads;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued asynchronously on it (hipGraph-capturable);
 *   - return value: 0 = ok, NMARL_EINVAL = bad argument, NMARL_EHIP = launch
 *     failed (hipGetLastError() != hipSuccess); no exception crosses the ABI;
 *   - E = number of lock-stepped environment replicas, N = agents per replica,
 *     A = actions per agent, T = n_step, H = LSTM width;
 *   - all arrays are row-major with the LAST index fastest.
 */
#ifndef NMARL_H
#define NMARL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMARL_OK 0
#define NMARL_EINVAL (-1)
#define NMARL_EHIP (-2)

#define NMARL_CACC_N 8        /* vehicles per platoon: one 8-lane group per replica  */
#define NMARL_CACC_NF 5       /* features per vehicle, cacc_env.py:54-65              */
#define NMARL_CACC_OBS 15     /* own 5 + two neighbour slots of 5 (zero padded)       */

int nmarl_abi_version(void);
/* "NMARL_SRC_HASH=<hex>": hash of the sources this binary was compiled from (deeprl_network_amd/build.py);
 * the Python binding refuses a library whose hash differs from the sources next to it. */
const char* nmarl_source_hash(void);

/* ------------------------------------------------------------------------- */
/* CACC platoon environment -- replaces envs/cacc_env.py                      */
/* ------------------------------------------------------------------------- */

/* The scalars of CACCEnv._load_config (cacc_env.py:320-343). */
typedef struct nmarl_cacc_params {
    float dt;            /* control_interval_sec                                  */
    float h_min;         /* headway_min     (collision threshold, :42)            */
    float h_star;        /* headway_target                                        */
    float h_s;           /* headway_st      (OVM, :360-369)                       */
    float h_g;           /* headway_go                                            */
    float v_max;         /* speed_max                                             */
    float v_star;        /* speed_target                                          */
    float u_min;         /* accel_min                                             */
    float u_max;         /* accel_max                                             */
    float reward_a;      /* reward_v                                              */
    float reward_b;      /* reward_u                                              */
    float G;             /* collision_penalty                                     */
    int32_t T;           /* episode_length_sec / dt (600)                         */
    int32_t batch_size;  /* ENV_CONFIG batch_size: collision ends the episode only
                            at a multiple of it (cacc_env.py:231-233)             */
    int32_t scenario;    /* 0 = catchup (:285-299), 1 = slowdown (:306-318)       */
    int32_t train_mode;  /* 1: add the soft-collision term (:48-49)               */
    int32_t per_agent_reward; /* coop_gamma >= 0: reward is [E,N]; else the global
                            scalar is broadcast, reward is [E] (:236-237)         */
    int32_t compact_obs; /* 1: obs is [E,8,5], each vehicle's OWN features only (_get_veh_state,
                            :54-65; the 'ma2c' observation) and the consumer gathers the
                            neighbours; 0: the 'ia2c' pre-gathered [E,8,15] (:70-73)       */
} nmarl_cacc_params_t;

/*
 * CACCEnv.reset (cacc_env.py:166-189) + _init_catchup/_init_slowdown (:285-318)
 * + the first _get_state (:67-79), for the replicas selected by `mask`.
 *
 *   mask      [E] u8 or NULL (= all replicas)
 *   u0        [E] f32 uniforms in [0,1) that stand for the reference's single
 *             np.random.rand() draw (:294/:314), or NULL: then
 *             U = Philox4x32-10(key=seed, ctr=(env_id_base+e, 0, episode[e], 0))
 *             (contract: oracle/philox.py) and episode[e] is post-incremented.
 *   episode   [E] i32 per-replica episode counter (may be NULL iff u0 != NULL)
 *   h,v,u     [E,8] f32 headway / speed / constrained acceleration
 *   t         [E] i32 step in episode; collided [E] u8; v0_init [E] f32 (speed
 *             of the leading vehicle at t=0; its profile v0s[t] is analytic)
 *   obs       [E,8,15] f32 gathered observation (see nmarl_cacc_step); [E,8,5] with p->compact_obs
 *   fp        [E,8,A] f32 fingerprints, set to 1/A (:184) -- may be NULL
 */
int nmarl_cacc_reset(const nmarl_cacc_params_t* p, int64_t E,
                     const uint8_t* mask, const float* u0,
                     uint64_t seed, int64_t env_id_base, int32_t* episode,
                     float* h, float* v, float* u, int32_t* t, uint8_t* collided,
                     float* v0_init, float* obs, float* fp, int32_t A, void* stream);

/*
 * CACCEnv.step (cacc_env.py:191-242) fused with _get_accel (:31-38),
 * OVMCarFollowing.get_vh/get_accel (:360-385), _constrain_speed (:24-29),
 * _get_reward (:40-52) and _get_state/_get_veh_state (:54-79) for E replicas.
 *
 *   action    [E,8] u8 in 0..3 -> (alpha,beta) = a_map[action] (:275)
 *   obs       [E,8,5] f32 own features (p->compact_obs), or
 *             [E,8,15] f32: slot 0 = own 5 features, slots 1..2 = the
 *             neighbours' 5 features in ascending vehicle index, zero padded
 *             (the 'ia2c' concatenation of :70-73 and lstm_comm's xi,
 *             agents/utils.py:192-193; the 'ma2c' 5-vector is columns 0..4)
 *   reward    [E] f32 (global scalar) or [E,8] if per_agent_reward
 *   done      [E] u8;  global_reward [E] f32 (the 4th return value, :229)
 *   auto_reset != 0: a replica that reports done is re-initialised in the same
 *             launch exactly as nmarl_cacc_reset(u0 = NULL) would; `obs` then
 *             holds the first observation of the new episode.  (The reference
 *             discards next_ob of a finished episode, utils.py:188-190.)
 */
int nmarl_cacc_step(const nmarl_cacc_params_t* p, int64_t E, const uint8_t* action,
                    float* h, float* v, float* u, int32_t* t, uint8_t* collided,
                    float* v0_init, float* obs, float* reward, uint8_t* done,
                    float* global_reward, int32_t auto_reset, uint64_t seed,
                    int64_t env_id_base, int32_t* episode, void* stream);

/*
 * nmarl_cacc_step AND the next lock-step's input encoders in one launch (batched rollout, compact observation only):
 * after the step, for every agent i and replica e
 *     out[i,e,  0: 64] = act([x_i | x_nbr(i,0) | x_nbr(i,1)] @ w_ob[i] + b_ob[i])      x = the new observation [E,8,5]
 *     out[i,e, 64:128] = act([fp_nbr(i,0) | fp_nbr(i,1)] @ w_fp[i] + b_fp[i])           (n_parts == 2)
 * i.e. `fc` of policies.py:145 / 176-181 and w_ob / w_fp of agents/utils.py:186-199 -- the arithmetic of
 * nmarl_fc_fwd_multi, bit for bit.  w_ob [N][15][64], w_fp [N][8][64], b_* [N][64], fp [N][E][4] (the policies of the
 * lock-step just decided: next step's fingerprints), nbr_idx [N][2] (-1 padded), out [N][E][out_row] (agent stride
 * out_sn).  act as in nmarl_bias_act.  Missing neighbours contribute zeros.
 */
typedef struct nmarl_cacc_encode {
    const float *w_ob, *b_ob, *w_fp, *b_fp, *fp;
    const int32_t* nbr_idx;
    float* out;
    int64_t w_ob_sn, b_ob_sn, w_fp_sn, b_fp_sn, fp_sn, out_sn, out_row;
    int32_t act, n_parts;
} nmarl_cacc_encode_t;
int nmarl_cacc_step_encode(const nmarl_cacc_params_t* p, int64_t E, const uint8_t* action,
                           float* h, float* v, float* u, int32_t* t, uint8_t* collided,
                           float* v0_init, float* obs, float* reward, uint8_t* done,
                           float* global_reward, int32_t auto_reset, uint64_t seed,
                           int64_t env_id_base, int32_t* episode, const nmarl_cacc_encode_t* enc, void* stream);

/* ------------------------------------------------------------------------- */
/* Synthetic (SUMO-free) 5x5 ATSC grid -- contract of envs/atsc_env.py +      */
/* envs/large_grid_env.py; dynamics specified in oracle/grid_ref.py           */
/* ------------------------------------------------------------------------- */
#define NMARL_GRID_N 25       /* intersections, node i = row*5+col = nt{i+1}           */
#define NMARL_GRID_NF 12      /* wave features = signal links per node (:25-26)         */
#define NMARL_GRID_OBS 60     /* own 12 + 4 neighbour slots (ascending index, 0 padded) */
#define NMARL_GRID_LANES 6    /* physical incoming lanes per node                       */

typedef struct nmarl_grid_params {
    float norm_wave;          /* atsc_env.py:91-92                                       */
    float clip_wave;          /* atsc_env.py:93-94 (< 0: no clip)                        */
    float peak1;              /* peak_flow1, large_grid_env.py:50                        */
    float peak2;              /* peak_flow2                                              */
    int32_t T;                /* ceil(episode_length_sec / control_interval_sec) = 720   */
    int32_t per_agent_reward; /* coop_gamma >= 0 -> reward [E,25], else global [E]       */
    int32_t compact_obs;      /* 1: obs [E,25,12] = every node's OWN wave vector (what the reference hands an agent,
                                 atsc_env.py:253-262; the consumer gathers the neighbours); 0: the gathered [E,25,60] slab */
    int32_t objective;        /* atsc_env.py:87, 383-418: 0 `queue` (the shipped configs), 1 `wait`, 2 `hybrid` = queue + coef_wait * wait */
    float coef_wait;          /* atsc_env.py:96 (hybrid only)                            */
    float* head_wait;         /* [E,25,6] f32 state, REQUIRED when objective != 0 (else NULL): seconds the head vehicle of each lane
                                 has been standing (synthetic stand-in of getWaitingTime of the front vehicle, oracle/grid_ref.py
                                 step 6); reset to 0 with the replica                    */
} nmarl_grid_params_t;

/*
 * TrafficSimulator.reset (atsc_env.py:164-179) for the replicas selected by mask: empty
 * network (init_density = 0), prev_action = 0 (:509-513), t = 0, and the per-replica
 * demand scale xi[e,g] = 0.8 + 0.4*U for the 4 flow groups; U from u0 [E,4] or from
 * Philox4x32-10(key=seed, ctr=(env_id_base+e, 0, episode[e], 0)) words 0..3.
 * q, transit [E,25,6] f32; prev_action [E,25] u8; t [E] i32; xi [E,4]; obs [E,25,60] ([E,25,12] with p->compact_obs).
 */
int nmarl_grid_reset(const nmarl_grid_params_t* p, int64_t E, const uint8_t* mask, const float* u0,
                     uint64_t seed, int64_t env_id_base, int32_t* episode, float* q, float* transit,
                     uint8_t* prev_action, int32_t* t, float* xi, float* obs, void* stream);
/*
 * TrafficSimulator.step (atsc_env.py:181-207): phase = action[e,i] in 0..4, 2 s yellow
 * handling (:216-240), 5 s of store-and-forward traffic, `wave` observation (:420-462,
 * :502-504) and queue reward (:383-418).  obs [E,25,60]: slot 0 own 12 features, slots
 * 1..4 the neighbours' in ascending node index (lstm_ic3 / lstm_comm concatenation); with p->compact_obs obs [E,25,12]
 * = slot 0 only.
 * reward [E] (global) or [E,25]; done [E] u8 when t reaches T; auto_reset as for CACC.
 */
int nmarl_grid_step(const nmarl_grid_params_t* p, int64_t E, const uint8_t* action, float* q,
                    float* transit, uint8_t* prev_action, int32_t* t, float* xi, float* obs,
                    float* reward, uint8_t* done, float* global_reward, int32_t auto_reset,
                    uint64_t seed, int64_t env_id_base, int32_t* episode, void* stream);

/* ------------------------------------------------------------------------- */
/* Synthetic signalised NETWORK with heterogeneous intersections -- contract   */
/* of envs/atsc_env.py + envs/real_net_env.py (Monaco: 28 nodes, 2..6 phases   */
/* over 2..22 signal links); dynamics specified in oracle/realnet_ref.py       */
/* ------------------------------------------------------------------------- */
typedef struct nmarl_net_params {
    float norm_wave;          /* atsc_env.py:91-92 (config_*_net.ini: 1.0)               */
    float clip_wave;          /* < 0: no clip (config_*_net.ini: -1)                     */
    float flow_rate;          /* veh/h of one flow, real_net_env.py:147                  */
    int32_t T;                /* ceil(episode_length_sec / control_interval_sec) = 720   */
    int32_t per_agent_reward; /* coop_gamma >= 0 -> reward [E,N], else global [E]        */
} nmarl_net_params_t;

/* The static network (built by the host from the reference's NODES / PHASES tables): N <= 32 nodes, L <= 24 = widest
 * phase string, A <= 8 = most phases, m_max <= 8 = most listed neighbours.
 * n_s [N] (device) links per node.  image (device, 16-byte aligned, NMARL_NET_IMAGE_BYTES): all other tables packed in
 * the layout the step kernel keeps in LDS (rows padded to 24 links):
 *   OFF_GREEN  u8  [32][8][24]   0 r / 1 G / 2 g, indexed [node][phase][link]
 *   OFF_SRC    i16 [32][24]      feeding node of a link or -1 (external entry)
 *   OFF_GROUP  i8  [32][24]      flow group of an external link, else -1
 *   OFF_SHARE  f32 [32][24]      its share of the group's demand
 *   OFF_FAN    f32 [32]          number of links a node feeds
 *   OFF_DNPTR  i16 [33], OFF_DNPAIR i16 [768]   the links fed by each node, ascending, as node*24 + link
 *   OFF_NBR    i8  [32][8]       neighbour table of the nets (ascending index, -1 padded) */
#define NMARL_NET_OFF_GREEN 0
#define NMARL_NET_OFF_SRC 6144
#define NMARL_NET_OFF_GROUP 7680
#define NMARL_NET_OFF_SHARE 8448
#define NMARL_NET_OFF_FAN 11520
#define NMARL_NET_OFF_DNPTR 11648
#define NMARL_NET_OFF_DNPAIR 11728
#define NMARL_NET_OFF_NBR 13264
#define NMARL_NET_IMAGE_BYTES 13568
typedef struct nmarl_net_topo {
    int32_t N, L, A, m_max;
    const int32_t* n_s;
    const uint8_t* image;
} nmarl_net_topo_t;

/* TrafficSimulator.reset / step (atsc_env.py:164-207) as nmarl_grid_reset / nmarl_grid_step, for the network:
 * q, transit [E,N,L] f32; prev_action [E,N] u8; t [E]; xi [E,4]; obs [E,N,L*(1+m_max)] (slot 0 own `wave`, slots
 * 1.. the listed neighbours' in ascending node index, each L wide and zero padded -- the padded input layout of the
 * heterogeneous nets); action[e,i] in 0..n_a_i-1; reward [E] or [E,N]. */
int nmarl_net_reset(const nmarl_net_topo_t* tp, int64_t E, const uint8_t* mask, const float* u0, uint64_t seed,
                    int64_t env_id_base, int32_t* episode, float* q, float* transit, uint8_t* prev_action,
                    int32_t* t, float* xi, float* obs, void* stream);
int nmarl_net_step(const nmarl_net_params_t* p, const nmarl_net_topo_t* tp, int64_t E, const uint8_t* action,
                   float* q, float* transit, uint8_t* prev_action, int32_t* t, float* xi, float* obs,
                   float* reward, uint8_t* done, float* global_reward, int32_t auto_reset, uint64_t seed,
                   int64_t env_id_base, int32_t* episode, void* stream);

/* ------------------------------------------------------------------------- */
/* Neighbourhood aggregation over the fixed adjacency (agent-major [N,E,F])   */
/* ------------------------------------------------------------------------- */
/*
 * nbr_idx [N, m_max] i32 (device): the neighbours of agent i in ascending
 * agent index (the order of tf.boolean_mask(.., masks[i])), left packed,
 * -1 padded.  m_max <= 8, N*m_max <= 1024.
 *
 * gather: y[i, e, k*F + f] = x[nbr_idx[i,k], e, f]  (0 where padded)
 *   replaces boolean_mask + reshape of agents/utils.py:192-195 (lstm_comm:
 *   neighbour h / fingerprints / observations) and policies.py:305.
 * mean:   y[i, e, f] = mean_k x[nbr_idx[i,k], e, f]
 *   replaces tf.reduce_mean(tf.boolean_mask(out_m, masks[i])) of
 *   agents/utils.py:395 (lstm_ic3, CommNet).
 * *_bwd:  the exact adjoints (dx overwritten, deterministic, no atomics).
 */
int nmarl_nbr_gather_fwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                         const float* x, float* y, void* stream);
int nmarl_nbr_gather_bwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                         const float* dy, float* dx, void* stream);
int nmarl_nbr_mean_fwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                       const float* x, float* y, void* stream);
int nmarl_nbr_mean_bwd(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                       const float* dy, float* dx, void* stream);
/* The adjoints with a second addend, dx = (sum over the fan-in) + add [N,E,F]: the manual BPTT of the coupled nets adds
 * the recurrent part of dL/dh_{t-1} to the message part in the same pass (add may alias dx). */
int nmarl_nbr_gather_bwd_add(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                             const float* dy, const float* add, float* dx, void* stream);
int nmarl_nbr_mean_bwd_add(int64_t E, int32_t N, int32_t F, int32_t m_max, const int32_t* nbr_idx,
                           const float* dy, const float* add, float* dx, void* stream);
/*
 * One-hot of the neighbours' actions for the centralised critic:
 * y[i, e, k*A + a] = (action[e, nbr_idx[i,k]] == a); replaces
 * tf.one_hot(boolean_mask(action, mask_i)) of policies.py:66-68, 305 and
 * CACCEnv.get_neighbor_action (cacc_env.py:125-129).  action is env-major [E,N] u8;
 * y_agent_stride = floats between consecutive agents of y (>= E*m_max*A; lets y be
 * slot t of an [N,T,E,m_max*A] rollout buffer).
 */
int nmarl_nbr_onehot(int64_t E, int32_t N, int32_t A, int32_t m_max, const int32_t* nbr_idx,
                     const uint8_t* action, float* y, int64_t y_agent_stride, void* stream);

/* ------------------------------------------------------------------------- */
/* Policy / update pointwise kernels                                          */
/* ------------------------------------------------------------------------- */
/*
 * LSTM cell with done reset -- agents/utils.py:102-113 (lstm), 199-208
 * (lstm_comm), 401-408 (lstm_ic3).  z [N,E,4H] = s*Wx + (h*(1-done))*Wh (bias
 * NOT yet added), bias [N,4H], c_prev [N,E,H], done [E] f32 in {0,1}.
 *   i,f,o = sigmoid(z+b), u = tanh(z+b)   (gate order i,f,o,u)
 *   c' = f * (c_prev*(1-done)) + i*u ;  h' = o * tanh(c')
 * Every tensor has contiguous [E,W] panels and its own AGENT STRIDE `*_sn` in floats
 * (multiple of 4), so slot t of an [N,T,E,W] sequence buffer is passed in place.
 * z2 (may be NULL) is a second addend of the pre-activation (the x-side part s*Wx kept apart so
 * that the recurrent GEMM needs no copy+accumulate).
 * gates [N,E,4H] receives the post-activation i,f,o,u (saved for backward; may
 * alias z; NULL = inference, not written).  H % 4 == 0, 16-byte aligned panels.
 * bwd: dz [N,E,4H] (= d bias before the reduction over E), dc_prev [N,E,H];
 * dh / dh2 / dc_new may be NULL (= 0); dL/dh' = dh + dh2 (head part + recurrent part).
 */
int nmarl_lstm_cell_fwd(int64_t E, int32_t N, int32_t H, const float* z, int64_t z_sn,
                        const float* z2, int64_t z2_sn, const float* bias, int64_t bias_sn, const float* c_prev, int64_t c_prev_sn,
                        const float* done, float* gates, int64_t gates_sn, float* c_new,
                        int64_t c_new_sn, float* h_new, int64_t h_new_sn, void* stream);
int nmarl_lstm_cell_bwd(int64_t E, int32_t N, int32_t H, const float* gates, int64_t gates_sn,
                        const float* c_prev, int64_t c_prev_sn, const float* c_new, int64_t c_new_sn,
                        const float* done, const float* dh, int64_t dh_sn, const float* dh2,
                        int64_t dh2_sn, const float* dc_new, int64_t dc_sn, float* dz, int64_t dz_sn,
                        float* dc_prev, int64_t dc_prev_sn, void* stream);
/*
 * Fused recurrent GEMM + cell on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32), H = 64 only:
 *   z = zadd1 (+ zadd2) + (h_in * (1-done)) @ wh ;  (gates, c_new, h_new) = cell(z + bias, c_prev, done)
 * Same maths as a batched GEMM followed by nmarl_lstm_cell_fwd, but the [rows,4H] pre-activation never
 * reaches HBM.  h_in [N,E,H], wh [N,H,4H], bias [N,4H], zadd1/zadd2 [N,E,4H] (zadd2 may be NULL: the
 * x-side product s*Wx, and for NeurComm additionally the message term), c_prev/c_new/h_new [N,E,H],
 * gates [N,E,4H] or NULL; agent strides `*_sn` in floats (multiples of 4); h_new may alias h_in and c_new
 * may alias c_prev.  Returns NMARL_EINVAL for H != 64 (callers then use GEMM + nmarl_lstm_cell_fwd).
 */
int nmarl_lstm_step_fused(int64_t E, int32_t N, int32_t H, const float* h_in, int64_t h_sn,
                          const float* wh, int64_t wh_sn, const float* bias, int64_t bias_sn,
                          const float* zadd1, int64_t zadd1_sn, const float* zadd2, int64_t zadd2_sn,
                          const float* c_prev, int64_t c_prev_sn, const float* done, float* gates,
                          int64_t gates_sn, float* c_new, int64_t c_new_sn, float* h_new,
                          int64_t h_new_sn, void* stream);
/*
 * The fused step with an actor or critic head in its epilogue (H = 64, A <= 8): what
 * Trainer._get_policy / _get_value (utils.py:129-149) need from one lock-step, with h' still on chip.
 *   kind 1 (forward 'p', policies.py:50-57 + utils.py:135-141):
 *       pi = softmax(h' @ w + b), w [N,H,A], b [N,A]  -> pi_out [N,E,A] (the next fingerprint slot),
 *       action = nmarl_sample_actions' draw from pi (same modes / Philox counters) -> act_out [E,Ntot] u8
 *   kind 2 (forward 'v', policies.py:59-77):
 *       v = [h', onehot(neighbour actions)] @ w + b, w [N,H+m_max*A] (neighbour-major one-hot block, rows
 *       of absent neighbours unused), b [N,1]; neighbour actions are read from act_in [E,Ntot] u8 through
 *       nbr_idx [N,m_max] (-1 padded, as in nmarl_nbr_onehot)  -> v_out [N,E]
 *   kind 3 (forward 'p' AND forward 'v' of one lock-step, quirk Q1, for nets whose recurrence has no cross-agent
 *       term): kind 1, then the value re-step from the state just produced -- z = the same addend + (h'(1-done)) @ wh,
 *       cell from c'(1-done) -- and v_h = h'' @ w2[:H] + b2 -> v_out [N,E]; h'' / c'' are not stored (the reference
 *       discards them, policies.py:124-133).  The critic's neighbour-action term needs the other agents' draws of
 *       this lock-step: add it with nmarl_nbr_action_value_fwd(accumulate = 1).
 * Agent strides in floats; `u` as in nmarl_sample_actions (mode 0).  A NULL head or kind 0 is
 * nmarl_lstm_step_fused.  NMARL_EINVAL for A > 8 (callers compose GEMM + softmax + nmarl_sample_actions).
 */
typedef struct nmarl_head {
    int32_t kind, A, mode, m_max;
    const float* w; int64_t w_sn;
    const float* b; int64_t b_sn;
    float* pi_out; int64_t pi_sn;
    uint8_t* act_out;
    const float* u;
    uint64_t seed;
    int64_t env_id_base, step;
    const int64_t* step_dev;
    const uint8_t* act_in;
    const int32_t* nbr_idx;
    float* v_out; int64_t v_sn;
    const float* w2; int64_t w2_sn;      /* kind 3: the critic's weights / bias (w, b are the actor's) */
    const float* b2; int64_t b2_sn;
} nmarl_head_t;
int nmarl_lstm_step_fused_head(int64_t E, int32_t N, int32_t H, const float* h_in, int64_t h_sn,
                               const float* wh, int64_t wh_sn, const float* bias, int64_t bias_sn,
                               const float* zadd1, int64_t zadd1_sn, const float* zadd2, int64_t zadd2_sn,
                               const float* c_prev, int64_t c_prev_sn, const float* done, float* gates,
                               int64_t gates_sn, float* c_new, int64_t c_new_sn, float* h_new,
                               int64_t h_new_sn, const nmarl_head_t* head, void* stream);
/*
 * The fused step with the x-side product inside: the WHOLE pre-activation of agents/utils.py:102-113 (lstm),
 * 199-208 (lstm_comm), 401-408 (lstm_ic3) on the matrix cores,
 *   z = [x | h_in * (1-done)] @ [wx; wh] + bias (+ zadd1) (+ zadd2),   x [N,E,KX], KX in {0, 32, ..., 256},
 * then cell + optional head exactly as nmarl_lstm_step_fused_head (head NULL / kind 0: none; gates may be requested).
 * x: the LSTM input of one lock-step -- fc output (KX = n_fc), the [fcs | fcp] concatenation of policies.py:176-181
 * (KX = 2 n_fc), lstm_comm's [hx | hp | hm] (KX = 3 H), lstm_ic3's s (KX = H) -- with agent stride x_sn and row pitch
 * x_row >= KX (floats, multiples of 4; a column block of a wider buffer is read in place).  zadd1 / zadd2 may be NULL.
 * KX2 > 0: the LAST KX2 columns of x come from a second tensor x2 [N,E,KX2] (x then holds the first KX - KX2): the
 * value re-step of a coupled net re-uses the observation / fingerprint encodings and swaps in the re-computed message.
 * The weights come as the chunked image nmarl_lstm_wimage builds from wx [N,KX,4H] and wh [N,H,4H] (agent strides in
 * floats): per agent nmarl_lstm_wimage_floats(KX) = (KX+64)*320 floats, image[k][c][t] = W[k][16t+c] for t < 16, 4
 * floats of padding per (k,c); rebuild it whenever the weights change (once per update).  H = 64 only.
 */
int nmarl_lstm_wimage_floats(int32_t KX);
int nmarl_lstm_wimage(int32_t N, int32_t KX, const float* wx, int64_t wx_sn, const float* wh, int64_t wh_sn,
                      float* img, int64_t img_sn, void* stream);
int nmarl_lstm_step_x(int64_t E, int32_t N, int32_t H, int32_t KX, const float* x, int64_t x_sn, int64_t x_row,
                      int32_t KX2, const float* x2, int64_t x2_sn, int64_t x2_row,
                      const float* h_in, int64_t h_sn, const float* img, int64_t img_sn, const float* bias,
                      int64_t bias_sn, const float* zadd1, int64_t zadd1_sn, const float* zadd2, int64_t zadd2_sn,
                      const float* c_prev, int64_t c_prev_sn, const float* done, float* gates, int64_t gates_sn,
                      float* c_new, int64_t c_new_sn, float* h_new, int64_t h_new_sn, const nmarl_head_t* head,
                      void* stream);
/*
 * nmarl_lstm_step_x for the policy / value step of a COUPLED net, its message term computed inside the kernel from the
 * neighbours' previous, un-masked h (quirk Q3; h_in of the other agents, agent stride h_sn) instead of by separate
 * gather / GEMM / bias-activation launches:
 *   kind 1  lstm_comm (agents/utils.py:182-199): hm = relu([h_j : j in nbr(i)] @ w_msg + b_msg), K = 64*m_max <= 128;
 *           the LSTM input is [x (KX-64 columns: [hx | hp]) | hm]
 *   kind 2  lstm_ic3 (agents/utils.py:395-400): s = mean_j(h_j) @ w_msg + b_msg + enc, K = 64; the LSTM input is s (KX = 64)
 *   kind 3  lstm_dial (agents/utils.py:515-599): hm = relu([msg_j : j in nbr(i)] @ w_msg + b_msg), K = 64*m_max <= 128, from
 *           the SENDERS' message vectors src [N,E,64] (msg_j = relu(h_j w_mfc + b), one nmarl_fc_fwd of the caller on the
 *           un-masked previous h; agent stride src_sn, rows contiguous); s = hm + enc is the LSTM input (KX = 64).  out2 (may
 *           be NULL) receives hm, out receives s.  src may not overlap out / out2; h_new MAY be h_in (the pre-phase reads no
 *           h).  Heads 1 and 2 only (the re-step's message vectors need the senders' NEW h).  Head 1 with next_out != NULL
 *           also runs the sender layer on the new h: next_out [N,E,64] (agent stride next_out_sn, rows contiguous; may not
 *           overlap src) = relu(h_new @ w_mfc + b_mfc), next_img = nmarl_lstm_msg_wimage of w_mfc [N,64,64], next_b [N,64]
 *           -- the src of the value re-step and of the next lock-step's policy step, without an fc launch in between.
 * w_msg comes as the image of nmarl_lstm_msg_wimage (K*64 floats per agent, K <= 256: image[k][c][t] = w_msg[k][4c+t]); nbr_idx
 * [N,m_max] (-1 padded, ascending); enc [N,E,64] with row pitch enc_row (kind 2).  out (may be NULL): where the 64
 * computed columns are stored for the update's backward ([N,E,64] view, row pitch out_row).  head: kind 1 or 2.
 * NO ALIASING: the pre-phase reads the other agents' panels of h_in while their blocks write h_new, so no [E,64] panel
 * of h_new may overlap a panel of h_in (NMARL_EINVAL otherwise); step in place only without the in-kernel message term.
 * head kind 3 (policy step AND the value re-step of quirk Q1 in one launch, Trainer._get_policy + _get_value,
 * utils.py:129-149): the re-step's message term is computed from the neighbours' NEW h, which their blocks hand over
 * inside the launch (write-through stores + one flag per wave, lstm_mfma.hip).  Needs msg->sync: a buffer of
 * nmarl_lstm_step_sync_words(E, N) 32-bit words the caller zeroes once and then leaves alone (word 2 becomes non-zero
 * if a block ever waited in vain, i.e. the launch shared the device with other work); every block must be resident, so
 * N * ceil(E / 128) may not exceed nmarl_handoff_capacity(1, K) (NMARL_EINVAL otherwise: use the two launches).  v_out
 * receives the critic's h part only (as nmarl_lstm_step_x with kind 3); gates / c_new / h_new / out are the POLICY step's.
 * msg->status (may be NULL): the hand-off status words (see nmarl_handoff_capacity below) -- word 0 is set, and stays
 * set, when a wave gives up waiting; nmarl_rmsprop_tf_clip_guarded then applies nothing (fail closed).
 */
typedef struct nmarl_msg {
    int32_t kind, m_max, K, pad_;
    const int32_t* nbr_idx;
    const float* img; int64_t img_sn;
    const float* b; int64_t b_sn;
    const float* enc; int64_t enc_sn, enc_row;
    float* out; int64_t out_sn, out_row;
    uint32_t* sync;     /* head kind 3 only: nmarl_lstm_step_sync_words(E, N) words, zeroed ONCE by the caller */
    /* head kind 3 + kind 2 only, ob != NULL: lstm_ic3's observation encoder enc = tanh([x_i | x_nbr] W_ob + b_ob)
     * (agents/utils.py:395-399) runs inside the launch too and WRITES enc before using it: ob [E][N][ob_F] the env's compact
     * observation (row pitch ob_row floats), ob_nbr [N, ob_segs] = own index then the neighbours (ascending, -1 padded),
     * ob_F % 4 == 0, ob_F * ob_segs <= 64; ob_img = nmarl_lstm_msg_wimage of W_ob zero-padded to 64 rows, ob_b [N,64] */
    const float* ob; int64_t ob_row;
    int32_t ob_F, ob_segs;
    const int32_t* ob_nbr;
    const float* ob_img; int64_t ob_img_sn;
    const float* ob_b; int64_t ob_b_sn;
    int32_t* status;    /* head kind 3 only, may be NULL: hand-off status words (word 0 <- 1 when a wave gives up) */
    const float* src; int64_t src_sn;               /* kind 3: the senders' message vectors [N,E,64] */
    float* out2; int64_t out2_sn, out2_row;         /* kind 3, may be NULL: hm before enc is added ([N,E,64] view) */
    const float* next_img; int64_t next_img_sn;     /* kind 3, head 1, with next_out: image of w_mfc */
    const float* next_b; int64_t next_b_sn;
    float* next_out; int64_t next_out_sn;           /* kind 3, head 1, may be NULL: relu(h_new @ w_mfc + b_mfc) */
    float* mean_out; int64_t mean_out_sn, mean_out_row;   /* kind 2, may be NULL: the policy step's mean_j(h_j) rows ([N,E,64] view), the
                                                           * input of the message layer -- its weight gradient is mean(h)^T d1 */
    /* head kind 3, kinds 1 / 2 (round 6): the value re-step's message term -- from the neighbours' NEW, un-masked h (quirk Q3) -- IS
     * the message term of the NEXT lock-step's policy step (utils.py:129-149 called again at t + 1 on the states t left).  carry_out
     * [N,E,64] (rows contiguous, may be NULL) receives it (kind 1: relu(m W + b); kind 2: mean W + b, before enc is added); carry_in
     * (may be NULL: compute it) hands the previous launch's over, and the launch then reads no neighbour row and multiplies nothing in
     * front of its K loop; mean_out is not written then -- mean_next (kind 2, may be NULL; layout as mean_out) of the previous launch
     * was.  carry_in may be carry_out (a row is read and written by the same lane).  The caller computes (carry_in = NULL) whenever
     * the states were changed between the two launches (a batch boundary's reset of finished replicas). */
    const float* carry_in; int64_t carry_in_sn;
    float* carry_out; int64_t carry_out_sn;
    float* mean_next; int64_t mean_next_sn, mean_next_row;
} nmarl_msg_t;
/*
 * The input encoders of a lock-step INSIDE the policy + value launch (head kind 3, no message term; round 5): IA2C-FP on
 * CACC, policies.py:176-181 -- s = [relu([x_i | x_nbr] W_ob + b_ob) | relu([pi_nbr] W_fp + b_fp)] (KX = 128) is formed by a
 * register-only matrix-core pre-phase from the env's COMPACT observation and the previous-step policies, written once to
 * `out` (the update's saved LSTM input; may be NULL: bootstrap step) and consumed as the K loop's first four chunks -- no
 * encoder launch, no re-read of s.  Replaces nmarl_fc_fwd_multi / the encoder half of nmarl_cacc_step_encode in front of
 * nmarl_lstm_step_x; same function up to fp32 summation order (the matrix cores add four products at a time).
 *   ob [E][N][F = 5] (row pitch ob_row floats), fp [N][E][A = 4] (agent stride fp_sn); w_ob [N][15][64], b_ob [N][64],
 *   w_fp [N][8][64], b_fp [N][64]: the parameter tensors as they are (agent strides *_sn); out [N][E][128] view (agent
 *   stride out_sn, row pitch out_row); nbr: HOST copy of the neighbour table [N][m_max = 2], ascending, -1 padded (N <= 32).
 * Round 6 -- the observation encoder ALONE (w_fp = NULL; KX = 64, out [N][E][64]): IA2C (policies.py:145, `fc(ob, 'fc', n_fc)`:
 *   m_max = 2, w_ob [N][15][64]) and ConseNet (policies.py:381-390, the agent's own five features only: m_max = 0, w_ob [N][5][64],
 *   nbr ignored); fp / b_fp / relu_bits unused (NULL).
 */
typedef struct nmarl_step_enc {
    const float* ob; int64_t ob_row;
    const float* fp; int64_t fp_sn;
    const float *w_ob, *b_ob, *w_fp, *b_fp;
    int64_t w_ob_sn, b_ob_sn, w_fp_sn, b_fp_sn;
    float* out; int64_t out_sn, out_row;
    int32_t F, A, m_max, pad_;
    int32_t nbr[64];
    /* The CACC env step of THIS lock-step inside the launch too (env != NULL; N = 8, A <= 4): CACCEnv.step (envs/cacc_env.py:191-242)
     * for the actions the launch draws -- one launch per lock-step.  Every drawn action is added into its replica's hand-off word
     * cnt[e] by ONE atomic (2 bits of payload per agent + an arrival count above them); the lane whose add finds N - 1 earlier
     * arrivals holds all N actions in the returned value and steps that replica (nmarl_cacc_step's arithmetic, bit for bit),
     * writing state, reward / done / global reward of lock-step t and the compact observation of lock-step t + 1 -- arguments as
     * for nmarl_cacc_step with p->compact_obs = 1.  Nothing is read back, no wave waits for another: no residency requirement, no
     * failure mode.  cnt: nmarl_lstm_step_env_words(E) uint32 words, zeroed ONCE by the caller (the kernel leaves them zero). */
    const nmarl_cacc_params_t* env;
    float *h, *v, *u; int32_t* t; uint8_t* collided; float* v0_init;
    float* obs_out; float* reward; uint8_t* done; float* global_reward;
    int32_t auto_reset, pad2_; uint64_t seed; int64_t env_id_base; int32_t* episode;
    uint32_t* cnt;
    /* relu_bits (may be NULL): [N][E][4] uint32 with agent stride relu_bits_sn words -- which of the 128 encoder outputs of a row are
     * > 0: bit 4 t + i of word q <=> out[row, 16 t + 4 q + i] > 0.  nmarl_fc_bwd_pair takes the relu derivative from these 16 bytes
     * per row instead of re-reading the 512-byte row of `out`. */
    uint32_t* relu_bits; int64_t relu_bits_sn;
} nmarl_step_enc_t;
int nmarl_lstm_step_env_words(int64_t E);
int nmarl_lstm_step_x_enc(int64_t E, int32_t N, int32_t H, int32_t KX, const float* h_in, int64_t h_sn, const float* img,
                          int64_t img_sn, const float* bias, int64_t bias_sn, const float* c_prev, int64_t c_prev_sn,
                          const float* done, float* gates, int64_t gates_sn, float* c_new, int64_t c_new_sn, float* h_new,
                          int64_t h_new_sn, const nmarl_head_t* head, const nmarl_step_enc_t* enc, void* stream);
int nmarl_lstm_msg_wimage(int32_t N, int32_t K, const float* w_msg, int64_t w_sn, float* img, int64_t img_sn, void* stream);
int nmarl_lstm_step_sync_words(int64_t E, int32_t N);
int nmarl_lstm_step_x_msg(int64_t E, int32_t N, int32_t H, int32_t KX, const float* x, int64_t x_sn, int64_t x_row,
                          const float* h_in, int64_t h_sn, const float* img, int64_t img_sn, const float* bias,
                          int64_t bias_sn, const float* c_prev, int64_t c_prev_sn, const float* done, float* gates,
                          int64_t gates_sn, float* c_new, int64_t c_new_sn, float* h_new, int64_t h_new_sn,
                          const nmarl_head_t* head, const nmarl_msg_t* msg, void* stream);
/*
 * CommNet's lock-step on the synthetic 5x5 grid with the ENV STEP as a role of the same launch (round 6; utils.py:163-197,
 * envs/atsc_env.py:181-207, agents/utils.py:344-417): nmarl_lstm_step_x_msg (head kind 3, message kind 2, msg->ob = the compact
 * observation encoder inside) launched with nmarl_lstm_step_grid_env_blocks(E, N) blocks MORE than the N x ceil(E / 128) LSTM
 * blocks -- the compute units those leave idle (56 of 256 at 25 x 1024).  Every LSTM wave adds its drawn actions into its replicas'
 * hand-off words (words [E][2] u64: 3 bits per agent, an arrival count above bit 59; nmarl_lstm_step_grid_words(E) words zeroed
 * ONCE, left zero by every launch); an env block waits (bounded) for the 25 arrivals of its groups of 16 replicas and then runs
 * nmarl_grid_step's own device code on them (bit-identical; arguments as for nmarl_grid_step with compact_obs = 1, objective =
 * queue), writing state, reward, done and the observation of lock-step t + 1 while the LSTM blocks are in their value re-steps.
 * A time-out raises msg->status word 0 (fail closed, like the hand-off of the new h).  env_blocks = 0: not available (no idle
 * compute unit) -- use nmarl_lstm_step_x_msg + nmarl_grid_step.
 */
typedef struct nmarl_grid_env {
    const nmarl_grid_params_t* params;
    float *q, *transit; uint8_t* prev_action; int32_t* t; float* xi;
    float* obs_out; float* reward; uint8_t* done; float* global_reward;
    int32_t auto_reset, pad_; uint64_t seed; int64_t env_id_base; int32_t* episode;
    uint64_t* words;
} nmarl_grid_env_t;
int nmarl_lstm_step_grid_words(int64_t E);
int nmarl_lstm_step_grid_env_blocks(int64_t E, int32_t N);
int nmarl_lstm_step_x_msg_grid(int64_t E, int32_t N, int32_t H, int32_t KX, const float* x, int64_t x_sn, int64_t x_row,
                               const float* h_in, int64_t h_sn, const float* img, int64_t img_sn, const float* bias,
                               int64_t bias_sn, const float* c_prev, int64_t c_prev_sn, const float* done, float* gates,
                               int64_t gates_sn, float* c_new, int64_t c_new_sn, float* h_new, int64_t h_new_sn,
                               const nmarl_head_t* head, const nmarl_msg_t* msg, const nmarl_grid_env_t* genv, void* stream);
/*
 * NeurComm's WHOLE lock-step in one launch (round 6; agents/utils.py:118-217 `lstm_comm`, utils.py:163-197, envs/cacc_env.py:191-242):
 * nmarl_lstm_step_x_msg with head kind 3 and message kind 1, plus `enc` as in nmarl_lstm_step_x_enc -- the two input encoders
 * [relu(x~ W_ob + b) | relu(p~ W_fp + b)] run in the launch's pre-phase and, with enc->env, the CACC env step behind the action
 * draw.  x (KX = 192: [N][E] rows of pitch x_row, agent stride x_sn) is the S slot of the saved activations: its first 128 columns
 * are WRITTEN here by the encoders (every lane then reads back, as its K-loop operands, exactly the 16-byte pieces it wrote), the
 * last 64 by the message pre-phase (msg->out).  enc->out must be NULL or x.  Replaces nmarl_cacc_step_encode + nmarl_lstm_step_x_msg.
 */
int nmarl_lstm_step_x_msg_enc(int64_t E, int32_t N, int32_t H, int32_t KX, float* x, int64_t x_sn, int64_t x_row,
                              const float* h_in, int64_t h_sn, const float* img, int64_t img_sn, const float* bias,
                              int64_t bias_sn, const float* c_prev, int64_t c_prev_sn, const float* done, float* gates,
                              int64_t gates_sn, float* c_new, int64_t c_new_sn, float* h_new, int64_t h_new_sn,
                              const nmarl_head_t* head, const nmarl_msg_t* msg, const nmarl_step_enc_t* enc, void* stream);
/*
 * One reverse step of the unrolled LSTM training graph (agents/utils.py:102-113, 199-208, 401-408, 585-593), the cell
 * backward and the dgrad product fused on the matrix cores (H = 64):
 *   dz = d cell/d z from gates / c_prev / c_new / done and dL/dh' = dh + dh2, dL/dc' = dc_in   (as nmarl_lstm_cell_bwd)
 *   [dx | dhd] = dz @ [wxm; wh]^T;   dhd *= (1 - done) if apply_keep;   dx = 0 where mask <= 0 (mask optional)
 * KM = 0: only dhd [N,E,64] (uncoupled nets: the recurrent part of dL/dh_{t-1}); KM = 64: also dx [N,E,64], the
 * gradient of the h-dependent part of the LSTM input (lstm_comm's message third hm with mask = hm for its relu;
 * lstm_ic3's / lstm_dial's s with mask = NULL resp. hm).  dz [N,E,4H] and dc_prev [N,E,H] are written as by
 * nmarl_lstm_cell_bwd.  The weights come as the image nmarl_lstm_bptt_wimage builds from wxm [N,KM,4H] (the rows of the
 * x-side weight that meet that input part; NULL for KM = 0) and wh [N,H,4H]: nmarl_lstm_bptt_wimage_floats(KM) =
 * 256*(KM+64) floats per agent; rebuild it whenever the weights change.  mask: row pitch mask_row >= 64.
 */
int nmarl_lstm_bptt_wimage_floats(int32_t KM);
int nmarl_lstm_bptt_wimage(int32_t N, int32_t KM, const float* wxm, int64_t wxm_sn, const float* wh, int64_t wh_sn,
                           float* img, int64_t img_sn, void* stream);
int nmarl_lstm_bptt_step(int64_t E, int32_t N, int32_t H, int32_t KM, const float* gates, int64_t gates_sn,
                         const float* c_prev, int64_t c_prev_sn, const float* c_new, int64_t c_new_sn,
                         const float* done, const float* dh, int64_t dh_sn, const float* dh2, int64_t dh2_sn,
                         const float* dc_in, int64_t dc_sn, const float* img, int64_t img_sn, float* dz,
                         int64_t dz_sn, float* dc_prev, int64_t dc_prev_sn, float* dx, int64_t dx_sn,
                         const float* mask, int64_t mask_sn, int64_t mask_row, float* dhd, int64_t dhd_sn,
                         int32_t apply_keep, void* stream);
/* The same step, also ADDING the column sums of dz (the LSTM bias gradient of this step) to db_part
 * [N][nmarl_lstm_bptt_step_parts(E)][256] (agent stride db_sn; may be NULL): the caller zeroes it before the first reverse
 * step, orders the steps of one recurrence on one stream and sums over the parts at the end -- no pass over dZ. */
int nmarl_lstm_bptt_step_parts(int64_t E);
int nmarl_lstm_bptt_step_db(int64_t E, int32_t N, int32_t H, int32_t KM, const float* gates, int64_t gates_sn,
                            const float* c_prev, int64_t c_prev_sn, const float* c_new, int64_t c_new_sn,
                            const float* done, const float* dh, int64_t dh_sn, const float* dh2, int64_t dh2_sn,
                            const float* dc_in, int64_t dc_sn, const float* img, int64_t img_sn, float* dz,
                            int64_t dz_sn, float* dc_prev, int64_t dc_prev_sn, float* dx, int64_t dx_sn,
                            const float* mask, int64_t mask_sn, int64_t mask_row, float* dhd, int64_t dhd_sn,
                            int32_t apply_keep, float* db_part, int64_t db_sn, void* stream);
/*
 * The whole reverse recurrence of one update in ONE launch, for nets whose recurrence has no cross-agent term (lstm,
 * agents/utils.py:102-113 unrolled by policies.py:99-100): T reverse steps of nmarl_lstm_bptt_step(KM = 0,
 * apply_keep = 1) with dL/dc, the recurrent dL/dh, c_{t-1} and the weight image kept on chip between steps.
 *   gates [N][T][E][4H], dz likewise (agent stride *_sn, step stride *_st, floats);  c_all [N][T+1][E][H] (c_all[t] =
 *   the cell state step t started from);  done [T][E];  dh_ext [N][T][E][H] = dL/dh_t from the heads;
 *   img = nmarl_lstm_bptt_wimage(KM = 0) of the current wh.
 * c_t (for tanh(c_t)) is RECOMPUTED as gf * (c_all[t] * (1 - done_t)) + gi * gu, operation for operation what the forward
 * kernels store (c_all[T] is never read; this and nmarl_lstm_bptt_coupled): gates / c_all must be the trace of a forward pass.
 * Outputs: dz (every step: the weight-gradient GEMMs need it);  db_part [N][nmarl_lstm_bptt_seq_blocks(E)][4H] or
 * NULL: per-block column sums of dz over all T steps and the block's rows (bias gradient = their sum over blocks);
 * dh0 / dc0 [N][E][H] or NULL: dL/d(h, c) of the initial state.  Pointers 16-byte aligned, strides % 4 == 0.
 */
int nmarl_lstm_bptt_seq_blocks(int64_t E);
int nmarl_lstm_bptt_seq(int32_t T, int64_t E, int32_t N, int32_t H, const float* gates, int64_t gates_sn,
                        int64_t gates_st, const float* c_all, int64_t c_sn, int64_t c_st, const float* done,
                        const float* dh_ext, int64_t dh_sn, int64_t dh_st, const float* img, int64_t img_sn,
                        float* dz, int64_t dz_sn, int64_t dz_st, float* db_part, int64_t db_sn, float* dh0,
                        int64_t dh0_sn, float* dc0, int64_t dc0_sn, void* stream);
/* nmarl_lstm_bptt_seq with the heads' dL/dh formed inside the kernel (round 6): dy8 [N][T][E][8] = [d logits | d v | 0] of
 * nmarl_heads_loss (agent stride dy_sn, step stride dy_st), hw [N][64][O] = [pi_w | v_w[:64]] (O = A + 1 <= 8):
 * dL/dh_t(heads) = dy_t hw^T is two more k-steps of the step's transposed product -- 32 bytes per row-step read instead of 256. */
int nmarl_lstm_bptt_seq_dy(int32_t T, int64_t E, int32_t N, int32_t H, const float* gates, int64_t gates_sn,
                           int64_t gates_st, const float* c_all, int64_t c_sn, int64_t c_st, const float* done,
                           const float* dy8, int64_t dy_sn, int64_t dy_st, const float* hw, int64_t hw_sn, int32_t O,
                           const float* img, int64_t img_sn, float* dz, int64_t dz_sn, int64_t dz_st, float* db_part,
                           int64_t db_sn, float* dh0, int64_t dh0_sn, float* dc0, int64_t dc0_sn, void* stream);

/*
 * The reverse recurrence of a COUPLED net's update -- NeurComm (lstm_comm, agents/utils.py:182-208; unrolled training graph
 * of policies.py:330-331) or CommNet (lstm_ic3, agents/utils.py:395-408) -- in ONE launch: per step the work of
 * nmarl_lstm_bptt_seq plus the adjoint of the message term,
 *     [dx | dh] = dz_t @ [wxm; wh]^T;   D1_t = dx (kind 1: * (hm_t > 0));   M_t = D1_t @ w_msg^T   [E, K]
 *     dL/dh_{t-1}[i] = dh[i] (1 - done_t) + sum_{(a, k): i is neighbour k of a} M_t[a][:, 64 k : 64 k + 64]        (kind 1)
 *                    = dh[i] (1 - done_t) + sum_{a: i in nbr(a)} M_t[a] / |nbr(a)|                               (kind 2)
 * M_t crosses agents (blocks) inside the launch through `ring` ([ring_slots][N][E][K] floats, zero it once after
 * allocating; ring_slots >= T for the one-launch form -- every step's messages get their own slot, a consumer never
 * re-reads an address inside a launch because the per-XCD L2s are not coherent -- else >= 2: step-wise launches)
 * with write-through stores and one flag per (agent, 128-row tile, wave) in `ws` (nmarl_lstm_bptt_coupled_ws_words(E, N)
 * 32-bit words; zeroed by the call).  kind 1: K = 64 m_max (<= 128), mask = hm [N][T][E][..] (row pitch mask_row);
 * kind 2: K = 64.  img = nmarl_lstm_bptt_wimage(KM = 64) of [wxm; wh]; img_m = nmarl_lstm_bptt_msg_wimage of w_msg
 * [N][K][64].  rev_agent / rev_col / rev_w [N][r_row]: for every agent the (source agent, first column of its slot in the
 * source's message row, weight) triples of the sum above, padded with (own index, 0, 0.0f) to r_row = 2 (r_max <= 2) or
 * 4 entries.  symmetric: i in nbr(a) <=> a in nbr(i).
 * Outputs: dz [N][T][E][4H], d1 [N][T][E][H], db_part [N][tiles][4H] / dbm_part [N][tiles][H] (column sums of dz / d1 per
 * 128-row tile; bias gradients = their sums over tiles), dhr_io / dc_io [N][E][H] (scratch; on return dL/d(h, c) of the
 * state the sequence started from, without the message part).  ws word [N * tiles * 8] is non-zero afterwards if a wave
 * gave up waiting for a neighbour's block (results invalid); that word is STICKY -- the call zeroes the flag words only --
 * and so is word 0 of `status` (may be NULL; see nmarl_handoff_capacity), which the guarded optimiser step consults.
 * mode 0: one launch if the grid (N * tiles blocks, one per CU) is resident at once, the relation symmetric and
 * ring_slots >= T, else T
 * launches of one step each (same kernel, same results); 1 / 2 force the one-launch / step-wise form (tests).
 */
typedef struct nmarl_bptt_coupled {
    int32_t kind, N, T, H, m_max, r_max, r_row, symmetric, mode, ring_slots;
    int64_t E;
    const float *gates, *c_all, *done, *dh_ext, *img, *img_m, *mask;
    float *dz, *d1, *ring, *db_part, *dbm_part, *dhr_io, *dc_io;
    void* ws;
    int32_t* status;    /* may be NULL: hand-off status words (word 0 <- 1 when a wave gives up) */
    const int32_t *rev_agent, *rev_col;
    const float* rev_w;
    int64_t gates_sn, gates_st, c_sn, c_st, dh_sn, dh_st, img_sn, imgm_sn, mask_sn, mask_st, mask_row, dz_sn, dz_st, d1_sn, d1_st,
        ring_sn, ring_slot, db_sn, dbm_sn, io_sn;
    /* round 6: EITHER dh_ext (the heads' dL/dh as a tensor) OR dy8 [N][T][E][8] = [d logits | d v | 0] of nmarl_heads_loss + the
     * heads' weights hw [N][64][O] (O <= 8): the kernel then forms dL/dh_t(heads) = dy_t hw^T itself (as nmarl_lstm_bptt_seq_dy) */
    const float *dy8, *hw;
    int64_t dy_sn, dy_st, hw_sn;
    int32_t O, pad2_;
} nmarl_bptt_coupled_t;
int nmarl_lstm_bptt_msg_wimage(int32_t N, int32_t K, const float* w_msg, int64_t w_sn, float* img, int64_t img_sn, void* stream);
int nmarl_lstm_bptt_coupled_ws_words(int64_t E, int32_t N);
int nmarl_lstm_bptt_coupled(const nmarl_bptt_coupled_t* p, void* stream);
/*
 * y[n,r,:W] = act(x[n,r,:] + bias[n,:]) for x [N,rows,W] (agent strides in floats, W % 4 == 0);
 * act 0 none / 1 relu / 2 tanh: the bias + activation of `fc` (agents/utils.py:65-73) and of the
 * lstm_comm / lstm_ic3 encoders (agents/utils.py:196-198, 400) after a plain batched GEMM.
 * y may alias x (in place) or be a column block of a wider buffer (row pitch y_row >= W floats):
 * FPPolicy / lstm_comm write their partial encodings side by side so that the LSTM input product is
 * ONE GEMM over the concatenation (tf.concat of policies.py:181 / agents/utils.py:199) without a copy.
 */
int nmarl_bias_act(int64_t rows, int32_t N, int32_t W, const float* x, int64_t x_sn, const float* bias,
                   int64_t bias_sn, int32_t act, float* y, int64_t y_sn, int64_t y_row, void* stream);
/*
 * lstm_dial's message adjoint of ONE reverse step of the update, all agents in one launch -- the backward of
 * agents/utils.py:560-580 (msg_j = relu(h_j w_mfc + b) on the sender, hm_i = relu([msg_j : j in nbr(i)] w_msg + b_msg) on the
 * receiver, s_i = enc_i + hm_i), given ds = dL/ds of this step (nmarl_lstm_bptt_step's dx) and dhd = dz @ wh^T:
 *   d1_i   = ds_i * (hm_i > 0)
 *   dmsg_j = sum over (i, k) with nbr(i, k) == j of d1_i @ w_msg_i[64 k : 64 k + 64, :]^T
 *   d2_j   = dmsg_j * (msg_j > 0)
 *   dh_j   = dhd_j + d2_j @ w_mfc_j^T
 * All of ds, hm, msg, dhd, d1, d2, dh are [N,E,64] panels (agent strides in floats, rows contiguous, 16-byte aligned); the
 * outputs may not overlap the inputs.  img_msg_t: nmarl_lstm_msg_wimage of the [N, 64 m_max, 64] tensor whose row block k is
 * w_msg[:, 64 k : 64 k + 64, :]^T; img_mfc_t: the same image of w_mfc^T.  rev_agent / rev_col / rev_w [N, r_row] (r_row 2 or
 * 4): for every agent its sources (i, 64 k) with weight 1, padded with (own index, 0, weight 0); entries must be valid
 * (agents < N, columns < 64 m_max) -- device memory, not checked.  m_max <= 4.
 * b1_part / b2_part (both or neither): [N][nmarl_dial_msg_adjoint_parts(E)][64] running partial column sums of d1 / d2 --
 * the gradients of b_msg / b_mfc; the caller zeroes them before the first reverse step, every call ADDS this step's sums
 * (calls of one recurrence must be ordered on one stream), and sums over the parts at the end.
 */
int nmarl_dial_msg_adjoint_parts(int64_t E);
int nmarl_dial_msg_adjoint(int64_t E, int32_t N, int32_t m_max, const float* ds, int64_t ds_sn, const float* hm,
                           int64_t hm_sn, const float* msg, int64_t msg_sn, const float* dhd, int64_t dhd_sn,
                           const float* img_msg_t, int64_t img_msg_sn, const float* img_mfc_t, int64_t img_mfc_sn,
                           const int32_t* rev_agent, const int32_t* rev_col, const float* rev_w, int32_t r_row,
                           float* d1, int64_t d1_sn, float* d2, int64_t d2_sn, float* dh, int64_t dh_sn,
                           float* b1_part, int64_t b1_sn, float* b2_part, int64_t b2_sn, void* stream);
/*
 * lstm_dial's own-action term (agents/utils.py:577: one_hot(argmax(p_i), n_h), added to the encoded observation at :579):
 * y[n,r,argmax_a p[n,r,a]] += scale[n] (scale NULL: 1; lstm_dial_hetero, agents/utils.py:676-688, gives agents without
 * neighbours no such term: scale 0).  p [N,rows,A] with contiguous rows (agent stride p_sn), y an [N,rows,W] view
 * (A <= W, row pitch y_row); the first maximum wins, like tf.argmax.
 */
int nmarl_onehot_argmax_add(int64_t rows, int32_t N, int32_t A, int32_t W, const float* p, int64_t p_sn,
                            const float* scale, float* y, int64_t y_sn, int64_t y_row, void* stream);
/*
 * Small-input fully connected encoder layer for all agents and rows in one launch (fc, agents/utils.py:65-73;
 * call sites policies.py:145, 177-180 and agents/utils.py:186-198, 395-400, 566-575):
 *   fwd:  y[n,r,:64] = act(x[n,r,:F] @ w[n] + b[n]),  F <= 64, J = 64 outputs, act as in nmarl_bias_act.
 *   bwd:  g = dy * act'(y);  dw[n] = x[n]^T g  [F,64],  db[n] = sum_r g  [64]   (x is data: no dx).
 * x [N,rows,F] with agent stride x_sn and row pitch x_row (so the env-major observation slab [rows,N,F] is read in
 * place: x_sn = F, x_row = N*F); y / dy [N,rows,64] with row pitch >= 64 (a column block of the concatenated
 * encoding, tf.concat of policies.py:181); w [N,F,64], b [N,64] with agent strides.  bwd is deterministic: blocks
 * write partial sums into `partial` [N, nmarl_fc_bwd_chunks(rows,N), F+1, 64] and a second kernel adds them in order.
 */
int nmarl_fc_fwd(int64_t rows, int32_t N, int32_t F, int32_t J, const float* x, int64_t x_sn, int64_t x_row,
                 const float* w, int64_t w_sn, const float* b, int64_t b_sn, int32_t act, float* y,
                 int64_t y_sn, int64_t y_row, void* stream);
/* Up to NMARL_FC_MAX_PARTS such layers in one launch, layer p writing columns [64p, 64p+64) of y (the whole
 * h-independent encoding of a lock-step: fcs || fcp of policies.py:176-181, w_ob || w_fp of agents/utils.py:186-199).
 * nbr_idx != NULL: the layer's input is gathered through the neighbour table, x~[n,r,k*A+a] = x[nbr_idx[n,k],r,a]
 * with x [N,rows,gather_A] (the previous-step policies), F = m_max*gather_A  (cacc_env.py:244-248 get_fingerprint +
 * models.py:171-179). */
#define NMARL_FC_MAX_PARTS 4
typedef struct nmarl_fc_part {
    const float* x; int64_t x_sn, x_row;
    int32_t F, gather_A, m_max, pad_;
    const int32_t* nbr_idx;
    const float* w; int64_t w_sn;
    const float* b; int64_t b_sn;
} nmarl_fc_part_t;
int nmarl_fc_fwd_multi(int64_t rows, int32_t N, int32_t n_parts, const nmarl_fc_part_t* parts, int32_t act,
                       float* y, int64_t y_sn, int64_t y_row, void* stream);
int nmarl_fc_bwd_chunks(int64_t rows, int32_t N);
/* The backward of BOTH layers of a two-part encoding [act(x_0 w_0 + b_0) | act(x_1 w_1 + b_1)] (fcs || fcp of policies.py:176-181)
 * in one pass over dy [N,rows,128] (16-byte aligned rows): parts[0..1] name the inputs as in nmarl_fc_fwd_multi (F <= 16; w / b
 * unused), y [N,rows,128] the saved encoding.  relu_bits != NULL (act = relu only) replaces y by a bit image its producer wrote,
 * [N,rows,4] uint32 with agent stride bits_sn words: bit 4 t + i of word q of a row <=> y[row, 16 t + 4 q + i] > 0
 * (nmarl_step_enc_t.relu_bits) -- 16 bytes per row instead of 512.  partial [N, nmarl_fc_bwd_chunks(rows,N), 2, 17, 64];
 * dwb [N, 2, 17, 64] receives dw of part p in rows 0..F_p-1 (zeros up to row 15) and db in row 16.  The sums are formed in the
 * order of two nmarl_fc_bwd calls (bit-identical results). */
int nmarl_fc_bwd_pair(int64_t rows, int32_t N, const nmarl_fc_part_t* parts, const float* y, int64_t y_sn, int64_t y_row,
                      const uint32_t* relu_bits, int64_t bits_sn, const float* dy, int64_t dy_sn, int64_t dy_row, int32_t act,
                      float* partial, float* dwb, void* stream);
int nmarl_fc_bwd(int64_t rows, int32_t N, int32_t F, int32_t J, const float* x, int64_t x_sn, int64_t x_row,
                 const float* y, int64_t y_sn, int64_t y_row, const float* dy, int64_t dy_sn, int64_t dy_row,
                 int32_t act, float* partial, float* dw, int64_t dw_sn, float* db, int64_t db_sn, void* stream);
/* nmarl_fc_bwd with the layer's input gathered over a neighbour table inside the kernel, like nmarl_fc_fwd_multi's parts:
 * x [*, rows, gather_A] (agent stride x_sn, row pitch x_row), nbr_idx [N, m_max] (-1 = absent: zeros), F = gather_A * m_max
 * -- the update reads the env's compact observation slab / the fingerprints in place (policies.py:171-174). */
int nmarl_fc_bwd_gather(int64_t rows, int32_t N, int32_t gather_A, int32_t m_max, const int32_t* nbr_idx, int32_t J,
                        const float* x, int64_t x_sn, int64_t x_row, const float* y, int64_t y_sn, int64_t y_row,
                        const float* dy, int64_t dy_sn, int64_t dy_row, int32_t act, float* partial, float* dw, int64_t dw_sn,
                        float* db, int64_t db_sn, void* stream);
/*
 * Backward of the thin actor / critic head layers y = h @ w + b over all rows of the update (policies.py:50-77):
 * h [N,rows,64], w [N,64,O] (O <= 8), dL/dy given as dy [N,rows,O1] (contiguous rows) plus, optionally, dy2 [N,rows]
 * for the last column (O = O1 + 1: the actor's logits and the critic's value arrive as separate gradients)
 *   ->  dh = dy @ w^T [N,rows,64],  dw = h^T dy [N,64,O],  db = sum_r dy [N,O].
 * One streaming pass; deterministic (partial [N, nmarl_fc_bwd_chunks(rows,N), 65, O]).
 */
int nmarl_thin_linear_bwd(int64_t rows, int32_t N, int32_t H, int32_t O, const float* h, int64_t h_sn,
                          const float* dy, int64_t dy_sn, const float* dy2, int64_t dy2_sn, const float* w, int64_t w_sn,
                          float* partial, float* dh, int64_t dh_sn, float* dw, int64_t dw_sn, float* db, int64_t db_sn,
                          void* stream);
/*
 * The update's actor / critic heads, the A2C loss and the heads' backward in ONE streaming pass over h (round 6; policies.py:20-30,
 * 50-77): logits = h w[:, :A] + b[:A], v = h w[:, A] + b[A] + va (w [N,64,A+1] = [pi_w | v_w[:64]], b [N,A+1], va [N,rows] the
 * critic's neighbour-action term of nmarl_nbr_action_value_fwd); the loss terms of nmarl_a2c_loss_fwd -> loss_out [N,3]; d logits,
 * d v of nmarl_a2c_loss_bwd with g_up = 1 -> dy8 [N,rows,8] = [d logits | d v | 0 ..] and dv [N,rows]; dw [N,64,A+1] / db [N,A+1]
 * as nmarl_thin_linear_bwd; dh [N,rows,64] (may be NULL: the one-launch BPTT kernels expand dy8 themselves, nmarl_lstm_bptt_seq_dy).
 * Replaces a skinny GEMM, two loss passes and nmarl_thin_linear_bwd: h is read once.  A + 1 <= 8.  Deterministic (partial
 * [N, nmarl_fc_bwd_chunks(rows,N), 65 (A+1) + 3], fixed-order sums).
 */
int nmarl_heads_loss(int64_t rows, int32_t N, int32_t H, int32_t A, const float* h, int64_t h_sn, const float* w, int64_t w_sn,
                     const float* b, int64_t b_sn, const float* va, const uint8_t* action, const float* adv, const float* R,
                     float v_coef, float e_coef, float* partial, float* loss_out, float* dy8, float* dv, float* dh,
                     int64_t dh_sn, float* dw, int64_t dw_sn, float* db, int64_t db_sn, void* stream);

/*
 * Neighbour-action term of the centralised critic, v += one_hot(neighbours' actions) @ w_a (policies.py:59-77),
 * without the one-hot tensor: action [rows,N] u8, nbr_idx [N,m_max] (-1 padded), w_a [N,m_max*A].
 *   fwd: va[n,r] (+)= sum_k w_a[n, k*A + action[r, nbr_idx[n,k]]]   (accumulate != 0: added to va)
 *   bwd: dw_a[n, k*A + a] = sum of dv[n,r] over the rows with action[r, nbr_idx[n,k]] == a   (m_max*A <= 32;
 *        partial [N, nmarl_fc_bwd_chunks(rows,N), m_max*A], fixed-order sums)
 */
int nmarl_nbr_action_value_fwd(int64_t rows, int32_t N, int32_t A, int32_t m_max, const int32_t* nbr_idx,
                               const uint8_t* action, const float* w, int64_t w_sn, float* va, int32_t accumulate,
                               void* stream);
int nmarl_nbr_action_value_bwd(int64_t rows, int32_t N, int32_t A, int32_t m_max, const int32_t* nbr_idx,
                               const uint8_t* action, const float* dv, float* partial, float* dw, int64_t dw_sn,
                               void* stream);
/*
 * Action draw of Trainer._get_policy (utils.py:135-141) for all (replica, agent):
 * pi [N,E,A] -> action [E,N] u8.
 *   mode 0: np.random.choice == searchsorted(cumsum(pi)/sum, u, 'right') with the
 *           caller's uniforms u [E,N] (legacy global-RNG stream, E = 1);
 *   mode 1: same with u = Philox4x32-10(key=seed, ctr=(env_id_base+e, n>>2, step, 1))
 *           word n&3 (contract: oracle/philox.py);
 *   mode 2: np.argmax (deterministic test policy, utils.py:140).
 * The Philox step is `step` + *step_dev (step_dev: device i64, may be NULL = 0): a captured
 * hipGraph of the rollout bakes the slot offset into `step` and advances the device base once per
 * batch, without re-capturing.
 */
int nmarl_sample_actions(int64_t E, int32_t N, int32_t A, const float* pi, const float* u, int32_t mode,
                         uint64_t seed, int64_t env_id_base, int64_t step, const int64_t* step_dev,
                         uint8_t* action, void* stream);
/*
 * A2C loss of Policy.prepare_loss (policies.py:20-30, 232-255) for all agents over rows = T*E:
 *   pi = softmax(logits); log_pi = log(clip(pi,1e-10,1)); H = -sum pi log_pi
 *   loss_out[n] = { -mean(log_pi[a] ADV),  0.5 v_coef mean((R-v)^2),  -e_coef mean(H) }
 * logits [N,rows,A] (A <= 8; agent stride l_sn, row pitch l_row: a column block of the heads' output), v / adv / R
 * [N,rows], action [rows,N] u8.  bwd: dlogits [N,rows,A], dv [N,rows] = g_up[n] * d(sum of the three terms), the
 * clip passing the gradient where pi >= 1e-10 (tf.clip_by_value).  partial: [N, nmarl_a2c_loss_chunks(rows,N), 3].
 */
int nmarl_a2c_loss_chunks(int64_t rows, int32_t N);
int nmarl_a2c_loss_fwd(int64_t rows, int32_t N, int32_t A, const float* logits, int64_t l_sn, int64_t l_row,
                       const float* v, const uint8_t* action, const float* adv, const float* R, float v_coef,
                       float e_coef, float* partial, float* loss_out, void* stream);
int nmarl_a2c_loss_bwd(int64_t rows, int32_t N, int32_t A, const float* logits, int64_t l_sn, int64_t l_row,
                       const float* v, const uint8_t* action, const float* adv, const float* R, float v_coef,
                       float e_coef, const float* g_up, float* dlogits, float* dv, void* stream);
/*
 * n-step return and advantage -- OnPolicyBuffer._add_R_Adv / _add_s_R_Adv
 * (agents/utils.py:763-775, 800-816) and the MultiAgent variants (837-855,
 * 888-912).  r [T,E] (alpha < 0: global reward) or [T,E,N] (alpha >= 0: spatial
 * discount over dist [N,N] i32), v [T,N,E] rollout values, done_post [T,E] u8
 * (done AFTER each step), R_end [N,E]; outputs R, Adv [N,T,E].  float64 scan.
 */
int nmarl_nstep_return(int64_t E, int32_t N, int32_t T, const float* r, const float* v,
                       const uint8_t* done_post, const float* R_end, double gamma, double alpha,
                       const int32_t* dist, float* R_out, float* adv_out, void* stream);
/*
 * tf.clip_by_global_norm + tf.train.RMSPropOptimizer.apply_gradients
 * (policies.py:32-39, 257-264; TF-1.12 ApplyRMSProp, ms0 = 1, momentum 0) on a
 * flat [G,P] parameter buffer; G = number of independent optimisers (IA2C: one
 * per agent, MA2C: 1).  g is first multiplied by grad_scale (data-parallel
 * mean).  scratch: [G,64] f32.  lr_dev (device scalar) overrides lr if != NULL.
 *   norm_g = ||g_g||;  g *= max_norm * min(1/norm_g, 1/max_norm)   (max_norm > 0)
 *   ms += (g*g - ms)*(1-rho);  w -= lr * g / sqrt(ms + eps)
 */
/*
 * In-launch hand-off (nmarl_lstm_step_x_msg head kind 3, nmarl_lstm_bptt_coupled one-launch form): blocks wait for flags
 * other blocks of the SAME launch publish, so every block must be co-resident.
 *   nmarl_handoff_capacity(which, K): the number of blocks of that kernel the device holds at once =
 *     hipOccupancyMaxActiveBlocksPerMultiprocessor(kernel, 512 threads, its dynamic LDS) x compute units   (which 1: the
 *     lock-step kernel with a message image of K floats x 64; 2: the coupled BPTT kernel, message rows of K floats);
 *     the launchers refuse / fall back to launch-per-step forms above it.  NMARL_TEST_FAKE_CUS=<n> in the environment
 *     replaces the device's compute-unit count (tests: a "smaller GPU").  < 0: error.
 *   status words (int32 x 4, caller-owned, zeroed once): [0] set to 1 -- and left set -- by any hand-off kernel whose wave
 *     gave up waiting (bounded spins: a neighbour block was not running), [1] number of optimiser steps
 *     nmarl_rmsprop_tf_clip_guarded refused since.  While [0] != 0 the guarded step changes neither w nor ms: a batch
 *     computed from a broken hand-off can not reach the weights.  The host clears [0] after switching to the
 *     launch-per-step forms and re-running the batch (deeprl_network_amd/utils.py BatchedTrainer.run_batch).
 *   nmarl_test_handoff_fault(nth): test hook -- the nth hand-off launch from now (1 = the next; 0 disarms) runs with block 0
 *     never publishing its flags and 4096 spins, i.e. its neighbours time out as if block 0 were not resident.
 */
int nmarl_handoff_capacity(int32_t which, int32_t K);
int nmarl_test_handoff_fault(int32_t nth);
int nmarl_rmsprop_tf_clip_guarded(int32_t G, int64_t P, float* w, const float* g, float* ms, float* scratch,
                                  const float* lr_dev, float lr, float rho, float eps, float max_norm, float grad_scale,
                                  float* grad_norm_out, int32_t* status, void* stream);
int nmarl_rmsprop_tf_clip(int32_t G, int64_t P, float* w, const float* g, float* ms, float* scratch,
                          const float* lr_dev, float lr, float rho, float eps, float max_norm,
                          float grad_scale, float* grad_norm_out, void* stream);

/*
 * Between two n_step batches of the batched loop (two launches instead of ~45 elementwise ones):
 *  - episode statistics as Trainer.run keeps them (utils.py:228-229: mean / std of an episode's global rewards): g [T][E]
 *    the batch's global rewards, done [E] u8 the flags of its last lock-step; per replica running ep_sum / ep_sq / ep_len
 *    [E] f64, closed into fin [4] f64 += (episodes, sum of means, sum of stds, episodes shorter than T_env = collisions,
 *    cacc_env.py:231-233) and zeroed where done;
 *  - the hand-over to the next batch: for finished replicas `model.reset()` / a fresh episode's fingerprint
 *    (policies.py:151-154, cacc_env.py:184): h_fw, c_fw [N][E][H] zeroed, fp_0 <- fp_uniform [N][A]; for all:
 *    h_bw, c_bw <- h_fw, c_fw (states_bw <- states_fw, policies.py:115), fp_0 [N][E][A] <- fp_T, x_0 [E][N*F] <- x_T
 *    (slot T of the rollout buffers becomes slot 0), done_pre [E] f32 <- done.
 * Everything the call writes is written by it alone: guarded by `skip_if` it is the commit point of a batch.
 */
typedef struct nmarl_batch_epilogue {
    int64_t E;
    int32_t N, H, A, F, T, T_env;
    const float* g;
    const uint8_t* done;
    double *ep_sum, *ep_sq, *ep_len, *fin;
    float *h_fw, *c_fw, *h_bw, *c_bw;
    const float *fp_T, *fp_uniform, *x_T;
    float *fp_0, *x_0, *done_pre;
    double* scratch;          /* [NMARL_EPILOGUE_SCRATCH] f64: per-block partial statistics */
    const int32_t* skip_if;   /* NULL, or a device word: while it is != 0 the call changes NOTHING (the hand-off status word of
                               * a model on in-launch hand-off kernels: a batch whose hand-off timed out is neither counted nor
                               * handed over -- the state the batch started from stays in place for the host's re-run) */
} nmarl_batch_epilogue_t;
#define NMARL_EPILOGUE_SCRATCH 4096
int nmarl_batch_epilogue(const nmarl_batch_epilogue_t* p, void* stream);

/*
 * n (<= NMARL_COPY_MAX) device-to-device copies dst[k] <- src[k] of bytes[k] bytes in ONE kernel launch; dst / src / bytes
 * are HOST arrays (read at call time), the ranges of one pair must not overlap.  Replaces what the reference does with NumPy
 * assignments between its Python-side buffers (agents/utils.py:732-761 `OnPolicyBuffer`, utils.py:163-197) where this
 * library's callers keep device-resident state: inside a captured hipGraph the copy is a KERNEL node like every other launch
 * of this library (a hipMemcpyAsync would be a memcpy node).  skip_if: NULL, or a device word -- while it is != 0 nothing is
 * copied (the trainer's start-of-batch snapshot is not overwritten once the hand-off status word is raised).
 */
#define NMARL_COPY_MAX 16
int nmarl_copy_multi(int32_t n, void* const* dst, const void* const* src, const int64_t* bytes, const int32_t* skip_if,
                     void* stream);

/*
 * Measurement (kernel durations against the roofline; the reference has no counterpart): nmarl_timestamp stores the
 * device's constant-rate wall clock into *out (device pointer) from a one-thread kernel on `stream`; nmarl_timestamp_rate_khz
 * returns the clock's rate in kHz (hipDeviceAttributeWallClockRate of the current device; < 0: error).  Two stamps around a
 * launch inside a captured hipGraph = that launch's duration where it runs (bench.py `roofline_bptt`).
 */
int nmarl_timestamp(uint64_t* out, void* stream);
int nmarl_timestamp_rate_khz(void);

#ifdef __cplusplus
}
#endif
#endif /* NMARL_H */
