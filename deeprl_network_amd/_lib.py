"""ctypes binding of libnmarl_hip.so (include/nmarl.h).

The product path has NO CPU fallback: if the HIP library is missing this module
raises at import, and every op raises if handed a non-HIP tensor.
torch is imported first so that the HIP runtime already mapped by torch
(libamdhip64.so.7) is the one the library binds to -- one runtime per process.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL, see above)

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libnmarl_hip.so')

if not os.path.exists(LIB_PATH):
    raise ImportError(
        'deeprl_network_amd: %s not found. Build it with `python -m deeprl_network_amd.build` '
        '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)

from . import build as _build  # noqa: E402

if _build.built_hash() != _build.source_hash():
    raise ImportError(
        'deeprl_network_amd: %s is stale (built from sources %s, current sources %s). Rebuild it with '
        '`python -m deeprl_network_amd.build`.' % (LIB_PATH, _build.built_hash(), _build.source_hash()))

# Same-box A/B of two builds of the kernels (tools/ab_build.sh): NMARL_LIB_AB=<path of another libnmarl_hip.so with the same C-ABI>
# is loaded INSTEAD (it is by definition not built from the current sources; every symbol and the ABI version are still checked below)
LIB_PATH = os.environ.get('NMARL_LIB_AB') or LIB_PATH
lib = C.CDLL(LIB_PATH)

ABI_VERSION = 1


class CaccParams(C.Structure):
    """nmarl_cacc_params_t (include/nmarl.h)."""
    _fields_ = [(n, C.c_float) for n in
                ('dt', 'h_min', 'h_star', 'h_s', 'h_g', 'v_max', 'v_star', 'u_min', 'u_max',
                 'reward_a', 'reward_b', 'G')] + \
               [(n, C.c_int32) for n in ('T', 'batch_size', 'scenario', 'train_mode', 'per_agent_reward', 'compact_obs')]


class Head(C.Structure):
    """nmarl_head_t (include/nmarl.h): actor / critic head of the fused step's epilogue."""
    _fields_ = [('kind', C.c_int32), ('A', C.c_int32), ('mode', C.c_int32), ('m_max', C.c_int32),
                ('w', C.c_void_p), ('w_sn', C.c_int64), ('b', C.c_void_p), ('b_sn', C.c_int64),
                ('pi_out', C.c_void_p), ('pi_sn', C.c_int64), ('act_out', C.c_void_p), ('u', C.c_void_p),
                ('seed', C.c_uint64), ('env_id_base', C.c_int64), ('step', C.c_int64), ('step_dev', C.c_void_p),
                ('act_in', C.c_void_p), ('nbr_idx', C.c_void_p), ('v_out', C.c_void_p), ('v_sn', C.c_int64),
                ('w2', C.c_void_p), ('w2_sn', C.c_int64), ('b2', C.c_void_p), ('b2_sn', C.c_int64)]


class Msg(C.Structure):
    """nmarl_msg_t (include/nmarl.h): the in-kernel message term of a coupled net's policy / value step."""
    _fields_ = [('kind', C.c_int32), ('m_max', C.c_int32), ('K', C.c_int32), ('pad_', C.c_int32), ('nbr_idx', C.c_void_p),
                ('img', C.c_void_p), ('img_sn', C.c_int64), ('b', C.c_void_p), ('b_sn', C.c_int64),
                ('enc', C.c_void_p), ('enc_sn', C.c_int64), ('enc_row', C.c_int64),
                ('out', C.c_void_p), ('out_sn', C.c_int64), ('out_row', C.c_int64), ('sync', C.c_void_p),
                ('ob', C.c_void_p), ('ob_row', C.c_int64), ('ob_F', C.c_int32), ('ob_segs', C.c_int32), ('ob_nbr', C.c_void_p),
                ('ob_img', C.c_void_p), ('ob_img_sn', C.c_int64), ('ob_b', C.c_void_p), ('ob_b_sn', C.c_int64),
                ('status', C.c_void_p), ('src', C.c_void_p), ('src_sn', C.c_int64),
                ('out2', C.c_void_p), ('out2_sn', C.c_int64), ('out2_row', C.c_int64),
                ('next_img', C.c_void_p), ('next_img_sn', C.c_int64), ('next_b', C.c_void_p), ('next_b_sn', C.c_int64),
                ('next_out', C.c_void_p), ('next_out_sn', C.c_int64),
                ('mean_out', C.c_void_p), ('mean_out_sn', C.c_int64), ('mean_out_row', C.c_int64),
                ('carry_in', C.c_void_p), ('carry_in_sn', C.c_int64), ('carry_out', C.c_void_p), ('carry_out_sn', C.c_int64),
                ('mean_next', C.c_void_p), ('mean_next_sn', C.c_int64), ('mean_next_row', C.c_int64)]


class StepEnc(C.Structure):
    """nmarl_step_enc_t (include/nmarl.h): the input encoders of a lock-step inside the policy + value launch."""
    _fields_ = [('ob', C.c_void_p), ('ob_row', C.c_int64), ('fp', C.c_void_p), ('fp_sn', C.c_int64),
                ('w_ob', C.c_void_p), ('b_ob', C.c_void_p), ('w_fp', C.c_void_p), ('b_fp', C.c_void_p),
                ('w_ob_sn', C.c_int64), ('b_ob_sn', C.c_int64), ('w_fp_sn', C.c_int64), ('b_fp_sn', C.c_int64),
                ('out', C.c_void_p), ('out_sn', C.c_int64), ('out_row', C.c_int64),
                ('F', C.c_int32), ('A', C.c_int32), ('m_max', C.c_int32), ('pad_', C.c_int32), ('nbr', C.c_int32 * 64),
                ('env', C.POINTER(CaccParams)), ('h', C.c_void_p), ('v', C.c_void_p), ('u', C.c_void_p), ('t', C.c_void_p),
                ('collided', C.c_void_p), ('v0_init', C.c_void_p), ('obs_out', C.c_void_p), ('reward', C.c_void_p),
                ('done', C.c_void_p), ('global_reward', C.c_void_p), ('auto_reset', C.c_int32), ('pad2_', C.c_int32),
                ('seed', C.c_uint64), ('env_id_base', C.c_int64), ('episode', C.c_void_p), ('cnt', C.c_void_p),
                ('relu_bits', C.c_void_p), ('relu_bits_sn', C.c_int64)]


class NetParams(C.Structure):
    """nmarl_net_params_t (include/nmarl.h)."""
    _fields_ = [('norm_wave', C.c_float), ('clip_wave', C.c_float), ('flow_rate', C.c_float), ('T', C.c_int32),
                ('per_agent_reward', C.c_int32)]


class NetTopo(C.Structure):
    """nmarl_net_topo_t (include/nmarl.h): sizes, links per node and the packed table image (device pointers)."""
    _fields_ = [('N', C.c_int32), ('L', C.c_int32), ('A', C.c_int32), ('m_max', C.c_int32), ('n_s', C.c_void_p),
                ('image', C.c_void_p)]


# layout of the packed network image (include/nmarl.h NMARL_NET_OFF_*)
NET_OFF = dict(green=0, src=6144, group=7680, share=8448, fan=11520, dnptr=11648, dnpair=11728, nbr=13264)
NET_IMAGE_BYTES = 13568


class FcPart(C.Structure):
    """nmarl_fc_part_t (include/nmarl.h): one layer of nmarl_fc_fwd_multi."""
    _fields_ = [('x', C.c_void_p), ('x_sn', C.c_int64), ('x_row', C.c_int64), ('F', C.c_int32), ('gather_A', C.c_int32),
                ('m_max', C.c_int32), ('pad_', C.c_int32), ('nbr_idx', C.c_void_p), ('w', C.c_void_p), ('w_sn', C.c_int64),
                ('b', C.c_void_p), ('b_sn', C.c_int64)]


FC_MAX_PARTS = 4


class CaccEncode(C.Structure):
    """nmarl_cacc_encode_t (include/nmarl.h): the encoders fused behind the CACC step."""
    _fields_ = ([(k, C.c_void_p) for k in ('w_ob', 'b_ob', 'w_fp', 'b_fp', 'fp', 'nbr_idx', 'out')] +
                [(k, C.c_int64) for k in ('w_ob_sn', 'b_ob_sn', 'w_fp_sn', 'b_fp_sn', 'fp_sn', 'out_sn', 'out_row')] +
                [('act', C.c_int32), ('n_parts', C.c_int32)])


class BatchEpilogue(C.Structure):
    """nmarl_batch_epilogue_t (include/nmarl.h)."""
    _fields_ = ([('E', C.c_int64)] + [(k, C.c_int32) for k in ('N', 'H', 'A', 'F', 'T', 'T_env')] +
                [(k, C.c_void_p) for k in ('g', 'done', 'ep_sum', 'ep_sq', 'ep_len', 'fin', 'h_fw', 'c_fw', 'h_bw', 'c_bw', 'fp_T',
                                           'fp_uniform', 'x_T', 'fp_0', 'x_0', 'done_pre', 'scratch', 'skip_if')])


class BpttCoupled(C.Structure):
    """nmarl_bptt_coupled_t (include/nmarl.h): arguments of nmarl_lstm_bptt_coupled."""
    _fields_ = ([(k, C.c_int32) for k in ('kind', 'N', 'T', 'H', 'm_max', 'r_max', 'r_row', 'symmetric', 'mode', 'ring_slots')] +
                [('E', C.c_int64)] +
                [(k, C.c_void_p) for k in ('gates', 'c_all', 'done', 'dh_ext', 'img', 'img_m', 'mask', 'dz', 'd1', 'ring', 'db_part',
                                           'dbm_part', 'dhr_io', 'dc_io', 'ws', 'status', 'rev_agent', 'rev_col', 'rev_w')] +
                [(k, C.c_int64) for k in ('gates_sn', 'gates_st', 'c_sn', 'c_st', 'dh_sn', 'dh_st', 'img_sn', 'imgm_sn', 'mask_sn',
                                          'mask_st', 'mask_row', 'dz_sn', 'dz_st', 'd1_sn', 'd1_st', 'ring_sn', 'ring_slot', 'db_sn',
                                          'dbm_sn', 'io_sn')] +
                [('dy8', C.c_void_p), ('hw', C.c_void_p), ('dy_sn', C.c_int64), ('dy_st', C.c_int64), ('hw_sn', C.c_int64),
                 ('O', C.c_int32), ('pad2_', C.c_int32)])


class GridParams(C.Structure):
    """nmarl_grid_params_t (include/nmarl.h)."""
    _fields_ = [('norm_wave', C.c_float), ('clip_wave', C.c_float), ('peak1', C.c_float), ('peak2', C.c_float),
                ('T', C.c_int32), ('per_agent_reward', C.c_int32), ('compact_obs', C.c_int32),
                ('objective', C.c_int32), ('coef_wait', C.c_float), ('head_wait', C.c_void_p)]


class GridEnv(C.Structure):
    """nmarl_grid_env_t (include/nmarl.h): the grid env step as a role of CommNet's lock-step launch."""
    _fields_ = ([('params', C.POINTER(GridParams))] +
                [(k, C.c_void_p) for k in ('q', 'transit', 'prev_action', 't', 'xi', 'obs_out', 'reward', 'done', 'global_reward')] +
                [('auto_reset', C.c_int32), ('pad_', C.c_int32), ('seed', C.c_uint64), ('env_id_base', C.c_int64),
                 ('episode', C.c_void_p), ('words', C.c_void_p)])


_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_u64 = C.c_uint64
_f32 = C.c_float

# name -> argtypes; every symbol include/nmarl.h declares (tests/test_abi.py checks both ways)
SIGNATURES = {
    'nmarl_abi_version': [],
    'nmarl_cacc_reset': [C.POINTER(CaccParams), _i64, _p, _p, _u64, _i64, _p,
                         _p, _p, _p, _p, _p, _p, _p, _p, _i32, _p],
    'nmarl_cacc_step': [C.POINTER(CaccParams), _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                        _i32, _u64, _i64, _p, _p],
    'nmarl_cacc_step_encode': [C.POINTER(CaccParams), _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                               _i32, _u64, _i64, _p, C.POINTER(CaccEncode), _p],
    'nmarl_grid_reset': [C.POINTER(GridParams), _i64, _p, _p, _u64, _i64, _p, _p, _p, _p, _p, _p, _p, _p],
    'nmarl_grid_step': [C.POINTER(GridParams), _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _u64, _i64, _p, _p],
    'nmarl_net_reset': [C.POINTER(NetTopo), _i64, _p, _p, _u64, _i64, _p, _p, _p, _p, _p, _p, _p, _p],
    'nmarl_net_step': [C.POINTER(NetParams), C.POINTER(NetTopo), _i64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i32, _u64, _i64,
                       _p, _p],
    'nmarl_nbr_gather_fwd': [_i64, _i32, _i32, _i32, _p, _p, _p, _p],
    'nmarl_nbr_gather_bwd': [_i64, _i32, _i32, _i32, _p, _p, _p, _p],
    'nmarl_nbr_mean_fwd': [_i64, _i32, _i32, _i32, _p, _p, _p, _p],
    'nmarl_nbr_mean_bwd': [_i64, _i32, _i32, _i32, _p, _p, _p, _p],
    'nmarl_nbr_gather_bwd_add': [_i64, _i32, _i32, _i32, _p, _p, _p, _p, _p],
    'nmarl_nbr_mean_bwd_add': [_i64, _i32, _i32, _i32, _p, _p, _p, _p, _p],
    'nmarl_nbr_onehot': [_i64, _i32, _i32, _i32, _p, _p, _p, _i64, _p],
    'nmarl_lstm_cell_fwd': [_i64, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _p],
    'nmarl_lstm_step_fused': [_i64, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p,
                              _i64, _p, _i64, _p],
    'nmarl_lstm_step_fused_head': [_i64, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64,
                                   _p, _i64, _p, _i64, C.POINTER(Head), _p],
    'nmarl_lstm_wimage_floats': [_i32],
    'nmarl_lstm_wimage': [_i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p],
    'nmarl_lstm_step_x': [_i64, _i32, _i32, _i32, _p, _i64, _i64, _i32, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p,
                          _i64, _p, _i64, _p, _i64, C.POINTER(Head), _p],
    'nmarl_lstm_msg_wimage': [_i32, _i32, _p, _i64, _p, _i64, _p],
    'nmarl_lstm_step_sync_words': [_i64, _i32],
    'nmarl_lstm_step_x_msg': [_i64, _i32, _i32, _i32, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64,
                              C.POINTER(Head), C.POINTER(Msg), _p],
    'nmarl_lstm_step_x_enc': [_i64, _i32, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64,
                              C.POINTER(Head), C.POINTER(StepEnc), _p],
    'nmarl_lstm_step_x_msg_enc': [_i64, _i32, _i32, _i32, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64,
                                  C.POINTER(Head), C.POINTER(Msg), C.POINTER(StepEnc), _p],
    'nmarl_lstm_step_env_words': [_i64],
    'nmarl_lstm_step_grid_words': [_i64],
    'nmarl_lstm_step_grid_env_blocks': [_i64, _i32],
    'nmarl_lstm_step_x_msg_grid': [_i64, _i32, _i32, _i32, _p, _i64, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64,
                                   C.POINTER(Head), C.POINTER(Msg), C.POINTER(GridEnv), _p],
    'nmarl_lstm_bptt_wimage_floats': [_i32],
    'nmarl_lstm_bptt_wimage': [_i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p],
    'nmarl_lstm_bptt_step': [_i64, _i32, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p,
                             _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _i32, _p],
    'nmarl_lstm_bptt_step_db': [_i64, _i32, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p,
                                _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _p, _i64, _i32, _p, _i64, _p],
    'nmarl_lstm_bptt_step_parts': [_i64],
    'nmarl_lstm_bptt_seq_blocks': [_i64],
    'nmarl_lstm_bptt_seq': [_i32, _i64, _i32, _i32, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i64, _p, _i64, _p, _i64, _i64,
                            _p, _i64, _p, _i64, _p, _i64, _p],
    'nmarl_lstm_bptt_seq_dy': [_i32, _i64, _i32, _i32, _p, _i64, _i64, _p, _i64, _i64, _p, _p, _i64, _i64, _p, _i64, _i32, _p, _i64, _p, _i64, _i64,
                               _p, _i64, _p, _i64, _p, _i64, _p],
    'nmarl_lstm_bptt_msg_wimage': [_i32, _i32, _p, _i64, _p, _i64, _p],
    'nmarl_lstm_bptt_coupled_ws_words': [_i64, _i32],
    'nmarl_lstm_bptt_coupled': [C.POINTER(BpttCoupled), _p],
    'nmarl_fc_fwd': [_i64, _i32, _i32, _i32, _p, _i64, _i64, _p, _i64, _p, _i64, _i32, _p, _i64, _i64, _p],
    'nmarl_fc_fwd_multi': [_i64, _i32, _i32, C.POINTER(FcPart), _i32, _p, _i64, _i64, _p],
    'nmarl_dial_msg_adjoint': [_i64, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _i32,
                               _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p],
    'nmarl_dial_msg_adjoint_parts': [_i64],
    'nmarl_onehot_argmax_add': [_i64, _i32, _i32, _i32, _p, _i64, _p, _p, _i64, _i64, _p],
    'nmarl_fc_bwd_chunks': [_i64, _i32],
    'nmarl_fc_bwd': [_i64, _i32, _i32, _i32, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i32, _p, _p, _i64, _p, _i64, _p],
    'nmarl_fc_bwd_pair': [_i64, _i32, _p, _p, _i64, _i64, _p, _i64, _p, _i64, _i64, _i32, _p, _p, _p],
    'nmarl_fc_bwd_gather': [_i64, _i32, _i32, _i32, _p, _i32, _p, _i64, _i64, _p, _i64, _i64, _p, _i64, _i64, _i32, _p, _p, _i64, _p, _i64, _p],
    'nmarl_heads_loss': [_i64, _i32, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _i64, _p, _i64, _p,
                         _i64, _p],
    'nmarl_thin_linear_bwd': [_i64, _i32, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _p],
    'nmarl_nbr_action_value_fwd': [_i64, _i32, _i32, _i32, _p, _p, _p, _i64, _p, _i32, _p],
    'nmarl_nbr_action_value_bwd': [_i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i64, _p],
    'nmarl_bias_act': [_i64, _i32, _i32, _p, _i64, _p, _i64, _i32, _p, _i64, _i64, _p],
    'nmarl_lstm_cell_bwd': [_i64, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p],
    'nmarl_a2c_loss_chunks': [_i64, _i32],
    'nmarl_a2c_loss_fwd': [_i64, _i32, _i32, _p, _i64, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p],
    'nmarl_a2c_loss_bwd': [_i64, _i32, _i32, _p, _i64, _i64, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p],
    'nmarl_sample_actions': [_i64, _i32, _i32, _p, _p, _i32, _u64, _i64, _i64, _p, _p, _p],
    'nmarl_nstep_return': [_i64, _i32, _i32, _p, _p, _p, _p, C.c_double, C.c_double, _p, _p, _p, _p],
    'nmarl_rmsprop_tf_clip': [_i32, _i64, _p, _p, _p, _p, _p, _f32, _f32, _f32, _f32, _f32, _p, _p],
    'nmarl_rmsprop_tf_clip_guarded': [_i32, _i64, _p, _p, _p, _p, _p, _f32, _f32, _f32, _f32, _f32, _p, _p, _p],
    'nmarl_handoff_capacity': [_i32, _i32],
    'nmarl_test_handoff_fault': [_i32],
    'nmarl_batch_epilogue': [C.POINTER(BatchEpilogue), _p],
    'nmarl_timestamp': [_p, _p],
    'nmarl_timestamp_rate_khz': [],
    'nmarl_copy_multi': [_i32, C.POINTER(_p), C.POINTER(_p), C.POINTER(_i64), _p, _p],
}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)   # AttributeError here == missing export: fail loudly
    _fn.argtypes = _args
    _fn.restype = C.c_int

if lib.nmarl_abi_version() != ABI_VERSION:
    raise ImportError('libnmarl_hip.so ABI %d != binding ABI %d: rebuild' %
                      (lib.nmarl_abi_version(), ABI_VERSION))


class NmarlError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise NmarlError('%s failed: %s' % (what, {-1: 'NMARL_EINVAL', -2: 'NMARL_EHIP'}.get(rc, rc)))


def ptr(t, dtype=None, strided=False):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise NmarlError('nmarl ops need HIP device tensors (got %s); there is no CPU path' % t.device)
    if t.device.index != torch.cuda.current_device():
        raise NmarlError('tensor on %s but the current device is cuda:%d: call torch.cuda.set_device first '
                         '(kernels launch on the current device\'s stream)' % (t.device, torch.cuda.current_device()))
    if not strided and not t.is_contiguous():
        raise NmarlError('nmarl ops need contiguous tensors')
    if dtype is not None and t.dtype != dtype:
        raise NmarlError('expected dtype %s, got %s' % (dtype, t.dtype))
    return t.data_ptr()


def stream():
    """torch's current stream of the CURRENT device: launches are made there, so every tensor handed to an op must
    live on that device (`ptr` checks it -- a model built on cuda:1 without torch.cuda.set_device(1) fails loudly
    instead of launching on the wrong GPU)."""
    return torch.cuda.current_stream().cuda_stream
