"""PMC target for rocprofv3 --pmc passes: a calibration pair (known-size copies: 16 B/lane vectorised and
4 B/lane scalar) followed by the CACC / grid step kernels at a size far beyond the 256 MB Infinity Cache.
Usage (one counter group per run, as MI355X_MICROARCH.md prescribes):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o pmc_fetch --output-format csv -- python tools/pmc_env.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from helpers import cacc_config, grid_config
from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
from deeprl_network_amd.envs.large_grid_env import LargeGridBatchEnv

n = 1 << 28                                    # 1 GiB of fp32
x = torch.ones(n + 4, device='cuda')
y = torch.empty(n + 4, device='cuda')
for _ in range(3):
    y[:n].copy_(x[:n])                         # vectorised: 16 B / lane
for _ in range(3):
    y[:n].copy_(x[1:n + 1])                    # misaligned source: scalar 4 B / lane loads
torch.cuda.synchronize()
E = int(os.environ.get('PMC_E', 1 << 21))
COMPACT = os.environ.get('PMC_COMPACT', '1') == '1'          # the batched engine's variant: obs [E,8,5]
env = CACCBatchEnv(cacc_config()['ENV_CONFIG'], num_envs=E)
if COMPACT:
    env.set_compact_obs(True)
env.reset()
e = torch.arange(E, device='cuda')[:, None]
a = torch.arange(8, device='cuda')[None, :]
acts = [((e + 3 * a + s) % 4).to(torch.uint8).contiguous() for s in range(4)]
for s in range(8):
    env.step(acts[s % 4], auto_reset=True)
torch.cuda.synchronize()
Es = 4096                                      # the bench workload: cacc_step_kernel<64,0>
senv = CACCBatchEnv(cacc_config()['ENV_CONFIG'], num_envs=Es)
if COMPACT:
    senv.set_compact_obs(True)
senv.reset()
for s in range(16):
    senv.step(acts[s % 4][:Es].contiguous(), auto_reset=True)
torch.cuda.synchronize()
Eg = int(os.environ.get('PMC_EG', 1 << 17))
ga = [torch.randint(0, 5, (Eg, 25), dtype=torch.uint8, device='cuda') for _ in range(2)]
for compact in (False, True):                  # grid_step_kernel<1,false> (gathered slab), then <1,true> (compact: the batched engine's)
    genv = LargeGridBatchEnv(grid_config()['ENV_CONFIG'], num_envs=Eg)
    if compact:
        genv.set_compact_obs(True)
    genv.reset()
    for s in range(6):
        genv.step(ga[s % 2], auto_reset=True)
    torch.cuda.synchronize()
    del genv
# the CACC step with the next lock-step's encoders behind it (cacc_step_encode_kernel) at the bench shape
from deeprl_network_amd import ops
import numpy as np
from deeprl_network_amd.agents import models
cpf = cacc_config(agent='ia2c_fp', scenario='catchup', n_step=60, reward_norm=800.0)
fenv = CACCBatchEnv(cpf['ENV_CONFIG'], num_envs=Es)
fenv.set_compact_obs(True)
fenv.reset()
np.random.seed(12)
fm = models.IA2C_FP(fenv.n_s_ls, fenv.n_a_ls, fenv.neighbor_mask, fenv.distance_mask, fenv.coop_gamma, 10 ** 9, cpf['MODEL_CONFIG'],
                    seed=12, num_envs=Es)
fm.enable_saved_activations(); fm.enable_compact_obs()
spec = fm.policy.fused_env_encode(fm.buf_fp[1], fm.encode_target(1))
for s in range(16):
    fenv.step(acts[s % 4][:Es].contiguous(), auto_reset=True, encode=spec)
torch.cuda.synchronize()
# the fused MFMA LSTM lock-step at the bench shape (x-side policy + value kernel, as the rollout launches it)
N, El, H, A, KX = 8, 4096, 64, 4, 128
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()                                   # noqa: E731
h, c, x = r(N, El, H) * 0.3, r(N, El, H) * 0.3, torch.relu(r(N, El, KX))
wx, wh, b = r(N, KX, 4 * H) * 0.1, r(N, H, 4 * H) * 0.1, r(N, 4 * H) * 0.1
pi_w, pi_b, v_w, v_b = r(N, H, A), r(N, A), r(N, H + 2 * A, 1), r(N, 1)
nbr = torch.tensor([[max(i - 1, 0), min(i + 1, N - 1)] for i in range(N)], dtype=torch.int32).cuda()
done = torch.zeros(El, device='cuda')
img = ops.lstm_wimage(wx, wh)
ho, co, gates = torch.empty_like(h), torch.empty_like(c), torch.empty(N, El, 4 * H, device='cuda')
pi, act, v = torch.empty(N, El, A, device='cuda'), torch.zeros(El, N, dtype=torch.uint8, device='cuda'), torch.empty(N, El, device='cuda')
for s in range(12):
    ops.lstm_step_policy_value(h, None, b, None, None, c, done, pi_w, pi_b, pi, act, v_w, v_b, nbr, A, v, mode=2, xs=(x, None, img),
                               h_out=ho, c_out=co, gates=gates, defer_action_term=True)
torch.cuda.synchronize()
# round 5: the WHOLE lock-step of IA2C-FP in one launch (lstm_step_x_kernel<3, 0, 1>: input encoders in the pre-phase, env step behind
# the draw) as the rollout launches it -- two eager rollouts of the bench workload's trainer (2 x 61 launches)
try:
    from deeprl_network_amd.utils import BatchedTrainer, Counter
    tenv = CACCBatchEnv(cpf['ENV_CONFIG'], num_envs=Es)
    np.random.seed(12)
    tm = models.IA2C_FP(tenv.n_s_ls, tenv.n_a_ls, tenv.neighbor_mask, tenv.distance_mask, tenv.coop_gamma, 10 ** 9, cpf['MODEL_CONFIG'],
                        seed=12, num_envs=Es)
    ttr = BatchedTrainer(tenv, tm, Counter(10 ** 12, 10 ** 12, 10 ** 12), use_graph=False)
    assert ttr.enc_in_kernel and ttr.env_in_kernel
    for s in range(2):
        ttr._rollout()
    torch.cuda.synchronize()
    del ttr, tm, tenv
except Exception as ex:                        # a side measurement: never fail the pass
    print('one-launch lock-step skipped:', ex)
# the update's recurrence in one launch (nmarl_lstm_bptt_seq) at the bench shape: T = 60 reverse steps
T = 60
rd = lambda *s: torch.randn(*s, device='cuda')        # on the device: big host copies would show up as copyBuffer launches  # noqa: E731
Gs = torch.cat([torch.sigmoid(rd(N, T, El, 3 * H)), torch.tanh(rd(N, T, El, H))], dim=-1)
Cs, Ds, dZs = rd(N, T + 1, El, H), rd(N, T, El, H), torch.empty(N, T, El, 4 * H, device='cuda')
dones = torch.zeros(T, El, device='cuda')
bimg = ops.lstm_bptt_wimage(None, wh)
# (round 6: the product form takes the heads' dL/dh as dy8 and expands it inside -- lstm_bptt_seq_kernel<true>)
hd = (rd(N, T * El, 8) * 1e-3, rd(N, H, 5) * 0.1)
for s in range(5):
    ops.bptt_seq(Gs, Cs, dones, None, bimg, dZs, head_dy=hd)
torch.cuda.synchronize()
# the update's heads + loss + heads' backward in one pass over h (nmarl_heads_loss) at the bench shape
try:
    A_ = 4
    nbr2 = torch.tensor([[max(i - 1, 0), min(i + 1, N - 1)] for i in range(N)], dtype=torch.int32, device='cuda')
    hh = torch.tanh(rd(N, T * El, H))
    acts = torch.randint(0, A_, (T * El, N), device='cuda').to(torch.uint8)
    for s in range(5):
        ops.heads_loss(hh, rd(N, H, A_) * .1, rd(N, A_) * .1, rd(N, H + 2 * A_, 1) * .1, rd(N, 1) * .1, acts, nbr2, A_, rd(N, T * El), rd(N, T * El),
                       0.5, 0.01, want_dh=False)
    torch.cuda.synchronize()
    del hh, acts
except Exception as ex:
    print('heads_loss skipped:', ex)
# the coupled nets' reverse recurrence in one launch (nmarl_lstm_bptt_coupled, NeurComm on the line graph) at the bench shape
import sys as _sys
_sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_gpu_ops import _topology
nbr_idx, _ = ops.neighbor_table(_topology(N, 'line'), 'cuda')
wxm, wmsg = rd(N, H, 4 * H) * 0.1, rd(N, 2 * H, H) * 0.15
Ss = torch.relu(rd(N, T, El, 3 * H))
D1s = torch.empty(N, T, El, H, device='cuda')
ws_, wm_ = (wxm, wh, ops.lstm_bptt_wimage(wxm, wh)), (wmsg, ops.lstm_bptt_msg_wimage(wmsg))
rev = ops.reverse_neighbor_table(nbr_idx, ops.COUPLED_NC)
for s in range(4):      # (the product form: the heads' dL/dh as dy8, expanded inside -- kernel<8,2,true,true>)
    ops.bptt_coupled(ops.COUPLED_NC, rev, 2, Gs, Cs, dones, None, ws_, wm_, Ss[..., 2 * H:], dZs, D1s, head_dy=hd)
torch.cuda.synchronize()
ops.check_coupled_status()
# the coupled nets' lock-step in ONE launch (lstm_step_x_kernel<4,1>: NeurComm on the line graph) at the bench shape
try:
    if ops.step_handoff_supported(N, El, 'cuda'):
        wx4, w_msg, b_msg = r(N, 3 * H, 4 * H) * 0.15, r(N, 2 * H, H) * 0.15, r(N, H) * 0.1
        img4, mimg = ops.lstm_wimage(wx4, wh), ops.lstm_msg_wimage(w_msg)
        slot = torch.relu(r(N, El, 3 * H))
        sync = ops.step_sync_words(N, El, 'cuda')
        msg = dict(kind=ops.MSG_GATHER_RELU, nbr_idx=nbr_idx, w_msg=w_msg, b_msg=b_msg, img=mimg, out=slot[:, :, 2 * H:], sync=sync)
        v_w4 = r(N, H + nbr_idx.shape[1] * A, 1)
        for s in range(12):
            ops.lstm_step_policy_value(h, None, b, None, None, c, done, pi_w, pi_b, pi, act, v_w4, v_b, nbr_idx, A, v, mode=2,
                                       xs=(slot[:, :, :2 * H], None, img4, None, msg), h_out=ho, c_out=co, gates=gates,
                                       defer_action_term=True)
        torch.cuda.synchronize()
        ops.check_coupled_status()
except Exception as ex:                        # a side measurement: never fail the pass
    print('one-launch coupled step skipped:', ex)
# CommNet on the 5 x 5 grid (BASELINE configs[3]: 25 agents x 1024 replicas): the one-launch lock-step with the observation encoder
# inside (lstm_step_x_kernel<4,2>) and the coupled BPTT of its update (lstm_bptt_coupled_kernel<4,4,false>, T = 120)
try:
    Ng, Egd, Ag, Tg, Fo = 25, 1024, 5, 120, 12
    nbr_g, _ = ops.neighbor_table(_topology(Ng, 'grid'), 'cuda')
    hg, cg = r(Ng, Egd, H) * 0.3, r(Ng, Egd, H) * 0.3
    whg, bg = r(Ng, H, 4 * H) * 0.1, r(Ng, 4 * H) * 0.1
    if ops.step_handoff_supported(Ng, Egd, 'cuda', K=H):
        wxg, wmg, bmg = r(Ng, H, 4 * H) * 0.15, r(Ng, H, H) * 0.15, r(Ng, H) * 0.1
        imgg, mimgg = ops.lstm_wimage(wxg, whg), ops.lstm_msg_wimage(wmg)
        nbr_self = torch.cat([torch.arange(Ng, dtype=torch.int32, device='cuda').view(-1, 1), nbr_g], dim=1)
        w_ob, b_ob = r(Ng, Fo * nbr_self.shape[1], H) * 0.3, r(Ng, H) * 0.1
        oimg = ops.lstm_ob_wimage(w_ob, torch.zeros(Ng, 64, H, device='cuda'))
        xo, enc_slot, s_slot = r(Egd, Ng, Fo), torch.zeros(Ng, Egd, H, device='cuda'), torch.zeros(Ng, Egd, H, device='cuda')
        syncg = ops.step_sync_words(Ng, Egd, 'cuda')
        msgg = dict(kind=ops.MSG_MEAN_ADD, nbr_idx=nbr_g, w_msg=wmg, b_msg=bmg, img=mimgg, enc=enc_slot, out=s_slot, sync=syncg,
                    ob=dict(x=xo, nbr=nbr_self, img=oimg, b=b_ob))
        pig, actg, vg = torch.empty(Ng, Egd, Ag, device='cuda'), torch.zeros(Egd, Ng, dtype=torch.uint8, device='cuda'), torch.empty(Ng, Egd, device='cuda')
        hog, cog, gg = torch.empty_like(hg), torch.empty_like(cg), torch.empty(Ng, Egd, 4 * H, device='cuda')
        pwg, pbg, vwg, vbg = r(Ng, H, Ag), r(Ng, Ag), r(Ng, H + nbr_g.shape[1] * Ag, 1), r(Ng, 1)
        doneg = torch.zeros(Egd, device='cuda')
        for s in range(12):
            ops.lstm_step_policy_value(hg, None, bg, None, None, cg, doneg, pwg, pbg, pig, actg, vwg, vbg, nbr_g, Ag, vg, mode=2,
                                       xs=(None, None, imgg, None, msgg), h_out=hog, c_out=cog, gates=gg, defer_action_term=True)
        torch.cuda.synchronize()
        ops.check_coupled_status()
    Gg = torch.cat([torch.sigmoid(rd(Ng, Tg, Egd, 3 * H)), torch.tanh(rd(Ng, Tg, Egd, H))], dim=-1)
    Cg, Dg = rd(Ng, Tg + 1, Egd, H), rd(Ng, Tg, Egd, H)
    dZg, D1g = torch.empty(Ng, Tg, Egd, 4 * H, device='cuda'), torch.empty(Ng, Tg, Egd, H, device='cuda')
    donesg = torch.zeros(Tg, Egd, device='cuda')
    wxmg, wmsgg = rd(Ng, H, 4 * H) * 0.1, rd(Ng, H, H) * 0.15
    wsg, wmgg = (wxmg, whg, ops.lstm_bptt_wimage(wxmg, whg)), (wmsgg, ops.lstm_bptt_msg_wimage(wmsgg))
    revg = ops.reverse_neighbor_table(nbr_g, ops.COUPLED_IC3)
    hdg = (rd(Ng, Tg * Egd, 8) * 1e-3, rd(Ng, H, 6) * 0.1)
    for s in range(4):
        ops.bptt_coupled(ops.COUPLED_IC3, revg, nbr_g.shape[1], Gg, Cg, donesg, None, wsg, wmgg, None, dZg, D1g, head_dy=hdg)
    torch.cuda.synchronize()
    ops.check_coupled_status()
except Exception as ex:                        # a side measurement: never fail the pass
    print('grid CommNet kernels skipped:', ex)
# lstm_dial on the line graph at the bench shape (8 x 4096): the policy step with the receiver layer in its pre-phase and the sender
# layer of the new h in its epilogue (lstm_step_x_kernel<1,3>), and the message adjoint of one reverse step (dial_msg_adjoint_kernel<2>)
try:
    wxd, wmd, bmd = r(N, H, 4 * H) * 0.15, r(N, 2 * H, H) * 0.15, r(N, H) * 0.1
    mfw, mfb = r(N, H, H) * 0.2, r(N, H) * 0.1
    imgd, mimgd, fimg = ops.lstm_wimage(wxd, wh), ops.lstm_msg_wimage(wmd), ops.lstm_msg_wimage(mfw)
    srcd, encd = torch.relu(r(N, El, H)), torch.relu(r(N, El, H))
    sv = torch.zeros(N, 4, El, H, device='cuda')
    msgd = dict(kind=ops.MSG_DIAL, nbr_idx=nbr_idx, w_msg=wmd, b_msg=bmd, img=mimgd, enc=encd, src=srcd, out=sv[:, 0], out2=sv[:, 1],
                next=dict(img=fimg, b=mfb, out=sv[:, 2]))
    for s in range(12):
        ops.lstm_step_policy(h, None, b, None, None, c, done, co, ho, pi_w, pi_b, pi, act, mode=2, xs=(None, None, imgd, None, msgd), gates=gates)
    torch.cuda.synchronize()
    revd = ops.reverse_neighbor_table(nbr_idx, ops.COUPLED_NC)
    imgs_d = ops.dial_adjoint_images(wmd, mfw)
    dsd, dhdd = rd(N, El, H), rd(N, El, H)
    d1d, d2d, dhd_o = (torch.empty(N, El, H, device='cuda') for _ in range(3))
    parts_d = ops.dial_adjoint_bias_parts(N, El, 'cuda')
    for s in range(12):
        ops.dial_msg_adjoint(dsd, sv[:, 1], sv[:, 2], dhdd, wmd, mfw, nbr_idx, imgs_d, revd, d1d, d2d, dhd_o, bias_parts=parts_d)
    torch.cuda.synchronize()
except Exception as ex:                        # a side measurement: never fail the pass
    print('lstm_dial kernels skipped:', ex)
print('done', E, Eg)
