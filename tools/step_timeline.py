"""Shader-clock timeline of the x-side LSTM lock-step kernel: builds csrc/lstm_mfma.hip with -DNMARL_STEP_TIMELINE into
tools/dbg/libstep_tl.so (instrumentation build, not the product), runs the policy + value step at the bench shape and
prints, for block 0, the stamps of every wave relative to the block's first stamp (cycles):
  0 entry | 1 prologue done | 2+2t chunk-tick t computed | 3+2t its barrier passed | 20 K loop done | 21 cell epilogue done
  22 head done | 23 re-step MFMAs done | 24 re-step cell done | 25 end.      python tools/step_timeline.py [head]
head 4: the coupled nets' one-launch policy + value step (NeurComm shape, line graph): 26 published | 23 h part of the re-step
done | 27 neighbours' flags seen | 28 message term | 29 message W chunks staged | 30 message chunks done | 33 cell maths |
24 tile written | 34 / 35 critic dot products / shuffles | 25 end (before the block's generation update)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, 'tools', 'dbg')
SO = os.path.join(OUT, 'libstep_tl.so')
SRC = os.path.join(ROOT, 'deeprl_network_amd', 'csrc', 'lstm_mfma.hip')

if '--build' in sys.argv or not os.path.exists(SO):
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                           '-DNMARL_STEP_TIMELINE', SRC, os.path.join(os.path.dirname(SRC), 'lstm_bptt.hip'), '-o', SO])
    if '--build' in sys.argv:
        sys.exit(0)

import torch  # noqa: E402
from deeprl_network_amd import _lib, ops  # noqa: E402

dbg = C.CDLL(SO)
for name, args in _lib.SIGNATURES.items():
    if hasattr(dbg, name):
        getattr(dbg, name).argtypes = args
        getattr(dbg, name).restype = C.c_int
_lib.lib.nmarl_lstm_step_x = dbg.nmarl_lstm_step_x            # route the product wrappers through the instrumented build
_lib.lib.nmarl_lstm_wimage = dbg.nmarl_lstm_wimage
_lib.lib.nmarl_lstm_step_x_msg = dbg.nmarl_lstm_step_x_msg
_lib.lib.nmarl_lstm_step_x_enc = dbg.nmarl_lstm_step_x_enc
N, E, H, A, KX = 8, 4096, 64, 4, 128
head = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3
ENC = 'enc' in sys.argv                       # head 3 with the input encoders inside the launch (round 5)
GRID = 'grid' in sys.argv                     # head 4 on CommNet's grid shape: 25 x 1024 rows, KX = 64, 4 neighbours, encoder inside
if GRID:
    N, E, A = 25, 1024, 5
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()            # noqa: E731
h, c, x = r(N, E, H), r(N, E, H), torch.relu(r(N, E, KX))
wx, wh, b = r(N, KX, 4 * H) * 0.15, r(N, H, 4 * H) * 0.2, torch.zeros(N, 4 * H).cuda()
pi_w, pi_b, v_w, v_b = r(N, H, A), r(N, A), r(N, H + 2 * A, 1), r(N, 1)
nbr = torch.tensor([[max(i - 1, 0), min(i + 1, N - 1)] for i in range(N)], dtype=torch.int32).cuda()
done = torch.zeros(E).cuda()
img = ops.lstm_wimage(wx, wh)
co, ho, gates = torch.empty_like(c), torch.empty_like(h), torch.empty(N, E, 4 * H, device='cuda')
pi, act, v = torch.empty(N, E, A, device='cuda'), torch.zeros(E, N, dtype=torch.uint8, device='cuda'), torch.empty(N, E, device='cuda')
tl = torch.zeros(8 * 64, dtype=torch.int64, device='cuda')
dbg.nmarl_timeline_set.argtypes = [C.c_void_p, C.c_void_p]
dbg.nmarl_timeline_set(tl.data_ptr(), torch.cuda.current_stream().cuda_stream)


if head == 4 and GRID:
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_gpu_ops import _topology
    nbr4, _ = ops.neighbor_table(_topology(N, 'grid'), 'cuda')
    nbr = nbr4
    v_w = r(N, H + nbr4.shape[1] * A, 1)
    wx4, w_msg, b_msg = r(N, H, 4 * H) * 0.15, r(N, H, H) * 0.15, r(N, H) * 0.1
    img4, mimg = ops.lstm_wimage(wx4, wh), ops.lstm_msg_wimage(w_msg)
    Fo = 12
    nbr_self = torch.cat([torch.arange(N, dtype=torch.int32, device='cuda').view(-1, 1), nbr4], dim=1)
    w_ob, b_ob = r(N, Fo * nbr_self.shape[1], H) * 0.3, r(N, H) * 0.1
    oimg = ops.lstm_ob_wimage(w_ob, torch.zeros(N, 64, H, device='cuda'))
    xo, enc_slot, s_slot = r(E, N, Fo), torch.zeros(N, E, H, device='cuda'), torch.zeros(N, E, H, device='cuda')
    sync = ops.step_sync_words(N, E, 'cuda')
    msg = dict(kind=2, nbr_idx=nbr4, w_msg=w_msg, b_msg=b_msg, img=mimg, enc=enc_slot, out=s_slot, sync=sync,
               ob=dict(x=xo, nbr=nbr_self, img=oimg, b=b_ob))
    slot = None
elif head == 4:
    wx4, w_msg, b_msg = r(N, 3 * H, 4 * H) * 0.15, r(N, 2 * H, H) * 0.15, r(N, H) * 0.1
    img4, mimg = ops.lstm_wimage(wx4, wh), ops.lstm_msg_wimage(w_msg)
    slot = torch.relu(r(N, E, 3 * H))
    sync = ops.step_sync_words(N, E, 'cuda')
    nbr4 = torch.tensor([[i - 1 if i > 0 else i + 1, i + 1 if 0 < i < N - 1 else -1] for i in range(N)], dtype=torch.int32).cuda()
    msg = dict(kind=1, nbr_idx=nbr4, w_msg=w_msg, b_msg=b_msg, img=mimg, out=slot[:, :, 2 * H:], sync=sync)


if head == 4 and 'carry' in sys.argv:         # the message term handed over from the previous launch's re-step (kernel<4,.,.,2>)
    carry_buf = torch.zeros(N, E, H, device='cuda')
    msg.update(carry_in=carry_buf, carry_out=carry_buf)
    if GRID:
        msg['mean_next'] = torch.zeros(N, E, H, device='cuda')

if ENC:
    nbrs = [[j for j in (i - 1, i + 1) if 0 <= j < N] for i in range(N)]
    enc_spec = ops.step_enc_spec(r(E, N, 5), torch.softmax(r(N, E, A), -1), r(N, 15, H) * 0.3, r(N, H) * 0.1, r(N, 8, H) * 0.3, r(N, H) * 0.1,
                                 nbrs, out=None if 'noout' in sys.argv else x)


def run():
    if head == 4:
        ops.lstm_step_policy_value(h, None, b, None, None, c, done, pi_w, pi_b, pi, act, v_w, v_b, nbr, A, v, mode=2,
                                   xs=(None if GRID else slot[:, :, :2 * H], None, img4, None, msg), h_out=ho, c_out=co, gates=gates,
                                   defer_action_term=True)
    elif head == 3:
        ops.lstm_step_policy_value(h, None, b, None, None, c, done, pi_w, pi_b, pi, act, v_w, v_b, nbr, A, v, mode=2,
                                   xs=(enc_spec if ENC else x, None, img), h_out=ho, c_out=co, gates=gates, defer_action_term=True)
    else:
        ops.lstm_step_fused(h, None, b, None, None, c, done, gates, co, ho, xs=(x, None, img))


for _ in range(5):
    run()
torch.cuda.synchronize()
t = tl.cpu().view(8, 64)
t0 = int(t[:, 0].min())
names = {0: 'entry', 40: 'epoch read', 41: 'A/W loads issued', 42: 'chunk 0 in LDS', 43: 'chunk 1 in LDS', 44: 'head w in LDS',
         36: 'enc loads issued', 37: 'enc operands in', 50: 'enc MFMAs done', 51: 'enc parked', 38: 'enc done', 39: 'env step done', 45: 'msg loads issued', 46: 'msg img in LDS', 47: 'ob img in LDS', 1: 'prologue', 48: 'encoder done', 49: 'msg term done', 55: 'chunks 0/1 staged', 20: 'K loop done', 21: 'cell epilogue', 22: 'head', 23: 're-step MFMA', 24: 're-step cell', 25: 'end',
         26: 'published', 31: 'nbr rows asked', 27: 'flags seen', 52: 'nbr rows summed', 53: 'msg MFMAs done', 28: 'message term', 29: 'msg W staged', 30: 'msg chunks',
         33: 'cell math', 34: 'critic dots', 35: 'critic shfl'}
for i in range(2, 20, 2):
    names[i], names[i + 1] = 'tick %d computed' % ((i - 2) // 2), 'tick %d barrier' % ((i - 2) // 2)
print('stamp'.ljust(18) + ''.join(('wave %d' % w).rjust(9) for w in range(8)))
ORDER = [0, 40, 36, 41, 42, 43, 44, 45, 46, 37, 47, 1, 50, 51, 38, 48, 49, 55] + list(range(2, 21)) + [26, 21, 22, 31, 23, 27, 52, 53, 28, 29, 30, 33, 24, 34, 35, 25, 39]
print('flags up at the look (1 yes, 2 no)'.ljust(18) + ''.join(('%d' % int(t[w, 54])).rjust(9) for w in range(8)))
for i in ORDER:
    if i not in names:
        continue
    if int(t[:, i].max()) == 0:
        continue
    print(names[i].ljust(18) + ''.join(('%d' % (int(t[w, i]) - t0) if int(t[w, i]) else '-').rjust(9) for w in range(8)))
