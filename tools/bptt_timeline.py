"""Shader-clock timeline of one mid-sequence step of the coupled BPTT kernel (instrumentation build of csrc/lstm_bptt.hip +
lstm_mfma.hip with -DNMARL_STEP_TIMELINE into tools/dbg/libstep_tl.so; see tools/step_timeline.py): block 0's eight waves,
cycles relative to the wave's own step start, plus the whole kernel's cycle count (-> the shader clock it ran at).
    python tools/bptt_timeline.py [nc|grid|ic3]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
OUT = os.path.join(ROOT, 'tools', 'dbg')
SO = os.path.join(OUT, 'libstep_tl.so')
CS = os.path.join(ROOT, 'deeprl_network_amd', 'csrc')
if '--build' in sys.argv or not os.path.exists(SO):
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
                           '-DNMARL_STEP_TIMELINE', os.path.join(CS, 'lstm_mfma.hip'), os.path.join(CS, 'lstm_bptt.hip'), '-o', SO])
    if '--build' in sys.argv:
        sys.exit(0)
import torch  # noqa: E402
from deeprl_network_amd import _lib, ops  # noqa: E402
from test_gpu_ops import _topology  # noqa: E402

dbg = C.CDLL(SO)
for name in ('nmarl_lstm_bptt_coupled', 'nmarl_lstm_bptt_wimage', 'nmarl_lstm_bptt_msg_wimage', 'nmarl_lstm_bptt_seq'):
    getattr(dbg, name).argtypes = _lib.SIGNATURES[name]
    getattr(dbg, name).restype = C.c_int
    setattr(_lib.lib, name, getattr(dbg, name))
shape = ([a for a in sys.argv[1:] if not a.startswith('-')] or ['nc'])[0]
if shape == 'seq':
    # the uncoupled one-launch BPTT at the bench shape: whole-kernel cycle count -> the shader clock this box sustains for it
    N, E, T, H = 8, 4096, 60, 64
    G = torch.cat([torch.sigmoid(torch.randn(N, T, E, 3 * H, device='cuda')), torch.tanh(torch.randn(N, T, E, H, device='cuda'))], dim=-1)
    Cc, D = torch.randn(N, T + 1, E, H, device='cuda') * 0.5, torch.randn(N, T, E, H, device='cuda')
    dZ, done = torch.empty_like(G), torch.zeros(T, E, device='cuda')
    img = ops.lstm_bptt_wimage(None, torch.randn(N, H, 4 * H, device='cuda') * 0.1)
    tl = torch.zeros(8 * 32, dtype=torch.int64, device='cuda')
    dbg.nmarl_timeline_set_bptt.argtypes = [C.c_void_p, C.c_void_p]
    dbg.nmarl_timeline_set_bptt(tl.data_ptr(), torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        e0.record()
        ops.bptt_seq(G, Cc, done, D, img, dZ, want_db=False)
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    t = tl.cpu().view(8, 32)
    print('seq: N %d E %d T %d, %.1f us per call (instrumented build), kernel %d cycles (block 0, wave 0) -> %.2f GHz; %d cycles per step'
          % (N, E, T, us, int(t[0, 17] - t[0, 0]), float(t[0, 17] - t[0, 0]) / us / 1e3, int(t[0, 17] - t[0, 1]) // T))
    sys.exit(0)
kind, topo, N, E, T = {'nc': (ops.COUPLED_NC, 'line', 8, 4096, 60), 'grid': (ops.COUPLED_IC3, 'grid', 25, 1024, 120),
                       'ic3': (ops.COUPLED_IC3, 'line', 8, 4096, 60)}[shape]
H = 64
rd = lambda *s: torch.randn(*s, device='cuda')                # noqa: E731
nbr_idx, _ = ops.neighbor_table(_topology(N, topo), 'cuda')
m_max = nbr_idx.shape[1]
K = H * m_max if kind == ops.COUPLED_NC else H
G = torch.cat([torch.sigmoid(rd(N, T, E, 3 * H)), torch.tanh(rd(N, T, E, H))], dim=-1)
Cc, D = rd(N, T + 1, E, H) * 0.5, rd(N, T, E, H)
dZ, D1 = torch.empty(N, T, E, 4 * H, device='cuda'), torch.empty(N, T, E, H, device='cuda')
done = torch.zeros(T, E, device='cuda')
wxm, wh, wmsg = rd(N, H, 4 * H) * 0.1, rd(N, H, 4 * H) * 0.1, rd(N, K, H) * 0.15
S = torch.relu(rd(N, T, E, 3 * H))
ws, wm = (wxm, wh, ops.lstm_bptt_wimage(wxm, wh)), (wmsg, ops.lstm_bptt_msg_wimage(wmsg))
rev = ops.reverse_neighbor_table(nbr_idx, kind)
mask = S[..., 2 * H:] if kind == ops.COUPLED_NC else None
tl = torch.zeros(8 * 32, dtype=torch.int64, device='cuda')
dbg.nmarl_timeline_set_bptt.argtypes = [C.c_void_p, C.c_void_p]
dbg.nmarl_timeline_set_bptt(tl.data_ptr(), torch.cuda.current_stream().cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    e0.record()
    ops.bptt_coupled(kind, rev, m_max, G, Cc, done, D, ws, wm, mask, dZ, D1)
    e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3
t = tl.cpu().view(8, 32)
names = {2: 'step start', 3: 'flags seen', 4: 'messages added', 5: 'cell 0 (+loads)', 6: 'product 0', 7: 'cell 1 (+loads)', 8: 'product 1',
         9: 'cell 2 (+loads)', 10: 'product 2', 11: 'cell 3 (+loads)', 12: 'product 3', 13: 'D1 / bias sums', 14: 'message product',
         15: 'stores drained', 16: 'step end'}
print('%s: N %d E %d T %d, %.1f us per call (instrumented), kernel %d cycles (wave 0) -> %.2f GHz; prologue %d cycles'
      % (shape, N, E, T, us, int(t[0, 17] - t[0, 0]), float(t[0, 17] - t[0, 0]) / us / 1e3, int(t[0, 1] - t[0, 0])))
print('cycles since the wave\'s own step start (step t = T/2); mean step = %.0f cycles' % (float(t[0, 17] - t[0, 1]) / T))
print('stamp'.ljust(18) + ''.join(('wave %d' % w).rjust(9) for w in range(8)))
for i in range(2, 17):
    print(names[i].ljust(18) + ''.join(('%d' % int(t[w, i] - t[w, 2])).rjust(9) for w in range(8)))
print('step start vs wave 0'.ljust(20) + ''.join(('%d' % int(t[w, 2] - t[0, 2])).rjust(9) for w in range(8)))
