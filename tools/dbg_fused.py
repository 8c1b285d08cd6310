import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deeprl_network_amd import _lib
L = ctypes.CDLL(os.path.join(ROOT, 'build_ab', 'liblstm_dbg.so'))
L.nmarl_lstm_step_fused.argtypes = _lib.SIGNATURES['nmarl_lstm_step_fused']; L.nmarl_lstm_step_fused.restype = ctypes.c_int
N, E, H = 8, 4096, 64
h = torch.randn(N, E, H, device='cuda'); c = torch.randn(N, E, H, device='cuda'); z1 = torch.randn(N, E, 4*H, device='cuda')
wh = torch.randn(N, H, 4*H, device='cuda') * 0.2; b = torch.zeros(N, 4*H, device='cuda'); done = torch.zeros(E, device='cuda')
co, ho = torch.empty_like(c), torch.empty_like(h)
P = lambda t: t.data_ptr()
def run():
    rc = L.nmarl_lstm_step_fused(E, N, H, P(h), E*H, P(wh), H*4*H, P(b), 4*H, P(z1), E*4*H, None, 0, P(c), E*H, P(done), None, 0, P(co), E*H, P(ho), E*H, None)
    assert rc == 0, rc
for _ in range(5): run()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(50): run()
t1.record(); torch.cuda.synchronize()
print('us per call', t0.elapsed_time(t1) * 1e3 / 50)
out = (ctypes.c_ulonglong * 32)()
L.nmarl_debug_read(out)
v = list(out)
names = ['issue loads / stage W', 'barrier wait', 'B issues loads', 'h tile + wait data', 'bias add', 'MFMA loop', 'epilogue']
base = v[0]
for grp, off in (('A (wave 0)', 0), ('B (wave 4)', 8)):
    print(grp, 'start at +%d' % (v[off] - base))
    for i, n in enumerate(names[:6]):
        print('   %-24s %8d cycles (ends at +%d)' % (names[i] if i != 2 else names[2], v[off + i + 1] - v[off + i], v[off + i + 1] - base))
