"""Microbenchmark behind DESIGN.md's statement on the lock-step kernel's bound: builds mfma_valu_overlap.hip and times
matrix-only, vector-only and both wave groups (one wave of each kind per SIMD).  python tools/microbench/mfma_valu_overlap.py"""
import ctypes as C
import os
import subprocess
import torch

here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, '..', 'dbg', 'liboverlap.so')
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(here, 'mfma_valu_overlap.hip')):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-shared', os.path.join(here, 'mfma_valu_overlap.hip'), '-o', so])
lib = C.CDLL(so)
lib.overlap_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(256 * 512, device='cuda')
iters = 2000                  # per matrix wave: 32 000 MFMAs x 32 cycles of the pipe; per vector wave: 256 000 FMAs x 4 issue cycles
res = {}
for mode, name in ((1, 'matrix waves only'), (2, 'vector waves only'), (3, 'both')):
    for _ in range(2):
        lib.overlap_launch(out.data_ptr(), 256, iters, mode, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        lib.overlap_launch(out.data_ptr(), 256, iters, mode, torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    res[mode] = e0.elapsed_time(e1) / 5
    print('%-18s %.3f ms' % (name, res[mode]))
print('both / max(single) = %.2f,  both / sum(single) = %.2f' % (res[3] / max(res[1], res[2]), res[3] / (res[1] + res[2])))
fl = 256 * 4 * iters * 16 * 2 * 16 * 16 * 4
print('matrix waves alone: %.1f TFLOP/s (1 of 2 wave slots per SIMD)' % (fl / res[1] / 1e9))
