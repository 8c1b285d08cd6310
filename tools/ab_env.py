"""A/B harness for the CACC step kernel: builds variants of csrc/cacc.hip with -D knobs into build_ab/,
loads them side by side (ctypes) and times them INTERLEAVED in one process on the same device state
(different gpurun boxes / DVFS states are not comparable).  Usage: python tools/ab_env.py [E]"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
CSRC = os.path.join(ROOT, 'deeprl_network_amd', 'csrc')
OUT = os.path.join(ROOT, 'build_ab')

VARIANTS = {
    'base': [],
    'ntS1': ['-DNMARL_CACC_NT_SMALL=1'],
    'ntS2': ['-DNMARL_CACC_NT_SMALL=2'],
    'blkS128': ['-DNMARL_CACC_BLOCK_SMALL=128'],
    'blkS256': ['-DNMARL_CACC_BLOCK_SMALL=256'],
    'ntL1': ['-DNMARL_CACC_NT_LARGE=1'],
}
USE_GRAPH = True


def build():
    os.makedirs(OUT, exist_ok=True)
    for name, flags in VARIANTS.items():
        lib = os.path.join(OUT, 'libcacc_%s.so' % name)
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared',
               os.path.join(CSRC, 'cacc.hip'), '-o', lib] + flags
        subprocess.check_call(cmd)
    print('built', list(VARIANTS))


def run(E):
    import torch
    from helpers import cacc_config
    from deeprl_network_amd import _lib
    from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
    env = CACCBatchEnv(cacc_config()['ENV_CONFIG'], num_envs=E)
    env.reset()
    e = torch.arange(E, device='cuda')[:, None]
    a = torch.arange(8, device='cuda')[None, :]
    acts = [((e + 3 * a + s) % 4).to(torch.uint8).contiguous() for s in range(4)]
    libs = {}
    for name in VARIANTS:
        L = ctypes.CDLL(os.path.join(OUT, 'libcacc_%s.so' % name))
        L.nmarl_cacc_step.argtypes = _lib.SIGNATURES['nmarl_cacc_step']
        L.nmarl_cacc_step.restype = ctypes.c_int
        libs[name] = L
    P = _lib.ptr

    def step(L, k):
        rc = L.nmarl_cacc_step(ctypes.byref(env.params), E, P(acts[k % 4]), P(env.h), P(env.v), P(env.u), P(env.t),
                               P(env.collided), P(env.v0_init), P(env.obs), P(env.reward), P(env.done),
                               P(env.global_reward), 1, env.seed, 0, P(env.episode), _lib.stream())
        assert rc == 0
    res = {n: [] for n in VARIANTS}
    graphs = {}
    if USE_GRAPH:      # 60 back-to-back launches per replay: removes the host launch cost (small E)
        side = torch.cuda.Stream()
        for name, L in libs.items():
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step(L, 0)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for k in range(60):
                    step(L, k)
            graphs[name] = g
    for rnd in range(5):
        for name, L in libs.items():
            reps = 20 if E <= (1 << 16) else 2
            run_once = (lambda: graphs[name].replay()) if USE_GRAPH else (lambda: [step(L, k) for k in range(60)])
            run_once()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(reps):
                run_once()
            t1.record()
            torch.cuda.synchronize()
            res[name].append(t0.elapsed_time(t1) * 1e3 / (60 * reps))
    for name, v in res.items():
        v = sorted(v)
        print('%-10s median %.1f us  (min %.1f max %.1f)  %.2f TB/s algorithmic(631 B)' %
              (name, v[len(v) // 2], v[0], v[-1], 631 * E / v[len(v) // 2] / 1e6))


if __name__ == '__main__':
    if sys.argv[1:2] == ['build']:
        build()
    else:
        run(int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 21)
