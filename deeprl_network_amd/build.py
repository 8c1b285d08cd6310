"""Build libnmarl_hip.so (the C-ABI of include/nmarl.h) for gfx950 with hipcc.

In-tree, explicit `hipcc -shared -fPIC`: the .so travels to the GPU box with the
repo snapshot (a JIT cache under ~/.cache would not).  hipcc cross-compiles
without a GPU, so this also is the CPU-side "does it build" check.
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libnmarl_hip.so')
ARCH = 'gfx950'
# -ffp-contract=off: fp32 arithmetic follows the oracle op by op (no implicit FMA)
FLAGS = ['-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-Wall', '-Wno-unused-function']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        [os.path.join(os.path.dirname(PKG), 'include', 'nmarl.h'), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=True):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '--offload-arch=' + ARCH] + FLAGS + sources() + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    build_native(force='--force' in sys.argv)
    print(LIB)
