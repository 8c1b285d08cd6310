"""Build libnmarl_hip.so (the C-ABI of include/nmarl.h) for gfx950 with hipcc.

In-tree, explicit `hipcc -c` per source (in parallel, objects cached under build/) + `hipcc -shared`: the .so
travels to the GPU box with the repo snapshot (a JIT cache under ~/.cache would not).  hipcc cross-compiles
without a GPU, so this also is the CPU-side "does it build" check.
"""
import glob
import hashlib
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


def dependencies():
    return sources() + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(os.path.dirname(PKG), 'include', 'nmarl.h')]


def source_hash():
    """sha256 over every file the library is compiled from; baked into the .so (nmarl_source_hash) and compared
    by _lib at import, so a library older than its sources is refused instead of silently mis-binding."""
    h = hashlib.sha256()
    for f in dependencies():
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:32]


def built_hash():
    """The hash string baked into the existing library, read from the file (no dlopen)."""
    if not os.path.exists(LIB):
        return None
    blob = open(LIB, 'rb').read()
    i = blob.find(b'NMARL_SRC_HASH=')
    return blob[i + 15:i + 47].decode('ascii', 'replace') if i >= 0 else None


def stale():
    return built_hash() != source_hash()


OBJ_DIR = os.path.join(PKG, 'build')      # per-file objects (git-ignored): an edit recompiles the files it touches only


def _object(src, hipcc, want, verbose):
    """One translation unit -> build/<name>.<key>.o, compiled only if no object with this key (its source, every header,
    the flags, the baked hash where the file holds it) exists."""
    h = hashlib.sha256()
    for f in [src] + [d for d in dependencies() if d.endswith('.h')]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    flags = ['--offload-arch=' + ARCH] + [f for f in FLAGS if f != '-shared']
    if os.path.basename(src) == 'cacc.hip':                 # the file that exports nmarl_source_hash
        flags.append('-DNMARL_SRC_HASH_STR="NMARL_SRC_HASH=%s"' % want)
    h.update(' '.join(flags).encode())
    stem = os.path.splitext(os.path.basename(src))[0]
    obj = os.path.join(OBJ_DIR, '%s.%s.o' % (stem, h.hexdigest()[:16]))
    if not os.path.exists(obj):
        for old in glob.glob(os.path.join(OBJ_DIR, stem + '.*.o')):
            os.remove(old)
        cmd = [hipcc] + flags + ['-c', src, '-o', obj + '.tmp']
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(obj + '.tmp', obj)
    return obj


def build_native(force=False, verbose=True):
    if not force and not stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for old in glob.glob(os.path.join(OBJ_DIR, '*.o')):
            os.remove(old)
    want = source_hash()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda f: _object(f, hipcc, want, verbose), sources()))
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC'] + objs + ['-o', LIB + '.tmp']
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    build_native(force='--force' in sys.argv)
    print(LIB)
