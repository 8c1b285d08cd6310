"""The C-ABI boundary: every entry point declared in include/nmarl.h is exported by the built
libnmarl_hip.so and bound (with the right arity) by deeprl_network_amd/_lib.py, and vice versa.
No compute call is made (runs without a GPU)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'nmarl.h')


def declared():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\bint\s+(nmarl_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args == 'void' else args.count(',') + 1
    return out


def test_header_symbols_are_exported_and_bound():
    from deeprl_network_amd import _lib
    decl = declared()
    assert len(decl) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name, nargs in decl.items():
        assert hasattr(lib, name), 'declared but not exported: %s' % name
        assert name in _lib.SIGNATURES, 'declared but not bound: %s' % name
        assert len(_lib.SIGNATURES[name]) == nargs, '%s: header has %d args, binding %d' % (
            name, nargs, len(_lib.SIGNATURES[name]))
    assert set(_lib.SIGNATURES) == set(decl), 'bound but not declared: %r' % (set(_lib.SIGNATURES) - set(decl))


def test_exports_are_plain_c_symbols():
    from deeprl_network_amd import _lib
    nm = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in nm.splitlines() if ' T ' in line}
    assert set(declared()) <= exported
    assert not [s for s in exported if s.startswith('_Z') and 'nmarl' in s and 'kernel' not in s]


def test_struct_layouts_match_header():
    from deeprl_network_amd import _lib
    assert ctypes.sizeof(_lib.CaccParams) == 12 * 4 + 6 * 4
    assert ctypes.sizeof(_lib.GridParams) == 4 * 4 + 3 * 4 + 2 * 4 + 4 + 8      # (+ objective, coef_wait, padding, head_wait)
    assert _lib.lib.nmarl_abi_version() == _lib.ABI_VERSION


def test_library_is_built_from_the_current_sources():
    """The hash baked into the .so equals the hash of csrc/* + include/nmarl.h: a stale library (struct layouts
    or kernels older than the binding) is refused at import, and caught here before any GPU time is spent."""
    from deeprl_network_amd import _lib, build
    assert build.built_hash() == build.source_hash()
    fn = _lib.lib.nmarl_source_hash
    fn.restype = ctypes.c_char_p
    assert fn().decode() == 'NMARL_SRC_HASH=' + build.source_hash()


@pytest.mark.parametrize('ctype,cls', [('nmarl_head_t', 'Head'), ('nmarl_fc_part_t', 'FcPart'), ('nmarl_msg_t', 'Msg'),
                                       ('nmarl_cacc_params_t', 'CaccParams'), ('nmarl_grid_params_t', 'GridParams'),
                                       ('nmarl_bptt_coupled_t', 'BpttCoupled'), ('nmarl_batch_epilogue_t', 'BatchEpilogue'), ('nmarl_cacc_encode_t', 'CaccEncode'),
                                       ('nmarl_step_enc_t', 'StepEnc'), ('nmarl_grid_env_t', 'GridEnv')])
def test_struct_layouts_match_c_compiler(tmp_path, ctype, cls):
    """Every struct of the C-ABI as gcc lays it out == its ctypes mirror, field by field."""
    from deeprl_network_amd import _lib
    mirror = getattr(_lib, cls)
    fields = [n for n, _ in mirror._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "%s"\nint main(){printf("%%zu", sizeof(%s));\n' % (HEADER, ctype)
    for f in fields:
        src += 'printf(" %%zu", offsetof(%s, %s));\n' % (ctype, f)
    src += 'return 0;}\n'
    c = tmp_path / 'off.c'
    c.write_text(src)
    exe = str(tmp_path / 'off')
    subprocess.check_call(['gcc', str(c), '-o', exe])
    nums = [int(x) for x in subprocess.run([exe], capture_output=True, text=True).stdout.split()]
    assert nums[0] == ctypes.sizeof(mirror)
    assert nums[1:] == [getattr(mirror, f).offset for f in fields]


def test_net_image_layout_matches_header():
    from deeprl_network_amd import _lib
    src = open(HEADER).read()
    defs = dict(re.findall(r'#define NMARL_NET_(OFF_\w+|IMAGE_BYTES) (\d+)', src))
    assert int(defs['IMAGE_BYTES']) == _lib.NET_IMAGE_BYTES
    assert {k[4:].lower(): int(v) for k, v in defs.items() if k.startswith('OFF_')} == _lib.NET_OFF


def test_net_kernel_row_index_magic_is_exact():
    """csrc/realnet.hip maps a flat state index to its node row by (i * ceil(65536 / L)) >> 16."""
    for L in range(1, 25):
        m = (65536 + L - 1) // L
        assert all((i * m) >> 16 == i // L for i in range(32 * 24))


def test_ops_fail_loudly_without_gpu_tensors():
    import pytest
    import torch
    from deeprl_network_amd import _lib, ops
    with pytest.raises(_lib.NmarlError):
        ops.nbr_onehot(torch.zeros(2, 8, dtype=torch.uint8), torch.zeros(8, 2, dtype=torch.int32), 4)
    with pytest.raises(_lib.NmarlError):
        ops.rmsprop_tf_clip(torch.zeros(1, 4), torch.zeros(1, 4), torch.ones(1, 4), torch.zeros(1, 64), 1e-3, .99, 1e-5, 40)


def test_library_issues_no_memset_or_memcpy_calls():
    """Rule learnt in round 5 (profiles/r05_determinism.txt): inside a captured hipGraph a hipMemsetAsync becomes a
    memset NODE, and that node was not reliably ordered in front of the kernel behind it (NeurComm's captured update went
    nondeterministic after ~900 replays).  Every entry point of the library may be captured, so buffers are cleared by kernels."""
    import glob
    import re
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'deeprl_network_amd', 'csrc')
    hits = []
    for f in sorted(glob.glob(os.path.join(csrc, '*.hip')) + glob.glob(os.path.join(csrc, '*.h'))):
        for i, line in enumerate(open(f), 1):
            code = line.split('//')[0]
            if re.search(r'\bhip(Memset|Memcpy)\w*\s*\(', code):
                hits.append('%s:%d: %s' % (os.path.basename(f), i, line.strip()))
    assert not hits, hits
