"""Per-kernel register / scratch / LDS table of the HIP sources, from hipcc's -Rpass-analysis=kernel-resource-usage
(cross-compiles for gfx950 without a GPU).  python tools/resource_usage.py [file.hip ...] [--md] [--filter substr]
Used to keep the matrix-core kernels free of spills (profiles/rNN_resource_usage.md)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'deeprl_network_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-c', '-Rpass-analysis=kernel-resource-usage']
KEYS = ['TotalSGPRs', 'VGPRs', 'AGPRs', 'ScratchSize [bytes/lane]', 'Occupancy [waves/SIMD]', 'SGPRs Spill', 'VGPRs Spill',
        'LDS Size [bytes/block]']


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout
    return [re.sub(r'\(anonymous namespace\)::', '', x).replace('void ', '').split('(')[0] for x in out.strip().split('\n')]


def usage(src, extra=()):
    p = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950'] + FLAGS + list(extra) + [src, '-o', '/dev/null'],
                       capture_output=True, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr)
        raise SystemExit(p.returncode)
    rows, cur = [], None
    for line in p.stderr.splitlines():
        m = re.search(r'remark: +Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r'remark: +([A-Za-z ]+(?:\[[^\]]+\])?): (\d+)', line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    for r, d in zip(rows, demangle([r['name'] for r in rows])):
        r['name'] = d
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    md = '--md' in sys.argv
    flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
    if flt in args:
        args.remove(flt)
    files = args or sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdr = ['kernel', 'SGPR', 'VGPR', 'AGPR', 'scratch B/lane', 'waves/SIMD', 'SGPR spill', 'VGPR spill', 'LDS B']
    if md:
        print('| ' + ' | '.join(hdr) + ' |')
        print('|' + '---|' * len(hdr))
    for f in files:
        for r in usage(f):
            if flt and flt not in r['name']:
                continue
            vals = [r['name']] + [str(r.get(k, '')) for k in KEYS]
            print(('| ' + ' | '.join(vals) + ' |') if md else '%-58s ' % vals[0] + ' '.join('%6s' % v for v in vals[1:]))


if __name__ == '__main__':
    main()
