"""Kernel statistics from a rocprofv3 rocpd database (bench_results.db): per-kernel totals as CSV (the same columns
as rocprofv3's kernel_stats.csv) and, optionally, the kernel sequence of one A2C update.

    python tools/rocpd_stats.py gpurun_out/prof_x/bench_results.db [--steps 7] [--update]
"""
import csv
import re
import sqlite3
import sys
from collections import Counter


def short(n):
    n = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', n)
    return n.replace('_ZN2at6native', 'at::')[:78]


def main():
    path = sys.argv[1]
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 1
    cur = sqlite3.connect(path).cursor()
    q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from rocpd_kernel_dispatch d
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"""
    rows = list(cur.execute(q))
    tot = sum(r[2] for r in rows)
    out = path.replace('_results.db', '_kernel_stats.csv')
    with open(out, 'w') as f:
        w = csv.writer(f)
        w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage'])
        for r in rows:
            w.writerow([r[0], r[1], r[2], r[3], 100.0 * r[2] / tot])
    print('total kernel ms %.2f over %d kernels -> %s' % (tot / 1e6, sum(r[1] for r in rows), out))
    for r in rows[:32]:
        print('%-78s calls=%6d /%d=%6.2fms avg=%8.2fus' % (short(r[0]), r[1], steps, r[2] / steps / 1e6, r[3] / 1e3))
    if '--update' in sys.argv:
        seq = list(cur.execute("""select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d
                                  join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"""))
        idx = [i for i, r in enumerate(seq) if 'nstep_kernel' in r[0]]
        # one batch = from an update's return scan to the next one, with a rollout in between (the last updates of a bench.py trace are
        # its side measurements: update graphs replayed back to back)
        is_step = lambda r: 'lstm_step' in r[0] and any(h in r[0] for h in ('Li1E', 'Li3E', 'Li4E'))      # noqa: E731
        seg = None
        for k in range(len(idx) - 1, 0, -1):         # the shortest such stretch is a plain batch (the side measurements replay whole
            cand = seq[idx[k - 1]:idx[k]]           # rollout graphs several times between two updates)
            if any(is_step(r) for r in cand[6:]) and (seg is None or len(cand) < len(seg)):
                seg = cand
        if seg is None:
            print('no batch with a rollout found in the trace')
            return
        # the next rollout starts at its first lock-step launch (policy / policy + value heads: the update of a batch with saved
        # activations launches none), or at the short launches in front of it that belong to it: the weight images, a separate
        # encoder launch over E rows (absent where the lock-step kernel runs the encoders itself: round 5)
        first = next(i for i, r in enumerate(seg) if i > 5 and is_step(r))
        end = first
        while end > 6 and (seg[end - 1][2] - seg[end - 1][1]) < 30000 and any(k in seg[end - 1][0] for k in ('gather_fwd', 'fc_fwd', 'wimage')):
            end -= 1
        print('batch: %d kernels, span %.2f ms, busy %.2f ms; update: %d kernels, span %.2f ms, busy %.2f ms' % (
            len(seg), (seg[-1][2] - seg[0][1]) / 1e6, sum(r[2] - r[1] for r in seg) / 1e6, end,
            (seg[end - 1][2] - seg[0][1]) / 1e6, sum(r[2] - r[1] for r in seg[:end]) / 1e6))
        cnt, tt = Counter(), Counter()
        many = Counter(short(r[0]) for r in seg[:end])
        for r in seg[:end]:
            d = (r[2] - r[1]) / 1e3
            if d > 30 and many[short(r[0])] < 8:          # one-off big kernels in launch order; loops are summed below
                print('%9.1f us  %s' % (d, short(r[0])))
            else:
                cnt[short(r[0])] += 1
                tt[short(r[0])] += d
        for k, v in cnt.most_common(16):
            print('   %4d x %8.1f us total (avg %6.1f)  %s' % (v, tt[k], tt[k] / v, k))
        # the rollout part: per kernel totals (the hipGraph of n_step lock-steps + bootstrap)
        cnt, tt = Counter(), Counter()
        for r in seg[end:]:
            cnt[short(r[0])] += 1
            tt[short(r[0])] += (r[2] - r[1]) / 1e3
        print('rollout: %d kernels, span %.2f ms, busy %.2f ms' % (len(seg) - end, (seg[-1][2] - seg[end][1]) / 1e6,
                                                                  sum(tt.values()) / 1e3))
        for k, v in sorted(tt.items(), key=lambda kv: -kv[1])[:14]:
            print('   %4d x %8.1f us total (avg %6.1f)  %s' % (cnt[k], v, v / cnt[k], k))


if __name__ == '__main__':
    main()
