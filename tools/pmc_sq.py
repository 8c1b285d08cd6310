"""Issue / stall / matrix-pipe picture of the hot kernels from two rocprofv3 --pmc passes over tools/pmc_env.py (SQ counters):
    pass A: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
    pass B: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
    python tools/pmc_sq.py /tmp/pmcsq > profiles/rNN_pmc_sq.md
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts
cycles summed over SIMDs.  WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for path in glob.glob(d + '/*_counter_collection.csv'):
    for r in csv.DictReader(open(path)):
        vals[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
WANT = ['lstm_step_x_kernel<3, 0, 1', 'lstm_step_x_kernel<4, 1,', 'lstm_step_x_kernel<4, 2,', 'lstm_bptt_seq_kernel<true>', 'lstm_bptt_coupled_kernel<8, 2, true, true>',
        'lstm_bptt_coupled_kernel<4, 4, false, true>', 'heads_loss_kernel<false, 5>', 'cacc_step4_kernel', 'fc_bwd_pair']
print('# SQ counters of the hot kernels (rocprofv3 --pmc, two passes over tools/pmc_env.py; means over the launches of the harness)\n')
print('| kernel | us (profiled) | parked (WAIT_ANY) | issue stall (WAIT_INST_ANY) | issuing (ACTIVE_INST_ANY) | of which VALU | MFMA busy cycles / (SIMDs x kernel cycles) | VALU : MFMA wave-instructions per launch (millions) | LDS bank-conflict cycles / LDS active |')
print('|---|---:|---:|---:|---:|---:|---:|---:|---:|')


def m(k, c):
    v = vals[k].get(c)
    return sum(v[2:]) / max(len(v[2:]), 1) if v and len(v) > 2 else (sum(v) / len(v) if v else float('nan'))


for want in WANT:
    ks = [k for k in vals if want in k]
    for k in ks[:1]:
        wc = m(k, 'SQ_WAVE_CYCLES')
        us = sum(dur[k]) / len(dur[k]) / 1e3
        cyc = us * 2.4e3 * 1024            # 1024 SIMDs x kernel cycles at a nominal 2.4 GHz
        print('| `%s` | %.1f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f : %.2f | %.3f |' % (
            k.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0][:60], us, m(k, 'SQ_WAIT_ANY') / wc, m(k, 'SQ_WAIT_INST_ANY') / wc,
            m(k, 'SQ_ACTIVE_INST_ANY') / wc, m(k, 'SQ_ACTIVE_INST_VALU') / wc, m(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / cyc,
            m(k, 'SQ_INSTS_VALU') / 1e6, m(k, 'SQ_INSTS_MFMA') / 1e6,
            m(k, 'SQ_LDS_BANK_CONFLICT') / max(m(k, 'SQ_LDS_IDX_ACTIVE'), 1.0)))
print('\nFractions are of SQ_WAVE_CYCLES (time waves exist).  MFMA busy is against ALL 1024 SIMDs at a nominal 2.4 GHz: kernels that leave '
      'compute units idle (the grid lock-step: 200 + 56 blocks) or run below that clock read lower than their busy SIMDs are.')
