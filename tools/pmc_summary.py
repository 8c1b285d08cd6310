"""Turn the rocprofv3 --pmc passes of tools/pmc_env.py (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE runs as
MI355X_MICROARCH.md prescribes) into profiles/<tag>_pmc_traffic.{json,md}.
Corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes
of a coalesced stream -> x2; verified here on a known 1 GiB copy in the same run (`calibration`).  WRITE_SIZE is
used as reported (the same 1 GiB copy reads back exactly 1 GiB)."""
import collections
import csv
import json
import sys

d, tag = sys.argv[1], sys.argv[2]
out_dir = sys.argv[3] if len(sys.argv) > 3 else 'profiles'


def load(counter):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open('%s/pmc_%s_counter_collection.csv' % (d, counter))):
        out[r['Kernel_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    return out


F, W = load('FETCH_SIZE'), load('WRITE_SIZE')


def stat(name_part, skip=2, first=None):
    ks = [k for k in F if name_part in k]
    assert len(ks) == 1, (name_part, ks)
    k = ks[0]
    f = [x for x, _ in F[k]][skip:first]
    w = [x for x, _ in W[k]][skip:first]
    us = [t / 1e3 for _, t in F[k]][skip:first]
    return dict(kernel=k[:80], launches=len(f), fetch_KiB=sum(f) / len(f), write_KiB=sum(w) / len(w),
                us_per_launch_profiled=sum(us) / len(us))


cal = stat('__amd_rocclr_copyBuffer', skip=0, first=6)          # the six 1 GiB calibration copies at the start of pmc_env.py
fetch_corr = (1 << 20) / cal['fetch_KiB']          # known: 1 GiB read per launch
write_corr = (1 << 20) / cal['write_KiB']
res = {'calibration': dict(cal, known_bytes_each_way=1 << 30, fetch_correction=fetch_corr, write_correction=write_corr),
       'fetch_factor_used': 2.0, 'write_factor_used': 1.0, 'kernels': {}}
import os
COMPACT = os.environ.get('PMC_COMPACT', '1') == '1'
pk, bcacc = ('cacc_step_compact', 351) if COMPACT else ('cacc_step', 631)
for key, part, E, balg in ((pk + '_E2p21', 'cacc_step4_kernel<256' if COMPACT else 'cacc_step_kernel<256', 1 << 21, bcacc), (pk + '_E4096', 'cacc_step_kernel<64', 4096, bcacc),
                           ('grid_step_E2p17', 'grid_step_kernel<1, false>', 1 << 17, 7548),
                           ('grid_step_compact_E2p17', 'grid_step_kernel<1, true>', 1 << 17, 3708),
                           # env 351 + fingerprints read 128 + encoded LSTM input written 8 x 128 x 4
                           ('cacc_step_encode_E4096', 'cacc_step_encode_kernel', 4096, 351 + 128 + 4096)):
    try:
        s = stat(part)
    except AssertionError as ex:
        print('skipped', key, ex)
        continue
    traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
    s.update(replicas=E, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / E,
             algorithmic_bytes_per_replica=balg, traffic_over_algorithmic=traffic / E / balg)
    res['kernels'][key] = s
try:        # fused MFMA LSTM lock-step (x-side, policy + value heads): "replica" = one (agent, replica) row;
    # algorithmic bytes per row: x 512 + h, c in 512 + h', c' out 512 + gates 1024 + pi 16 + v 4 + action 1
    s = stat('lstm_step_x_kernel<3, 0, 0,')
    traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
    rows = 8 * 4096
    s.update(replicas=rows, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / rows,
             algorithmic_bytes_per_replica=2581, traffic_over_algorithmic=traffic / rows / 2581)
    res['kernels']['lstm_step_x_N8_E4096'] = s
except (AssertionError, ZeroDivisionError) as ex:
    print('no lstm step in this collection:', ex)
try:        # round 5, the whole lock-step in one launch (encoders + policy + value + env step): per (agent, replica) row -- own compact
    # observation 20 + own previous policy 16 + h, c 512 read; encoded LSTM input 512 + h', c' 512 + gates 1024 + pi 16 + v 4 + action 1
    # written; the hand-off word's atomic 8; the env step's 351 B per replica = 44 per row
    s = stat('lstm_step_x_kernel<3, 0, 1,')
    traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
    rows = 8 * 4096
    balg = 36 + 512 + 512 + 512 + 1024 + 21 + 8 + 44
    s.update(replicas=rows, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / rows,
             algorithmic_bytes_per_replica=balg, traffic_over_algorithmic=traffic / rows / balg)
    res['kernels']['lstm_step_x_enc_env_N8_E4096'] = s
except (AssertionError, ZeroDivisionError) as ex:
    print('no one-launch lock-step in this collection:', ex)
try:        # the coupled nets' lock-step in one launch (NeurComm, line graph): per (agent, replica) row x [hx | hp] 512 + own h, c 512 +
    # the neighbours' h before (512) and after (512) the step read; h', c' 512 + gates 1024 + message term 256 + pi 16 + v 4 + action 1 written
    s = stat('lstm_step_x_kernel<4, 1,')
    traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
    rows = 8 * 4096
    s.update(replicas=rows, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / rows,
             algorithmic_bytes_per_replica=3861, traffic_over_algorithmic=traffic / rows / 3861)
    res['kernels']['lstm_step_x4_nc_N8_E4096'] = s
except (AssertionError, ZeroDivisionError) as ex:
    print('no one-launch coupled step in this collection:', ex)
try:        # the whole reverse recurrence in one launch: "replica" = one (agent, replica, step) row of T = 60 steps;
    # algorithmic bytes per row: gates 1024 + c 256 + dy8 32 read (round 6: the heads' dL/dh is expanded inside), dz 1024 written
    s = stat('lstm_bptt_seq_kernel<true>')
    traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
    rows = 8 * 4096 * 60
    s.update(replicas=rows, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / rows,
             algorithmic_bytes_per_replica=2336, traffic_over_algorithmic=traffic / rows / 2336)
    res['kernels']['lstm_bptt_seq_N8_E4096_T60'] = s
except (AssertionError, ZeroDivisionError) as ex:
    print('no bptt_seq in this collection:', ex)
try:        # the update's heads + loss + heads' backward in one pass: per (agent, replica, step) row h 256 + va / adv / R 12 + action 1 read,
    # dy8 32 + dv 4 written
    s = stat('heads_loss_kernel<false, 5>')
    traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
    rows = 8 * 4096 * 60
    s.update(replicas=rows, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / rows,
             algorithmic_bytes_per_replica=305, traffic_over_algorithmic=traffic / rows / 305)
    res['kernels']['heads_loss_N8_E4096_T60'] = s
except (AssertionError, ZeroDivisionError) as ex:
    print('no heads_loss in this collection:', ex)
try:        # the coupled nets' reverse recurrence in one launch (NeurComm, line graph): per (agent, replica, step) row gates 1024 +
    # c 256 + dL/dh 256 + relu mask 256 + 2 neighbours' message slots 512 read; dz 1024 + d1 256 + message row 512 written
    s = stat('lstm_bptt_coupled_kernel<8, 2, true, true>')          # (dy8 form: 32 instead of 256 B of the heads' dL/dh per row-step)
    traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
    rows = 8 * 4096 * 60
    s.update(replicas=rows, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / rows,
             algorithmic_bytes_per_replica=3872, traffic_over_algorithmic=traffic / rows / 3872)
    res['kernels']['lstm_bptt_coupled_nc_N8_E4096_T60'] = s
except (AssertionError, ZeroDivisionError) as ex:
    print('no bptt_coupled in this collection:', ex)
try:        # CommNet on the 5 x 5 grid, 25 x 1024 rows: the one-launch lock-step with the observation encoder inside.  Per (agent, replica)
    # row: own + neighbours' compact observation (1 + 3.2) x 48 read; own h, c 512 + the neighbours' h before and after 2 x 3.2 x 256 read;
    # encoder output 256 + LSTM input 256 + h', c' 512 + gates 1024 + pi 20 + v 4 + action 1 written (3.2 = mean degree of the grid)
    s = stat('lstm_step_x_kernel<4, 2,')
    traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
    rows = 25 * 1024
    balg = int(4.2 * 48 + 512 + 2 * 3.2 * 256 + 256 + 256 + 512 + 1024 + 25)
    s.update(replicas=rows, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / rows,
             algorithmic_bytes_per_replica=balg, traffic_over_algorithmic=traffic / rows / balg)
    res['kernels']['lstm_step_x4_ic3_N25_E1024'] = s
except (AssertionError, ZeroDivisionError) as ex:
    print('no one-launch grid step in this collection:', ex)
try:        # its coupled BPTT (T = 120): per (agent, replica, step) row gates 1024 + c 256 + dL/dh 256 + 3.2 neighbours' message rows
    # 3.2 x 256 read; dz 1024 + d1 256 + message row 256 written
    s = stat('lstm_bptt_coupled_kernel<4, 4, false, true>')
    traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
    rows = 25 * 1024 * 120
    balg = int(1024 + 256 + 32 + 3.2 * 256 + 1024 + 256 + 256)
    s.update(replicas=rows, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / rows,
             algorithmic_bytes_per_replica=balg, traffic_over_algorithmic=traffic / rows / balg)
    res['kernels']['lstm_bptt_coupled_ic3_N25_E1024_T120'] = s
except (AssertionError, ZeroDivisionError) as ex:
    print('no grid bptt_coupled in this collection:', ex)
for key, part, balg in (
        # lstm_dial's policy step (line graph: 1.75 senders per agent on average): enc 256 + own h, c 512 + the senders' message vectors
        # 1.75 x 256 read; s 256 + hm 256 + h', c' 512 + gates 1024 + pi 16 + action 1 + the new message vectors 256 written
        ('lstm_step_x13_dial_N8_E4096', 'lstm_step_x_kernel<1, 3,', int(256 + 512 + 1.75 * 256 + 256 + 256 + 512 + 1024 + 17 + 256)),
        # its message adjoint of one reverse step: own ds, hm, msg, dhd 1024 + 1.75 sources x (ds, hm) 512 read; d1, d2, dh 768 written
        ('dial_msg_adjoint_N8_E4096', 'dial_msg_adjoint_kernel<2>', int(1024 + 1.75 * 512 + 768))):
    try:
        s = stat(part)
        traffic = (2.0 * s['fetch_KiB'] + s['write_KiB']) * 1024
        rows = 8 * 4096
        s.update(replicas=rows, traffic_bytes_per_launch=traffic, traffic_bytes_per_replica=traffic / rows,
                 algorithmic_bytes_per_replica=balg, traffic_over_algorithmic=traffic / rows / balg)
        res['kernels'][key] = s
    except (AssertionError, ZeroDivisionError) as ex:
        print('no %s in this collection:' % part, ex)
json.dump(res, open('%s/%s_pmc_traffic.json' % (out_dir, tag), 'w'), indent=1)
with open('%s/%s_pmc_traffic.md' % (out_dir, tag), 'w') as f:
    f.write('# HBM traffic of the env-step kernels (and the fused LSTM step) from rocprofv3 PMC passes\n\n'
            'commands: `rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python tools/pmc_env.py` and the same with '
            '`--pmc WRITE_SIZE` (separate runs).\n\ncalibration on a 1 GiB `copy_` in the same runs: FETCH_SIZE = %.0f KiB '
            '(x%.2f needed), WRITE_SIZE = %.0f KiB (x%.2f) -> the guide\'s gfx950 x2 FETCH correction is applied, WRITE as is.\n\n'
            % (cal['fetch_KiB'], fetch_corr, cal['write_KiB'], write_corr))
    f.write('| kernel | replicas | FETCH KiB | WRITE KiB | traffic B/replica | algorithmic B/replica | ratio | us/launch (profiled) |\n|---|---:|---:|---:|---:|---:|---:|---:|\n')
    for k, s in res['kernels'].items():
        f.write('| %s | %d | %.0f | %.0f | %.1f | %d | %.3f | %.1f |\n' % (k, s['replicas'], s['fetch_KiB'], s['write_KiB'],
                s['traffic_bytes_per_replica'], s['algorithmic_bytes_per_replica'], s['traffic_over_algorithmic'], s['us_per_launch_profiled']))
print(json.dumps(res['kernels'], indent=1))
