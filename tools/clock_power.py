"""What clock and power does the chip hold under each of the batch's three kernel classes?  (pin down the box-to-box
spread of the HBM-bound update kernels with data.)  For ~2.5 s each: the one-launch BPTT of the uncoupled update
(nmarl_lstm_bptt_seq, 5 GB of HBM traffic per launch), the matrix-core lock-step kernel (nmarl_lstm_step_x, 85 MB per launch) and
the whole batch (BatchedTrainer.run_batch) run back to back on the bench shapes while a thread samples `rocm-smi` (shader clock,
memory clock, average socket power, temperature, power cap); the launch durations come from HIP events around the same loop.
    python tools/clock_power.py  -> table on stdout (profiles/rNN_clock_power.txt)"""
import configparser
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def smi():
    """One sample: dict of whatever this rocm-smi offers (field names differ between releases)."""
    out = {}
    try:
        raw = subprocess.run(['rocm-smi', '-d', '0', '--showclocks', '--showpower', '--showtemp', '--showmaxpower', '--json'],
                             capture_output=True, text=True, timeout=10).stdout
        d = json.loads(raw[raw.index('{'):])
        card = d[sorted(d)[0]]
        def mhz(v):
            v = str(v).lower()
            return float(v.strip('() ').replace('mhz', '')) if 'mhz' in v else None
        for k, v in card.items():
            kl = k.lower()
            try:
                if 'sclk' in kl and mhz(v) is not None:
                    out['sclk_MHz'] = mhz(v)
                elif 'mclk' in kl and mhz(v) is not None:
                    out['mclk_MHz'] = mhz(v)
                elif 'power' in kl and 'max' in kl:
                    out['cap_W'] = float(v)
                elif 'power' in kl and '(w)' in kl:
                    out['power_W'] = float(v)
                elif 'temperature' in kl and ('junction' in kl or 'hotspot' in kl):
                    out['Tj_C'] = float(v)
            except ValueError:
                pass
    except Exception as ex:          # never fail the measurement over a monitoring field
        out['error'] = repr(ex)[:80]
    return out


def sampled(body, seconds=2.5):
    """Run body() in a loop for `seconds`; -> (us per call, list of smi samples taken while it ran)."""
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            samples.append(smi())
    th = threading.Thread(target=poll)
    body()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    th.start()
    t0, n = time.time(), 0
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(8):
            body()
        n += 8
        torch.cuda.current_stream().synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    return e0.elapsed_time(e1) * 1e3 / n, samples


def summary(samples):
    keys = ('sclk_MHz', 'mclk_MHz', 'power_W', 'cap_W', 'Tj_C')
    cols = []
    for k in keys:
        v = [s[k] for s in samples if k in s]
        cols.append('%s %s' % (k, '-' if not v else '%.0f (%.0f..%.0f)' % (np.mean(v), min(v), max(v))))
    return ', '.join(cols) + ', %d samples' % len(samples)


def main():
    import bench
    from deeprl_network_amd import ops
    import argparse
    args = argparse.Namespace(envs=0, no_graph=False)
    cp = configparser.ConfigParser()
    cp.read(os.path.join(ROOT, 'config', 'config_ia2c_fp_catchup.ini'))
    dev = torch.device('cuda', 0)
    E, env, model, trainer = bench.make_job(args, cp, dev, 0, 1, None)
    for _ in range(3):
        trainer.run_batch()
    torch.cuda.synchronize()
    print('device: %s' % torch.cuda.get_device_name(0))
    print('idle            : %s' % summary([smi() for _ in range(3)]))
    us, s = sampled(trainer.run_batch)
    print('whole batch     : %8.1f us per batch;  %s' % (us, summary(s)))
    p = model.policy
    G, C = model.G_buf, model.C_all
    N, T, _, H4 = G.shape
    dHs = torch.randn(N, T, E, H4 // 4, device=dev)
    dZ = torch.empty_like(G)
    done = torch.zeros(T, E, device=dev)
    img = ops.lstm_bptt_wimage(None, p.params[p.k_wh])
    us, s = sampled(lambda: ops.bptt_seq(G, C, done, dHs, img, dZ, want_db=False))
    nbytes = N * T * E * 2560
    print('one-launch BPTT : %8.1f us per launch = %.2f TB/s = %.2f of 8 TB/s;  %s' % (us, nbytes / us / 1e6, nbytes / us / 1e6 / 8.0, summary(s)))
    h, c = torch.randn(N, E, 64, device=dev) * 0.3, torch.randn(N, E, 64, device=dev) * 0.3
    x = torch.relu(torch.randn(N, E, 128, device=dev))
    pi, act, v = torch.empty(N, E, 4, device=dev), torch.zeros(E, N, dtype=torch.uint8, device=dev), torch.empty(N, E, device=dev)
    gates, ho, co = torch.empty(N, E, 256, device=dev), torch.empty_like(h), torch.empty_like(c)
    zd = torch.zeros(E, device=dev)
    p.refresh_wimage()
    us, s = sampled(lambda: p.step_policy_value(x, h, c, zd, pi, act, v, h_out=ho, c_out=co, gates=gates, defer_action_term=True,
                                                mode=ops.SAMPLE_PHILOX, seed=1, env_id_base=0, step=0))
    print('LSTM lock-step  : %8.1f us per launch (eager launches, host-bound gaps included);  %s' % (us, summary(s)))


if __name__ == '__main__':
    main()
