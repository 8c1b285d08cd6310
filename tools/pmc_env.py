"""PMC target for rocprofv3 --pmc passes: a calibration pair (known-size copies: 16 B/lane vectorised and
4 B/lane scalar) followed by the CACC / grid step kernels at a size far beyond the 256 MB Infinity Cache.
Usage (one counter group per run, as MI355X_MICROARCH.md prescribes):
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o pmc_fetch --output-format csv -- python tools/pmc_env.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from helpers import cacc_config, grid_config
from deeprl_network_amd.envs.cacc_env import CACCBatchEnv
from deeprl_network_amd.envs.large_grid_env import LargeGridBatchEnv

n = 1 << 28                                    # 1 GiB of fp32
x = torch.ones(n + 4, device='cuda')
y = torch.empty(n + 4, device='cuda')
for _ in range(3):
    y[:n].copy_(x[:n])                         # vectorised: 16 B / lane
for _ in range(3):
    y[:n].copy_(x[1:n + 1])                    # misaligned source: scalar 4 B / lane loads
torch.cuda.synchronize()
E = int(os.environ.get('PMC_E', 1 << 21))
env = CACCBatchEnv(cacc_config()['ENV_CONFIG'], num_envs=E)
env.reset()
e = torch.arange(E, device='cuda')[:, None]
a = torch.arange(8, device='cuda')[None, :]
acts = [((e + 3 * a + s) % 4).to(torch.uint8).contiguous() for s in range(4)]
for s in range(8):
    env.step(acts[s % 4], auto_reset=True)
torch.cuda.synchronize()
Es = 4096                                      # the bench workload: cacc_step_kernel<64,0>
senv = CACCBatchEnv(cacc_config()['ENV_CONFIG'], num_envs=Es)
senv.reset()
for s in range(16):
    senv.step(acts[s % 4][:Es].contiguous(), auto_reset=True)
torch.cuda.synchronize()
Eg = int(os.environ.get('PMC_EG', 1 << 17))
genv = LargeGridBatchEnv(grid_config()['ENV_CONFIG'], num_envs=Eg)
genv.reset()
ga = [torch.randint(0, 5, (Eg, 25), dtype=torch.uint8, device='cuda') for _ in range(2)]
for s in range(6):
    genv.step(ga[s % 2], auto_reset=True)
torch.cuda.synchronize()
# the fused MFMA LSTM step at the bench shape ([8, 4096, 64] state; in place like the rollout) and its head variants
from deeprl_network_amd import ops
N, El, H, A = 8, 4096, 64, 4
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).cuda()                                   # noqa: E731
h, c, z = r(N, El, H) * 0.3, r(N, El, H) * 0.3, r(N, El, 4 * H)
wh, b = r(N, H, 4 * H) * 0.1, r(N, 4 * H) * 0.1
done = torch.zeros(El, device='cuda')
for s in range(12):
    ops.lstm_step_fused(h, wh, b, z, None, c, done, None, c, h)
torch.cuda.synchronize()
print('done', E, Eg)
