"""Small host-side utilities of the reference's agents/utils.py that stay on the host.

The n-step buffers of the reference (OnPolicyBuffer / MultiAgentOnPolicyBuffer,
agents/utils.py:722-912) are device tensors owned by agents.models (buf_*), their
return scan is the HIP kernel nmarl_nstep_return; only the lr schedule lives here.
"""


class Scheduler:
    """Constant / linear-decay schedule (agents/utils.py:917-930)."""

    def __init__(self, val_init, val_min=0, total_step=0, decay='linear'):
        self.val = val_init
        self.N = float(total_step)
        self.val_min = val_min
        self.decay = decay
        self.n = 0

    def get(self, n_step):
        self.n += n_step
        if self.decay == 'linear':
            return max(self.val_min, self.val * (1 - self.n / self.N))
        return self.val
