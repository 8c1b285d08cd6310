"""Host-side pieces of the reference's agents/utils.py that stay on the host.

The n-step buffers of the reference (OnPolicyBuffer / MultiAgentOnPolicyBuffer,
agents/utils.py:722-912) are device tensors owned by agents.models (buf_*), their
return scan is the HIP kernel nmarl_nstep_return; only the lr schedule lives here.
"""


class Scheduler:
    """Learning rate as a function of the lock-steps consumed (agents/utils.py:917-930; same constructor, `get(n)` advances
    by n and THEN evaluates, `n` is what a checkpoint stores).  `at` / `rewind` / `constant` serve the batched engine: a
    batch that is re-run after a hand-off time-out takes its steps back, and a captured update reads the rate from a
    device scalar that only has to be refreshed when the schedule is not constant."""

    def __init__(self, val_init, val_min=0, total_step=0, decay='linear'):
        self.n = 0
        self._v0, self._floor, self._span, self._kind = val_init, val_min, float(total_step), decay

    @property
    def constant(self):
        return self._kind != 'linear'

    def at(self, n):
        return self._v0 if self.constant else max(self._floor, self._v0 * (1 - n / self._span))

    def get(self, n_step):
        self.n += n_step
        return self.at(self.n)

    def rewind(self, n_step):
        self.n -= n_step
