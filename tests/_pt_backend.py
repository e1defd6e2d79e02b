"""Oracle-backed stand-in for the GPU batch, for CPU tests of the tempering driver (test infrastructure)."""
import numpy as np

import _oracle as O


class OracleBackend:
    """Implements the backend protocol of isingmontecarlo_amd.tempering.TemperingContainer with oracle replicas.
    `replica_offset` = global id of local replica 0, like isingmc_config.replica_offset."""

    def __init__(self, model, nreplicas, capacity, cutoff, seed, replica_offset=0):
        self.model = model
        self.nreplicas = nreplicas
        self.reps = [O.Replica(model, capacity, cutoff, seed, replica_offset + r) for r in range(nreplicas)]
        self.nrows = nreplicas
        self.rows = np.arange(nreplicas)
        self.acc = np.zeros((nreplicas, 8), dtype=np.uint64)

    def get_n(self):
        return np.array([r.n for r in self.reps], dtype=np.uint32)

    def get_cutoff(self):
        return np.array([r.cutoff for r in self.reps], dtype=np.uint32)

    def set_cutoffs(self, cut):
        for r, c in zip(self.reps, cut):
            assert r.set_cutoff(int(c)) == 0

    def set_accumulator_rows(self, nrows, rows):
        if nrows != self.nrows:
            self.acc = np.zeros((nrows, 8), dtype=np.uint64)
            self.nrows = nrows
        self.rows = np.asarray(rows).astype(np.int64)

    def run(self, t, betas, sampling_freq=1, flags=0):
        for i, r in enumerate(self.reps):
            r.reset_accumulators()
            r.timesteps(t, float(betas[i]), sampling_freq, flags)
            self.acc[self.rows[i]] += r.accumulators()

    def accumulators(self):
        return self.acc.copy()

    def reset_accumulators(self):
        self.acc[:] = 0

    def get_offset(self):
        return self.model.offset

    def state_ref(self):
        return np.array([r.state() for r in self.reps], dtype=np.uint8)

    def verify(self):
        return np.array([r.verify() for r in self.reps])
