"""world_size-2 gloo worker for tests/test_tempering_cpu.py: each rank owns half of the temperatures."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import _lattices as lat  # noqa: E402
import _oracle as O  # noqa: E402
from _pt_backend import OracleBackend  # noqa: E402
import isingmontecarlo_amd as im  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    e, j = lat.split(lat.one_d_periodic(6, -1.0))
    m = O.Model(6, e, j, 1.0, 0.0)
    betas = np.array([0.5, 0.9, 1.3, 2.0])
    K, seed = 2, 777
    per = len(betas) // world * K
    backend = OracleBackend(m, per, 2048, 6, seed, replica_offset=rank * per)
    tc = im.TemperingContainer(backend, betas, K, seed)
    for _ in range(10):
        tc.timesteps(2)
        tc.tempering_step()
    n_all = tc.coll.all_gather_u32(backend.get_n())
    acc = tc.slot_accumulators()
    if rank == 0:
        np.savez(sys.argv[1], swaps=tc.get_total_swaps(), config_at=tc.config_at, n=n_all, acc=acc)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
