"""world_size-2 worker for tests/test_gpu_tempering.py: both ranks on GPU 0, each owns half of the temperatures; the
neighbour exchange of isingmc_pt_step runs through the host-staged transport (torch.distributed gloo)."""
import os
import sys

import numpy as np
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import _lattices as lat  # noqa: E402
import isingmontecarlo_amd as im  # noqa: E402


def main():
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    betas = np.array([0.5, 0.8, 1.1, 1.5, 2.0, 2.6])
    K, seed, steps, sweeps = 4, 2468, 15, 2
    edges = lat.two_d_periodic(4)
    per = len(betas) // world * K
    g = im.QmcIsingGraph(edges, 1.0, 0.0, 16, seed, nreplicas=per, capacity=4096, replica_offset=rank * per, device=0)
    tc = im.NativeTemperingContainer(g, betas, K, seed)
    for _ in range(steps):
        tc.timesteps(sweeps)
        tc.tempering_step()
    swaps = tc.get_total_swaps()
    ok = tc.verify()
    np.savez(sys.argv[1] + f".rank{rank}.npz", swaps=swaps, ok=ok, slot_of=tc.slot_of, config_of=tc.config_of, n=g.get_n(), cutoff=g.get_cutoff(),
             state=g.state_ref(), ops=np.array([np.pad(g.export_ops(r), (0, 4096 - len(g.export_ops(r)))) for r in range(per)]),
             acc=g.accumulators())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
