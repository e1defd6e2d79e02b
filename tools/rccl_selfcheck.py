#!/usr/bin/env python3
"""Self-check of the RCCL transport of the native tempering step (isingmc_pt_attach_nccl: ncclCommInitRank, grouped ncclSend /
ncclRecv of boundary counts and packed configurations, ncclAllReduce of the cutoffs).  STATUS: that transport has never run —
every box this project has had holds one GPU, and RCCL refuses two ranks on one device — so this tool is what to run FIRST on a
multi-GPU node:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 tools/rccl_selfcheck.py

One rank per GPU.  Every rank builds the SAME tempering problem twice from the same seeds — container A exchanges through RCCL on
device buffers, container B through the host-staged torch.distributed transport that the 2-rank tests cover — runs K sweeps +
tempering steps on both and requires identical labels (slot_of), configuration identities (config_of), operator counts, cutoffs,
p=0 states and operator words on every rank, and at least one configuration that crossed a rank boundary.  Exit code 0 = identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist  # (torchrun started this process before anything touched the GPU)


def main():
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world < 2 or torch.cuda.device_count() < world:
        print("rccl_selfcheck: needs >= 2 ranks, one GPU each (found %d ranks, %d devices): nothing checked" % (world, torch.cuda.device_count()))
        return 2
    torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    import _lattices as lat
    import isingmontecarlo_amd as im
    from isingmontecarlo_amd.tempering import NativeTemperingContainer
    tper, K, seed, steps = 3, 4, 2468, 20
    betas = np.geomspace(0.5, 2.6, tper * world)
    per = tper * K
    out = []
    for use_rccl in (True, False):
        g = im.QmcIsingGraph(lat.two_d_periodic(4), 1.0, 0.0, 16, seed, nreplicas=per, capacity=4096, replica_offset=rank * per, device=local)
        tc = NativeTemperingContainer(g, betas, K, seed)
        if use_rccl:
            tc.attach_rccl()
        for _ in range(steps):
            tc.timesteps(2)
            tc.tempering_step()
        ops = np.stack([np.pad(g.export_ops(r), (0, 4096 - len(g.export_ops(r)))) for r in range(per)])
        out.append((tc.slot_of.copy(), tc.config_of.copy(), g.get_n(), g.get_cutoff(), g.state_ref(), ops, tc.get_total_swaps(), bool(tc.verify())))
    a, b = out
    same = all(np.array_equal(x, y) for x, y in zip(a[:6], b[:6])) and a[6] == b[6] and a[7] and b[7]
    moved = int((a[1] // per != rank).sum())  # configurations that started on another rank
    flag = torch.tensor([int(same), moved], device="cuda")
    dist.all_reduce(flag[0:1], op=dist.ReduceOp.MIN)
    dist.all_reduce(flag[1:2], op=dist.ReduceOp.SUM)
    if rank == 0:
        print("rccl_selfcheck: %s; %d swaps, %d configurations sit on a rank they did not start on" % ("identical to the host-staged transport" if int(flag[0]) else "MISMATCH", a[6], int(flag[1])))
    dist.destroy_process_group()
    return 0 if int(flag[0]) and int(flag[1]) > 0 else 1


if __name__ == "__main__":
    sys.exit(main())
