#!/usr/bin/env python3
"""BASELINE configs[3] on one rank's share: 64x64 TFIM parallel tempering, NT temperatures x NC walkers per GPU, one
native tempering step (isingmc_pt_step: label swaps inside the rank's temperature block, point-to-point exchange of the
boundary walkers with the neighbouring ranks) after every sweep.  Prints one JSON line.

Single GPU:   python tools/bench_tempering.py
Several GPUs: python -m torch.distributed.run --nproc-per-node G --master-addr 127.0.0.1 tools/bench_tempering.py [--rccl]
              (temperatures are sharded over the ranks; --rccl attaches the library's own RCCL communicator: ncclSend / ncclRecv
              on device buffers, one rank per GPU; without it the exchange is staged through torch.distributed)
--window B W : betas geometric in [B / W, B * W] instead of [beta-min, beta-max] (a grid on which 64x64 swaps are accepted)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import _lattices as lat
import isingmontecarlo_amd as im
from isingmontecarlo_amd.tempering import NativeTemperingContainer

ap = argparse.ArgumentParser()
ap.add_argument("--L", type=int, default=64); ap.add_argument("--ntemps", type=int, default=32, help="temperatures per GPU")
ap.add_argument("--nchains", type=int, default=32); ap.add_argument("--beta-min", type=float, default=0.5)
ap.add_argument("--beta-max", type=float, default=16.0); ap.add_argument("--equilibrate", type=int, default=60)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--seed", type=int, default=1234)
ap.add_argument("--rccl", action="store_true"); ap.add_argument("--window", type=float, nargs=2, default=None)
a = ap.parse_args()
world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
ntemps = a.ntemps * world
if a.window:
    a.beta_min, a.beta_max = a.window[0] / a.window[1], a.window[0] * a.window[1]
betas = np.geomspace(a.beta_min, a.beta_max, ntemps)
L, N = a.L, a.L * a.L
R = a.ntemps * a.nchains
cap = 1 << int(np.ceil(np.log2(2.0 * a.beta_max * 5.2 * N + 4 * N)))
g = im.QmcIsingGraph(lat.two_d_ferro(L), 1.0, 0.0, N, a.seed, nreplicas=R, capacity=cap, replica_offset=rank * R, device=local)
tc = NativeTemperingContainer(g, betas, a.nchains, a.seed, flags=0)
if a.rccl and world > 1:
    tc.attach_rccl()  # (never executed on hardware so far: run tools/rccl_selfcheck.py first on a multi-GPU node)
for _ in range(a.equilibrate):
    tc.timesteps(1); tc.tempering_step(count_swaps=False)
g.reset_accumulators()
swaps0 = tc.get_total_swaps()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps):
    tc.timesteps(1); tc.tempering_step(count_swaps=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
acc = g.accumulators().astype(np.float64)
if rank == 0:
    print(json.dumps({"workload": f"configs[3] share of one GPU: {L}x{L} TFIM parallel tempering, {a.ntemps} temperatures x {a.nchains} walkers per GPU, "
                                  f"{ntemps} temperatures in [{a.beta_min:.4g}, {a.beta_max:.4g}] over {world} GPU(s), native tempering step "
                                  f"(isingmc_pt_step, {'RCCL point-to-point' if a.rccl and world > 1 else 'single rank' if world == 1 else 'host-staged transport'}) after every sweep",
                      "n_gpus": world, "steps": a.steps, "ms_per_sweep_plus_tempering_step": dt * 1e3 / a.steps,
                      "swaps_per_step": (tc.get_total_swaps() - swaps0) / a.steps,
                      "spin_op_updates_per_s_this_rank": float(acc[:, 4].sum() + acc[:, 5].sum()) / dt,
                      "mean_cutoff": float(g.get_cutoff().mean()), "max_cutoff": int(g.get_cutoff().max()),
                      "launch": g.launch_info(), "all_verified": bool(g.verify().all())}))
