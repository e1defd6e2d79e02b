#!/usr/bin/env python3
"""bench.py — headline benchmark of the SSE sweep on MI355X (BASELINE.json metric).

A "step" is one SSE sweep (timestep) of every replica resident on a GPU.  Workload = BASELINE.json
configs[1]: 32x32 periodic square-lattice TFIM, J=-1 (ferromagnet in the reference's sign convention),
Gamma=1, h=0, beta=16, 1024 independent replicas per GPU, driven like the reference's Qmc::timestep with
loop updates enabled (src/sse/qmc_runner.rs:363-377): diagonal sweep -> one directed loop -> cluster
flips -> free-spin randomisation.  Synthetic input = op-strings equilibrated on the device before the
warm-up (the cutoff must first grow from N to ~1.5 n; that growth is data preparation, not the workload).

value = spin-operator updates / second, whole job: one update = one op-string slot processed by one pass
(diagonal pass: cutoff M slots; off-diagonal passes: vertices visited), summed over all ranks' replicas.

Multi-GPU: replicas shard across ranks, no data-path collective; launched by the driver as
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`.  BASELINE's metric reads "1024 replicas at 1/2/4/8
GPUs"; `value` is the WEAK-scaling reading (1024 replicas on every GPU: the per-GPU work that fills the machine — the diagonal
launch runs 4 workgroups of 4 waves per CU, i.e. exactly 1024 replicas on 256 CUs; fewer replicas per GPU leave CUs idle), and
for N > 1 the line also carries the STRONG-scaling reading (1024 replicas in total, 1024/N per GPU) under `strong_scaling`,
measured in the same invocation.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# algorithmic HBM bytes per slot per sweep with one u32 word per slot:
#   diagonal pass 4 B read + 4 B written; cluster pass build 4 B read + apply 4 B read + 4 B written
#   => 20 B per slot.  The directed loop moves 8 B per VISITED VERTEX (a few hundred vertices per sweep against
#   1.3e5 slots): below 0.1 % of the sweep and not counted.
BYTES_PER_SLOT_DIAG, BYTES_PER_SLOT_CLUSTER = 8.0, 12.0


def lattice_edges(l):
    import _lattices as lat
    return lat.two_d_ferro(l)


def usable_cores():
    """Host cores this process may really use: the CPU affinity mask, capped by the cgroup CPU quota (the GPU boxes
    expose every logical CPU of the host but grant a quota of 16 of them to a one-GPU job)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); pr = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // pr))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(l, beta, flags, seed, budget_s=20.0, pmj3d=0, couplings=None):
    """Time the CPU oracle (a C restatement of the reference path, kind='port') on this box's host cores."""
    import _oracle
    import _lattices as lat
    _oracle.build()
    nthreads = max(1, min(_oracle.lib().ora_max_threads(), usable_cores()))
    eq = 60
    if pmj3d:  # configs[4] geometry: one +-J realisation per thread (the first rows of the GPU run's couplings)
        edges = lat.cubic_periodic(pmj3d)
        e, _ = lat.split(edges)
        nsite, eq = pmj3d ** 3, 30
        cap = 1 << int(np_ceil_log2(2.0 * beta * (len(edges) * 1.3 + nsite * 1.2) + 4 * nsite))
        models = [_oracle.Model(nsite, e, list(couplings[r % len(couplings)]), 1.0, 0.1) for r in range(nthreads)]
        reps = [_oracle.Replica(models[r], cap, nsite, seed, 1_000_000 + r) for r in range(nthreads)]
        what = f"{pmj3d}^3 +-J h=0.1"
    else:
        edges = lattice_edges(l)
        e, j = lat.split(edges)
        m = _oracle.Model(l * l, e, j, 1.0, 0.0)
        reps = [_oracle.Replica(m, 1 << 18, l * l, seed, 1_000_000 + r) for r in range(nthreads)]
        what = f"{l}x{l}"
    betas = [beta] * nthreads
    t0 = time.time()
    _oracle.batch_timesteps(reps, eq, betas, 1, flags, nthreads)  # equilibrate (untimed)
    t_eq = time.time() - t0
    per_sweep = max(t_eq / eq, 1e-4)
    sweeps = int(max(4, min(200, (budget_s - t_eq) / per_sweep))) if budget_s > t_eq else 4
    for r in reps:
        r.reset_accumulators()
    t0 = time.time()
    _oracle.batch_timesteps(reps, sweeps, betas, 1, flags, nthreads)
    dt = time.time() - t0
    upd = sum(int(r.accumulators()[4]) + int(r.accumulators()[5]) for r in reps)
    return {"value": upd / dt, "unit": "spin-op updates/s", "cores": nthreads, "kind": "port",
            "sample": f"{nthreads} replicas (one per thread) x {sweeps} sweeps of the same {what} beta={beta} "
                      f"workload after {eq} equilibration sweeps; C oracle, not the Rust binary",
            "sweeps_per_s_per_core": sweeps / dt, "host_logical_cpus": os.cpu_count()}


def np_ceil_log2(x):
    import math
    return math.ceil(math.log2(x))


METRIC = "spin-op updates/sec (whole node), 32\u00d732 TFIM, 1024 replicas at 1/2/4/8 GPUs"  # BASELINE.json, verbatim


def timed_run(g, steps, warmup, beta, flags, dist, torch):
    """W untimed warm-up steps, then exactly `steps` steps between barrier + synchronize on both sides; returns seconds."""
    if warmup:
        g.run(warmup, beta, flags=flags)
    g.reset_accumulators()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.run(steps, beta, flags=flags)  # EXACTLY K steps, returns after completion
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--replicas", type=int, default=1024, help="replicas per GPU")
    ap.add_argument("--L", type=int, default=32)
    ap.add_argument("--beta", type=float, default=16.0)
    ap.add_argument("--equilibrate", type=int, default=80, help="untimed sweeps that prepare the synthetic input")
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--k", type=int, default=0)
    ap.add_argument("--no-lds-tables", action="store_true")
    ap.add_argument("--global-tables", action="store_true", help="force the per-variable tables into HBM (with --waves: that wave count in the HBM-table mode)")
    ap.add_argument("--no-loop", action="store_true")
    ap.add_argument("--rvb", action="store_true", help="configs[2]: QmcIsingGraph::timestep with RVB sweeps (no directed loop)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--pmj3d", type=int, default=0, metavar="L",
                    help="secondary workload (configs[4] geometry): L^3 periodic cubic lattice, J=+-1 i.i.d. per replica, "
                         "Gamma=1, h=0.1, one disorder realisation per replica; use with --beta 4 --replicas 512")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="HBM bytes per launch from a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run (profiles/README.md)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import isingmontecarlo_amd as im

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # SSE_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks then share devices and the
    # scalar reductions go through CPU tensors); the driver's runs use nccl = RCCL, one rank per GPU.
    backend = os.environ.get("SSE_BENCH_BACKEND", "nccl")
    # one rank per GPU; if the launcher narrowed the visible devices of each rank to one, that one is device 0
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    flags = 0 if args.no_loop else im.FLAG_LOOP
    if args.rvb:
        flags = im.FLAG_RVB
    L, R, beta = args.L, args.replicas, args.beta
    edges = lattice_edges(L)
    n_est = beta * (3 * L * L + 2.2 * L * L)
    cap = 1 << int(np.ceil(np.log2(2.0 * n_est + 4 * L * L)))
    if args.pmj3d:
        import _lattices as lat3
        L = args.pmj3d
        edges = lat3.cubic_periodic(L)
        nsite = L ** 3
        rngJ = np.random.default_rng(args.seed + 7919 * rank)
        couplings = rngJ.choice([-1.0, 1.0], size=(R, len(edges)))
        couplings_cpu = couplings[:64].copy()
        n_est = beta * (len(edges) * 1.3 + nsite * 1.2)
        cap = 1 << int(np.ceil(np.log2(2.0 * n_est + 4 * nsite)))
        flags &= ~im.FLAG_LOOP
        g = im.QmcIsingGraph(edges, 1.0, 0.1, nsite, args.seed, nreplicas=R, capacity=cap, replica_offset=rank * R,
                             device=local_rank, waves_per_replica=args.waves, slots_per_lane=args.k, couplings=couplings,
                             cfg_flags=(im.CFG_GLOBAL_TABLES | im.CFG_NO_LDS_TABLES) if args.global_tables else 0)
    else:
        g = im.QmcIsingGraph(edges, 1.0, 0.0, L * L, args.seed, nreplicas=R, capacity=cap,
                             replica_offset=rank * R, device=local_rank, waves_per_replica=args.waves, slots_per_lane=args.k,
                             cfg_flags=im.CFG_NO_LDS_TABLES if args.no_lds_tables else 0)
    # data preparation: equilibrate (cutoff growth + thermalisation), untimed.  FLAG_PREP runs the identical
    # kernel under its "data preparation" symbol so that rocprofv3 --stats averages only the measured launches.
    for _ in range(0, args.equilibrate, 10):  # in chunks: the LDS union-find capacity adapts between launches
        g.run(min(10, args.equilibrate), beta, flags=(flags & ~im.FLAG_RVB) | im.FLAG_PREP)
    # Two kernel launches per sweep (isingmc_hip.hip run()): diagonal pass + directed loop (sse::sweep_kernel<..,PASSES=1>)
    # and cluster + free spins (sse::sweep_kernel<..,PASSES=2>).  A "launch" in the
    # roofline object is one launch of the dominant kernel = its pass over all R replicas of this rank.
    dt = timed_run(g, args.steps, args.warmup, beta, flags, dist, torch)

    pass_ms = g.last_pass_ms()
    g_rvb_ms = g.last_rvb_ms()
    acc = g.accumulators().astype(np.float64)
    updates = float(acc[:, 4].sum() + acc[:, 5].sum())
    slots = float(acc[:, 5].sum())  # sum over replicas and steps of the cutoff M
    kernel_ms, launches = g.last_kernel_ms()
    mean_M = slots / (R * args.steps)
    mean_n = float(acc[:, 0].sum() / max(1.0, acc[:, 1].sum()))
    t = torch.tensor([dt, updates, slots, kernel_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if dist is not None:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, updates, slots_all = float(tmax[0]), float(tsum[1]), float(tsum[2])
    else:
        slots_all = slots
    launch_info = g.launch_info()
    energy = -(acc[:, 0] / np.maximum(acc[:, 1], 1)) / beta + g.get_offsets()
    # ---- strong-scaling reading of the same metric (N > 1): 1024 replicas in total, R/N per GPU ----
    strong = None
    if world > 1 and not args.pmj3d and not args.rvb and R % world == 0:
        del g
        Rs = R // world
        gs = im.QmcIsingGraph(edges, 1.0, 0.0, L * L, args.seed, nreplicas=Rs, capacity=cap, replica_offset=rank * Rs,
                              device=local_rank, waves_per_replica=args.waves, slots_per_lane=args.k)
        for _ in range(0, args.equilibrate, 10):
            gs.run(min(10, args.equilibrate), beta, flags=flags | im.FLAG_PREP)
        dts = timed_run(gs, args.steps, args.warmup, beta, flags, dist, torch)
        accs = gs.accumulators().astype(np.float64)
        ts = torch.tensor([dts, float(accs[:, 4].sum() + accs[:, 5].sum())], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        tm = ts.clone(); dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        tsu = ts.clone(); dist.all_reduce(tsu, op=dist.ReduceOp.SUM)
        strong = {"scaling": "strong", "replicas_total": R, "replicas_per_gpu": Rs, "value": float(tsu[1]) / float(tm[0]),
                  "ms_per_step": float(tm[0]) * 1e3 / args.steps,
                  "note": "1024/N replicas per GPU leave (N-1)/N of the diagonal launch's workgroup slots empty"}

    if rank == 0:
        # per kernel: algorithmic bytes per launch / average launch duration (HIP events recorded around every
        # launch on the launch stream by the library, isingmc_last_pass_ms)
        (ms_diag, ms_rest), (l_diag, l_rest) = pass_ms
        ms_rvb, l_rvb = g_rvb_ms
        ms_rest -= ms_rvb; l_rest -= l_rvb  # the RVB sweep is a kernel of its own: reported on its own line
        kernels = []
        diag_name = ("sse::sweep_fast_kernel<K,0,false,false> (diagonal pass + directed loop)" if launch_info.get("fast_diagonal") and not (flags & im.FLAG_HEATBATH)
                     else "sse::sweep_kernel<W,K,MODE,0,PASSES=1> (diagonal pass + directed loop)")
        # the cluster launch of the headline geometry is sse::cluster_kernel followed by the general kernel for the replicas it
        # flagged (its workgroups leave at once otherwise: a few microseconds); both are inside the timed interval
        if launch_info.get("lean_cluster"):  # (the library counts the pair as one launch of this pass)
            rest_name = "sse::cluster_kernel<K,HAS_LONG,0> (cluster + free spins + sampling; + sse::sweep_kernel<4,K,1,0,2> for flagged replicas)"
        else:
            rest_name = "sse::sweep_kernel<W,K,MODE,0,PASSES=2> (cluster + free spins + sampling)"
        # the RVB sweep is two launches where that applies (growth of all attempts, then the attempts in order); the library counts
        # and times the pair as one launch of this pass
        info_now = g.launch_info() if args.rvb else {}
        rvb_name = (f"sse::rvb_grow_kernel<CL> + sse::rvb_main_kernel<{info_now.get('rvb_main_waves')},CL> (RVB sweep)" if info_now.get("rvb_split")
                    else "sse::sweep_kernel<16,K,MODE,0,PASSES=3> (RVB sweep, fused kernel)")
        for name, bps, ms, nl in ((diag_name, BYTES_PER_SLOT_DIAG, ms_diag, l_diag),
                                  # RVB sweep: find_constants reads the op-string twice (count + fill); window traffic not counted
                                  (rvb_name, 8.0, ms_rvb, l_rvb),
                                  (rest_name, BYTES_PER_SLOT_CLUSTER, ms_rest, l_rest)):
            if nl == 0:
                continue
            per_launch_bytes = bps * slots / args.steps
            per_launch_ms = ms / nl
            kernels.append({"kernel": name, "bytes_per_slot": bps, "launches": nl, "kernel_ms_per_launch": per_launch_ms,
                            "algorithmic_bytes_per_launch": per_launch_bytes,
                            "achieved_GBps": per_launch_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0})
        dom = max(kernels, key=lambda k: k["kernel_ms_per_launch"] * k["launches"])
        traffic = args.traffic_bytes
        if traffic is None and L == 32 and R == 1024 and beta == 16.0 and not args.rvb and not args.no_loop and not args.pmj3d:
            # HBM bytes per launch of the dominant kernel from the PMC passes of this same workload (cannot be
            # collected inside this process: rocprofv3 counters need their own runs), see profiles/README.md
            try:
                with open(os.path.join(ROOT, "profiles", "r03_traffic.json")) as f:
                    traffic = float(json.load(f)["offdiagonal" if "cluster" in dom["kernel"].split("(")[0] or "PASSES=2" in dom["kernel"] else "diagonal"])
            except (OSError, KeyError, ValueError):
                traffic = None
        achieved = dom["achieved_GBps"]
        bytes_per_sweep = sum(k["algorithmic_bytes_per_launch"] for k in kernels)
        sweep_ms = kernel_ms / args.steps
        nsites = (args.pmj3d ** 3) if args.pmj3d else L * L
        out = {
            "metric": METRIC,
            "value": updates / dt,
            "unit": "spin-op updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 op words, f64 acceptance arithmetic",
            "data": "synthetic (op-strings equilibrated on device from random spins, Philox seed %d)" % args.seed,
            "config": {"workload": (f"configs[4]{'' if L == 32 else ' geometry at reduced size'}: {L}^3 periodic cubic +-J (one disorder realisation per replica) Gamma=1 h=0.1 beta={beta}, "
                                    f"{R} replicas/GPU, QmcIsingGraph::timestep = diagonal + cluster + free spins") if args.pmj3d else
                                   (f"configs[{2 if args.rvb else 1}]: {L}x{L} periodic TFIM J=-1 Gamma=1 h=0 beta={beta}, {R} replicas/GPU, "
                                    f"{'QmcIsingGraph::timestep = diagonal + RVB sweep + ' if args.rvb else 'Qmc::timestep = diagonal + ' + ('directed loop + ' if not args.no_loop else '')}cluster + free spins"),
                       "replicas_per_gpu": R, "lattice": f"{L}^3" if args.pmj3d else f"{L}x{L}", "beta": beta,
                       "mean_cutoff_M": mean_M, "mean_n": mean_n, "sweeps_per_s": args.steps / dt,
                       "updates_diagonal_per_s_rank0": slots / dt,  # U_diag: slots visited by the diagonal pass
                       "updates_offdiagonal_per_s_rank0": float(acc[:, 4].sum()) / dt,  # U_off: cluster + loop vertices
                       "waves_per_replica": launch_info["waves_per_replica"], "waves_per_replica_offdiagonal": launch_info["waves_offdiag"],
                       "slots_per_lane": launch_info["slots_per_lane"],
                       "lds_bytes_per_workgroup": launch_info["lds_bytes"],
                       "tables": "HBM/L2 (per-variable tables exceed LDS)" if launch_info.get("global_tables") else "LDS",
                       "min_replicas_per_gpu_to_fill_the_chip": 1024,
                       "energy_per_site": float(energy.mean() / nsites),
                       "energy_per_site_sem": float(energy.std(ddof=1) / np.sqrt(R) / nsites) if R > 1 else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": dom["kernel"], "kernel_ms_per_launch": dom["kernel_ms_per_launch"], "launches": dom["launches"],
                         "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "bytes_per_slot": dom["bytes_per_slot"],
                         "all_kernels": kernels,
                         "whole_sweep": {"algorithmic_bytes": bytes_per_sweep, "ms": sweep_ms,
                                         "achieved_GBps": bytes_per_sweep / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0}},
        }
        if strong is not None:
            out["strong_scaling"] = strong
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(L, beta, flags, args.seed, budget_s=25.0 if (args.rvb or args.pmj3d) else 20.0,
                                               pmj3d=args.pmj3d, couplings=couplings_cpu if args.pmj3d else None)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
