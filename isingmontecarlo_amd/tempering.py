"""Parallel tempering over the batch engine.

Host-side mirror of qmc::sse::parallel_tempering::TemperingContainer
(src/sse/parallel_tempering/tempering_container.rs) for `nchains` independent chains ("walkers") of `ntemps`
temperatures that share one Hamiltonian.  Configurations never move: a swap exchanges temperature labels.

Global layout (independent of how the job is sharded over GPUs):
    configuration id  c = t0 * nchains + chain   (t0 = temperature the configuration started at)
    slot id           s = t  * nchains + chain   (t  = temperature index)
Rank g of G owns the configurations with t0 in its contiguous temperature block, i.e. a contiguous id range,
so `replica_offset` makes the device Philox streams identical to the single-GPU run.  One tempering step
all-gathers (n, cutoff) of every configuration (RCCL over xGMI when the backend is nccl: 8 B per replica),
after which every rank evaluates the same swap decisions (counter-based Philox keyed by step and pair) and
relabels its own configurations.  No op-string ever crosses a link.
"""
import ctypes as C

import numpy as np


def _lib():
    from . import load_library
    return load_library()


def pt_decide(seed, step, nchains, ntemps, betas, n_of_config, config_at):
    """TemperingContainer::tempering_step decisions (tempering_container.rs:121-149,241-302) through the C ABI.
    `config_at` (uint32[ntemps*nchains]) is permuted in place; returns the number of swaps."""
    b = np.ascontiguousarray(np.asarray(betas, dtype=np.float64))
    n = np.ascontiguousarray(np.asarray(n_of_config, dtype=np.uint32))
    assert config_at.dtype == np.uint32 and config_at.flags.c_contiguous
    sw = C.c_uint64(0)
    rc = _lib().isingmc_pt_decide(int(seed), int(step), int(nchains), int(ntemps), b.ctypes.data_as(C.POINTER(C.c_double)),
                                  n.ctypes.data_as(C.POINTER(C.c_uint32)),
                                  config_at.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(sw))
    if rc != 0:
        raise RuntimeError("isingmc_pt_decide failed")
    return int(sw.value)


class _Collective:
    """all-gather of small integer arrays: torch.distributed when initialised (nccl = RCCL on ROCm), else local."""

    def __init__(self, device=None):
        self.dist = None
        self.device = device
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                self.dist = dist
        except Exception:  # torch not importable: single process
            self.dist = None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1

    def all_gather_u32(self, local):
        if not self.dist:
            return local.copy()
        import torch
        dev = self.device if self.device is not None else ("cuda" if self.dist.get_backend() == "nccl" else "cpu")
        t = torch.from_numpy(local.astype(np.int64)).to(dev)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return torch.cat(outs).cpu().numpy().astype(np.uint32)

    def all_reduce_sum_u64(self, local):
        if not self.dist:
            return local.copy()
        import torch
        dev = self.device if self.device is not None else ("cuda" if self.dist.get_backend() == "nccl" else "cpu")
        t = torch.from_numpy(local.astype(np.int64)).to(dev)
        self.dist.all_reduce(t)
        return t.cpu().numpy().astype(np.uint64)


class TemperingContainer:
    """qmc::sse::parallel_tempering::TemperingContainer, batch-first.

    `backend` must offer: nreplicas, get_n(), get_cutoff(), set_cutoffs(arr), run(t, betas, sampling_freq, flags),
    set_accumulator_rows(nrows, rows), accumulators(), reset_accumulators(), get_offset(), state_ref(), verify().
    `QmcIsingGraph` is one; the CPU tests plug in an oracle-backed one.
    """

    def __init__(self, backend, betas, nchains, seed, flags=0, collective=None):
        self.b = backend
        self.betas = np.ascontiguousarray(np.asarray(betas, dtype=np.float64))
        self.ntemps, self.nchains = len(self.betas), int(nchains)
        self.seed, self.flags = int(seed), int(flags)
        self.coll = collective if collective is not None else _Collective()
        nconf = self.ntemps * self.nchains
        assert self.ntemps % self.coll.world == 0, "temperatures must divide evenly over the ranks"
        tper = self.ntemps // self.coll.world
        self.c0 = self.coll.rank * tper * self.nchains  # first global configuration id owned by this rank
        assert backend.nreplicas == tper * self.nchains, "backend must hold ntemps/world * nchains replicas"
        self.config_at = np.arange(nconf, dtype=np.uint32)  # slot -> configuration
        self.step = 0
        self.total_swaps = 0
        self._apply_labels()

    # ---- labels ----
    def _slot_of_config(self):
        inv = np.empty_like(self.config_at)
        inv[self.config_at] = np.arange(len(self.config_at), dtype=np.uint32)
        return inv

    def _apply_labels(self):
        slot = self._slot_of_config()[self.c0:self.c0 + self.b.nreplicas]
        self.local_betas = self.betas[slot // self.nchains]
        self.b.set_accumulator_rows(self.ntemps * self.nchains, slot)

    # ---- reference API ----
    def num_graphs(self):
        return self.ntemps * self.nchains

    def get_total_swaps(self):
        return self.total_swaps

    def timesteps(self, t, sampling_freq=1):
        """TemperingContainer::timesteps (:77-81) / parallel_timesteps (:366-371)."""
        self.b.run(int(t), self.local_betas, sampling_freq, self.flags)

    def tempering_step(self):
        """TemperingContainer::tempering_step (:121-149): equalise cutoffs per chain, then the a/b pair sets."""
        if self.ntemps <= 1:
            return 0
        n = self.coll.all_gather_u32(np.asarray(self.b.get_n(), dtype=np.uint32))
        cut = self.coll.all_gather_u32(np.asarray(self.b.get_cutoff(), dtype=np.uint32))
        chain = np.arange(len(n)) % self.nchains
        maxcut = np.zeros(self.nchains, dtype=np.uint32)
        np.maximum.at(maxcut, chain, cut)
        self.b.set_cutoffs(np.ascontiguousarray(maxcut[chain[self.c0:self.c0 + self.b.nreplicas]]))
        swaps = pt_decide(self.seed, self.step, self.nchains, self.ntemps, self.betas, n, self.config_at)
        self.step += 1
        self.total_swaps += swaps
        self._apply_labels()
        return swaps

    def timesteps_sample(self, timesteps, replica_swap_freq, sampling_freq):
        """parallel_timesteps_sample (:411-453): returns (states per slot, energy sums per slot).

        Like the reference (:187-189, :430-434) the second value is the SUM over blocks of E_block * t_block,
        not a mean.  States are sampled every `sampling_freq` steps and reported per temperature slot."""
        nslots = self.num_graphs()
        states = [[] for _ in range(nslots)]
        energy_acc = np.zeros(nslots)
        remaining, to_swap, to_sample = int(timesteps), int(replica_swap_freq), int(sampling_freq)
        while remaining > 0:
            t = min(to_sample, to_swap, remaining)
            self.b.reset_accumulators()
            self.timesteps(t)
            acc = self.coll.all_reduce_sum_u64(self.b.accumulators()).astype(np.float64)
            beta_of_slot = np.repeat(self.betas, self.nchains)
            with np.errstate(divide="ignore", invalid="ignore"):
                te = -(acc[:, 0] / acc[:, 1]) / beta_of_slot + self.b.get_offset()
            energy_acc += np.where(acc[:, 1] > 0, te, 0.0) * t
            to_sample -= t; to_swap -= t; remaining -= t
            if to_swap == 0:
                self.tempering_step()
                to_swap = int(replica_swap_freq)
            if to_sample == 0:
                st = self.b.state_ref()
                slot = self._slot_of_config()[self.c0:self.c0 + self.b.nreplicas]
                for i, s in enumerate(slot):
                    states[int(s)].append(st[i].copy())
                to_sample = int(sampling_freq)
        return states, energy_acc

    def slot_accumulators(self):
        """Global [ntemps*nchains][8] accumulator table (summed over ranks)."""
        return self.coll.all_reduce_sum_u64(self.b.accumulators())

    def verify(self):
        return bool(np.all(self.b.verify()))


class NativeTemperingContainer:
    """TemperingContainer (tempering_container.rs) over the library's native step, isingmc_pt_step: temperature blocks
    sharded over ranks, label swaps inside a block, point-to-point exchange of the boundary walkers between neighbouring
    ranks (RCCL ncclSend / ncclRecv after attach_rccl(); otherwise torch.distributed send / recv staged through host memory,
    which is what the tests use for two ranks on one GPU).  `graph` is this rank's QmcIsingGraph holding
    len(betas) / world * nchains replicas, built with replica_offset = rank * nreplicas.  With per-replica couplings the
    rows of `couplings` belong to the temperature slots and swaps weigh the two Hamiltonians (tempering_traits.rs:126-155)."""

    def __init__(self, graph, betas, nchains, seed, flags=0):
        from . import _PtLayout, _PtTransport, _PT_SENDRECV, _PT_ALLMAX
        self.g = graph
        self.betas = np.ascontiguousarray(np.asarray(betas, dtype=np.float64))
        self.ntemps, self.nchains, self.flags = len(self.betas), int(nchains), int(flags)
        self.dist = None
        try:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                self.dist = dist
        except Exception:
            self.dist = None
        self.rank = self.dist.get_rank() if self.dist else 0
        self.world = self.dist.get_world_size() if self.dist else 1
        self._cb = (_PT_SENDRECV(self._sendrecv), _PT_ALLMAX(self._allmax))  # keep the callbacks alive
        self._tr = _PtTransport(ctx=None, sendrecv=self._cb[0], allreduce_max_u32=self._cb[1])
        lay = _PtLayout(struct_size=C.sizeof(_PtLayout), ntemps=self.ntemps, nchains=self.nchains, rank=self.rank, world=self.world,
                        betas=self.betas.ctypes.data_as(C.POINTER(C.c_double)), seed=int(seed),
                        transport=C.pointer(self._tr) if self.world > 1 else None)
        graph._check(graph._lib.isingmc_pt_create(graph._h, C.byref(lay)))
        self.total_swaps_local = 0
        self._stale = True
        self._refresh()

    # Labels.  With device-side decisions (single rank, one Hamiltonian: the library's default there) the labels live in device memory;
    # these mirrors are refreshed when somebody looks at them, not after every step.
    @property
    def device_decisions(self):
        on = C.c_int(0)
        self.g._check(self.g._lib.isingmc_pt_get_device_decisions(self.g._h, C.byref(on)))
        return bool(on.value)

    def set_device_decisions(self, on):
        self.g._check(self.g._lib.isingmc_pt_set_device_decisions(self.g._h, 1 if on else 0))
        self._stale = True

    slot_of = property(lambda self: self._mirror("_slot_of"))
    local_betas = property(lambda self: self._mirror("_local_betas"))
    config_of = property(lambda self: self._mirror("_config_of"))

    def _mirror(self, name):
        if self._stale:
            self._refresh()
        return getattr(self, name)

    # ---- host-staged transport over torch.distributed (CPU tensors with gloo, device tensors with nccl) ----
    def _tensor_dev(self):
        return "cuda" if self.dist.get_backend() == "nccl" else "cpu"

    def _sendrecv(self, ctx, peer, sbuf, sbytes, rbuf, rbytes):
        try:
            import torch
            ts = torch.from_numpy(np.ctypeslib.as_array(C.cast(sbuf, C.POINTER(C.c_uint8)), shape=(sbytes,)).copy()).to(self._tensor_dev())
            tr = torch.empty(rbytes, dtype=torch.uint8, device=self._tensor_dev())
            ops = [self.dist.P2POp(self.dist.isend, ts, peer), self.dist.P2POp(self.dist.irecv, tr, peer)]
            if self.rank > peer:
                ops.reverse()
            for req in self.dist.batch_isend_irecv(ops):
                req.wait()
            C.memmove(rbuf, tr.cpu().numpy().ctypes.data, rbytes)
            return 0
        except Exception as e:  # never unwind through the C caller
            print("tempering transport:", e)
            return 1

    def _allmax(self, ctx, buf, count):
        try:
            import torch
            a = np.ctypeslib.as_array(buf, shape=(count,))
            t = torch.from_numpy(a.astype(np.int64)).to(self._tensor_dev())
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            a[:] = t.cpu().numpy().astype(np.uint32)
            return 0
        except Exception as e:
            print("tempering transport:", e)
            return 1

    def attach_rccl(self):
        """Create the library's own RCCL communicator (ncclCommInitRank; the id travels through torch.distributed) so that
        the neighbour exchange runs as ncclSend / ncclRecv on device buffers.  Needs one rank per GPU."""
        from . import _NcclId
        import torch
        nid = _NcclId()
        if self.rank == 0:
            self.g._check(self.g._lib.isingmc_pt_nccl_unique_id(C.byref(nid)))
        t = torch.frombuffer(bytearray(C.string_at(C.byref(nid), 128)), dtype=torch.uint8).clone().to(self._tensor_dev() if self.dist else "cpu")
        if self.dist:
            self.dist.broadcast(t, src=0)
        C.memmove(C.byref(nid), t.cpu().numpy().ctypes.data, 128)
        self.g._check(self.g._lib.isingmc_pt_attach_nccl(self.g._h, C.byref(nid)))

    # ---- labels ----
    def _refresh(self):
        R = self.g.nreplicas
        self._slot_of = np.zeros(R, dtype=np.uint32)
        self._local_betas = np.zeros(R, dtype=np.float64)
        self._config_of = np.zeros(R, dtype=np.uint32)
        self.g._check(self.g._lib.isingmc_pt_get_slots(self.g._h, self._slot_of.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                         self._local_betas.ctypes.data_as(C.POINTER(C.c_double)),
                                                         self._config_of.ctypes.data_as(C.POINTER(C.c_uint32))))
        self._stale = False
        # accumulators per temperature slot (the decision kernel keeps the rows up to date afterwards when it runs on the device)
        self.g.set_accumulator_rows(self.ntemps * self.nchains, self._slot_of)

    # ---- reference API ----
    def num_graphs(self):
        return self.ntemps * self.nchains

    def timesteps(self, t, sampling_freq=1):
        if int(t) > 0:
            self.g._check(self.g._lib.isingmc_pt_timesteps(self.g._h, int(t), int(sampling_freq), int(self.flags)))

    def tempering_step(self, count_swaps=True):
        """One tempering step; returns the swaps this rank counted (None with count_swaps=False, which spares the device-side path
        its only read-back: get_total_swaps() still has the total)."""
        if self.device_decisions:
            if count_swaps:
                sw = C.c_uint64(0)
                self.g._check(self.g._lib.isingmc_pt_step(self.g._h, C.byref(sw)))
            else:
                sw = None
                self.g._check(self.g._lib.isingmc_pt_step(self.g._h, None))
            self._stale = True
            return None if sw is None else int(sw.value)
        sw = C.c_uint64(0)
        self.g._check(self.g._lib.isingmc_pt_step(self.g._h, C.byref(sw)))
        self.total_swaps_local += int(sw.value)
        self._refresh()
        return int(sw.value)

    def get_total_swaps(self):
        if self.device_decisions:
            step, sw = C.c_uint64(0), C.c_uint64(0)
            self.g._check(self.g._lib.isingmc_pt_get_state(self.g._h, C.byref(step), C.byref(sw)))
            return int(sw.value)
        if not self.dist:
            return self.total_swaps_local
        import torch
        t = torch.tensor([self.total_swaps_local], dtype=torch.int64, device=self._tensor_dev())
        self.dist.all_reduce(t)
        return int(t.item())

    def verify(self):
        return bool(np.all(self.g.verify()))

    # ---- container-level save / load (tempering_container.rs:683-792 serialises the whole container) ----
    def save(self, path):
        """This rank's share: the batch checkpoint (`path`.batch.npz) plus labels, configuration identities and step counter."""
        self.g.save_checkpoint(path + ".batch.npz")
        step, sw = C.c_uint64(0), C.c_uint64(0)
        self.g._check(self.g._lib.isingmc_pt_get_state(self.g._h, C.byref(step), C.byref(sw)))
        np.savez(path + ".pt.npz", slot_of=self.slot_of, config_of=self.config_of, step=np.uint64(step.value), swaps=np.uint64(sw.value),
                 betas=self.betas, nchains=np.uint32(self.nchains), rank=np.uint32(self.rank), world=np.uint32(self.world),
                 swaps_local=np.uint64(self.total_swaps_local))

    def load(self, path):
        z = np.load(path + ".pt.npz", allow_pickle=False)
        if not np.array_equal(z["betas"], self.betas) or int(z["nchains"]) != self.nchains or int(z["rank"]) != self.rank or int(z["world"]) != self.world:
            raise RuntimeError("tempering checkpoint belongs to a different layout")
        self.g.load_checkpoint(path + ".batch.npz")
        so = np.ascontiguousarray(z["slot_of"].astype(np.uint32)); co = np.ascontiguousarray(z["config_of"].astype(np.uint32))
        self.g._check(self.g._lib.isingmc_pt_set_state(self.g._h, so.ctypes.data_as(C.POINTER(C.c_uint32)), co.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                        int(z["step"]), int(z["swaps"])))
        self.total_swaps_local = int(z["swaps_local"])
        self._stale = True
        self._refresh()
