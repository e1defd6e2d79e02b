"""Checkpoint export / import in the reference's serde field layout (feature "serialize").

`to_serde(graph, r)` returns, for replica r, a dict with exactly the fields and nesting that serde derives for
`SerializeQmcGraph<FastOps>` (src/sse/qmc_ising.rs:1010-1028) and its members `FastOpsTemplate` (fast_ops.rs:35-49),
`FastOpNodeTemplate` (:181-190), `BasicOp` (qmc_traits/op_container.rs:224-237), `OpType` (:165-173, externally tagged),
`PRel` (directed_loop.rs:12-17), `BondWeights` (heatbath.rs:10-12): `json.dumps` of it is what `serde_json::to_string` of the
reference struct would hold for the same configuration.  The engine stores no links: `previous_p` / `next_p` /
`previous_for_vars` / `next_for_vars` / `p_ends` / `var_ends` / `bond_counters` are rebuilt here from the exported op words.
`from_serde` installs such a dict into a replica (ops, p=0 state, cutoff), ignoring the link fields (FastOps::new_from_ops does
the same, fast_ops.rs:80-174).

The op manager's allocator is the default `DefaultFastOpAllocator` (FastOps = FastOpsTemplate<FastOp>, fast_ops.rs:18,35-49): nine
`Allocator<..>` members (fast_op_alloc.rs:29-39), each serialised as {instances: <length, u64: util/allocator.rs:31-38,142-162>,
gen_more}; a graph at rest has every pooled instance returned, so the lengths are the `max_in_flight` values of its `Default`
(fast_op_alloc.rs:43-56).

Status: FIELD NAMES AND NESTING of SerializeQmcGraph as read off the struct definitions (tests/test_abi_cpu.py walks a dict against a
schema written from those definitions); no file produced by the Rust crate exists in this project to compare with (no Rust toolchain,
SURVEY.md §8c) — byte-level compatibility with serde_json output is therefore unpinned.
"""
import numpy as np

from . import op_fields, op_make


def _bond_vars(graph, bond):
    e = graph.edges
    ne = len(e)
    if bond < ne:
        return [int(e[bond, 0]), int(e[bond, 1])]
    return [int((bond - ne) % graph.nvars)]


# DefaultFastOpAllocator::default() (fast_op_alloc.rs:43-56): pooled instances per member
DEFAULT_ALLOCATOR_POOLS = (("usize_alloc", 10), ("bool_alloc", 2), ("opside_alloc", 1), ("leg_alloc", 1), ("option_usize_alloc", 4),
                           ("f64_alloc", 1), ("bond_container_alloc", 2), ("bond_container_varpos_alloc", 2), ("binary_heap_alloc", 1))


def classical_bonds(graph):
    """make_classical_bonds (qmc_ising.rs:421-432): for every variable the bonds (edge numbers) it belongs to, in edge order."""
    lookup = [[] for _ in range(graph.nvars)]
    for bond, (a, b) in enumerate(graph.edges):
        lookup[int(a)].append(bond)
        lookup[int(b)].append(bond)
    return lookup


def to_serde(graph, r=0, total_rvb_successes=0, rvb_clusters_counted=0):
    words = graph.export_ops(r)
    state = graph.state_ref()[r]
    nvars, ne = graph.nvars, len(graph.edges)
    has_long = abs(float(graph.longitudinal_r[r]) if getattr(graph, "longitudinal_r", None) is not None else graph.longitudinal) > np.finfo(float).eps
    nbonds = ne + nvars + (nvars if has_long else 0)
    nodes = [None] * len(words)
    last_p, first_p = None, None
    var_first, var_last = [None] * nvars, [None] * nvars
    counters = [0] * nbonds
    for p, w in enumerate(words):
        f = op_fields(int(w))
        if f is None:
            continue
        bond, ib, ob = f
        vs = _bond_vars(graph, bond)
        k = len(vs)
        ins = [bool((ib >> j) & 1) for j in range(k)]
        outs = [bool((ob >> j) & 1) for j in range(k)]
        in_out = {"Diagonal": ins} if ins == outs else {"Offdiagonal": [ins, outs]}
        node = {"op": {"vars": vs, "bond": bond, "in_out": in_out, "constant": ne <= bond < ne + nvars},
                "previous_p": last_p, "next_p": None, "previous_for_vars": [None] * k, "next_for_vars": [None] * k}
        if last_p is not None:
            nodes[last_p]["next_p"] = p
        else:
            first_p = p
        for j, v in enumerate(vs):
            if var_last[v] is not None:
                pp, pj = var_last[v]
                node["previous_for_vars"][j] = {"p": pp, "relv": pj}
                nodes[pp]["next_for_vars"][pj] = {"p": p, "relv": j}
            else:
                var_first[v] = (p, j)
            var_last[v] = (p, j)
        nodes[p] = node
        last_p = p
        counters[bond] += 1
    var_ends = [None if var_first[v] is None else [{"p": var_first[v][0], "relv": var_first[v][1]}, {"p": var_last[v][0], "relv": var_last[v][1]}]
                for v in range(nvars)]
    ops_n = int(np.count_nonzero(words))
    manager = {"ops": nodes, "n": ops_n, "p_ends": None if first_p is None else [first_p, last_p], "var_ends": var_ends,
               "bond_counters": counters,
               "alloc": {name: {"instances": n, "gen_more": False} for name, n in DEFAULT_ALLOCATOR_POOLS}}
    J = np.asarray(graph.J)
    Jr = J[r] if J.ndim == 2 else J
    edges = [[[int(a), int(b)], float(j)] for (a, b), j in zip(graph.edges, Jr)]
    offset = float(graph.get_offsets()[r])
    run_rvb = bool(graph._flags & 8)
    gam = float(graph.transverse_r[r]) if getattr(graph, "transverse_r", None) is not None else graph.transverse
    hl = float(graph.longitudinal_r[r]) if getattr(graph, "longitudinal_r", None) is not None else graph.longitudinal
    # (set_run_rvb builds classical_bonds, qmc_ising.rs:435-447, and timestep unwraps it when run_rvb_steps is set, :705-708)
    return {"edges": edges, "transverse": gam, "longitudinal": hl, "state": [bool(s) for s in state],
            "cutoff": int(len(words)), "op_manager": manager, "total_energy_offset": offset, "nvars": nvars,
            "run_rvb_steps": run_rvb, "classical_bonds": classical_bonds(graph) if run_rvb else None,
            "total_rvb_successes": int(total_rvb_successes), "rvb_clusters_counted": int(rvb_clusters_counted),
            "bond_weights": None}


def from_serde(graph, d, r=0):
    """Install the configuration of a SerializeQmcGraph dict into replica r (same model required)."""
    if int(d["nvars"]) != graph.nvars or len(d["edges"]) != len(graph.edges):
        raise ValueError("serialized graph belongs to a different model")
    J = np.asarray(graph.J)
    Jr = J[r] if J.ndim == 2 else J
    for (vs, j), (a, b), jg in zip(d["edges"], graph.edges, Jr):
        if [int(a), int(b)] != [int(x) for x in vs]:
            raise ValueError("serialized graph has different edges")
        if float(j) != float(jg):
            raise ValueError("serialized graph has different couplings")
    gam = float(graph.transverse_r[r]) if getattr(graph, "transverse_r", None) is not None else graph.transverse
    hl = float(graph.longitudinal_r[r]) if getattr(graph, "longitudinal_r", None) is not None else graph.longitudinal
    if float(d["transverse"]) != gam or float(d["longitudinal"]) != hl:
        raise ValueError("serialized graph has different fields")
    ops = d["op_manager"]["ops"] if d.get("op_manager") else []
    words = np.zeros(max(int(d["cutoff"]), len(ops)), dtype=np.uint32)
    for p, node in enumerate(ops):
        if node is None:
            continue
        op = node["op"]
        io = op["in_out"]
        ins, outs = (io["Diagonal"], io["Diagonal"]) if "Diagonal" in io else io["Offdiagonal"]
        ib = sum(int(bool(x)) << j for j, x in enumerate(ins))
        ob = sum(int(bool(x)) << j for j, x in enumerate(outs))
        words[p] = op_make(int(op["bond"]), ib, ob)
    if d.get("state") is not None:
        graph.set_state(np.array([1 if s else 0 for s in d["state"]], dtype=np.uint8), r)
    graph.import_ops(words, r)
    graph.set_cutoff(int(d["cutoff"]), r)
