#!/usr/bin/env python3
"""Exact-diagonalisation golden values for small transverse-field Ising systems.

The reference's own tests pin no numbers, and the Rust crate cannot be built here,
so the oracle (and through it the HIP path) is pinned against exact results computed with numpy:

    H = sum_ij J_ij sz_i sz_j - Gamma sum_i sx_i - h sum_i sz_i        (src/lib.rs:29 of the reference)

For every case we store E = <H>, <|m|>, <m^2> with m = (1/N) sum_i sz_i measured in the sz basis
(what the SSE p=0 state samples) and <sx> = (1/N) sum_i <sx_i>.

Run:  python tests/golden/make_ed_golden.py   (writes tests/golden/ed_tfim.json)
"""
import json
import os

import numpy as np


def ring(l, j):
    return [(i, (i + 1) % l, j) for i in range(l)]


def lattice(lx, ly, jfun):
    f = lambda i, j: j * lx + i
    edges = []
    for i in range(lx):
        for j in range(ly):
            if lx > 2 or i == 0:
                edges.append((f(i, j), f((i + 1) % lx, j), jfun(i, j, 0)))
    for i in range(lx):
        for j in range(ly):
            if ly > 2 or j == 0:
                edges.append((f(i, j), f(i, (j + 1) % ly), jfun(i, j, 1)))
    return edges


def villain(i, j, d):
    # benches/end_to_end.rs:12-30: right bonds -1, down bonds +1 on even columns else -1
    return -1.0 if d == 0 else (1.0 if i % 2 == 0 else -1.0)


CASES = [
    dict(name="small_qmc_ring4", n=4, edges=[(0, 1, -1.0), (1, 2, 1.0), (2, 3, 1.0), (3, 0, 1.0)], gamma=1.0, h=0.0,
         betas=[1.0, 4.0]),
    dict(name="ring3_afm", n=3, edges=ring(3, 1.0), gamma=1.0, h=0.0, betas=[1.0]),
    dict(name="ring8_afm", n=8, edges=ring(8, 1.0), gamma=1.0, h=0.0, betas=[1.0, 4.0]),
    dict(name="ring8_fm", n=8, edges=ring(8, -1.0), gamma=1.0, h=0.0, betas=[1.0, 4.0]),
    dict(name="ring10_fm_g05", n=10, edges=ring(10, -1.0), gamma=0.5, h=0.0, betas=[2.0]),
    dict(name="lat3x3_villain", n=9, edges=lattice(3, 3, villain), gamma=1.0, h=0.0, betas=[1.0, 4.0]),
    dict(name="lat3x3_fm", n=9, edges=lattice(3, 3, lambda i, j, d: -1.0), gamma=1.0, h=0.0, betas=[1.0, 4.0]),
    dict(name="lat4x3_fm", n=12, edges=lattice(4, 3, lambda i, j, d: -1.0), gamma=1.5, h=0.0, betas=[2.0]),
    dict(name="ring6_fm_long", n=6, edges=ring(6, -1.0), gamma=1.0, h=0.3, betas=[1.0, 2.0]),
    dict(name="ring6_afm_neglong", n=6, edges=ring(6, 1.0), gamma=0.8, h=-0.5, betas=[1.5]),
    dict(name="single_bond", n=2, edges=[(0, 1, 1.0)], gamma=1.0, h=0.0, betas=[1.0]),
    dict(name="ring5_randmag", n=5, edges=[(0, 1, 0.7), (1, 2, -1.3), (2, 3, 1.9), (3, 4, -0.6), (4, 0, 1.1)],
         gamma=1.2, h=0.0, betas=[1.0, 3.0]),
]


def hamiltonian(n, edges, gamma, h):
    dim = 1 << n
    idx = np.arange(dim)
    sz = np.array([1.0 - 2.0 * ((idx >> v) & 1) for v in range(n)])  # bit 0 -> +1 ... sign irrelevant by symmetry
    # convention of the reference: state bool true = up.  Use bit=1 <-> up (+1):
    sz = -sz
    diag = np.zeros(dim)
    for a, b, j in edges:
        diag += j * sz[a] * sz[b]
    diag -= h * sz.sum(axis=0)
    hmat = np.diag(diag)
    for v in range(n):
        hmat[idx, idx ^ (1 << v)] -= gamma
    return hmat, sz


def thermal(n, edges, gamma, h, beta):
    hmat, sz = hamiltonian(n, edges, gamma, h)
    evals, evecs = np.linalg.eigh(hmat)
    wts = np.exp(-beta * (evals - evals.min()))
    z = wts.sum()
    energy = float((wts * evals).sum() / z)
    rho_diag = (evecs ** 2 * wts[None, :]).sum(axis=1) / z
    m = sz.sum(axis=0) / n
    dim = 1 << n
    idx = np.arange(dim)
    sx = 0.0
    rho = (evecs * wts[None, :]) @ evecs.T / z
    for v in range(n):
        sx += rho[idx, idx ^ (1 << v)].sum()
    return dict(beta=beta, energy=energy, abs_m=float((rho_diag * np.abs(m)).sum()),
                m2=float((rho_diag * m * m).sum()), m=float((rho_diag * m).sum()), sx=float(sx / n))


def main():
    out = []
    for c in CASES:
        rec = dict(name=c["name"], nvars=c["n"], edges=[[a, b] for a, b, _ in c["edges"]],
                   J=[j for _, _, j in c["edges"]], gamma=c["gamma"], h=c["h"], results=[])
        for beta in c["betas"]:
            rec["results"].append(thermal(c["n"], c["edges"], c["gamma"], c["h"], beta))
        out.append(rec)
        print(rec["name"], [(r["beta"], round(r["energy"], 6), round(r["abs_m"], 4), round(r["sx"], 4)) for r in rec["results"]])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ed_tfim.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
