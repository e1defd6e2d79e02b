"""Lattice builders with the reference's edge ordering (benches/end_to_end.rs:8-30)."""


def one_d_periodic(l, j=1.0):
    return [((i, (i + 1) % l), j) for i in range(l)]


def two_d_periodic(l, jfun=None):
    """Site (i,j) -> j*l+i; all right bonds first, then all down bonds; default = the benches' Villain pattern."""
    idx = [(i, j) for i in range(l) for j in range(l)]
    f = lambda i, j: j * l + i
    if jfun is None:
        jfun = lambda i, j, d: -1.0 if d == 0 else (1.0 if i % 2 == 0 else -1.0)
    right = [((f(i, j), f((i + 1) % l, j)), jfun(i, j, 0)) for i, j in idx]
    down = [((f(i, j), f(i, (j + 1) % l)), jfun(i, j, 1)) for i, j in idx]
    return right + down


def two_d_ferro(l):
    return two_d_periodic(l, lambda i, j, d: -1.0)


def split(edges):
    return [list(e) for e, _ in edges], [j for _, j in edges]


def cubic_periodic(l):
    """Periodic simple-cubic lattice, site (x,y,z) -> (z*l + y)*l + x, edges +x, +y, +z per site (J filled in by the caller)."""
    edges = []
    for z in range(l):
        for y in range(l):
            for x in range(l):
                a = (z * l + y) * l + x
                edges.append(((a, (z * l + y) * l + (x + 1) % l), 1.0))
                edges.append(((a, (z * l + (y + 1) % l) * l + x), 1.0))
                edges.append(((a, (((z + 1) % l) * l + y) * l + x), 1.0))
    return edges


def xxz_ring_interactions(n, t=1.0, jz=0.5, c=1.0, hx=0.3):
    """Generic-interaction test model on a ring of n sites (weights = matrix elements of -H_b + const, all >= 0):
    two-site bonds with hopping t (|01> <-> |10>), Ising part jz (favouring aligned spins) and constant c on the diagonal;
    one-site bonds hx*(sigma_x + 1).  Matrices in the reference's layout (index = out0 out1 in0 in1, qmc_runner.rs:666-679).
    Returns [(mat, vars), ...]."""
    import numpy as np
    ints = []
    for i in range(n):
        m = np.zeros(16)
        for s in range(4):  # s = (s0 s1), first variable most significant
            aligned = ((s >> 1) & 1) == (s & 1)
            m[(s << 2) | s] = c + (jz if aligned else 0.0)
        m[(0b01 << 2) | 0b10] = t
        m[(0b10 << 2) | 0b01] = t
        ints.append((m, (i, (i + 1) % n)))
    for i in range(n):
        ints.append((np.full(4, hx), (i,)))
    return ints


def exact_energy_from_interactions(n, ints, beta):
    """<H> at inverse temperature beta for H = -sum_b M_b (dense, n <= 10)."""
    import numpy as np
    dim = 1 << n
    H = np.zeros((dim, dim))
    for mat, vs in ints:
        k = len(vs)
        for s_in in range(dim):
            bits_in = [(s_in >> (n - 1 - v)) & 1 for v in vs]
            iin = 0
            for bbit in bits_in:
                iin = (iin << 1) | bbit
            for iout in range(1 << k):
                w = mat[(iout << k) | iin]
                if w == 0.0:
                    continue
                s_out = s_in
                for j, v in enumerate(vs):
                    ob = (iout >> (k - 1 - j)) & 1
                    s_out = (s_out & ~(1 << (n - 1 - v))) | (ob << (n - 1 - v))
                H[s_out, s_in] -= w
    ev = np.linalg.eigvalsh(H)
    wts = np.exp(-beta * (ev - ev.min()))
    return float((ev * wts).sum() / wts.sum())
