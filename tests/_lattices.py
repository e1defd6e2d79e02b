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
