"""Path B growth: the oracle against the exact law of the reference's recursive walk.

MultiviewSegmentation::getNewConnectedSet (src/libraries/MultiviewSegmentation.cpp:207-257) is transcribed
below as a probability tree (every randbetween(0,1) <= q becomes a branch).  The build ties the trial of an
undirected edge to the edge instead of to the order of the walk (DESIGN.md, K4 growth) and the kernel grows
breadth first; both rest on the claim that the grown set has the same distribution either way.  This test
checks that claim on a 5-vertex graph: 60 000 oracle chains take one split step (split probability 1, so the
grown set becomes a new segment and can be read back), grouped by their initial partition, and the empirical
distribution of the grown set is compared with the walk's exact law.  The GPU kernel is bit-exact against the
oracle (tests/test_multiview.py), so it inherits the result.
"""
from collections import defaultdict
from types import SimpleNamespace

import numpy as np

from tests import oracle_lib as O

EDGES = {(0, 1): 0.7, (0, 2): 0.45, (1, 2): 0.6, (1, 3): 0.8, (2, 3): 0.3, (3, 4): 0.55}
V = 5
N = 60000


def _graph():
    adj = [[] for _ in range(V)]
    for (a, b), q in EDGES.items():
        adj[a].append((b, q)); adj[b].append((a, q))
    for l in adj:
        l.sort()
    row_ptr = np.zeros(V + 1, np.uint32); col = []; q = []
    for v in range(V):
        row_ptr[v + 1] = row_ptr[v] + len(adj[v])
        col += [j for j, _ in adj[v]]; q += [p for _, p in adj[v]]
    rng = np.random.default_rng(5)
    pts = []; pt_ptr = [0]
    for v in range(V):
        xy = rng.uniform(-1, 1, (12, 2)) + [2.5 * v, 0]
        z = 2.0 + 0.1 * xy[:, 0] - 0.05 * xy[:, 1] + rng.normal(0, 0.01, 12)
        pts.append(np.column_stack([xy, z, np.ones(12)])); pt_ptr.append(pt_ptr[-1] + 12)
    mv = SimpleNamespace(n_vertices=V, points=np.concatenate(pts).astype(np.float32), pt_ptr=np.array(pt_ptr, np.uint32),
                         row_ptr=row_ptr, col=np.array(col, np.uint32), q=np.array(q))
    return adj, mv


def walk_law(adj, labels, start):
    """exact distribution of `elements` after getNewConnectedSet(start) (depth-first, one trial per visit)"""
    out = defaultdict(float)

    def rec(p, elements, stack):
        while stack:
            v, pos = stack[-1]
            if pos >= len(adj[v]):
                stack = stack[:-1]; continue
            j, q = adj[v][pos]
            stack = stack[:-1] + ((v, pos + 1),)
            if labels[j] != labels[v]:            # :224-227
                continue
            if j in elements:                     # :230-235
                continue
            rec(p * q, elements | {j}, stack + ((j, 0),))       # :239-247 trial succeeds, recurse
            p *= 1.0 - q                                        # fails: next out-edge
        out[elements] += p

    rec(1.0, frozenset([start]), ((start, 0),))
    return out


def test_grown_set_has_the_law_of_the_recursive_walk():
    adj, mv = _graph()
    ch = O.MvChains(mv, N, seed=9, max_new_segments=4, split_prob=1.0)
    cur = ch.labels(0)
    flags, _ = ch.run(1)
    own = ch.labels(1)
    assert ((flags[:, 0] & 2) != 0).all()         # every step was a split
    groups = defaultdict(lambda: defaultdict(int))
    for c, o in zip(cur, own):
        key = tuple(int(x) for x in c)
        groups[key][frozenset(np.nonzero(c != o)[0].tolist())] += 1
    checked = 0
    for labels, hist in groups.items():
        n = sum(hist.values())
        if n < 3000:
            continue
        law = defaultdict(float)
        for s in range(V):                        # uniform start vertex
            for e, p in walk_law(adj, labels, s).items():
                law[e] += p / V
        assert abs(sum(law.values()) - 1.0) < 1e-12
        assert set(hist) <= set(law)
        for e, p in law.items():
            tol = 5.0 * np.sqrt(p * (1 - p) / n) + 1e-3
            assert abs(hist.get(e, 0) / n - p) <= tol, (labels, sorted(e), hist.get(e, 0) / n, p)
        checked += 1
    assert checked >= 3
