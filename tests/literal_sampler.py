"""Exact one-step transition law of the reference's sampler on a small graph (test infrastructure).

An independent Python transcription of the reference's proposal and accept code, written as a probability
tree instead of a random walk: every `rand()` of the reference becomes a branch with its probability, so
one call returns the exact distribution of the next state.  It shares no code with oracle/mcmc_oracle.c;
tests/test_sampler_law.py checks the oracle's empirical k-step distributions against powers of this matrix.

Transcribed from (reference file:line):
  Segmentation::pickSuperPixel   src/libraries/Segmentation.cpp:151-183  uniform start, one discarded draw
  Segmentation::createComponent  src/libraries/Segmentation.cpp:59-149   FIFO growth, separation product
  Segmentation::selectSegment    src/libraries/Segmentation.cpp:10-57    new segment / neighbour / original
  Segment::mergeProbability      src/libraries/Graph.cpp:10-33
  Problem::propose               src/libraries/Problem.cpp:116-224       cases A, B, C
  Problem::jumpProbability       src/libraries/Problem.cpp:475-487       log target ratio
  FastMetropolisHastings::iterate include/MCMC.h:207-238                 accept rule

A state is a tuple of labels in canonical form (relabelled by first appearance, Segmentation.cpp:197-237).
The draws are treated as continuous uniforms; the 32-bit draws of the build differ from that by < 2^-31 per
comparison.
"""
import math
from collections import defaultdict

NEW_SEGMENT_PROB = 0.8      # Segmentation.cpp:16


def canonical(labels):
    m = {}
    return tuple(m.setdefault(x, len(m)) for x in labels)


def partitions(n):
    """all canonical label tuples of n items"""
    out = []

    def rec(prefix, k):
        if len(prefix) == n:
            out.append(tuple(prefix)); return
        for x in range(k + 1):
            rec(prefix + [x], max(k, x + 1))
    rec([], 0)
    return out


def _grow(nbrs, labels, start):
    """createComponent's FIFO loop as a tree: yields (probability, component frozenset, neighbour segments)."""
    seg0 = labels[start]
    results = []

    def expand(p, comp, queue, node, pos, nsegs):
        # continue the neighbour loop of `node` at position `pos` (map order = ascending id)
        while True:
            if node is None:
                if not queue:
                    results.append((p, frozenset(comp), frozenset(nsegs))); return
                node, queue, pos = queue[0], queue[1:], 0
            lst = nbrs[node]
            if pos >= len(lst):
                node = None; continue
            j, q = lst[pos]
            pos += 1
            if j in comp:                                   # :94-97
                continue
            if labels[j] != seg0:                           # :100-105
                nsegs = nsegs | {labels[j]}
                continue
            # :108-117 one draw, joins iff r < q
            if q > 0.0:
                expand(p * q, comp | {j}, queue + (j,), node, pos, nsegs)
            if q < 1.0:
                p = p * (1.0 - q)
                continue
            return

    expand(1.0, frozenset([start]), (start,), None, 0, frozenset())
    return results


def step_law(labels, nbrs, probL):
    """exact distribution of the state after one MCMC::iterate; nbrs[i] = [(j, q_ij)...] ascending j,
    probL(frozenset of superpixels) = Segment::probabilityL_."""
    S = len(labels)
    out = defaultdict(float)
    members = defaultdict(set)
    for i, l in enumerate(labels):
        members[l].add(i)
    for start in range(S):                                   # pickSuperPixel: rand() % S
        seg0 = labels[start]
        orig = frozenset(members[seg0])
        for p, comp, nsegs in _grow(nbrs, labels, start):
            p = p / S
            # separation product (:127-146): ordered pairs (i in comp, neighbour in orig but not comp)
            sep = 1.0
            for i in sorted(comp):
                for j, q in nbrs[i]:
                    if labels[j] == seg0 and j not in comp:
                        sep *= (1.0 - q)
            same = len(comp) == len(orig)
            choices = []                                    # (probability, kind, next segment label, ratio)
            rest = 1.0
            if not same:
                choices.append((NEW_SEGMENT_PROB, "new", None, NEW_SEGMENT_PROB))
                rest = 1.0 - NEW_SEGMENT_PROB
            n = len(nsegs) + 1
            ratio = (1.0 if same else 1.0 - NEW_SEGMENT_PROB) / n
            choices.append((rest / n, "orig", seg0, ratio))
            for sg in sorted(nsegs):
                choices.append((rest / n, "other", sg, ratio))
            for pc, kind, sg, ratio in choices:
                pp = p * pc
                if kind == "orig":                          # case A: same state returned, always "accepted"
                    out[canonical(labels)] += pp
                    continue
                if kind == "new":                           # case B
                    target = probL(comp) - probL(orig) + probL(orig - comp)
                    prop = ratio * (1.0 / sep)
                    new_labels = list(labels)
                    fresh = max(labels) + 1
                    for i in comp:
                        new_labels[i] = fresh
                else:                                       # case C
                    nxt = frozenset(members[sg])
                    merge = 1.0
                    for i in sorted(comp):
                        for j, q in nbrs[i]:
                            if labels[j] == sg:
                                merge *= (1.0 - q)
                    target = probL(nxt | comp) - probL(orig) + ((0.0 if same else probL(orig - comp)) - probL(nxt))
                    prop = ratio * (merge / sep)
                    new_labels = list(labels)
                    for i in comp:
                        new_labels[i] = sg
                aL = target + math.log(prop)
                acc = 1.0 if aL >= 0.0 else math.exp(aL)    # MCMC.h:218-227: reject iff aL < 0 and r > a
                out[canonical(new_labels)] += pp * acc
                out[canonical(labels)] += pp * (1.0 - acc)
    return dict(out)


def transition_matrix(n, nbrs, probL):
    import numpy as np
    states = partitions(n)
    index = {s: k for k, s in enumerate(states)}
    P = np.zeros((len(states), len(states)))
    for s in states:
        for t, p in step_law(s, nbrs, probL).items():
            P[index[s], index[t]] += p
    return states, index, P
