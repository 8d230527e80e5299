"""CPU tests of the multi-GPU layer with 2 gloo ranks.

The data path has no collective: a rank that owns chains [a, b) of a job reproduces
them bit for bit (Philox key = global chain id).  What is exchanged afterwards is
the best-energy labeling (all-gather) and the label marginals (all-reduce for a
shared frame, all-gather for disjoint frames).  Here each rank runs the CPU oracle
as the stand-in for its GPU (the GPU<->oracle bit-exactness is what
tests/test_gpu_parity.py checks); the collectives run over gloo.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mcmc_amd import distributed as D  # noqa: E402


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 1024, 4097):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0
            for (o0, c0), (o1, _) in zip(spans, spans[1:]):
                assert o0 + c0 == o1
            assert spans[-1][0] + spans[-1][1] == n
            counts = [c for _, c in spans]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        D.shard_range(4, 2, 2)


def test_single_process_paths():
    e = np.array([-5.0, -1.0, -1.0, -9.0]); lab = np.arange(12, dtype=np.uint16).reshape(4, 3)
    be, cid, bl, per = D.gather_best(e, lab, chain_offset=100)
    assert be == -1.0 and cid == 101 and bl.tolist() == [3, 4, 5] and per == [(-1.0, 101)]
    m = np.arange(6, dtype=np.uint64).reshape(3, 2)
    assert np.array_equal(D.reduce_marginals(m), m)
    assert D.merge_visits({1: 2}) == {1: 2}
    assert D.gather_rows(np.array([4, 5])).tolist() == [[4, 5]]
    assert D.sum_max(np.array([3]), np.array([9])).tolist() == [3, 9]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mcmc_amd.scene import make_scene
        from tests import oracle_lib as O
        sc = make_scene(width=160, height=120, n_planes=3, n_superpixels=40, seed=3)
        S = sc.n_superpixels
        fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
        _, qe = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
        total = 6
        off, cnt = D.shard_range(total, world, rank)
        ch = O.Chains(fit, valid, sc.row_ptr, sc.col, qe, cnt, seed=11, chain_offset=off)
        ch.run(30, 120)
        s = ch.scalars()
        be, cid, bl, per = D.gather_best(s["best_energy"], ch.labels(1, True), off)
        marg = D.reduce_marginals(ch.marginals(), shared_frames=True)
        cat = D.reduce_marginals(ch.marginals(), shared_frames=False)
        vis = {}
        for v in ch.visits():
            for h, c in v.items():
                vis[h] = vis.get(h, 0) + c
        vis = D.merge_visits(vis)
        rows = D.gather_rows(np.array([off, rank * 10 + 1], np.int64))
        sm = D.sum_max(np.array([cnt]), np.array([rank + 5]))
        q.put((rank, be, cid, bl.tolist(), marg.tolist(), cat.shape, s["trace_hash"].tolist(), per, vis,
               rows.tolist(), sm.tolist()))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_matches_single_process():
    import torch.multiprocessing as mp
    from mcmc_amd.scene import make_scene
    from tests import oracle_lib as O

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0

    # the same job in one process
    sc = make_scene(width=160, height=120, n_planes=3, n_superpixels=40, seed=3)
    S = sc.n_superpixels
    fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
    _, qe = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
    ch = O.Chains(fit, valid, sc.row_ptr, sc.col, qe, 6, seed=11)
    ch.run(30, 120)
    s = ch.scalars()
    # sharded chains are the same chains
    assert res[0][6] + res[1][6] == s["trace_hash"].tolist()
    # both ranks hold the same global best, and it is the single-process best
    be, cid, bl, _ = D.gather_best(s["best_energy"], ch.labels(1, True), 0)
    for r in res:
        assert r[1] == be and r[2] == cid and r[3] == bl.tolist()
        assert sorted(r[7]) == sorted(res[0][7]) and len(r[7]) == 2
    # marginals: all-reduce equals the single-process histogram; all-gather stacks rank blocks
    for r in res:
        assert r[4] == ch.marginals().tolist()
        assert r[5] == (2 * S, 64)
    tot = {}
    for v in ch.visits():
        for h, c in v.items():
            tot[h] = tot.get(h, 0) + c
    assert res[0][8] == tot and res[1][8] == tot
    for r in res:
        assert r[9] == [[0, 1], [3, 11]] and r[10] == [6, 6]


def _worker_frames(rank, world, port, q):
    """C5 shape: the FRAMES are sharded (two per rank, three chains each); marginals are all-gathered in frame order."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mcmc_amd.scene import make_scene
        from tests import oracle_lib as O
        total_frames, cpf = 2 * world, 3
        f_off, f_cnt = D.shard_range(total_frames, world, rank)
        margs, best_e, best_l, hashes = [], [], [], []
        smax = 0
        for f in range(f_off, f_off + f_cnt):
            sc = make_scene(width=160, height=120, n_planes=3, n_superpixels=36, seed=200 + f)
            S = sc.n_superpixels
            smax = max(smax, S)
            fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
            _, qe = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
            ch = O.Chains(fit, valid, sc.row_ptr, sc.col, qe, cpf, seed=5, chain_offset=f * cpf)
            ch.run(20, 60)
            m = np.zeros((64, 64), np.uint64); m[:S] = ch.marginals()          # frames padded to one s_max, as the library does
            margs.append(m)
            lab = np.full((cpf, 64), 0xFFFF, np.uint16); lab[:, :S] = ch.labels(1, True)
            best_e.append(ch.scalars()["best_energy"]); best_l.append(lab)
            hashes += ch.scalars()["trace_hash"].tolist()
        cat = D.reduce_marginals(np.stack(margs), shared_frames=False)
        be, cid, bl, per = D.gather_best(np.concatenate(best_e), np.concatenate(best_l), f_off * cpf)
        q.put((rank, cat.tolist(), be, cid, bl.tolist(), hashes, per))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_frame_sharding_matches_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_frames, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=90) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the same job on one rank (no process group: the single-process paths)
    q1 = ctx.Queue()
    port1 = _free_port()
    p1 = ctx.Process(target=_worker_frames_single, args=(port1, q1))
    p1.start()
    one = q1.get(timeout=180)
    p1.join(60)
    assert p1.exitcode == 0
    for r in res:
        assert r[1] == one[1]                       # gathered marginals: rank blocks in frame order = all four frames
        assert (r[2], r[3], r[4]) == (one[2], one[3], one[4])   # same global best on both ranks and on one rank
        assert len(r[6]) == 2
    assert res[0][5] + res[1][5] == one[5]          # the chains of a frame do not depend on who owns the frame


def _worker_frames_single(port, q):
    # world of one rank, still through the collectives (gloo)
    _worker_frames_as(0, 1, port, q, total_world=2)


def _worker_frames_as(rank, world, port, q, total_world):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mcmc_amd.scene import make_scene
        from tests import oracle_lib as O
        total_frames, cpf = 2 * total_world, 3
        margs, best_e, best_l, hashes = [], [], [], []
        for f in range(total_frames):
            sc = make_scene(width=160, height=120, n_planes=3, n_superpixels=36, seed=200 + f)
            S = sc.n_superpixels
            fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
            _, qe = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
            ch = O.Chains(fit, valid, sc.row_ptr, sc.col, qe, cpf, seed=5, chain_offset=f * cpf)
            ch.run(20, 60)
            m = np.zeros((64, 64), np.uint64); m[:S] = ch.marginals()
            margs.append(m)
            lab = np.full((cpf, 64), 0xFFFF, np.uint16); lab[:, :S] = ch.labels(1, True)
            best_e.append(ch.scalars()["best_energy"]); best_l.append(lab)
            hashes += ch.scalars()["trace_hash"].tolist()
        cat = D.reduce_marginals(np.stack(margs), shared_frames=False)
        be, cid, bl, per = D.gather_best(np.concatenate(best_e), np.concatenate(best_l), 0)
        q.put((rank, cat.tolist(), be, cid, bl.tolist(), hashes, per))
    finally:
        dist.destroy_process_group()
