"""Multi-GPU sharding of chains and frames.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL over xGMI
on ROCm, ``gloo`` in the CPU tests).  Chains never interact and frames never
interact, so the data path has NO collective: a rank owns a contiguous block of
global chain ids (C3: one shared frame, chains sharded) or a contiguous block of
frames (C5: frames sharded, all chains of a frame on the frame's rank), and the
Philox key is ``(seed, global chain id)``, so every rank count reproduces the
single-GPU chains bit for bit.

The only exchange happens after a run (or per reporting interval), and it is
small and latency-bound (<= 16 KB and <= 4 MB per rank):

* ``gather_best``      -- all-gather of each rank's best-energy labeling
                          ``{energy f64, global chain id, labels u16[S]}``; every
                          rank then holds the global arg-max (ties: lowest chain id).
* ``reduce_marginals`` -- all-reduce(sum) of the label-marginal histogram when the
                          ranks share one frame (C3); all-gather when every rank
                          owns different frames (C5).

The reference has no counterpart (single-threaded, no collectives); the
semantics being distributed are Statistics' visit counts / best state
(reference include/Statistics.h:33-70, src/exe/can05_sampler.cpp:23-45).
"""
from __future__ import annotations

import numpy as np


def shard_range(n_total: int, world: int, rank: int):
    """Contiguous block [offset, offset + count) of n_total units owned by `rank`.
    The first ``n_total % world`` ranks get one extra unit."""
    if world <= 0 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("shard_range: bad arguments")
    base, extra = divmod(n_total, world)
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def _device():
    import torch
    dist = _dist()
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def local_best(best_energy: np.ndarray, best_labels: np.ndarray, chain_offset: int):
    """This rank's best chain: (energy, global chain id, labels[S]).  Ties go to
    the lowest chain id so the result does not depend on the rank count."""
    e = np.asarray(best_energy, np.float64)
    if e.size == 0:
        return -np.inf, np.uint64(0xFFFFFFFFFFFFFFFF), np.zeros(best_labels.shape[1:], np.uint16)
    k = int(np.flatnonzero(e == e.max())[0])
    return float(e[k]), np.uint64(chain_offset + k), np.ascontiguousarray(best_labels[k], np.uint16)


def pack_best(e, cid, lab):
    """One packed int64 record: energy bits, global chain id, labels (4 per word) -- the layout of
    mcmc_chains_best_record_device."""
    S = lab.size
    rec = np.zeros(2 + (S + 3) // 4, np.int64)
    rec[0] = np.float64(e).view(np.int64)
    rec[1] = np.uint64(cid).astype(np.uint64).view(np.int64)
    padded = np.full(((S + 3) // 4) * 4, 0xFFFF, np.uint16)
    padded[:S] = lab
    rec[2:] = padded.view(np.int64)
    return rec


def _pick_best(allrec, S):
    es = allrec[:, 0].copy().view(np.float64)
    cids = allrec[:, 1].copy().view(np.uint64)
    order = np.lexsort((cids, -es))       # highest energy, then lowest chain id
    w = int(order[0])
    labels = allrec[w, 2:].copy().view(np.uint16)[:S]
    return float(es[w]), int(cids[w]), labels, [(float(a), int(b)) for a, b in zip(es, cids)]


def gather_best(best_energy, best_labels, chain_offset):
    """All-gather every rank's best labeling; returns (energy, global chain id,
    labels, per_rank) where per_rank lists every rank's (energy, chain id)."""
    import torch
    e, cid, lab = local_best(best_energy, best_labels, chain_offset)
    dist = _dist()
    if dist is None:
        return e, int(cid), lab, [(e, int(cid))]
    dev = _device()
    world = dist.get_world_size()
    rec = pack_best(e, cid, lab)
    out = torch.empty(world * rec.size, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, torch.from_numpy(rec).to(dev))
    return _pick_best(out.cpu().numpy().reshape(world, rec.size), lab.size)


class _DeviceMemory:
    """A span of device memory owned by the C library, as torch sees it (no copy)."""

    def __init__(self, ptr, n, typestr="<i8"):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_view(ptr, n_words):
    """int64 torch tensor over `n_words` 8-byte words of the library's device memory at `ptr`."""
    import torch
    return torch.as_tensor(_DeviceMemory(ptr, n_words), device=torch.device("cuda", torch.cuda.current_device()))


def gather_best_ctx(ctx, which=1, canonical=True):
    """gather_best for a mcmc_amd.host.Context: the arg-max over the rank's chains is taken on the device
    (mcmc_chains_get_best) and, over RCCL, the packed record goes from the library's device memory into the
    all-gather as it is -- one labeling per rank crosses the links, none crosses the host bus before the result."""
    dist = _dist()
    S = ctx.s_max
    if dist is None or dist.get_backend() != "nccl":
        e, _, cid, lab = ctx.best(which, canonical)
        if dist is None:
            return e, cid, lab, [(e, cid)]
        import torch
        world = dist.get_world_size()
        rec = pack_best(e, cid, lab)
        out = torch.empty(world * rec.size, dtype=torch.int64)
        dist.all_gather_into_tensor(out, torch.from_numpy(rec))
        return _pick_best(out.numpy().reshape(world, rec.size), S)
    import torch
    ptr, nbytes = ctx.best_record_device(which, canonical)
    ctx.synchronize()                      # the record is written on the context's stream
    mine = device_view(ptr, nbytes // 8)
    world = dist.get_world_size()
    out = torch.empty(world * mine.numel(), dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    return _pick_best(out.cpu().numpy().reshape(world, mine.numel()), S)


def reduce_marginals_ctx(ctx, n_frames, shared_frames=True):
    """reduce_marginals for a Context's frames [0, n_frames): over RCCL the counts are read where the library keeps
    them (device memory), summed / gathered on the device, and only the result is copied to the host."""
    dist = _dist()
    if dist is None or dist.get_backend() != "nccl":
        m = np.stack([ctx.marginals(f) for f in range(n_frames)])
        return reduce_marginals(m[0] if shared_frames and n_frames == 1 else m, shared_frames)
    import torch
    ptr, per_frame = ctx.marginals_device(0)
    ctx.synchronize()
    shape = (n_frames, ctx.s_max, per_frame // ctx.s_max)
    t = device_view(ptr, n_frames * per_frame).clone()      # the library's counts stay as they are
    if shared_frames:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        out = t.cpu().numpy().view(np.uint64).reshape(shape)
        return out[0] if n_frames == 1 else out
    world = dist.get_world_size()
    out = torch.empty(world * t.numel(), dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().view(np.uint64).reshape((world * n_frames,) + shape[1:])


def reduce_marginals(marg: np.ndarray, shared_frames: bool = True) -> np.ndarray:
    """Label marginals across ranks.  shared_frames=True: every rank sampled the
    SAME frames (chains sharded) -> element-wise sum.  False: ranks own different
    frames -> concatenation in rank order (every rank passes equally shaped input)."""
    import torch
    m = np.ascontiguousarray(marg, np.uint64)
    dist = _dist()
    if dist is None:
        return m.copy()
    dev = _device()
    t = torch.from_numpy(m.view(np.int64)).to(dev)
    if shared_frames:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy().view(np.uint64).reshape(m.shape)
    world = dist.get_world_size()
    out = torch.empty((world,) + tuple(t.shape), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out.view(-1), t.reshape(-1))
    return out.cpu().numpy().view(np.uint64).reshape((world * m.shape[0],) + m.shape[1:])


def gather_rows(row: np.ndarray) -> np.ndarray:
    """All-gather of one small int64 row per rank -> array [world, len(row)] in rank order."""
    import torch
    r = np.ascontiguousarray(row, np.int64).reshape(-1)
    dist = _dist()
    if dist is None:
        return r.reshape(1, -1).copy()
    dev = _device()
    world = dist.get_world_size()
    out = torch.empty(world * r.size, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(out, torch.from_numpy(r).to(dev))
    return out.cpu().numpy().reshape(world, r.size)


def sum_max(to_sum: np.ndarray, to_max: np.ndarray) -> np.ndarray:
    """All-reduce of two small int64 vectors, one by SUM and one by MAX; returns them concatenated."""
    import torch
    a = np.ascontiguousarray(to_sum, np.int64).reshape(-1)
    b = np.ascontiguousarray(to_max, np.int64).reshape(-1)
    dist = _dist()
    if dist is None:
        return np.concatenate([a, b])
    dev = _device()
    ta = torch.from_numpy(a.copy()).to(dev); tb = torch.from_numpy(b.copy()).to(dev)
    dist.all_reduce(ta, op=dist.ReduceOp.SUM)
    dist.all_reduce(tb, op=dist.ReduceOp.MAX)
    return np.concatenate([ta.cpu().numpy(), tb.cpu().numpy()])


def merge_visits(visits: dict) -> dict:
    """Statistics::update across ranks for ONE shared frame: sum of visit counts per
    state hash (small python dicts; object gather, once per run)."""
    dist = _dist()
    if dist is None:
        return dict(visits)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, visits)
    total = {}
    for p in parts:
        for h, c in p.items():
            total[h] = total.get(h, 0) + c
    return total
