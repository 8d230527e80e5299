#!/usr/bin/env python3
"""bench.py -- MCMC proposals evaluated / sec on synthetic planar RGB-D scenes.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1], "C2"): one synthetic 640x480 planar scene,
~500 superpixels (seed 1), 1024 independent chains per GPU.  One *step* is one
pass of the hot path over that batch, inputs already resident in HBM:
    K1 sp_moments + K1b local fit  ->  K2 edge probabilities  ->
    K3 chain_step for `--iters-per-step` sampler iterations of every chain
(statistics on).  value = proposals evaluated / s over all ranks; one proposal
= one FastMetropolisHastings::iterate of one chain.
Multi-GPU: chains are sharded by global chain id (weak scaling: 1024 chains per
GPU, Philox key = global id), no collective inside the timed region.

Extra objects on the JSON line:
  roofline     -- the moment-accumulation kernel (north_star's HBM target) on a
                  256-frame batch (C5 shape, 324.8 MB algorithmic), timed with HIP
                  events on the context's stream inside this process.
  chain_kernel -- K3's device time per launch and its algorithmic-byte rate
                  (latency/LDS-bound by design; reported, not the roofline claim).
  cpu_baseline -- the CPU oracle ("port" of the reference algorithm, oracle/)
                  timed single-threaded on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=1024, help="chains per GPU")
    ap.add_argument("--iters-per-step", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--roofline-frames", type=int, default=256)
    ap.add_argument("--roofline-reps", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-multiview", action="store_true")
    ap.add_argument("--no-c5", action="store_true")
    ap.add_argument("--mv-chains", type=int, default=1024)
    ap.add_argument("--mv-steps", type=int, default=20)
    return ap.parse_args()


def cpu_baseline(sc, seed, seconds):
    """Oracle (restatement of the reference algorithm) on this host, 1 thread."""
    from tests import oracle_lib as O
    S = sc.n_superpixels
    fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
    _, q = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
    burn, iters = 1000, 10000
    done = 0
    t_meas = 0.0
    k = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        ch = O.Chains(fit, valid, sc.row_ptr, sc.col, q, 1, seed, chain_offset=k)
        ch.run(burn, 0)
        t1 = time.perf_counter()
        ch.run(0, iters)
        t_meas += time.perf_counter() - t1
        done += iters
        k += 1
    cpu = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": done / t_meas, "unit": "proposals/s", "cores": 1, "kind": "port",
            "sample": f"{k} chains x ({burn} burn-in + {iters} measured) steps of the same C2 scene, "
                      f"oracle/mcmc_oracle.c -O2 -ffp-contract=off, 1 thread, {cpu}"}


def same_computation_check(sc, seed, total_steps, gpu_scalars, n=4):
    """show that the CPU baseline and the GPU run are the same computation.  The first `n` chains of
    the GPU job are re-run on the oracle with the same key and step count; the hash of the accept/case trace,
    the state hash and the energy bits must be identical.  Part of the cpu_baseline leg (not timed)."""
    from tests import oracle_lib as O
    S = sc.n_superpixels
    fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
    _, q = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
    ch = O.Chains(fit, valid, sc.row_ptr, sc.col, q, n, seed, chain_offset=0)
    ch.run(0, total_steps)
    o = ch.scalars()
    ok = all(np.array_equal(np.asarray(gpu_scalars[k][:n]).view(np.uint64) if k == "energy" else gpu_scalars[k][:n],
                            np.asarray(o[k]).view(np.uint64) if k == "energy" else o[k])
             for k in ("trace_hash", "state_hash", "energy", "n_accept"))
    return {"chains": n, "steps_per_chain": int(total_steps), "identical_to_gpu": bool(ok),
            "trace_hash_chain0": "%016x" % int(o["trace_hash"][0])}


def cpu_baseline_all_cores(sc, seed, seconds, threads):
    """The same oracle, chains spread over host threads (ctypes releases the GIL): what SURVEY section 8d
    asks for beside the single-threaded figure.  Each thread owns its chains; proposals / wall time."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import oracle_lib as O
    S = sc.n_superpixels
    fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
    _, q = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
    burn, iters = 1000, 10000

    def work(t):
        n = 0
        k = t
        t_meas = 0.0
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end:
            ch = O.Chains(fit, valid, sc.row_ptr, sc.col, q, 1, seed, chain_offset=1000 + k)
            ch.run(burn, 0)
            t1 = time.perf_counter()
            ch.run(0, iters)
            t_meas += time.perf_counter() - t1
            n += iters
            k += threads
        return n / t_meas if t_meas > 0 else 0.0
    with ThreadPoolExecutor(threads) as ex:
        rate = sum(ex.map(work, range(threads)))       # threads run side by side for the whole window
    return {"value": rate, "unit": "proposals/s", "cores": threads, "kind": "port",
            "sample": f"{threads} host threads side by side for {seconds:.0f} s, each running chains of (1000 burn-in + 10000 "
                      f"measured) steps of the same C2 scene; sum of the threads' measured rates"}


def k1_roofline(host, sc_list, n_frames, reps, device):
    """Moment-accumulation kernel on a batch of frames (C5 shape)."""
    W, H = sc_list[0].width, sc_list[0].height
    smax = max(s.n_superpixels for s in sc_list)
    ctx = host.Context(width=W, height=H, device=device)
    ctx.frames_alloc(n_frames, smax, 8)
    alg = 0
    for f in range(n_frames):
        sc = sc_list[f % len(sc_list)]
        ctx.frame_upload(f, sc.disp, sc.label, sc.n_superpixels)
        alg += W * H * 4 + sc.n_superpixels * 80
    ctx.set_timing(True)
    for _ in range(3):
        ctx.frames_moments()
    ctx.synchronize()
    ms = []
    for _ in range(reps):
        ctx.frames_moments()
        ms.append(ctx.last_kernel_ms(host.K_MOMENTS))
    # spot-check the batch result against frame 0 computed alone (bit-exact integers)
    m_batch = ctx.moments(n_frames - 1)
    ctx.frames_moments(n_frames - 1, 1)
    assert np.array_equal(m_batch, ctx.moments(n_frames - 1))
    ctx.close()
    avg = float(np.mean(ms))
    ach = alg / (avg * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "k1_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": "sp_moments_lr_kernel", "workload": f"{n_frames} frames x {W}x{H} (C5 batch)",
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg, "min_launch_ms": float(np.min(ms)),
            "launches": reps, "traffic": traffic}


def c5_share_line(host, sc_list, device, frames=32, cpf=64, iters=1000, steps=2):
    """BASELINE configs[4] ("C5"), one GPU's share of the node: 32 of the 256 frames x 64 chains each; a step =
    moments + local fits + edge probabilities of all frames, then `iters` sampler iterations of every chain.
    Secondary line (the headline stays C2): the shape a multi-frame deployment runs."""
    import torch
    W, H = sc_list[0].width, sc_list[0].height
    smax = max(s.n_superpixels for s in sc_list); emax = max(s.col.size for s in sc_list)
    ctx = host.Context(width=W, height=H, device=device)
    ctx.frames_alloc(frames, smax, emax)
    for f in range(frames):
        host.load_scene(ctx, f, sc_list[f % len(sc_list)])
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.chains_init(frames * cpf, chains_per_frame=cpf, seed=5)
    ctx.set_timing(True)

    def step():
        ctx.frames_moments(); ctx.frames_edge_probs(); ctx.chains_run(0, iters, sync=False)
    step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    sca = ctx.scalars()
    assert (sca["status"] == 0).all()
    out = {"workload": "C5 share: %d frames x %d chains, %dx%d, step = moments + edge probs + %d iterations/chain"
                       % (frames, cpf, W, H, iters),
           "proposals_per_s": frames * cpf * iters * steps / dt, "ms_per_step": dt / steps * 1e3,
           "chain_kernel_ms": ctx.last_kernel_ms(host.K_CHAIN), "moments_kernel_ms": ctx.last_kernel_ms(host.K_MOMENTS)}
    ctx.close()
    return out


def multiview_line(host, device, n_chains, steps):
    """Path B (BASELINE configs[3], "C4"): 3 synthetic views, ~500 superpixels each, `n_chains` chains x `steps`
    proposals of the PatternedMatcher loop (the reference runs 20, PatternedMatcher.cpp:29)."""
    from mcmc_amd.scene import make_multiview
    mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
    m = host.Multiview(device=device)
    m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
    m.chains_init(n_chains, seed=4, max_new_segments=4 * steps)
    m.chains_run(steps)                      # warm-up launch
    m.chains_run(steps)
    ms = m.last_run_ms()
    sc = m.scalars()
    assert (sc["status"] == 0).all()
    # one more launch with the phase counters on (not timed): points of SB1 u SB2 per proposal -> SURVEY 8(d)'s
    # algorithmic bytes for K4 (2 passes x 16 B per point); the chains share the points through L2 / registers,
    # so this rate may exceed the HBM peak -- it is reported, not claimed as a roofline fraction
    alg = None
    if n_chains >= 32:
        m.set_option("profile", 1)
        m.chains_run(steps)
        m.synchronize() if hasattr(m, "synchronize") else None
        pts = float(m.profile()[10]) / (n_chains * steps)
        alg = {"points_per_proposal": pts, "bytes_per_proposal": 32.0 * pts,
               "GB_per_s": 32.0 * pts * n_chains * steps / (ms * 1e-3) / 1e9,
               "evaluations_per_proposal": float(m.profile()[11]) / (n_chains * steps)}
    m.close()
    return {"workload": "C4: 3 views x 640x480, %d vertices, %d points, %d directed edges, %d chains x %d steps"
                        % (mv.n_vertices, mv.points.shape[0], mv.col.size, n_chains, steps),
            "proposals_per_s": n_chains * steps / (ms * 1e-3), "ms_per_launch": ms,
            "accept_rate": float(sc["n_accept"].sum()) / (2.0 * n_chains * steps),
            "algorithmic": alg,
            "bound": "f64 VALU of the L1 residual pass (evaluated for all chains vertex by vertex, points read once per "
                     "vertex block); the per-chain growth is breadth first over four waves"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    # MCMC_BENCH_FORCE_DIST=1: rehearse the RCCL path with a single rank (one-GPU boxes)
    if world > 1 or os.environ.get("MCMC_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from mcmc_amd import host
    from mcmc_amd.scene import make_scene

    sc = make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=a.seed)
    S = sc.n_superpixels
    ctx = host.Context(width=640, height=480, device=local)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.frames_alloc(1, S, sc.col.size)
    host.load_scene(ctx, 0, sc)
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.chains_init(a.chains, seed=a.seed, chain_offset=rank * a.chains)
    ctx.set_timing(True)

    def step():
        ctx.frames_moments()
        ctx.frames_edge_probs()
        ctx.chains_run(0, a.iters_per_step, sync=False)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    chain_ms = []
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    chain_ms.append(ctx.last_kernel_ms(host.K_CHAIN))
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # after the timed region: the only exchange of the multi-GPU path, RCCL when N > 1
    from mcmc_amd import distributed as D
    sca = ctx.scalars()
    best_e, best_chain, best_labels, per_rank = D.gather_best(sca["best_energy"], ctx.labels(1, True), rank * a.chains)
    marg = D.reduce_marginals(ctx.marginals(0), shared_frames=True)
    n_valid = int((ctx.local_fit(0)[1] != 0).sum())
    # every valid superpixel is counted once per chain per measured sampler iteration, on every rank
    assert int(marg.sum()) == world * a.chains * a.iters_per_step * (a.warmup + a.steps) * n_valid
    assert (sca["status"] == 0).all(), "a chain reported an error status"
    total_steps = (a.warmup + a.steps) * a.iters_per_step
    assert (sca["n_case"].sum(axis=1) == total_steps).all()
    acc_rate = float(sca["n_accept"].sum()) / (a.chains * total_steps)
    proposals = world * a.chains * a.iters_per_step * a.steps
    out = {
        "metric": "MCMC proposals evaluated/sec (whole node) at 640x480 RGB-D; % HBM peak",
        "value": proposals / dt, "unit": "proposals/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C2: synthetic 640x480 planar RGB-D scene (seed %d), %d superpixels, "
                               "%d chains per GPU, step = moments + edge probs + %d sampler iterations/chain"
                               % (a.seed, S, a.chains, a.iters_per_step),
                   "chains_per_gpu": a.chains, "iters_per_step": a.iters_per_step, "superpixels": S,
                   "directed_edges": int(sc.col.size), "accept_rate": acc_rate,
                   "segments_mean": float(np.mean([len(np.unique(l[:S])) for l in ctx.labels(0, True)[:64]])),
                   "best_energy": best_e, "best_chain": best_chain, "ranks_gathered": len(per_rank)},
    }
    if rank == 0:
        k3_ms = chain_ms[-1]
        out["chain_kernel"] = {"kernel": "chain_run_kernel", "ms_per_launch": k3_ms,
                               "proposals_per_launch": a.chains * a.iters_per_step,
                               "us_per_proposal_per_chain": k3_ms * 1e3 / a.iters_per_step,
                               "bound": "latency (LDS-resident BFS); not an HBM claim"}
        out["frame_kernels_ms"] = {"sp_moments_1frame": ctx.last_kernel_ms(host.K_MOMENTS),
                                   "local_fit": ctx.last_kernel_ms(host.K_LOCAL_FIT),
                                   "edge_prob": ctx.last_kernel_ms(host.K_EDGE_PROB)}
    ctx.close()
    if rank == 0 and world == 1 and not a.no_multiview:
        out["multiview"] = multiview_line(host, local, a.mv_chains, a.mv_steps)
    if rank == 0 and world == 1:
        scs = None
        if not a.no_roofline or not a.no_c5:
            scs = [sc] + [make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=100 + i)
                          for i in range(3)]
        if not a.no_c5:
            out["c5_share"] = c5_share_line(host, scs, local)
        if not a.no_roofline:
            out["roofline"] = k1_roofline(host, scs, a.roofline_frames, a.roofline_reps, local)
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, a.seed, a.cpu_seconds)
            out["cpu_baseline"]["same_computation"] = same_computation_check(sc, a.seed, total_steps, sca)
            assert out["cpu_baseline"]["same_computation"]["identical_to_gpu"], "GPU chains differ from the oracle"
            nthr = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(sc, a.seed, min(a.cpu_seconds, 6.0), min(nthr, 16))
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
