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

`--gpus N` without an external launcher (no WORLD_SIZE in the environment) starts the N ranks itself: the parent
runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same
flags>` as a child process before anything touches a GPU, relays rank 0's JSON line and exits with the child's
code.  A world that differs from --gpus is an error (exit 2), never a silently smaller run.

`--config c2` (default, the headline): weak scaling, `--chains` chains per GPU.
`--config c3`: BASELINE configs[2], one 1920x1080 frame (~4500 superpixels), `--total-chains` (4096) chains sharded
              over the ranks by contiguous global id (strong scaling), marginals all-reduced.
`--config c5`: BASELINE configs[4], `--total-frames` (256) frames x 64 chains, frames sharded over the ranks
              (strong scaling), marginals all-gathered.

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


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains", type=int, default=1024, help="chains per GPU")
    ap.add_argument("--iters-per-step", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--roofline-frames", type=int, default=256)
    ap.add_argument("--roofline-reps", type=int, default=300)     # the plain mean of all of them: a stray 200 us launch (one in a few hundred) then moves it by 0.5 us, not 1.6
    ap.add_argument("--roofline-warmup", type=int, default=400)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-multiview", action="store_true")
    ap.add_argument("--no-c5", action="store_true")
    ap.add_argument("--mv-chains", type=int, default=1024)
    ap.add_argument("--mv-steps", type=int, default=20)
    ap.add_argument("--config", choices=["c2", "c3", "c5"], default="c2")
    ap.add_argument("--visit-capacity", type=int, default=0,
                    help="distinct states counted per chain (Statistics); 0 = sized from the run by the library")
    ap.add_argument("--total-chains", type=int, default=4096, help="c3: chains over all GPUs")
    ap.add_argument("--total-frames", type=int, default=256, help="c5: frames over all GPUs")
    ap.add_argument("--chains-per-frame", type=int, default=64, help="c5")
    ap.add_argument("--distinct-scenes", type=int, default=8, help="c5: distinct synthetic scenes cycled over the frames")
    return ap.parse_args(argv)


def launch_plan(a, argv, env):
    """How this invocation gets its ranks.  Returns None when this process is itself a rank (N == 1, or started
    by torchrun: WORLD_SIZE set), else (command, environment) of the child that starts N ranks.  Pure function of
    its arguments (tests/test_bench_launch.py) -- nothing here touches a GPU."""
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in env:
        if int(env["WORLD_SIZE"]) != a.gpus:
            print("bench.py: WORLD_SIZE=%s but --gpus %d" % (env["WORLD_SIZE"], a.gpus), file=sys.stderr)
            raise SystemExit(2)                   # the world that runs must be the one asked for
        return None
    if a.gpus == 1:
        return None
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    child = dict(env)
    child.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes on this host driver)
    child.setdefault("MASTER_ADDR", "127.0.0.1")
    return cmd, child


def run_children(cmd, env):
    """Parent of a self-launched multi-rank run: never touches the GPU, relays rank 0's JSON line."""
    import subprocess
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0:
        raise SystemExit(p.returncode)
    if line is None:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        raise SystemExit(3)
    print(line)
    raise SystemExit(0)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
    cpu = cpu_model()
    return {"value": done / t_meas, "unit": "proposals/s", "cores": 1, "kind": "port",
            "sample": f"{k} chains x ({burn} burn-in + {iters} measured) steps of the same C2 scene, "
                      f"oracle/mcmc_oracle.c -O2 -ffp-contract=off, 1 thread, {cpu}"}


def same_computation_check(scenes, cfg, cpf, seed, total_steps, firsts):
    """show that the CPU baseline and the GPU run are the same computation, and that a rank's chains
    are the chains a single-GPU run would have produced.  `firsts` holds, per rank, (global chain id, trace hash)
    of the rank's first chain; each is re-run on the oracle with the same key and step count on the frame that
    chain samples, and the hash of its accept / case / state trace must be identical.  Part of the cpu_baseline
    leg (not timed)."""
    from tests import oracle_lib as O
    tabs = {}
    ok, rows = True, []
    for cid, hi, lo in firsts.tolist():
        sc = scenes[(cid // cpf) % len(scenes)] if cfg == "c5" else scenes[0]
        if id(sc) not in tabs:
            S = sc.n_superpixels
            fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
            _, q = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
            tabs[id(sc)] = (fit, valid, q)
        fit, valid, q = tabs[id(sc)]
        ch = O.Chains(fit, valid, sc.row_ptr, sc.col, q, 1, seed, chain_offset=int(cid))
        ch.run(0, total_steps)
        want = int(ch.scalars()["trace_hash"][0])
        got = (int(hi) << 32) | int(lo)
        ok = ok and want == got
        rows.append({"chain": int(cid), "trace_hash": "%016x" % got})
    return {"chains": rows, "steps_per_chain": int(total_steps), "identical_to_gpu": bool(ok)}


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
                      f"measured) steps of the same C2 scene; sum of the threads' measured rates; {cpu_model()}"}


def k1_roofline(host, sc_list, n_frames, reps, device, warm=400):
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
    # untimed warm-up: a 70 us launch after an idle gap runs at a lower clock; the launch time settles after a few hundred
    # launches (tools/ab/k1_ab.py: the first 50 after 150 warm-up launches average 68.3 us, the next hundred 67.0; back to
    # 71-72 after 2 s of idle, which roofline.cold_launch_ms reports beside the steady figure)
    for _ in range(warm):
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
    # the same launch after an idle gap (lower clock, cold caches): the first 25 launches after 2 s without work
    ctx.synchronize()
    time.sleep(2.0)
    cold = []
    for _ in range(25):
        ctx.frames_moments()
        cold.append(ctx.last_kernel_ms(host.K_MOMENTS))
    # the ceiling of this access pattern: same launch shape, loads and waits, arithmetic replaced by an OR
    ctx.set_option("k1_debug", 1)
    sm = []
    for _ in range(warm + reps):
        ctx.frames_moments()
        sm.append(ctx.last_kernel_ms(host.K_MOMENTS))
    stream_ms = float(np.mean(sm[warm:]))
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
            "median_launch_ms": float(np.median(ms)), "p90_launch_ms": float(np.percentile(ms, 90)),
            "max_launch_ms": float(np.max(ms)), "launches_over_1p5x_median": int((np.asarray(ms) > 1.5 * np.median(ms)).sum()),
            "launches": reps, "warmup_launches": warm, "traffic": traffic,
            "cold_launch_ms": float(np.mean(cold)), "cold_launch_note": "mean of the first 25 launches after 2 s idle",
            "cold_frac": alg / (float(np.mean(cold)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "stream_only_GBps": alg / (stream_ms * 1e-3) / 1e9, "stream_only_launch_ms": stream_ms,
            "stream_only_note": "the same kernel with the arithmetic replaced by an OR of the words read: what this "
                                "launch shape and access pattern reach on their own; achieved / stream_only = %.2f"
                                % (stream_ms / avg)}


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
    ctx.reserve_visits((steps + 1) * iters)      # Statistics keeps every state: room for the whole run, as in the headline job
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


def k3_bound_model(us_per_step, variant):
    """K3 has no byte stream: a chain step is ~2400 vector instructions on LDS-resident tables, several waves per chain
    (the further ones evaluate the next steps speculatively), one chain per SIMD at C2 (1024 chains, 1024 SIMDs).
    Its bound is the SIMD's vector pipe.  This is a MODEL FROM A RECORDED PROFILE, not a live counter: instruction
    totals come from profiles/k3_counts.json (tools/prof_k3.sh: rocprofv3 --pmc over tools/k3_profile.py, the same
    kernel and launch shape), the step time is live; the fractions are only emitted when the kernel variant that ran
    is the one that was profiled."""
    path = os.path.join(ROOT, "profiles", "k3_counts.json")
    try:
        rec = json.load(open(path))
    except Exception:
        return {"kind": "no recorded profile (profiles/k3_counts.json missing)", "achieved_us_per_step": us_per_step}
    t = rec["totals"]
    per = float(rec["chains"]) * float(rec["steps_per_chain"])
    valu = t.get("SQ_INSTS_VALU", 0.0) / per
    others = (t.get("SQ_INSTS_SALU", 0.0) + t.get("SQ_INSTS_LDS", 0.0) + t.get("SQ_INSTS_VMEM_RD", 0.0)) / per
    # the vector pipe: its instructions at the 4.8 cycles the f64 / 64-bit-multiply / lane-move mix of this kernel issues
    # at with several waves per SIMD (profiles/r02_valu_rate_k3.txt); scalar, LDS and memory instructions issue beside it
    valu_us = valu * 4.8 / 2400.0
    out = {"kind": "model from a recorded profile: vector instructions per committed chain step x 4.8 cycles at 2.4 GHz, one chain per SIMD",
           "profiled_kernel": rec["kernel"], "profiled_variant": rec.get("variant"), "ran_variant": variant,
           "valu_instructions_per_chain_step": valu, "other_instructions_per_chain_step": others,
           "achieved_us_per_step": us_per_step, "source": "profiles/k3_counts.json"}
    if rec.get("variant") == variant:
        out["valu_pipe_us_per_step"] = valu_us
        out["frac"] = valu_us / us_per_step
        out["note"] = ("frac = vector-pipe time of the recorded instruction count / live step time; all waves of a chain, "
                       "wasted speculation included")
    else:
        out["note"] = "the kernel variant that ran is not the profiled one: no fraction"
    return out


def multiview_line(host, device, n_chains, steps):
    """Path B (BASELINE configs[3], "C4"): 3 synthetic views, ~500 superpixels each, `n_chains` chains x `steps`
    proposals of the PatternedMatcher loop (the reference runs 20, PatternedMatcher.cpp:29)."""
    from mcmc_amd.scene import make_multiview
    mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
    m = host.Multiview(device=device)
    m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
    m.chains_init(n_chains, seed=4, max_new_segments=4 * steps)
    for _ in range(3):                       # untimed warm-up launches (the first ones run on a clock that is still rising)
        m.chains_run(steps)
    ms_all = []
    for _ in range(5):                       # the median of five launches of `steps` iterations
        m.chains_run(steps)
        ms_all.append(m.last_run_ms())
    ms = sorted(ms_all)[2]
    sc = m.scalars()
    assert (sc["status"] == 0).all()
    # one more launch with the phase counters on (not timed): points of SB1 u SB2 per proposal -> SURVEY 8(d)'s
    # algorithmic bytes for K4 (2 passes x 16 B per point); the chains share the points through L2 / registers,
    # so this rate may exceed the HBM peak -- it is reported, not claimed as a roofline fraction
    alg = None
    l1 = None
    if n_chains >= 32:
        m.set_option("profile", 1)
        m.chains_run(steps)
        m.synchronize() if hasattr(m, "synchronize") else None
        p = m.profile().astype(float)
        pts = p[10] / (n_chains * steps)
        alg = {"points_per_proposal": pts, "bytes_per_proposal": 32.0 * pts,
               "GB_per_s": 32.0 * pts * n_chains * steps / (ms * 1e-3) / 1e9,
               "evaluations_per_proposal": p[11] / (n_chains * steps),
               "point_residuals_per_proposal": p[12] / (n_chains * steps)}
        # K4's bound is the f64 matrix pipe: a residual |n.p + d| is 4 FMAs of one v_mfma_f64_16x16x4_f64 lane-slot,
        # 78.6 TFLOP/s dense f64 matrix peak (= 64 cycles per instruction and SIMD, tools/micro/mfma_f64_rate.hip).
        # Algorithmic flops = 8 per residual asked for; the whole iteration (growth, plane fits, L1 pass, accept) is
        # timed, so this is the fraction of the peak that the ITERATION reaches; the L1 kernel alone is in
        # profiles/r02_multiview_kernel_stats.csv + r02_multiview_pmc.txt (MfmaUtil).
        flops = 8.0 * p[12]
        l1 = {"bound": "mfma", "dtype": "f64", "peak": 78.6, "unit": "TFLOP/s",
              "achieved_whole_iteration": flops / (ms * 1e-3) / 1e12, "frac_whole_iteration": flops / (ms * 1e-3) / 78.6e12,
              "matrix_instructions_per_iteration": 16.0 * p[21] / steps if p[19] else None,
              "slots_per_residual": 16.0 * p[21] * 256.0 / p[12] if p[12] and p[19] else None,
              "note": "matrix-instruction count from one timed item in 16 (diagnostic counters); per-kernel times and MfmaUtil of "
                      "the L1 kernel are in profiles/ (rocprofv3), the first half of the iteration (component growth) is latency bound"}
    # the same graph with four times the chains: 1024 chains are one round of four chains per CU and leave the
    # iteration at twice a single chain's latency; more chains in flight raise the rate
    more = None
    if n_chains == 1024:
        m.set_option("profile", 0)
        more = []
        for nc in (2048, 4096):
            m.chains_init(nc, seed=4, max_new_segments=4 * steps)
            for _ in range(2):
                m.chains_run(steps)
            msn_all = []
            for _ in range(5):
                m.chains_run(steps)
                msn_all.append(m.last_run_ms())
            msn = sorted(msn_all)[2]
            more.append({"chains": nc, "proposals_per_s": nc * steps / (msn * 1e-3), "ms_per_launch": msn})
    m.close()
    # the first half of the iteration (component growth + plane fits) has no byte stream and no matrix work: what is
    # known about its bound comes from recorded experiments, not from a live counter
    l1_kernel = {"kind": "recorded, not live: profiles/r03_multiview_kernel_stats.csv, r03_multiview_pmc.txt, r03_mfma_f64_abs_sum.txt, r03_l1_timeline.txt",
                 "kernel_us_at_1024_chains_one_group": 50.1, "matrix_instructions_per_launch": 874029,
                 "matrix_pipe_us_at_64_cycles": 21.6, "frac_of_f64_matrix_peak": 0.45,
                 "at_more_chains": [{"chains": 2048, "kernel_us": 89.4, "frac_of_f64_matrix_peak": 0.51},
                                    {"chains": 4096, "kernel_us": 169.4, "matrix_instructions_per_launch": 3536840, "frac_of_f64_matrix_peak": 0.54}],
                 "ceiling_frac_with_abs_sums_on_the_same_pipe": 0.73, "frac_of_that_ceiling": 0.62,
                 "note": "the four v_add_f64 |d| per matrix instruction share the f64 pipe with it: a SIMD saturates at 85-88 cycles "
                         "per product + additions (64 for the product alone); the rest is two round trips per item before its first "
                         "product and the spread of the SIMDs' totals (DESIGN.md section 5, K4)"}
    first_half = {"kind": "recorded experiments, not live: profiles/r03_multiview_experiments.txt",
                  "growth_ahead": "up to 3072 chains the growth of step t+1 runs on a second stream beside the plane fits and the "
                                  "residual pass of step t (it needs nothing of the steps before it); the figures below are of "
                                  "the kernels run one after the other",
                  "fused_kernel_us_at_1024_chains": 108.9, "what_it_replaces_us": [99.9, 18.6],
                  "kernel_us_at_1024_chains": 99.9, "of_which_growth_us": 75.0, "plane_fits_and_hand_over_us": 22.0,
                  "trial_random_numbers_us": 6.0,
                  "instructions_per_chain_step": {"valu": 20850, "salu": 12090, "lds": 1710, "vmem": 940},
                  "wave_cycles_waiting_share": 0.59,
                  "one_chain_alone_us": 65.8, "slowest_over_mean_growth": 2.2,
                  "issue_bound_us": 68.3, "frac_of_issue_bound": 68.3 / 99.9,
                  "note": "bound: the SIMD's instruction issue over all pipes (36.5 M wave-instructions per launch / 1024 SIMDs x 4.6 "
                          "cycles = 68 us) and the launch's slowest chain (2.2x the mean growth); moving instructions between pipes "
                          "(+27 % vector, -9 % scalar) or a third of the row trips left the time unchanged, requesting four rows "
                          "ahead took 11 us off"}
    return {"more_chains": more, "first_half_model": first_half, "l1_kernel": l1_kernel,
            "workload": "C4: 3 views x 640x480, %d vertices, %d points, %d directed edges, %d chains x %d steps"
                        % (mv.n_vertices, mv.points.shape[0], mv.col.size, n_chains, steps),
            "proposals_per_s": n_chains * steps / (ms * 1e-3), "ms_per_launch": ms, "ms_per_launch_all": ms_all,
            "accept_rate": float(sc["n_accept"].sum()) / (8.0 * n_chains * steps),      # eight launches by then
            "algorithmic": alg, "l1_pass": l1,
            "bound": "iteration = growth (latency: breadth first over four waves per chain, bitmap adjacency rows) + plane fits + "
                     "L1 residual pass on the f64 matrix cores (one (chain, plane) pair per column, 16 points per instruction)"}


def main():
    argv = sys.argv[1:]
    a = parse(argv)
    plan = launch_plan(a, argv, os.environ)
    if plan is not None:
        run_children(*plan)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # MCMC_BENCH_ONE_GPU=1: a rehearsal of the N-rank job on a box with one GPU -- every rank computes on device 0 and the
    # collectives go over gloo (RCCL refuses two ranks on one device).  Same sharding, gathers and checks; the rate it
    # prints is no measurement of anything (the ranks share the GPU) and the line says so.
    one_gpu = bool(os.environ.get("MCMC_BENCH_ONE_GPU")) and world > 1
    if one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    # MCMC_BENCH_FORCE_DIST=1: rehearse the RCCL path with a single rank (one-GPU boxes)
    if world > 1 or os.environ.get("MCMC_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:      # the one-rank rehearsal, started without a launcher
            os.environ.setdefault("MASTER_PORT", "29533"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from mcmc_amd import host
    from mcmc_amd import distributed as D
    from mcmc_amd.scene import make_scene

    # ---- the job of this rank: frames resident in HBM, chains by global id --------------------------------
    cfg = a.config
    if cfg == "c2":       # weak scaling: every GPU gets a.chains chains of the one C2 frame
        scenes = [make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=a.seed)]
        frames_local, cpf, chain_off, n_local = 1, a.chains, rank * a.chains, a.chains
        total_chains, scaling, shared = world * a.chains, "weak", True
        seed = a.seed
        workload = ("C2: synthetic 640x480 planar RGB-D scene (seed %d), %d superpixels, %d chains per GPU, step = "
                    "moments + edge probs + %d sampler iterations/chain"
                    % (a.seed, scenes[0].n_superpixels, a.chains, a.iters_per_step))
    elif cfg == "c3":     # strong scaling: 4096 chains of one 1920x1080 frame, contiguous blocks of global ids
        scenes = [make_scene(width=1920, height=1080, n_planes=24, n_superpixels=4000, seed=2)]
        chain_off, n_local = D.shard_range(a.total_chains, world, rank)
        frames_local, cpf, total_chains, scaling, shared = 1, n_local, a.total_chains, "strong", True
        seed = 2
        workload = ("C3: synthetic 1920x1080 scene (seed 2), %d superpixels, %d chains sharded over %d GPU(s), step = "
                    "moments + edge probs + %d sampler iterations/chain"
                    % (scenes[0].n_superpixels, a.total_chains, world, a.iters_per_step))
    else:                 # c5, strong scaling: 256 frames x 64 chains, contiguous blocks of frames
        scenes = [make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=100 + i)
                  for i in range(max(1, a.distinct_scenes))]
        f_off, frames_local = D.shard_range(a.total_frames, world, rank)
        cpf = a.chains_per_frame
        chain_off, n_local = f_off * cpf, frames_local * cpf
        total_chains, scaling, shared = a.total_frames * cpf, "strong", False
        seed = 5
        workload = ("C5: %d frames of 640x480 (%d distinct synthetic scenes, seeds 100.., cycled) x %d chains, frames "
                    "sharded over %d GPU(s), step = moments + edge probs of the rank's frames + %d sampler "
                    "iterations/chain" % (a.total_frames, len(scenes), cpf, world, a.iters_per_step))
    if n_local <= 0:
        raise SystemExit("bench.py: rank %d has nothing to do (more ranks than units)" % rank)
    if cfg == "c5" and a.total_frames % world != 0:
        # the marginal all-gather stacks equal blocks in rank order; an uneven split would need padding
        print("bench.py: --total-frames %d does not divide over %d ranks" % (a.total_frames, world), file=sys.stderr)
        raise SystemExit(2)
    W, H = scenes[0].width, scenes[0].height
    smax = max(s.n_superpixels for s in scenes); emax = max(s.col.size for s in scenes)
    sc = scenes[0]
    S = sc.n_superpixels
    ctx = host.Context(width=W, height=H, device=local, visit_capacity=a.visit_capacity)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.frames_alloc(frames_local, smax, emax)
    first_frame = (chain_off // cpf) if cfg == "c5" else 0
    for f in range(frames_local):
        host.load_scene(ctx, f, scenes[(first_frame + f) % len(scenes)])
    ctx.frames_moments(); ctx.frames_edge_probs()
    ctx.chains_init(n_local, chains_per_frame=cpf, seed=seed, chain_offset=chain_off)
    # Statistics keeps every visited state: make room for the whole run now, so no step has to grow the table
    ctx.reserve_visits((a.warmup + a.steps) * a.iters_per_step)
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
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    k3_ms = ctx.last_kernel_ms(host.K_CHAIN)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- after the timed region: the only exchange of the multi-GPU path, RCCL when N > 1 -----
    sca = ctx.scalars()
    total_steps = (a.warmup + a.steps) * a.iters_per_step
    assert (sca["status"] == 0).all(), "a chain reported an error status"
    assert (sca["n_case"].sum(axis=1) == total_steps).all()
    # the rank's arg-max is taken on the device (mcmc_chains_get_best): one labeling per rank enters the all-gather,
    # straight from the library's device memory when the backend is RCCL
    best_e, best_chain, best_labels, per_rank = D.gather_best_ctx(ctx, which=1, canonical=True)
    host_e = sca["best_energy"]
    assert best_e >= float(host_e.max()) and (len(per_rank) > 1 or best_chain == chain_off + int(np.flatnonzero(host_e == host_e.max())[0]))
    if shared:      # every rank sampled the same frame: all-reduce(sum)
        marg = D.reduce_marginals_ctx(ctx, 1, shared_frames=True)
        n_valid = int((ctx.local_fit(0)[1] != 0).sum())
        # every valid superpixel is counted once per chain per measured sampler iteration, on every rank
        assert int(marg.sum()) == total_chains * total_steps * n_valid
        frames_gathered = 1
    else:           # disjoint frames: all-gather, rank blocks in frame order (equal shares, checked above)
        marg = D.reduce_marginals_ctx(ctx, frames_local, shared_frames=False)
        frames_gathered = int(marg.shape[0])
        assert frames_gathered == a.total_frames
        per_frame = marg.reshape(a.total_frames, -1).sum(axis=1)
        nv = [int((ctx.local_fit(f)[1] != 0).sum()) for f in range(frames_local)]
        assert all(int(per_frame[first_frame + f]) == cpf * total_steps * nv[f] for f in range(frames_local))
    # one chain per rank (its first): the trace hash a single-GPU run gives the same global chain id is
    # recomputed on the oracle in the cpu_baseline leg below
    first = np.array([chain_off, int(sca["trace_hash"][0]) >> 32, int(sca["trace_hash"][0]) & 0xFFFFFFFF], np.int64)
    firsts = D.gather_rows(first)
    overflow = D.sum_max(np.array([int(sca["overflow"].sum())], np.int64), np.array([int(sca["overflow"].max())], np.int64))
    acc_rate = float(sca["n_accept"].sum()) / (n_local * total_steps)
    proposals = total_chains * a.iters_per_step * a.steps
    out = {
        "metric": "MCMC proposals evaluated/sec (whole node) at 640x480 RGB-D; % HBM peak",
        "value": proposals / dt, "unit": "proposals/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "name": cfg,
                   "chains_per_gpu": n_local, "chains_total": total_chains, "frames_per_gpu": frames_local,
                   "iters_per_step": a.iters_per_step, "superpixels": S,
                   "directed_edges": int(sc.col.size), "accept_rate": acc_rate,
                   "segments_mean": float(np.mean([len(np.unique(l[:S])) for l in ctx.labels(0, True)[:min(64, cpf)]])),
                   "best_energy": best_e, "best_chain": best_chain, "ranks_gathered": len(per_rank),
                   "marginal_frames_gathered": frames_gathered,
                   "chain_kernel_variant": ctx.last_variant()},
        **({"rehearsal": "MCMC_BENCH_ONE_GPU: %d ranks on ONE GPU over gloo -- sharding, gathers and checks as on N GPUs; "
                         "the rate is not a measurement" % world} if one_gpu else {}),
        "statistics": {"visit_capacity": ctx.visit_capacity(), "overflow_sum": int(overflow[0]), "overflow_max": int(overflow[1]),
                       "measured_steps_per_chain": int(total_steps),
                       "note": "Statistics::update visits that found the per-chain visit table full; 0 = every state counted"},
    }
    ctx_variant = ctx.last_variant()
    if rank == 0:
        out["chain_kernel"] = {"kernel": "chain_run_kernel", "ms_per_launch": k3_ms,
                               "proposals_per_launch": n_local * a.iters_per_step,
                               "us_per_proposal_per_chain": k3_ms * 1e3 / a.iters_per_step,
                               "bound": "vector-pipe issue (LDS-resident BFS, four waves per chain in step order); not an HBM claim",
                               "bound_model": k3_bound_model(k3_ms * 1e3 / a.iters_per_step, ctx_variant)}
        out["frame_kernels_ms"] = {"sp_moments": ctx.last_kernel_ms(host.K_MOMENTS),
                                   "local_fit": ctx.last_kernel_ms(host.K_LOCAL_FIT),
                                   "edge_prob": ctx.last_kernel_ms(host.K_EDGE_PROB)}
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()      # nothing below communicates; the other ranks are done
    if rank == 0:
        c2s = scenes if cfg != "c3" else []
        if (not a.no_roofline or (world == 1 and cfg == "c2" and not a.no_c5)) and len(c2s) < 4:
            c2s = c2s + [make_scene(width=640, height=480, n_planes=8, n_superpixels=500, seed=100 + i)
                         for i in range(4 - len(c2s))]
        # the roofline kernel first: measured after the secondary lines (which allocate and free gigabytes: 4096 multiview
        # chains, 2048 C5 chains) a few of its hundred launches take 100+ us (same median, mean 1.5 us higher)
        if not a.no_roofline:
            out["roofline"] = k1_roofline(host, c2s, a.roofline_frames, a.roofline_reps, local, a.roofline_warmup)
        if world == 1 and cfg == "c2" and not a.no_multiview:
            out["multiview"] = multiview_line(host, local, a.mv_chains, a.mv_steps)
        if world == 1 and cfg == "c2" and not a.no_c5:
            out["c5_share"] = c5_share_line(host, c2s, local)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, seed, a.cpu_seconds)
        if not a.no_cpu_baseline:
            # the CPU baseline and the GPU figure are the same computation -- the first chain of
            # every rank re-run on the oracle with the same global id (what a single-GPU run would give it)
            chk = same_computation_check(scenes, cfg, cpf, seed, total_steps, firsts)
            if world == 1:
                out["cpu_baseline"]["same_computation"] = chk
            else:
                out["cross_rank_check"] = chk
            assert chk["identical_to_gpu"], "GPU chains differ from the oracle"
        if world == 1 and cfg == "c2" and not a.no_cpu_baseline:
            ncpu = os.cpu_count() or 1
            # the box's CPU share is what OMP_NUM_THREADS says (16 per GPU there); without it, every CPU up to 64
            nthr = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "0")) or min(ncpu, 64), ncpu))
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(sc, seed, min(a.cpu_seconds, 6.0), nthr)
            out["cpu_baseline_all_cores"]["host_cpus"] = ncpu
        print(json.dumps(out))


if __name__ == "__main__":
    main()
