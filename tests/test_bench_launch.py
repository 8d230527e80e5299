"""bench.py's launch logic, without a GPU: `--gpus N` must really run N ranks (it silently measured one).
launch_plan() is a pure function of (args, argv, environment); the parent never touches the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def test_single_gpu_runs_in_process():
    a = bench.parse([])
    assert a.gpus == 1 and bench.launch_plan(a, [], {}) is None
    a = bench.parse(["--gpus", "1", "--steps", "3"])
    assert bench.launch_plan(a, ["--gpus", "1", "--steps", "3"], {"HOME": "/x"}) is None


def test_multi_gpu_without_launcher_spawns_torchrun():
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "c3"]
    a = bench.parse(argv)
    cmd, env = bench.launch_plan(a, argv, {"PATH": "/bin"})
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == argv                      # the ranks get exactly the flags the parent got
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["PATH"] == "/bin"
    assert "WORLD_SIZE" not in env and "RANK" not in env


def test_rank_started_by_torchrun_does_not_respawn():
    argv = ["--gpus", "8"]
    a = bench.parse(argv)
    assert bench.launch_plan(a, argv, {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}) is None


def test_world_size_must_match_gpus():
    a = bench.parse(["--gpus", "8"])
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(a, ["--gpus", "8"], {"WORLD_SIZE": "2", "RANK": "0"})
    assert e.value.code == 2
    a = bench.parse(["--gpus", "1"])
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(a, [], {"WORLD_SIZE": "4", "RANK": "0"})
    assert e.value.code == 2


def test_parent_relays_rank0_line_and_exit_code(tmp_path):
    fake = tmp_path / "fake_ranks.py"
    fake.write_text("import sys, json\nprint('noise from a rank')\n"
                    "print(json.dumps({'metric': 'm', 'value': 1.5, 'n_gpus': 2}))\nsys.exit(int(sys.argv[1]))\n")
    run = ("import sys; sys.path.insert(0, %r); import bench; bench.run_children([sys.executable, %r, sys.argv[1]], None)"
           % (ROOT, str(fake)))
    ok = subprocess.run([sys.executable, "-c", run, "0"], capture_output=True, text=True)
    assert ok.returncode == 0
    assert json.loads(ok.stdout.strip()) == {"metric": "m", "value": 1.5, "n_gpus": 2}   # exactly ONE line on stdout
    assert "noise from a rank" in ok.stderr
    bad = subprocess.run([sys.executable, "-c", run, "7"], capture_output=True, text=True)
    assert bad.returncode == 7 and bad.stdout.strip() == ""


def _run_bench(args, env_extra):
    env = dict(os.environ); env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["c3", "c5"])
def test_sharded_configs_through_rccl_one_rank(cfg):
    """The multi-GPU modes of bench.py at reduced size with the process group really initialised (backend nccl =
    RCCL, one rank: this box has one GPU): chains / frames are placed by shard_range, the best labeling and the
    marginals go through the collectives from device memory, and the first chain is re-run on the oracle."""
    port = 29600 + (os.getpid() % 300) + (0 if cfg == "c3" else 1)
    args = ["--config", cfg, "--steps", "1", "--warmup", "1", "--iters-per-step", "60", "--no-roofline", "--cpu-seconds", "1"]
    args += ["--total-chains", "64"] if cfg == "c3" else ["--total-frames", "8", "--chains-per-frame", "16", "--distinct-scenes", "2"]
    out = _run_bench(args, {"MCMC_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                            "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    c = out["config"]
    assert out["n_gpus"] == 1 and c["name"] == cfg and c["ranks_gathered"] == 1
    assert out["scaling"] == "strong" and out["value"] > 0
    assert c["chains_total"] == (64 if cfg == "c3" else 8 * 16)
    assert c["marginal_frames_gathered"] == (1 if cfg == "c3" else 8)
    assert out["cpu_baseline"]["same_computation"]["identical_to_gpu"] is True
    assert out["statistics"]["overflow_sum"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n", [("c2", 2), ("c3", 2), ("c5", 2), ("c3", 4), ("c5", 4)])
def test_two_ranks_on_one_gpu_rehearsal(cfg, n):
    """`bench.py --gpus N` (two and four ranks) as the driver would start it on a node -- the parent spawns torch.distributed.run --
    on a box with ONE GPU (MCMC_BENCH_ONE_GPU: both ranks compute on device 0, collectives over gloo; RCCL refuses two
    ranks on a device).  Everything but the transport is what an N-GPU run does: chains / frames placed by global id,
    barrier and max-over-ranks timing, the best labeling and the marginals gathered from both ranks and checked against
    the step counts, rank 0's first chain re-run on the oracle.  The rate it prints measures nothing."""
    args = ["--gpus", str(n), "--config", cfg, "--steps", "1", "--warmup", "1", "--iters-per-step", "60", "--no-roofline",
            "--cpu-seconds", "1"]
    if cfg == "c2":
        args += ["--chains", "32", "--no-multiview"]
    elif cfg == "c3":
        args += ["--total-chains", "64"]
    else:
        args += ["--total-frames", "8", "--chains-per-frame", "16", "--distinct-scenes", "2"]
    out = _run_bench(args, {"MCMC_BENCH_ONE_GPU": "1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    c = out["config"]
    assert out["n_gpus"] == n and c["name"] == cfg and c["ranks_gathered"] == n and "rehearsal" in out
    assert out["scaling"] == ("weak" if cfg == "c2" else "strong") and out["value"] > 0
    assert c["chains_total"] == {"c2": 32 * n, "c3": 64, "c5": 128}[cfg]
    assert c["chains_per_gpu"] == c["chains_total"] // n
    assert c["marginal_frames_gathered"] == (8 if cfg == "c5" else 1)
    chk = out["cross_rank_check"]                 # the first chain of EACH rank against the oracle, by global chain id
    assert chk["identical_to_gpu"] is True and len(chk["chains"]) == n
    assert sorted(r["chain"] for r in chk["chains"]) == [k * c["chains_per_gpu"] for k in range(n)]
    assert out["statistics"]["overflow_sum"] == 0


def test_c5_refuses_uneven_frame_shares():
    a = bench.parse(["--gpus", "3", "--config", "c5", "--total-frames", "8"])
    assert a.total_frames % 3 != 0      # main() exits 2 before any work; the check itself needs ranks, see bench.py
