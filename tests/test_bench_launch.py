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
