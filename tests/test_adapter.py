"""The C++ adapter (include/mcmc_amd_adapter.hpp): compiles against the C ABI (CPU),
and on the GPU the PROBLEM-concept path, the batched path and the oracle agree."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "adapter_smoke")


def build_exe():
    src = os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp")
    hdrs = [os.path.join(ROOT, "include", h) for h in ("mcmc_amd.h", "mcmc_amd_adapter.hpp")]
    if os.path.exists(EXE) and all(os.path.getmtime(EXE) >= os.path.getmtime(p) for p in [src] + hdrs):
        return EXE
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                           "-L", os.path.join(ROOT, "mcmc_amd"), "-lmcmc_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "mcmc_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_adapter_compiles_and_links():
    from mcmc_amd import host
    host.load()
    assert os.path.exists(build_exe())


@pytest.mark.gpu
def test_adapter_matches_batched_path_and_oracle(tmp_path, scene_c2_small):
    from tests import oracle_lib as O
    sc = scene_c2_small
    exe = EXE if os.path.exists(EXE) else build_exe()
    path = tmp_path / "scene.bin"
    with open(path, "wb") as f:
        np.array([sc.width, sc.height, sc.n_superpixels, sc.col.size], np.int32).tofile(f)
        sc.disp.astype(np.int16).tofile(f); sc.label.astype(np.uint16).tofile(f)
        sc.row_ptr.astype(np.uint32).tofile(f); sc.col.astype(np.uint32).tofile(f)
    steps, seed = 120, 4
    out = subprocess.run([exe, str(path), str(steps), str(seed)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = dict((l.split()[0], l.split()) for l in out.stdout.strip().splitlines())
    S = sc.n_superpixels
    fit, valid, _ = O.local_fit(O.moments(sc.disp, sc.label, S), sc.width, sc.height)
    _, q = O.edge_probs(fit, valid, sc.row_ptr, sc.col)
    orc = O.Chains(fit, valid, sc.row_ptr, sc.col, q, 1, seed)
    orc.run(0, steps)
    s = orc.scalars()
    want_hash = int(s["state_hash"][0]); want_trace = int(s["trace_hash"][0]); want_acc = int(s["n_accept"][0])
    # both adapter paths reproduce the oracle's chain 0
    for key in ("problem", "batched"):
        assert int(lines[key][2]) == want_hash, key
        assert int(lines[key][6]) == want_trace, key
    assert int(lines["batched"][4]) == want_acc
    # Problem::propose hands over the real ratios of the step (Problem.cpp:213-217), bit for bit the oracle's
    t, p = orc.last_ratios()
    assert float.fromhex(lines["ratios"][1]) == t[0] and float.fromhex(lines["ratios"][2]) == p[0]
    assert np.isfinite(t[0]) and np.isfinite(p[0]) and p[0] > 0
