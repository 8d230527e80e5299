"""CPU tests of the boundary: the C-ABI library loads without a GPU and exports
every symbol include/mcmc_amd.h declares; argument checking that needs no device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = []
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not h.endswith(".h"):
            continue
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms += re.findall(r"\b(mcmc_[a-z0-9_]+)\s*\(", text)
    return sorted(set(syms))


def test_library_exports_every_declared_symbol():
    from mcmc_amd import host
    L = host.load()
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(L, s), f"libmcmc_amd.so does not export {s}"


def test_default_params_and_version():
    from mcmc_amd import host
    L = host.load()
    p = host.Params()
    L.mcmc_default_params(C.byref(p))
    # include/MyMath.h:28-32, Segmentation.cpp:16
    assert (p.width, p.height, p.disparity_variance, p.new_segment_prob) == (640, 480, 7.0, 0.8)
    assert b"gfx950" in L.mcmc_version()


def test_null_arguments_are_rejected_without_a_device():
    from mcmc_amd import host
    L = host.load()
    assert L.mcmc_ctx_create(0, None, None) == -1
    assert L.mcmc_ctx_destroy(None) == -1
    assert L.mcmc_frames_alloc(None, 1, 1, 1) == -1
    assert L.mcmc_chains_run(None, 0, 1) == -3
    assert L.mcmc_last_kernel_ms(None, 0) < 0


def test_product_does_not_touch_the_oracle():
    # the oracle is test infrastructure: nothing shipped may import, include or link it
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "mcmc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                t = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"oracle_lib|libmcmc_oracle|oracle/|orc_[a-z]", t):
                    bad.append(f)
    for f in os.listdir(os.path.join(ROOT, "include")):
        t = open(os.path.join(ROOT, "include", f), errors="ignore").read()
        if re.search(r"oracle_lib|libmcmc_oracle|orc_[a-z]", t):
            bad.append(f)
    assert not bad, bad
