"""ctypes loader for the CPU oracle (oracle/_build/libmcmc_oracle.so).

Test infrastructure only: the product package (mcmc_amd/) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# MCMC_ORACLE_SO: an alternative build of the same sources (e.g. one compiled with -fsanitize=undefined)
SO = os.environ.get("MCMC_ORACLE_SO") or os.path.join(ROOT, "oracle", "_build", "libmcmc_oracle.so")

_lib = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C")
i16p = np.ctypeslib.ndpointer(np.int16, flags="C")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C")


def build():
    """Compile the oracle with gcc if the .so is missing or stale."""
    if os.environ.get("MCMC_ORACLE_SO"):
        return
    srcs = [os.path.join(ROOT, "oracle", f) for f in os.listdir(os.path.join(ROOT, "oracle"))
            if f.endswith((".c", ".cpp"))]
    if os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(s) for s in srcs):
        return
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(SO)
    L.orc_philox4x32_10.argtypes = [u32p, u32p, u32p]
    L.orc_det_log.restype = C.c_double; L.orc_det_log.argtypes = [C.c_double]
    L.orc_det_exp.restype = C.c_double; L.orc_det_exp.argtypes = [C.c_double]
    L.orc_moments.argtypes = [i16p, u16p, C.c_int, C.c_int, C.c_int, i64p]
    L.orc_local_fit.argtypes = [i64p, C.c_int, C.c_int, C.c_int, C.c_double, f64p, u8p, f64p]
    L.orc_segment_probability.argtypes = [f64p, C.c_int, u32p, C.c_int, C.c_double,
                                          C.POINTER(C.c_double), f64p, C.POINTER(C.c_double)]
    L.orc_edge_probs.argtypes = [f64p, u8p, C.c_int, u32p, u32p, C.c_double, f64p]
    L.orc_chains_create.restype = C.c_void_p
    L.orc_chains_create.argtypes = [C.c_int, f64p, u8p, u32p, u32p, f64p, C.c_double, C.c_double,
                                    C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
    L.orc_chains_free.argtypes = [C.c_void_p]
    L.orc_chains_run.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.orc_chains_get_labels.argtypes = [C.c_void_p, C.c_int, C.c_int, u16p]
    L.orc_chains_get_scalars.argtypes = [C.c_void_p] + [C.c_void_p] * 10
    L.orc_chains_get_visits.argtypes = [C.c_void_p, u64p, u64p, i32p]
    L.orc_chains_get_marginals.argtypes = [C.c_void_p, u64p]
    L.orc_chains_get_last_ratios.argtypes = [C.c_void_p, f64p, f64p]
    L.orc_chains_recompute_energy.argtypes = [C.c_void_p, f64p]
    L.orc_mix64.restype = C.c_uint64; L.orc_mix64.argtypes = [C.c_uint64]
    L.orc_wave_sum.restype = C.c_double; L.orc_wave_sum.argtypes = [f64p]
    L.orc_wave_prod.restype = C.c_double; L.orc_wave_prod.argtypes = [f64p]
    L.orc_det_sqrt.restype = C.c_double; L.orc_det_sqrt.argtypes = [C.c_double]
    L.orc_det_rsqrt.restype = C.c_double; L.orc_det_rsqrt.argtypes = [C.c_double]
    L.orc_mv_vertex_moments.argtypes = [f32p, u32p, C.c_int, f64p]
    L.orc_mv_fit_segment.argtypes = [C.c_int, f32p, u32p, f64p, u32p, C.c_int, f64p,
                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.orc_mv_pmerge.restype = C.c_double
    L.orc_mv_pmerge.argtypes = [f32p, f32p, f32p, C.c_int, C.c_double]
    L.orc_mv_chains_create.restype = C.c_void_p
    L.orc_mv_chains_create.argtypes = [C.c_int, f32p, u32p, f64p, u32p, u32p, f64p, C.c_int, C.c_uint32,
                                       C.c_uint32, C.c_int, C.c_double]
    L.orc_mv_chains_free.argtypes = [C.c_void_p]
    L.orc_mv_chains_run.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    L.orc_mv_chains_get_labels.argtypes = [C.c_void_p, C.c_int, u32p]
    L.orc_mv_chains_get_scalars.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.orc_mv_chains_get_planes.argtypes = [C.c_void_p, C.c_int, f64p]
    L.orc_mv_world_graph.restype = C.c_longlong
    L.orc_mv_world_graph.argtypes = [C.c_int, C.c_int, f32p, f32p, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                     C.c_double, u32p, C.c_void_p, C.c_void_p]
    L.orc_bilateral.argtypes = [i16p, u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, u8p]
    L.orc_create_edges.restype = C.c_longlong
    L.orc_create_edges.argtypes = [i16p, u8p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, i32p, i32p, f32p]
    L.orc_segment_graph.argtypes = [C.c_int, C.c_longlong, i32p, i32p, f32p, C.c_float, C.c_int, i32p]
    L.orc_superpixels.argtypes = [i32p, C.c_int, C.c_int, u16p, C.POINTER(C.c_int), u32p, u32p, C.c_longlong,
                                  C.POINTER(C.c_uint32)]
    _lib = L
    return L


# ---------------------------------------------------------------- convenience
def moments(disp, label, S):
    H, W = disp.shape
    out = np.zeros((S, 10), np.int64)
    lib().orc_moments(np.ascontiguousarray(disp, np.int16), np.ascontiguousarray(label, np.uint16),
                      W, H, S, out)
    return out


def local_fit(mom, W, H, var=7.0):
    S = mom.shape[0]
    fit = np.zeros((S, 16), np.float64)
    valid = np.zeros(S, np.uint8)
    lerr = np.zeros(S, np.float64)
    lib().orc_local_fit(np.ascontiguousarray(mom), W, H, S, var, fit, valid, lerr)
    return fit, valid, lerr


def segment_probability(fit, members, var=7.0):
    p = C.c_double(); e = C.c_double()
    th = np.zeros(3, np.float64)
    m = np.ascontiguousarray(members, np.uint32)
    st = lib().orc_segment_probability(np.ascontiguousarray(fit), fit.shape[0], m, m.size, var,
                                       C.byref(p), th, C.byref(e))
    return st, p.value, th, e.value


def edge_probs(fit, valid, row_ptr, col, var=7.0):
    q = np.zeros(col.size, np.float64)
    st = lib().orc_edge_probs(np.ascontiguousarray(fit), np.ascontiguousarray(valid), fit.shape[0],
                              np.ascontiguousarray(row_ptr, np.uint32),
                              np.ascontiguousarray(col, np.uint32), var, q)
    return st, q


class Chains:
    """Oracle sampler over one frame (mirrors mcmc_amd.host.Chains)."""

    def __init__(self, fit, valid, row_ptr, col, q, n_chains, seed, chain_offset=0,
                 var=7.0, new_seg_prob=0.8, hcap=4096, lmax=64):
        self.S = fit.shape[0]
        self.n = n_chains
        self.hcap, self.lmax = hcap, lmax
        # keep references alive: the oracle borrows these buffers
        self._keep = [np.ascontiguousarray(fit), np.ascontiguousarray(valid),
                      np.ascontiguousarray(row_ptr, np.uint32), np.ascontiguousarray(col, np.uint32),
                      np.ascontiguousarray(q)]
        self.h = lib().orc_chains_create(self.S, *self._keep, var, new_seg_prob, n_chains, seed,
                                         chain_offset, hcap, lmax)

    def run(self, burn_in, iters, trace=False):
        tr = np.zeros((self.n, burn_in + iters), np.uint8) if trace else None
        st = lib().orc_chains_run(self.h, burn_in, iters,
                                  tr.ctypes.data_as(C.c_void_p) if trace else None)
        return st, tr

    def labels(self, which=0, canonical=True):
        out = np.zeros((self.n, self.S), np.uint16)
        lib().orc_chains_get_labels(self.h, which, int(canonical), out)
        return out

    def scalars(self):
        n = self.n
        d = dict(energy=np.zeros(n), best_energy=np.zeros(n), state_hash=np.zeros(n, np.uint64),
                 trace_hash=np.zeros(n, np.uint64), n_accept=np.zeros(n, np.uint64),
                 n_case=np.zeros((n, 3), np.uint64), status=np.zeros(n, np.int32),
                 top_count=np.zeros(n, np.uint64), top_hash=np.zeros(n, np.uint64),
                 overflow=np.zeros(n, np.uint64))
        lib().orc_chains_get_scalars(self.h, *[v.ctypes.data_as(C.c_void_p) for v in d.values()])
        return d

    def visits(self):
        hh = np.zeros((self.n, self.hcap), np.uint64)
        cc = np.zeros((self.n, self.hcap), np.uint64)
        nn = np.zeros(self.n, np.int32)
        lib().orc_chains_get_visits(self.h, hh, cc, nn)
        return [dict(zip(hh[k, :nn[k]].tolist(), cc[k, :nn[k]].tolist())) for k in range(self.n)]

    def marginals(self):
        out = np.zeros((self.S, self.lmax), np.uint64)
        lib().orc_chains_get_marginals(self.h, out)
        return out

    def last_ratios(self):
        t = np.zeros(self.n); p = np.zeros(self.n)
        lib().orc_chains_get_last_ratios(self.h, t, p)
        return t, p

    def recompute_energy(self):
        out = np.zeros(self.n)
        lib().orc_chains_recompute_energy(self.h, out)
        return out

    def __del__(self):
        try:
            lib().orc_chains_free(self.h)
        except Exception:
            pass


# ---------------------------------------------------------------- Path B
def mv_vertex_moments(points, pt_ptr):
    V = pt_ptr.size - 1
    out = np.zeros((V, 10))
    lib().orc_mv_vertex_moments(np.ascontiguousarray(points, np.float32), np.ascontiguousarray(pt_ptr, np.uint32), V, out)
    return out


def mv_fit_segment(points, pt_ptr, vm, members):
    V = pt_ptr.size - 1
    pl = np.zeros(5); p = C.c_double(); e = C.c_double()
    m = np.ascontiguousarray(members, np.uint32)
    st = lib().orc_mv_fit_segment(V, np.ascontiguousarray(points, np.float32), np.ascontiguousarray(pt_ptr, np.uint32),
                                  np.ascontiguousarray(vm), m, m.size, pl, C.byref(p), C.byref(e))
    assert st == 0
    return pl, p.value, e.value


def mv_pmerge(t1, t2, w, temperature=8.0):
    t1 = np.ascontiguousarray(t1, np.float32); t2 = np.ascontiguousarray(t2, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    return lib().orc_mv_pmerge(t1, t2, w, t1.size, temperature)


class MvChains:
    """Oracle Path-B sampler (mirrors mcmc_amd.host.Multiview)."""

    def __init__(self, mv, n_chains, seed, chain_offset=0, max_new_segments=1024, split_prob=0.10):
        self.V = mv.n_vertices; self.n = n_chains
        self._keep = [np.ascontiguousarray(mv.points, np.float32), np.ascontiguousarray(mv.pt_ptr, np.uint32), None,
                      np.ascontiguousarray(mv.row_ptr, np.uint32), np.ascontiguousarray(mv.col, np.uint32),
                      np.ascontiguousarray(mv.q, np.float64)]
        self._keep[2] = mv_vertex_moments(self._keep[0], self._keep[1])
        self.h = lib().orc_mv_chains_create(self.V, *self._keep, n_chains, seed, chain_offset, max_new_segments,
                                            split_prob)

    def run(self, steps):
        f = np.zeros((self.n, steps), np.uint8); a = np.zeros((self.n, steps))
        lib().orc_mv_chains_run(self.h, steps, f.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p))
        return f, a

    def labels(self, which=0):
        out = np.zeros((self.n, self.V), np.uint32)
        lib().orc_mv_chains_get_labels(self.h, which, out)
        return out

    def scalars(self):
        n = self.n
        d = dict(n_accept=np.zeros(n, np.uint64), n_split=np.zeros(n, np.uint64), n_skip=np.zeros(n, np.uint64),
                 trace_hash=np.zeros(n, np.uint64), n_segments=np.zeros(n, np.int32), status=np.zeros(n, np.int32))
        lib().orc_mv_chains_get_scalars(self.h, *[v.ctypes.data_as(C.c_void_p) for v in d.values()])
        return d

    def planes(self, chain, n_segments):
        out = np.zeros((n_segments, 5))
        lib().orc_mv_chains_get_planes(self.h, chain, out)
        return out

    def __del__(self):
        try:
            lib().orc_mv_chains_free(self.h)
        except Exception:
            pass


# ---- front end (oracle/frontend_oracle.c, frontend_fh_oracle.cpp) --------------------------
def bilateral(disp, colors, window=9, spatial_stdev=5.0, disparity_stdev=3.0):
    H, W = disp.shape
    out = np.zeros((H, W, 3), np.uint8)
    rc = lib().orc_bilateral(np.ascontiguousarray(disp, np.int16), np.ascontiguousarray(colors, np.uint8), W, H,
                             window, spatial_stdev, disparity_stdev, out)
    return rc, out


def create_edges(disp, smoothed, color_factor=1.0, disparity_factor=1.0, distance_factor=1.0):
    H, W = disp.shape
    a = np.zeros(4 * W * H, np.int32); b = np.zeros(4 * W * H, np.int32); w = np.zeros(4 * W * H, np.float32)
    n = lib().orc_create_edges(np.ascontiguousarray(disp, np.int16), np.ascontiguousarray(smoothed, np.uint8), W, H,
                               color_factor, disparity_factor, distance_factor, a, b, w)
    return a[:n].copy(), b[:n].copy(), w[:n].copy()


def segment_graph(nv, a, b, w, c, tie_order=0):
    """tie_order 0: equal weights in creation order (the product's device sort); 1: this C++ runtime's std::sort (the product's default)."""
    root = np.zeros(nv, np.int32)
    sets = lib().orc_segment_graph(nv, a.size, np.ascontiguousarray(a, np.int32), np.ascontiguousarray(b, np.int32),
                                   np.ascontiguousarray(w, np.float32), float(c), int(tie_order), root)
    return sets, root


def superpixels(root, W, H, col_cap=None):
    label = np.zeros((H, W), np.uint16)
    S = C.c_int(0); e2 = C.c_uint32(0)
    cap = col_cap if col_cap is not None else 16 * W * H
    rp = np.zeros(W * H + 1, np.uint32); col = np.zeros(cap, np.uint32)
    rc = lib().orc_superpixels(np.ascontiguousarray(root, np.int32), W, H, label, C.byref(S), rp, col, cap, C.byref(e2))
    return rc, label, S.value, rp[:S.value + 1].copy(), col[:e2.value].copy()


def mv_world_graph(theta, weights, n0, in_row_ptr=None, in_col=None, in_thr=0.01, thr=0.2, T=8.0):
    theta = np.ascontiguousarray(theta, np.float32); weights = np.ascontiguousarray(weights, np.float32)
    V, dim = theta.shape
    rp = np.zeros(V + 1, np.uint32)
    irp = None if in_row_ptr is None else np.ascontiguousarray(in_row_ptr, np.uint32)
    ic = None if in_col is None else np.ascontiguousarray(in_col, np.uint32)
    pi = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    n = lib().orc_mv_world_graph(V, dim, theta, weights, n0, pi(irp), pi(ic), in_thr, thr, T, rp, None, None)
    col = np.zeros(n, np.uint32); prob = np.zeros(n, np.float64)
    lib().orc_mv_world_graph(V, dim, theta, weights, n0, pi(irp), pi(ic), in_thr, thr, T, rp, pi(col), pi(prob))
    return rp, col, prob
