"""ctypes binding of include/mcmc_amd.h (libmcmc_amd.so).

Thin: every method is one C-ABI call plus numpy buffer plumbing.  There is no
CPU fallback -- if the HIP library is missing or a call fails this raises.
The C++ adapter with the reference's Problem / Segmentation / MCMC<PROBLEM>
shapes lives in include/mcmc_amd_adapter.hpp; this module is what tests/ and
bench.py drive.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCMC_AMD_LIB") or os.path.join(HERE, "libmcmc_amd.so")   # MCMC_AMD_LIB: tuning builds only

NOLABEL = 0xFFFF
K_MOMENTS, K_LOCAL_FIT, K_EDGE_PROB, K_CHAIN = 0, 1, 2, 3

ERRORS = {-1: "MCMC_E_BADARG", -2: "MCMC_E_DEVICE", -3: "MCMC_E_STATE", -4: "MCMC_E_NAN_RATIO",
          -5: "MCMC_E_INF_ENERGY", -6: "MCMC_E_TOO_LARGE", -7: "MCMC_E_NOSUPERPIXELS", -8: "MCMC_E_RANGE",
          -9: "MCMC_E_OVERFLOW"}


class McmcError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"{ERRORS.get(code, code)}: {text}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("disparity_variance", C.c_double),
                ("new_segment_prob", C.c_double), ("marginal_bins", C.c_int32),
                ("visit_capacity", C.c_int32)]


_lib = None


class OversegParams(C.Structure):
    """mcmc_overseg_params of include/mcmc_amd.h (OverSegmentationParameters)."""
    _fields_ = [("window_size", C.c_int), ("spatial_stdev", C.c_double), ("disparity_stdev", C.c_double),
                ("weight_threshold", C.c_int), ("color_factor", C.c_double), ("disparity_factor", C.c_double),
                ("distance_factor", C.c_double)]


def _ptr(a, dtype):
    if a is None:
        return None
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
    return a.ctypes.data_as(C.c_void_p)


def load():
    """Load libmcmc_amd.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                          " (hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i, u32, u64, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_double
    sig = {
        "mcmc_default_params": (None, [C.POINTER(Params)]),
        "mcmc_ctx_create": (i, [i, C.POINTER(Params), C.POINTER(vp)]),
        "mcmc_ctx_destroy": (i, [vp]),
        "mcmc_last_error": (C.c_char_p, [vp]),
        "mcmc_ctx_set_stream": (i, [vp, vp]),
        "mcmc_ctx_synchronize": (i, [vp]),
        "mcmc_version": (C.c_char_p, []),
        "mcmc_frames_alloc": (i, [vp, i, i, i]),
        "mcmc_frame_upload": (i, [vp, i, vp, vp, i]),
        "mcmc_frame_upload_device": (i, [vp, i, vp, vp, i]),
        "mcmc_frame_set_graph": (i, [vp, i, vp, vp, u32]),
        "mcmc_frames_moments": (i, [vp, i, i]),
        "mcmc_frames_edge_probs": (i, [vp, i, i]),
        "mcmc_frame_get_moments": (i, [vp, i, vp]),
        "mcmc_frame_get_local_fit": (i, [vp, i, vp, vp, vp]),
        "mcmc_frame_get_edge_probs": (i, [vp, i, vp]),
        "mcmc_segment_probability": (i, [vp, i, vp, i, C.POINTER(dbl), vp, C.POINTER(dbl)]),
        "mcmc_segment_probability_noalloc": (i, [vp, i, vp, i, C.POINTER(dbl), vp, C.POINTER(dbl)]),
        "mcmc_chains_init": (i, [vp, i, i, u32, u32]),
        "mcmc_chains_run": (i, [vp, u64, u64]),
        "mcmc_chains_set_trace": (i, [vp, u64]),
        "mcmc_chains_get_trace": (i, [vp, vp]),
        "mcmc_chains_get_labels": (i, [vp, i, i, vp]),
        "mcmc_chains_get_scalars": (i, [vp] + [vp] * 10),
        "mcmc_chains_get_best": (i, [vp, i, i, C.POINTER(dbl), C.POINTER(u64), C.POINTER(u64), vp]),
        "mcmc_chains_best_record_device": (i, [vp, i, i, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "mcmc_frame_marginals_device": (i, [vp, i, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "mcmc_chains_get_visits": (i, [vp, vp, vp]),
        "mcmc_chains_visit_capacity": (i, [vp]),
        "mcmc_chains_reserve_visits": (i, [vp, u64]),
        "mcmc_chains_get_last_ratios": (i, [vp, vp, vp]),
        "mcmc_frame_get_marginals": (i, [vp, i, vp]),
        "mcmc_chains_recompute_energy": (i, [vp, vp]),
        "mcmc_last_kernel_ms": (dbl, [vp, i]),
        "mcmc_set_timing": (i, [vp, i]),
        "mcmc_ctx_set_option": (i, [vp, C.c_char_p, i]),
        "mcmc_chains_get_profile": (i, [vp, vp]),
        "mcmc_selftest_math": (i, [vp, vp, vp, i, vp, vp, vp]),
        "mcmc_selftest_philox": (i, [vp, vp, vp, vp]),
        "mcmc_selftest_wave_sums": (i, [vp, vp, i, vp, vp]),
        "mcmc_mv_create": (i, [i, C.POINTER(vp)]),
        "mcmc_mv_destroy": (i, [vp]),
        "mcmc_mv_last_error": (C.c_char_p, [vp]),
        "mcmc_mv_synchronize": (i, [vp]),
        "mcmc_mv_set_graph": (i, [vp, i, vp, vp, vp, vp, vp, u32]),
        "mcmc_mv_vertex_moments": (i, [vp]),
        "mcmc_mv_get_vertex_moments": (i, [vp, vp]),
        "mcmc_mv_fit_segment": (i, [vp, vp, i, vp, C.POINTER(dbl), C.POINTER(dbl)]),
        "mcmc_mv_pmerge": (i, [vp, vp, vp, vp, i, i, dbl, vp]),
        "mcmc_mv_build_world_graph": (i, [vp, i, i, vp, vp, i, vp, vp, dbl, dbl, dbl, vp, vp, vp, u32, C.POINTER(u32)]),
        "mcmc_mv_chains_init": (i, [vp, i, u32, u32, i, dbl]),
        "mcmc_mv_chains_run": (i, [vp, u64]),
        "mcmc_mv_set_option": (i, [vp, C.c_char_p, i]),
        "mcmc_mv_chains_set_trace": (i, [vp, u64]),
        "mcmc_mv_chains_get_trace": (i, [vp, vp, vp]),
        "mcmc_mv_chains_get_labels": (i, [vp, i, vp]),
        "mcmc_mv_chains_get_scalars": (i, [vp] + [vp] * 6),
        "mcmc_mv_chains_get_planes": (i, [vp, i, vp]),
        "mcmc_mv_chains_get_profile": (i, [vp, vp]),
        "mcmc_chains_last_variant": (i, [vp]),
        "mcmc_mv_last_run_ms": (dbl, [vp]),
        "mcmc_overseg_default_params": (None, [C.POINTER(OversegParams)]),
        "mcmc_overseg_create": (i, [i, i, i, C.POINTER(vp)]),
        "mcmc_overseg_destroy": (i, [vp]),
        "mcmc_overseg_last_error": (C.c_char_p, [vp]),
        "mcmc_overseg_run": (i, [vp, vp, vp, C.POINTER(OversegParams), C.POINTER(i), C.POINTER(u32)]),
        "mcmc_overseg_get_smoothed": (i, [vp, vp]),
        "mcmc_overseg_get_edges": (C.c_longlong, [vp, vp, vp, vp]),
        "mcmc_overseg_get_roots": (i, [vp, vp]),
        "mcmc_overseg_get_labels": (i, [vp, vp]),
        "mcmc_overseg_get_graph": (i, [vp, vp, vp]),
        "mcmc_overseg_device_buffers": (i, [vp, C.POINTER(vp), C.POINTER(vp)]),
        "mcmc_overseg_last_ms": (i, [vp] + [C.POINTER(C.c_float)] * 4),
        "mcmc_overseg_last_sort_ms": (i, [vp] + [C.POINTER(C.c_float)] * 2),
        "mcmc_overseg_set_option": (i, [vp, C.c_char_p, i]),
        "mcmc_overseg_last_tie_count": (C.c_longlong, [vp]),
        "mcmc_overseg_batch_create": (i, [i, i, i, i, C.POINTER(vp)]),
        "mcmc_overseg_batch_destroy": (i, [vp]),
        "mcmc_overseg_batch_last_error": (C.c_char_p, [vp]),
        "mcmc_overseg_batch_set_option": (i, [vp, C.c_char_p, i]),
        "mcmc_overseg_batch_run": (i, [vp, i, vp, vp, C.POINTER(OversegParams), vp, vp]),
        "mcmc_overseg_batch_get_graph": (i, [vp, i, vp, vp]),
        "mcmc_overseg_batch_device_buffers": (i, [vp, i, C.POINTER(vp), C.POINTER(vp)]),
        "mcmc_overseg_batch_get_labels": (i, [vp, i, vp]),
        "mcmc_overseg_batch_tie_count": (C.c_longlong, [vp, i]),
        "mcmc_overseg_batch_last_wall_ms": (C.c_float, [vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


class Context:
    """One context per (thread, GPU): frames resident in HBM plus chains over them."""

    def __init__(self, width=640, height=480, device=0, disparity_variance=7.0, new_segment_prob=0.8,
                 marginal_bins=64, visit_capacity=0):
        L = load()
        self.L = L
        self.p = Params(width, height, disparity_variance, new_segment_prob, marginal_bins, visit_capacity)
        self.h = C.c_void_p()
        rc = L.mcmc_ctx_create(device, C.byref(self.p), C.byref(self.h))
        if rc != 0:
            raise McmcError(rc, "mcmc_ctx_create failed (no GPU, or bad parameters)")
        self.width, self.height = width, height
        self.n_frames = self.s_max = self.e2_max = 0
        self.S = []
        self.E2 = []
        self.n_chains = 0

    def _ck(self, rc):
        if rc != 0:
            raise McmcError(rc, self.L.mcmc_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.mcmc_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- frames
    def frames_alloc(self, n_frames, s_max, e2_max):
        self._ck(self.L.mcmc_frames_alloc(self.h, n_frames, s_max, e2_max))
        self.n_frames, self.s_max, self.e2_max = n_frames, s_max, e2_max
        self.S = [0] * n_frames
        self.E2 = [0] * n_frames

    def frame_upload(self, frame, disp, label, n_superpixels):
        disp = np.ascontiguousarray(disp, np.int16)
        label = np.ascontiguousarray(label, np.uint16)
        assert disp.size == self.width * self.height == label.size
        self._ck(self.L.mcmc_frame_upload(self.h, frame, _ptr(disp, np.int16), _ptr(label, np.uint16),
                                          n_superpixels))
        self.S[frame] = n_superpixels

    def frame_upload_device(self, frame, disp_ptr, label_ptr, n_superpixels):
        self._ck(self.L.mcmc_frame_upload_device(self.h, frame, C.c_void_p(disp_ptr), C.c_void_p(label_ptr),
                                                 n_superpixels))
        self.S[frame] = n_superpixels

    def frame_set_graph(self, frame, row_ptr, col):
        row_ptr = np.ascontiguousarray(row_ptr, np.uint32)
        col = np.ascontiguousarray(col, np.uint32)
        self._ck(self.L.mcmc_frame_set_graph(self.h, frame, _ptr(row_ptr, np.uint32), _ptr(col, np.uint32),
                                             col.size))
        self.E2[frame] = int(col.size)

    def frames_moments(self, first=0, count=None):
        self._ck(self.L.mcmc_frames_moments(self.h, first, self.n_frames - first if count is None else count))

    def frames_edge_probs(self, first=0, count=None):
        self._ck(self.L.mcmc_frames_edge_probs(self.h, first, self.n_frames - first if count is None else count))

    def moments(self, frame):
        out = np.zeros((self.S[frame], 10), np.int64)
        self._ck(self.L.mcmc_frame_get_moments(self.h, frame, _ptr(out, np.int64)))
        return out

    def local_fit(self, frame):
        S = self.S[frame]
        fit = np.zeros((S, 16)); valid = np.zeros(S, np.uint8); lerr = np.zeros(S)
        self._ck(self.L.mcmc_frame_get_local_fit(self.h, frame, _ptr(fit, np.float64), _ptr(valid, np.uint8),
                                                 _ptr(lerr, np.float64)))
        return fit, valid, lerr

    def edge_probs(self, frame):
        q = np.zeros(self.E2[frame])
        self._ck(self.L.mcmc_frame_get_edge_probs(self.h, frame, _ptr(q, np.float64)))
        return q

    def segment_probability(self, frame, members):
        m = np.ascontiguousarray(members, np.uint32)
        p = C.c_double(); e = C.c_double(); th = np.zeros(3)
        self._ck(self.L.mcmc_segment_probability(self.h, frame, _ptr(m, np.uint32), m.size, C.byref(p),
                                                 _ptr(th, np.float64), C.byref(e)))
        return p.value, th, e.value

    def segment_probability_noalloc(self, frame, members):
        """mcmc_segment_probability through the library's static scratch: no device allocation."""
        m = np.ascontiguousarray(members, np.uint32)
        p = C.c_double(); e = C.c_double(); th = np.zeros(3)
        self._ck(self.L.mcmc_segment_probability_noalloc(self.h, frame, _ptr(m, np.uint32), m.size, C.byref(p),
                                                         _ptr(th, np.float64), C.byref(e)))
        return p.value, th, e.value

    # ---- chains
    def chains_init(self, n_chains, chains_per_frame=None, seed=1, chain_offset=0):
        cpf = n_chains if chains_per_frame is None else chains_per_frame
        self._ck(self.L.mcmc_chains_init(self.h, n_chains, cpf, seed, chain_offset))
        self.n_chains = n_chains
        self.chains_per_frame = cpf

    def chains_set_trace(self, capacity_steps):
        self._ck(self.L.mcmc_chains_set_trace(self.h, capacity_steps))

    def chains_run(self, burn_in, iters, sync=True):
        self._ck(self.L.mcmc_chains_run(self.h, burn_in, iters))
        self._last_steps = burn_in + iters
        if sync:
            self.synchronize()

    def trace(self):
        out = np.zeros((self.n_chains, self._last_steps), np.uint8)
        self._ck(self.L.mcmc_chains_get_trace(self.h, _ptr(out, np.uint8)))
        return out

    def labels(self, which=0, canonical=True):
        out = np.zeros((self.n_chains, self.s_max), np.uint16)
        self._ck(self.L.mcmc_chains_get_labels(self.h, which, int(canonical), _ptr(out, np.uint16)))
        return out

    def scalars(self):
        n = self.n_chains
        d = dict(energy=np.zeros(n), best_energy=np.zeros(n), state_hash=np.zeros(n, np.uint64),
                 trace_hash=np.zeros(n, np.uint64), n_accept=np.zeros(n, np.uint64),
                 n_case=np.zeros((n, 3), np.uint64), status=np.zeros(n, np.int32),
                 top_count=np.zeros(n, np.uint64), top_hash=np.zeros(n, np.uint64),
                 overflow=np.zeros(n, np.uint64))
        self._ck(self.L.mcmc_chains_get_scalars(self.h, *[v.ctypes.data_as(C.c_void_p) for v in d.values()]))
        return d

    def best(self, which=1, canonical=True):
        """The one labeling of the job (arg-max on the device; which = 1 best energy, 2 most visited):
        (energy, visits, global chain id, labels[s_max])."""
        e = C.c_double(); v = C.c_uint64(); k = C.c_uint64()
        lab = np.zeros(self.s_max, np.uint16)
        self._ck(self.L.mcmc_chains_get_best(self.h, which, int(canonical), C.byref(e), C.byref(v), C.byref(k),
                                             _ptr(lab, np.uint16)))
        return e.value, int(v.value), int(k.value), lab

    def best_record_device(self, which=1, canonical=True):
        """(device pointer, bytes) of the packed record [key, global chain id, labels] for a collective."""
        d = C.c_void_p(); n = C.c_size_t()
        self._ck(self.L.mcmc_chains_best_record_device(self.h, which, int(canonical), C.byref(d), C.byref(n)))
        return d.value, int(n.value)

    def marginals_device(self, frame):
        """(device pointer, count) of the frame's uint64 label-marginal counts."""
        d = C.c_void_p(); n = C.c_size_t()
        self._ck(self.L.mcmc_frame_marginals_device(self.h, frame, C.byref(d), C.byref(n)))
        return d.value, int(n.value)

    def visit_capacity(self):
        return int(self.L.mcmc_chains_visit_capacity(self.h))

    def reserve_visits(self, measured_steps):
        self._ck(self.L.mcmc_chains_reserve_visits(self.h, int(measured_steps)))

    def last_ratios(self):
        """(targetRatio log, proposalRatio) of every chain's last step: Problem::propose's out-parameters."""
        t = np.zeros(self.n_chains); p = np.zeros(self.n_chains)
        self._ck(self.L.mcmc_chains_get_last_ratios(self.h, _ptr(t, np.float64), _ptr(p, np.float64)))
        return t, p

    def ordered_states(self, chain):
        """Statistics::getOrderedStates (Statistics.h:50-70): (count, hash) of one chain, most visited first."""
        v = self.visits()[chain]
        return sorted(((c, h) for h, c in v.items()), key=lambda t: (-t[0], t[1]))

    def visits(self):
        cap = self.visit_capacity()
        hh = np.zeros((self.n_chains, cap), np.uint64); cc = np.zeros((self.n_chains, cap), np.uint64)
        self._ck(self.L.mcmc_chains_get_visits(self.h, _ptr(hh, np.uint64), _ptr(cc, np.uint64)))
        return [{int(h): int(c) for h, c in zip(hh[k], cc[k]) if c} for k in range(self.n_chains)]

    def marginals(self, frame):
        out = np.zeros((self.s_max, self.p.marginal_bins), np.uint64)
        self._ck(self.L.mcmc_frame_get_marginals(self.h, frame, _ptr(out, np.uint64)))
        return out

    def recompute_energy(self):
        out = np.zeros(self.n_chains)
        self._ck(self.L.mcmc_chains_recompute_energy(self.h, _ptr(out, np.float64)))
        return out

    def last_variant(self):
        """10 * table placement + chains per workgroup of the last chains_run (diagnostic)."""
        return int(self.L.mcmc_chains_last_variant(self.h))

    # ---- plumbing / measurement
    def set_stream(self, hip_stream_handle):
        self._ck(self.L.mcmc_ctx_set_stream(self.h, C.c_void_p(hip_stream_handle)))

    def synchronize(self):
        self._ck(self.L.mcmc_ctx_synchronize(self.h))

    def set_timing(self, on=True):
        self._ck(self.L.mcmc_set_timing(self.h, int(on)))

    def set_option(self, name, value):
        self._ck(self.L.mcmc_ctx_set_option(self.h, name.encode(), int(value)))

    def chain_profile(self):
        out = np.zeros(16, np.uint64)
        self._ck(self.L.mcmc_chains_get_profile(self.h, _ptr(out, np.uint64)))
        return out

    def last_kernel_ms(self, kernel):
        return float(self.L.mcmc_last_kernel_ms(self.h, kernel))

    def selftest_math(self, x, y):
        x = np.ascontiguousarray(x, np.float64); y = np.ascontiguousarray(y, np.float64)
        ol = np.zeros_like(x); oe = np.zeros_like(x); od = np.zeros_like(x)
        self._ck(self.L.mcmc_selftest_math(self.h, _ptr(x, np.float64), _ptr(y, np.float64), x.size,
                                           _ptr(ol, np.float64), _ptr(oe, np.float64), _ptr(od, np.float64)))
        return ol, oe, od

    def selftest_wave_sums(self, x):
        """x[n_waves][11][64] -> (plain, folded), each [n_waves][11]: the chain kernel's wave sums, both forms."""
        x = np.ascontiguousarray(x, np.float64)
        assert x.ndim == 3 and x.shape[1:] == (11, 64)
        plain = np.zeros((x.shape[0], 11)); folded = np.zeros((x.shape[0], 11))
        self._ck(self.L.mcmc_selftest_wave_sums(self.h, _ptr(x, np.float64), x.shape[0], _ptr(plain, np.float64),
                                                _ptr(folded, np.float64)))
        return plain, folded

    def selftest_philox(self, ctr, key):
        ctr = np.ascontiguousarray(ctr, np.uint32); key = np.ascontiguousarray(key, np.uint32)
        out = np.zeros(4, np.uint32)
        self._ck(self.L.mcmc_selftest_philox(self.h, _ptr(ctr, np.uint32), _ptr(key, np.uint32),
                                             _ptr(out, np.uint32)))
        return out


def load_scene(ctx: Context, frame: int, scene):
    """Upload one mcmc_amd.scene.Scene and its adjacency into `frame`."""
    ctx.frame_upload(frame, scene.disp, scene.label, scene.n_superpixels)
    ctx.frame_set_graph(frame, scene.row_ptr, scene.col)


class Multiview:
    """Path B: an SPGraph resident in HBM plus chains over it (mcmc_mv_* of include/mcmc_amd.h)."""

    def __init__(self, device=0):
        self.L = load()
        self.h = C.c_void_p()
        rc = self.L.mcmc_mv_create(device, C.byref(self.h))
        if rc != 0:
            raise McmcError(rc, "mcmc_mv_create failed (no GPU?)")
        self.V = 0
        self.n_chains = 0

    def _ck(self, rc):
        if rc != 0:
            raise McmcError(rc, self.L.mcmc_mv_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.mcmc_mv_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_graph(self, pt_ptr, points, row_ptr, col, q):
        pt_ptr = np.ascontiguousarray(pt_ptr, np.uint32); points = np.ascontiguousarray(points, np.float32)
        row_ptr = np.ascontiguousarray(row_ptr, np.uint32); col = np.ascontiguousarray(col, np.uint32)
        q = np.ascontiguousarray(q, np.float64)
        assert points.ndim == 2 and points.shape[1] == 4
        V = pt_ptr.size - 1
        self._ck(self.L.mcmc_mv_set_graph(self.h, V, _ptr(pt_ptr, np.uint32), _ptr(points, np.float32),
                                          _ptr(row_ptr, np.uint32), _ptr(col, np.uint32), _ptr(q, np.float64),
                                          col.size))
        self.V = V
        self._ck(self.L.mcmc_mv_vertex_moments(self.h))

    def vertex_moments(self):
        out = np.zeros((self.V, 10))
        self._ck(self.L.mcmc_mv_get_vertex_moments(self.h, _ptr(out, np.float64)))
        return out

    def fit_segment(self, members):
        m = np.ascontiguousarray(members, np.uint32)
        pl = np.zeros(5); p = C.c_double(); e = C.c_double()
        self._ck(self.L.mcmc_mv_fit_segment(self.h, _ptr(m, np.uint32), m.size, _ptr(pl, np.float64),
                                            C.byref(p), C.byref(e)))
        return pl, p.value, e.value

    def pmerge(self, t1, t2, w, temperature=8.0):
        t1 = np.ascontiguousarray(t1, np.float32); t2 = np.ascontiguousarray(t2, np.float32)
        w = np.ascontiguousarray(w, np.float32)
        n, dim = t1.shape
        out = np.zeros(n)
        self._ck(self.L.mcmc_mv_pmerge(self.h, _ptr(t1, np.float32), _ptr(t2, np.float32), _ptr(w, np.float32),
                                       dim, n, temperature, _ptr(out, np.float64)))
        return out

    def build_world_graph(self, theta, weights, n_first_scan, in_row_ptr=None, in_col=None, in_view_threshold=0.01,
                          merge_threshold=0.2, temperature=8.0):
        """getPlanarAdjacencyGraph + mergeNewScanGraph on the GPU: returns (row_ptr, col, prob)."""
        theta = np.ascontiguousarray(theta, np.float32); weights = np.ascontiguousarray(weights, np.float32)
        V, dim = theta.shape
        rp = np.zeros(V + 1, np.uint32); e2 = C.c_uint32(0)
        irp = None if in_row_ptr is None else np.ascontiguousarray(in_row_ptr, np.uint32)
        ic = None if in_col is None else np.ascontiguousarray(in_col, np.uint32)
        args = (self.h, V, dim, _ptr(theta, np.float32), _ptr(weights, np.float32), n_first_scan, _ptr(irp, np.uint32),
                _ptr(ic, np.uint32), in_view_threshold, merge_threshold, temperature)
        rc = self.L.mcmc_mv_build_world_graph(*args, _ptr(rp, np.uint32), None, None, 0, C.byref(e2))
        if rc not in (0, -6):
            self._ck(rc)
        col = np.zeros(e2.value, np.uint32); prob = np.zeros(e2.value, np.float64)
        if e2.value:
            self._ck(self.L.mcmc_mv_build_world_graph(*args, _ptr(rp, np.uint32), _ptr(col, np.uint32),
                                                      _ptr(prob, np.float64), e2.value, C.byref(e2)))
        return rp, col, prob

    def chains_init(self, n_chains, seed=1, chain_offset=0, max_new_segments=1024, split_prob=0.10):
        self._ck(self.L.mcmc_mv_chains_init(self.h, n_chains, seed, chain_offset, max_new_segments, split_prob))
        self.n_chains = n_chains

    def set_option(self, name, value):
        self._ck(self.L.mcmc_mv_set_option(self.h, name.encode(), int(value)))

    def chains_set_trace(self, cap):
        self._ck(self.L.mcmc_mv_chains_set_trace(self.h, cap))

    def chains_run(self, steps, sync=True):
        self._ck(self.L.mcmc_mv_chains_run(self.h, steps))
        self._last_steps = steps
        if sync:
            self._ck(self.L.mcmc_mv_synchronize(self.h))

    def trace(self):
        f = np.zeros((self.n_chains, self._last_steps), np.uint8); a = np.zeros((self.n_chains, self._last_steps))
        self._ck(self.L.mcmc_mv_chains_get_trace(self.h, _ptr(f, np.uint8), _ptr(a, np.float64)))
        return f, a

    def labels(self, which=0):
        out = np.zeros((self.n_chains, self.V), np.uint32)
        self._ck(self.L.mcmc_mv_chains_get_labels(self.h, which, _ptr(out, np.uint32)))
        return out

    def scalars(self):
        n = self.n_chains
        d = dict(n_accept=np.zeros(n, np.uint64), n_split=np.zeros(n, np.uint64), n_skip=np.zeros(n, np.uint64),
                 trace_hash=np.zeros(n, np.uint64), n_segments=np.zeros(n, np.int32), status=np.zeros(n, np.int32))
        self._ck(self.L.mcmc_mv_chains_get_scalars(self.h, *[v.ctypes.data_as(C.c_void_p) for v in d.values()]))
        return d

    def planes(self, chain, n_segments):
        out = np.zeros((n_segments, 5))
        self._ck(self.L.mcmc_mv_chains_get_planes(self.h, chain, _ptr(out, np.float64)))
        return out

    def profile(self):
        out = np.zeros(32, np.uint64)
        self._ck(self.L.mcmc_mv_chains_get_profile(self.h, _ptr(out, np.uint64)))
        return out

    def last_run_ms(self):
        return float(self.L.mcmc_mv_last_run_ms(self.h))


class OverSegmenter:
    """Over-segmentation front end (mcmc_overseg_* of include/mcmc_amd.h): colours + disparities ->
    label map, superpixel count and neighbour graph, ready for Context.frame_upload_device /
    frame_set_graph.  Mirrors OverSegmentation::overSegment steps 1-4."""

    def __init__(self, width=640, height=480, device=0):
        self.L = load()
        self.h = C.c_void_p()
        self.W, self.H = width, height
        rc = self.L.mcmc_overseg_create(device, width, height, C.byref(self.h))
        if rc != 0:
            raise McmcError(rc, "mcmc_overseg_create failed (no GPU?)")
        self.S = 0
        self.e2 = 0

    def _ck(self, rc):
        if rc != 0:
            raise McmcError(rc, self.L.mcmc_overseg_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.mcmc_overseg_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def default_params(self):
        p = OversegParams()
        self.L.mcmc_overseg_default_params(C.byref(p))
        return p

    def run(self, colors, disp, params=None):
        """colors: (H, W, 3) uint8 in B, G, R order; disp: (H, W) int16.  Returns (S, e2)."""
        p = params if params is not None else self.default_params()
        colors = np.ascontiguousarray(colors, np.uint8); disp = np.ascontiguousarray(disp, np.int16)
        assert colors.shape == (self.H, self.W, 3) and disp.shape == (self.H, self.W)
        S = C.c_int(0); e2 = C.c_uint32(0)
        self._ck(self.L.mcmc_overseg_run(self.h, _ptr(colors, np.uint8), _ptr(disp, np.int16), C.byref(p),
                                         C.byref(S), C.byref(e2)))
        self.S, self.e2 = S.value, e2.value
        return self.S, self.e2

    def smoothed(self):
        out = np.zeros((self.H, self.W, 3), np.uint8)
        self._ck(self.L.mcmc_overseg_get_smoothed(self.h, _ptr(out, np.uint8)))
        return out

    def edges(self):
        n = self.L.mcmc_overseg_get_edges(self.h, None, None, None)
        if n < 0:
            self._ck(int(n))
        a = np.zeros(n, np.int32); b = np.zeros(n, np.int32); w = np.zeros(n, np.float32)
        self.L.mcmc_overseg_get_edges(self.h, _ptr(a, np.int32), _ptr(b, np.int32), _ptr(w, np.float32))
        return a, b, w

    def roots(self):
        out = np.zeros(self.H * self.W, np.int32)
        self._ck(self.L.mcmc_overseg_get_roots(self.h, _ptr(out, np.int32)))
        return out

    def labels(self):
        out = np.zeros((self.H, self.W), np.uint16)
        self._ck(self.L.mcmc_overseg_get_labels(self.h, _ptr(out, np.uint16)))
        return out

    def graph(self):
        rp = np.zeros(self.S + 1, np.uint32); col = np.zeros(self.e2, np.uint32)
        self._ck(self.L.mcmc_overseg_get_graph(self.h, _ptr(rp, np.uint32), _ptr(col, np.uint32)))
        return rp, col

    def device_buffers(self):
        d = C.c_void_p(); l = C.c_void_p()
        self._ck(self.L.mcmc_overseg_device_buffers(self.h, C.byref(d), C.byref(l)))
        return d.value, l.value

    def last_ms(self):
        f = [C.c_float(0) for _ in range(4)]
        self._ck(self.L.mcmc_overseg_last_ms(self.h, *[C.byref(x) for x in f]))
        g = [C.c_float(0) for _ in range(2)]
        self._ck(self.L.mcmc_overseg_last_sort_ms(self.h, *[C.byref(x) for x in g]))
        out = dict(zip(("filter", "edges", "union_find", "graph"), (x.value for x in f)))
        out["device_sort"], out["host_scan"] = g[0].value, g[1].value
        return out

    def tie_count(self):
        """Edges of the last run whose weight equals their predecessor's in scan order."""
        return int(self.L.mcmc_overseg_last_tie_count(self.h))

    def set_option(self, name, value):
        """"tie_order": 1 = std::sort on the host, the reference's call (default); 0 = device sort, equal weights in
        creation order (faster; a layout of this library's own wherever weights tie)."""
        self._ck(self.L.mcmc_overseg_set_option(self.h, name.encode(), int(value)))


class OverSegmenterBatch:
    """mcmc_overseg_batch_*: OverSegmentation::overSegment over a batch of frames, n_workers frames at a time (one host
    thread, stream and set of buffers each); every frame's planes stay on the device, its neighbour graph on the host."""

    def __init__(self, width=640, height=480, n_workers=8, device=0):
        self.L = load()
        self.h = C.c_void_p()
        self.W, self.H = width, height
        rc = self.L.mcmc_overseg_batch_create(device, width, height, n_workers, C.byref(self.h))
        if rc != 0:
            raise McmcError(rc, "mcmc_overseg_batch_create failed (no GPU?)")
        self.S = np.zeros(0, np.int32)
        self.e2 = np.zeros(0, np.uint32)

    def _ck(self, rc):
        if rc != 0:
            raise McmcError(rc, self.L.mcmc_overseg_batch_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.mcmc_overseg_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        self._ck(self.L.mcmc_overseg_batch_set_option(self.h, name.encode(), int(value)))

    def run(self, colors, disp, params=None):
        """colors: (F, H, W, 3) uint8 in B, G, R order; disp: (F, H, W) int16.  Returns (S[F], e2[F])."""
        if params is None:
            params = OversegParams()
            self.L.mcmc_overseg_default_params(C.byref(params))
        colors = np.ascontiguousarray(colors, np.uint8); disp = np.ascontiguousarray(disp, np.int16)
        F = disp.shape[0]
        assert colors.shape == (F, self.H, self.W, 3) and disp.shape == (F, self.H, self.W)
        self.S = np.zeros(F, np.int32); self.e2 = np.zeros(F, np.uint32)
        self._ck(self.L.mcmc_overseg_batch_run(self.h, F, _ptr(colors, np.uint8), _ptr(disp, np.int16), C.byref(params),
                                               _ptr(self.S, np.int32), _ptr(self.e2, np.uint32)))
        return self.S, self.e2

    def graph(self, frame):
        rp = np.zeros(int(self.S[frame]) + 1, np.uint32); col = np.zeros(int(self.e2[frame]), np.uint32)
        self._ck(self.L.mcmc_overseg_batch_get_graph(self.h, frame, _ptr(rp, np.uint32), _ptr(col, np.uint32)))
        return rp, col

    def labels(self, frame):
        out = np.zeros((self.H, self.W), np.uint16)
        self._ck(self.L.mcmc_overseg_batch_get_labels(self.h, frame, _ptr(out, np.uint16)))
        return out

    def device_buffers(self, frame):
        d = C.c_void_p(); l = C.c_void_p()
        self._ck(self.L.mcmc_overseg_batch_device_buffers(self.h, frame, C.byref(d), C.byref(l)))
        return d.value, l.value

    def tie_count(self, frame):
        n = self.L.mcmc_overseg_batch_tie_count(self.h, frame)
        if n < 0:
            self._ck(int(n))
        return int(n)

    def last_wall_ms(self):
        return float(self.L.mcmc_overseg_batch_last_wall_ms(self.h))
