"""When and where the items of the L1 kernel run (diagnostic build -DMC_MV_ERR_TIMELINE, loaded with MCMC_AMD_LIB)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_multiview
mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
m = host.Multiview()
m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
m.vertex_moments()
m.set_option("groups", int(os.environ.get("MVG", 1)))      # one group: one L1 launch per iteration
m.set_option("growth_ahead", int(os.environ.get("GROWTH_AHEAD", 0)))      # and no growth kernel beside it
m.chains_init(int(os.environ.get("CHAINS", 1024)), seed=1)
m.chains_run(6)
m.chains_run(1)
lib = host.load()
buf = np.zeros(16384 * 4, dtype=np.uint64)
rc = lib.mcmc_mv_debug_timeline(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
t = buf.reshape(-1, 4)
t = t[t[:, 1] > 0]
# only the last launch's items are valid where the queue was as long; take those whose start is within 1 ms of the latest
last = t[:, 0].max()
t = t[(last - t[:, 0]) < 8000]
if os.environ.get("TL_DUMP"):
    np.save(os.environ["TL_DUMP"], t)
t0 = t[:, 0].min()
st = (t[:, 0] - t0).astype(float) / 100.0      # us (100 MHz)
en = (t[:, 1] - t0).astype(float) / 100.0
hw = t[:, 2]
simd = ((hw >> 32) & 0xF) * 1000000 + ((hw >> 13) & 7) * 100000 + ((hw >> 12) & 1) * 10000 + ((hw >> 8) & 0xF) * 100 + ((hw >> 4) & 3)
n = t[:, 3].astype(float)
print(f"items {len(t)}  span {en.max():.1f} us  item time mean {np.mean(en - st):.1f} max {np.max(en - st):.1f}  matrix instructions per item mean {n.mean():.0f} max {n.max():.0f}")
print("distinct SIMDs", len(np.unique(simd)))
edges = np.arange(0, en.max() + 4, 4.0)
print("time us : waves running : items started : matrix instructions in flight (by item share)")
rate = (n / np.maximum(en - st, 0.01))
for a in edges:
    run = (st <= a) & (en > a)
    print(f"{a:6.0f} {run.sum():6d} {((st >= a) & (st < a + 4)).sum():6d} {rate[run].sum() * 64 / 2400 / 1024:8.2f}")
# per SIMD: when its last item ends, how many it had
ends = {}
cnt = {}
work = {}
for s_, e_, n_ in zip(simd, en, n):
    ends[s_] = max(ends.get(s_, 0), e_); cnt[s_] = cnt.get(s_, 0) + 1; work[s_] = work.get(s_, 0) + n_
print("per SIMD matrix instructions percentiles 0/10/50/90/100:", np.percentile(np.array(list(work.values())), [0, 10, 50, 90, 100]))
e = np.array(list(ends.values())); c = np.array(list(cnt.values())); w = np.array(list(work.values()))
print("per SIMD: last end us percentiles 10/50/90/100:", np.percentile(e, [10, 50, 90, 100]).round(1), " items", c.min(), c.mean().round(2), c.max(),
      " matrix instructions", w.min(), w.mean().round(0), w.max(), " -> us of matrix pipe at 64 cycles, 2.4 GHz:", (w.max() * 64 / 2400).round(1))
per = (en - st) * 2400 / n
print("cycles per matrix instruction and wave (2.4 GHz), by start time: first 10 us %.0f, later %.0f" % (per[st < 10].mean(), per[st >= 10].mean()))

# where the first workgroups of XCD 0 went (item index -> workgroup is the kernel's mapping; shown by start order)
o = np.argsort(t[:, 0], kind="stable")
x0 = [(int((hw[i] >> 13) & 7), int((hw[i] >> 12) & 1), int((hw[i] >> 8) & 0xF), int((hw[i] >> 4) & 3), int(n[i])) for i in o if ((hw[i] >> 32) & 0xF) == 0][:48]
print("XCD 0, by start time: (se, sh, cu, simd, matrix instructions)", x0)
