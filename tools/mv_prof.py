"""Path B at C4 size for rocprofv3: the three kernels of the split iteration (or the fused one with FUSED=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_amd import host
from mcmc_amd.scene import make_multiview
mv = make_multiview(n_views=3, width=640, height=480, n_planes=8, n_superpixels=500, seed=4)
m = host.Multiview()
m.set_graph(mv.pt_ptr, mv.points, mv.row_ptr, mv.col, mv.q)
m.vertex_moments()
if os.environ.get("GROWTH_AHEAD"):             # the growth a step ahead on a stream of its own (the default up to 3072 chains)
    m.set_option("growth_ahead", int(os.environ["GROWTH_AHEAD"]))
if os.environ.get("MVG"):                      # groups of chains (the default is two from 1024 chains on)
    m.set_option("groups", int(os.environ["MVG"]))
if os.environ.get("SPLIT_GROWTH"):
    m.set_option("split_growth", 1)
if os.environ.get("FUSED"):
    m.set_option("split_step", 0)
m.chains_init(int(os.environ.get("CHAINS", 1024)), seed=1)
m.chains_run(2)
if os.environ.get("PHASES"):
    m.set_option("profile", 1)
m.chains_run(int(os.environ.get("STEPS", 10)))
print("ms per launch", m.last_run_ms())
for _ in range(int(os.environ.get("REPEAT", 0))):
    m.chains_run(int(os.environ.get("STEPS", 10)))
    print("ms per launch (again)", m.last_run_ms())
print("segments per chain (mean)", float(m.scalars()["n_segments"].mean()))
if os.environ.get("PHASES"):
    p = m.profile().astype(float)
    n = float(os.environ.get("CHAINS", 1024)) * float(os.environ.get("STEPS", 10))
    names = ["set-up", "start", "growth", "move/neighbours", "planes", "hand-over", "write-back", "(the profile's own counting)"]
    for k, nm in enumerate(names):
        print(f"{nm:16s} {p[k] / n:10.0f} cycles per chain-step")
    print(f"levels {p[8] / n:.1f}  grown vertices {p[9] / n:.1f}")
    print(f"growth, wave 0 (coarse timers): whole loop {p[30] / n:.0f} cycles per chain-step (slowest chain-step {p[31]:.0f}); scanning the frontier "
          f"{(p[24] - p[25]) / n:.0f} for {p[29] / n:.1f} trips, {p[28] / n:.1f} full batches of trials {p[25] / n:.0f}, "
          f"the level's last batch {p[26] / n:.0f}, level barriers {p[27] / n:.0f}")
    st = float(os.environ.get("STEPS", 10))
    if p[19]:
        print(f"L1 kernel per iteration (one item in 16 is timed): items {16 * p[19] / st:.0f}  blocks {16 * p[20] / st:.0f}  matrix instructions {16 * p[21] / st:.0f}  "
              f"residual slots / residuals asked {16 * p[21] * 256 / max(p[12], 1):.2f}")
        print(f"  per item: before the loop {p[16] / p[19]:.0f} cycles, loop {p[17] / p[19]:.0f}, after {p[18] / p[19]:.0f}; "
              f"waves with work {16 * p[22] / st:.0f}, cycles per such wave {p[23] / max(p[22], 1):.0f}")
