"""Print the main figures of a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print("value %.1f M/s  ms_per_step %.3f  variant %s" % (d["value"] / 1e6, d["ms_per_step"], d["config"]["chain_kernel_variant"]))
ck = d.get("chain_kernel", {})
print("k3 us per proposal per chain %.3f" % ck.get("us_per_proposal_per_chain", -1), {k: v for k, v in ck.get("bound_model", {}).items() if k.endswith("frac")})
if "c5_share" in d: print("c5 share: %.1f M/s, chain kernel %.2f ms" % (d["c5_share"]["proposals_per_s"] / 1e6, d["c5_share"]["chain_kernel_ms"]))
if "roofline" in d: print("K1 roofline frac %.4f  avg %.2f us  stream-only %.0f GB/s" % (d["roofline"]["frac"], d["roofline"]["avg_launch_ms"] * 1e3, d["roofline"]["stream_only_GBps"]))
if d.get("multiview"): print("multiview %.2f M/s, %.3f ms per launch" % (d["multiview"]["proposals_per_s"] / 1e6, d["multiview"]["ms_per_launch"]), d["multiview"].get("more_chains"))
if "cpu_baseline" in d: print("cpu %.0f /s, same computation %s" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["same_computation"]["identical_to_gpu"]))
print("statistics", d.get("statistics"))
