#!/bin/bash
# rocprofv3 PMC passes over tools/k3_profile.py (K3 at C2). Usage: tools/prof_k3.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/k3pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/p1 -o p -- python3 $GRAFT_REPO_ROOT/tools/k3_profile.py > $OUT/p1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/p2 -o p -- python3 $GRAFT_REPO_ROOT/tools/k3_profile.py > $OUT/p2.log 2>&1 || exit 1
python3 - <<PY
import csv, collections, json
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("p1", "p2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open("$OUT/%s/p_counter_collection.csv" % d)):
        if "chain_run_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        print(k)
        for c, v in sorted(dd.items()):
            print("   %-22s %16.0f  (%d launches)" % (c, sum(v) / len(v), len(v)))
            tot[k][c] = sum(v)
# what bench.py's chain_kernel.bound_model reads: instruction totals of the kernel the headline job launches (the first
# one here: tools/k3_profile.py runs 2000 + ITERS steps of it for CHAINS chains before the diagnostic build)
import os
chains = int(os.environ.get("CHAINS", 1024)); steps = 2000 + int(os.environ.get("ITERS", 500))
k = [k for k in tot if "true, true, true" not in k][0]
import re
m = re.search(r"<(\d+), (true|false), (true|false), (true|false), (true|false), (\d+)>", k)
wpb, tab, rec, topo, ns = int(m.group(1)), m.group(2) == "true", m.group(3) == "true", m.group(5) == "true", int(m.group(6))
place = 1 if (tab and rec) else ((4 if (ns > 1 and wpb == 4) else 2) if tab else (3 if topo else 0))
variant = 10 * place + wpb + (100 * (ns - 2) if ns > 1 else 0)      # mcmc_chains_last_variant's number
out = {"kernel": k.strip(), "variant": variant, "chains": chains, "steps_per_chain": steps, "source": "tools/prof_k3.sh (rocprofv3 --pmc, two passes)",
       "totals": {c: tot[k][c] for c in sorted(tot[k])}}
json.dump(out, open("$OUT/k3_counts.json", "w"), indent=1)
PY
