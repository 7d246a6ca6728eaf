#!/bin/bash
# Memory-pipeline counters of the default bench (texture addresser, L1 = TCP, data return = TD):
# two passes, each within the per-block counter limits.  Run on the GPU box from the repo root.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_mem
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-secondary > $O/p1.log 2>&1
timeout -k 10 150 rocprofv3 --pmc TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum --output-format csv -d $O/p2 -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-secondary > $O/p2.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "pyrlk" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in agg:
    for c in sorted(agg[k]):
        print("%-24s %-44s %16.0f per launch" % (k[-24:], c, agg[k][c] / n[k][c]))
PY
