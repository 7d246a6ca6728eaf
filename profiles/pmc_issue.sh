#!/bin/bash
# Issue-side SQ counters of the default bench (what the waves of the LK kernel spend their time on).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_issue
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $O/a -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-secondary > $O/a.log 2>&1
timeout -k 10 150 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_CYCLES --output-format csv -d $O/b -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-secondary > $O/b.log 2>&1
python3 $R/profiles/summarize_pmc.py $O/summary.json $O/a/*/*_counter_collection.csv $O/b/*/*_counter_collection.csv > /dev/null
python3 -c "
import json;d=json.load(open('$O/summary.json'))['pyrlk_track_w21'];print(json.dumps(d,indent=1))"
