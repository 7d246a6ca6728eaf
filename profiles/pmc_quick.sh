#!/bin/bash
# One SQ counter pass of the default bench (quick check of instruction counts after a kernel change).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_quick
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-secondary > $O/sq.log 2>&1
python3 $R/profiles/summarize_pmc.py $O/summary.json $O/sq/*/*_counter_collection.csv > /dev/null
python3 -c "
import json;d=json.load(open('$O/summary.json'))['pyrlk_track_w21'];print(json.dumps(d,indent=1))"
