#!/bin/bash
# Same-box A/B of the two readings of LK's window sums (float tree against exact integer totals).
#   box: bash tools/ab_sums.sh [rounds]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in $(seq 1 ${1:-3}); do
  for v in tree exact; do
    timeout -k 10 200 python bench.py --cpu-seconds 0 --skip-secondary --skip-stream --soak 0 --lk-sums $v > gpurun_out/ab_sums.json
    python -c "
import json;d=json.load(open('gpurun_out/ab_sums.json'));print('$v', round(d['value']/1e6,2), 'M/s  step', round(d['ms_per_step'],4), 'ms  LK', round(d['roofline']['avg_launch_ms'],4), 'ms  tracked', d['config']['tracked_per_step'])"
  done
done
