#!/bin/bash
# Same-box sweep of ONE library build over several values of an environment switch read at context creation.
#   box:  bash tools/sweep_env.sh KTLVO_SCHARR_CHUNK "1 2 4 8" [rounds]
set -e
cd "$(dirname "$0")/.."
VAR=$1
for round in $(seq 1 ${3:-2}); do
  for v in $2; do
    env $VAR=$v timeout -k 10 200 python bench.py --cpu-seconds 0 --skip-secondary --soak 0 > gpurun_out/ab.json 2>/dev/null
    python -c "
import json;d=json.load(open('gpurun_out/ab.json'));k=d['kernels'];print('$VAR=$v', round(d['value']/1e6,2), 'M/s  step', round(d['ms_per_step'],4), 'ms  LK', round(d['roofline']['avg_launch_ms'],4), ' K1', round(k['pyr_down_u8']['avg_ms_per_step'],4), ' K2', round(k['scharr_planes']['avg_ms_per_step'],4))"
  done
done
