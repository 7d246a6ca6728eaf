#!/bin/bash
# Same-box A/B of ONE library build under two settings of an environment switch read at context creation
# (boxes differ by several per cent, so numbers from different gpurun calls do not compare).
#   box:  bash tools/ab_env.sh KTLVO_LK_STAY 0 1 [rounds]
# Prints the bench step under each setting, then requires the fixed-batch workload's per-pair tables
# (counts, CRC-32 of status bytes and of positions) to be byte-identical under both.
set -e
cd "$(dirname "$0")/.."
VAR=$1; A=$2; B=$3
for round in $(seq 1 ${4:-3}); do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 200 python bench.py --cpu-seconds 0 --skip-secondary --soak 0 > gpurun_out/ab.json
    python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('$VAR=$v', round(d['value']/1e6,2), 'M/s  step', round(d['ms_per_step'],4), 'ms  LK', round(d['roofline']['avg_launch_ms'],4), 'ms')"
  done
done
for v in $A $B; do
  env $VAR=$v timeout -k 10 300 python bench.py --workload batch --width 1241 --height 376 --features 2000 --min-dist 10 \
      --batch-pairs 48 --steps 2 --warmup 1 --dump-pairs gpurun_out/pairs_$v.npy > gpurun_out/ab_$v.json
done
python -c "
import numpy as np
a, b = np.load('gpurun_out/pairs_$A.npy'), np.load('gpurun_out/pairs_$B.npy')
print('tables equal:', np.array_equal(a, b), a.shape, 'tracked', int(a[:, 2].sum()))
assert np.array_equal(a, b)"
