#!/bin/bash
# Same-box check that two library builds (variants/lib_head.so, variants/lib_new.so; tools/ab_bench.sh build) give
# byte-identical tracks: the fixed-batch workload's per-pair table (counts and CRC-32 of status bytes and positions).
set -e
cd "$(dirname "$0")/.."
cp slam_amd/libktlvo_hip.so /tmp/ktlvo_keep.so
for v in head new; do
  cp variants/lib_$v.so slam_amd/libktlvo_hip.so
  KTLVO_AB_OLD_LIBRARY=1 timeout -k 10 300 python bench.py --workload batch --width 1241 --height 376 --features 2000 --min-dist 10 \
      --batch-pairs 48 --steps 2 --warmup 1 --dump-pairs gpurun_out/pairs_$v.npy > gpurun_out/ab_$v.json
done
cp /tmp/ktlvo_keep.so slam_amd/libktlvo_hip.so
python -c "
import numpy as np
a, b = np.load('gpurun_out/pairs_head.npy'), np.load('gpurun_out/pairs_new.npy')
print('tables equal:', np.array_equal(a, b), a.shape, 'tracked', int(a[:, 2].sum()))
assert np.array_equal(a, b)"
