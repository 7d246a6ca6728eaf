#!/bin/bash
# A/B of the committed library against the working tree on ONE GPU box (boxes differ by several
# per cent, so numbers from different gpurun calls do not compare).
#   here:  bash tools/ab_bench.sh build      -> variants/lib_head.so, variants/lib_new.so
#   box:   bash tools/ab_bench.sh run [rounds]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p variants
  make -C slam_amd/csrc >/dev/null && cp slam_amd/libktlvo_hip.so variants/lib_new.so
  git stash -q && make -C slam_amd/csrc >/dev/null && cp slam_amd/libktlvo_hip.so variants/lib_head.so
  git stash pop -q && make -C slam_amd/csrc >/dev/null
  exit 0
fi
cp slam_amd/libktlvo_hip.so /tmp/ktlvo_keep.so
for round in $(seq 1 ${2:-3}); do
  for v in head new; do
    cp variants/lib_$v.so slam_amd/libktlvo_hip.so
    timeout -k 10 200 python bench.py --cpu-seconds 0 --skip-secondary --soak 0 > gpurun_out/ab.json
    python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('$v', round(d['value']/1e6,2), 'M/s  step', round(d['ms_per_step'],4), 'ms  LK', round(d['roofline']['avg_launch_ms'],4), 'ms  K1', round(d['kernels']['pyr_down_u8']['avg_ms_per_step'],4), ' K2', round(d['kernels']['scharr_planes']['avg_ms_per_step'],4))"
  done
done
cp /tmp/ktlvo_keep.so slam_amd/libktlvo_hip.so
