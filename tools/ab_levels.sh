set -e
cp slam_amd/libktlvo_hip.so /tmp/ktlvo_keep.so
for round in 1 2; do
  for v in head new; do
    cp variants/lib_$v.so slam_amd/libktlvo_hip.so
    for ml in 3 4; do
      KTLVO_AB_OLD_LIBRARY=1 timeout -k 10 200 python bench.py --cpu-seconds 0 --skip-secondary --skip-stream --soak 0 --max-level $ml > gpurun_out/ab.json
      python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('$v maxLevel $ml', round(d['value']/1e6,2), 'M/s  LK', round(d['roofline']['avg_launch_ms'],4))"
    done
  done
done
cp /tmp/ktlvo_keep.so slam_amd/libktlvo_hip.so
