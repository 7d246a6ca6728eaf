#!/bin/bash
# Run on the GPU box from the repo root: kernel trace + counter passes of the default bench step, and counter
# passes of the corner kernels.  Counter passes are separate runs (rocprofv3 does not split counters; FETCH_SIZE
# and WRITE_SIZE do not fit one TCC pass).  Outputs land in gpurun_out/; summaries are copied into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-seconds 0 --skip-secondary --skip-stream --soak 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 20 --warmup 3 > $O/trace.log 2>&1
grep '"metric"' $O/trace.log > $O/bench_under_trace.json || true
pass() { # name, counters...
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- $B --steps 3 --warmup 1 > $O/pmc_$name.log 2>&1
}
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY
pass grbm GRBM_GUI_ACTIVE
for mode in frame grid batch; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout -s KILL 100 rocprofv3 --pmc $c --output-format csv -d $O/det_${mode}_$n -- python3 $R/tools/detect_once.py $mode > $O/det_${mode}_$n.log 2>&1
  done
done
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/det_trace_frame -- python3 $R/tools/detect_once.py frame > $O/det_trace_frame.log 2>&1
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/det_trace_grid -- python3 $R/tools/detect_once.py grid > $O/det_trace_grid.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/det_trace_batch -- python3 $R/tools/detect_once.py batch > $O/det_trace_batch.log 2>&1
cd $R
python3 profiles/summarize_pmc.py $O/pmc_summary.json $(find $O/pmc_* -name "*counter_collection.csv") > $O/summary.log
python3 profiles/summarize_pmc.py $O/det_frame_summary.json $(find $O/det_frame_* -name "*counter_collection.csv") > $O/det_frame.log
python3 profiles/summarize_pmc.py $O/det_grid_summary.json $(find $O/det_grid_* -name "*counter_collection.csv") > $O/det_grid.log
python3 profiles/summarize_pmc.py $O/det_batch_summary.json $(find $O/det_batch_* -name "*counter_collection.csv") > $O/det_batch.log
echo collected
