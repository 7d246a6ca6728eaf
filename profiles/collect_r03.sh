#!/bin/bash
# Run on the GPU box from the repo root: kernel traces and counter passes of the default bench step, of the frame
# path, of the stream step (detection and pyramid build side by side) and of the corner kernels.  Counter passes are
# separate runs (rocprofv3 does not split counters; FETCH_SIZE and WRITE_SIZE do not fit one TCC pass), every pass
# under a limit that first asks the run to end (so the Python child does not stay behind holding the GPU), and a
# pass that fails ends the script before anything is summarised from it.
# Outputs land in gpurun_out/prof_r03; the summaries are then copied into profiles/ by hand.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-seconds 0 --skip-secondary --skip-stream --soak 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 20 --warmup 3 > $O/trace.log 2>&1
grep '"metric"' $O/trace.log > $O/bench_under_trace.json
pass() { # name, counters...
  local name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -- $B --steps 3 --warmup 1 > $O/pmc_$name.log 2>&1
}
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY
pass grbm GRBM_GUI_ACTIVE
# the same counters for the exact-sum reading of LK (an option; summarised into a file of its own)
passx() {
  local name=$1; shift
  timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $O/pmcx_$name -- $B --lk-sums exact --steps 3 --warmup 1 > $O/pmcx_$name.log 2>&1
}
passx sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
passx fetch FETCH_SIZE
passx write WRITE_SIZE
passx grbm GRBM_GUI_ACTIVE
for mode in frame grid batch; do
  for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout -k 10 100 rocprofv3 --pmc $c --output-format csv -d $O/det_${mode}_$n -- python3 $R/tools/detect_once.py $mode > $O/det_${mode}_$n.log 2>&1
  done
done
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/det_trace_frame -- python3 $R/tools/detect_once.py frame > $O/det_trace_frame.log 2>&1
timeout -k 10 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/det_trace_grid -- python3 $R/tools/detect_once.py grid > $O/det_trace_grid.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/det_trace_batch -- python3 $R/tools/detect_once.py batch > $O/det_trace_batch.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/frame_path -- python3 $R/tools/time_frame_path.py 40 > $O/frame_path.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/stream_side -- python3 $R/tools/time_stream_step.py 127 4 side exact > $O/stream_side.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/stream_turn -- python3 $R/tools/time_stream_step.py 127 4 turn exact > $O/stream_turn.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/stream_one -- python3 $R/tools/time_stream_step.py 127 4 one exact > $O/stream_one.log 2>&1
cd $R
python3 tools/trace_timeline.py $(ls $O/stream_one/*/*kernel_trace.csv | head -1) 17 > $O/stream_one_timeline.txt
python3 tools/trace_timeline.py $(ls $O/stream_side/*/*kernel_trace.csv | head -1) 17 > $O/stream_side_timeline.txt
python3 tools/trace_timeline.py $(ls $O/stream_turn/*/*kernel_trace.csv | head -1) 17 > $O/stream_turn_timeline.txt
python3 profiles/summarize_pmc.py $O/pmc_summary.json $(find $O/pmc_* -name "*counter_collection.csv") > $O/summary.log
python3 profiles/summarize_pmc.py $O/pmc_summary_exact_sums.json $(find $O/pmcx_* -name "*counter_collection.csv") > $O/summary_exact.log
python3 profiles/summarize_pmc.py $O/det_frame_summary.json $(find $O/det_frame_* -name "*counter_collection.csv") > $O/det_frame.log
python3 profiles/summarize_pmc.py $O/det_grid_summary.json $(find $O/det_grid_* -name "*counter_collection.csv") > $O/det_grid.log
python3 profiles/summarize_pmc.py $O/det_batch_summary.json $(find $O/det_batch_* -name "*counter_collection.csv") > $O/det_batch.log
echo collected
