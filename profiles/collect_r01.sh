#!/bin/bash
# Run on the GPU box from the repo root: kernel trace + three PMC passes of the default bench.
# Counter passes are separate runs (rocprofv3 does not split counters; FETCH_SIZE and WRITE_SIZE
# do not fit one TCC pass).  Outputs land in gpurun_out/; copy the summaries into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_r01
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-seconds 0 --skip-secondary > $O/trace.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-secondary > $O/pmc_sq.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-secondary > $O/pmc_fetch.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-secondary > $O/pmc_write.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $O/pmc_lds -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-seconds 0 --skip-secondary > $O/pmc_lds.log 2>&1
grep '"metric"' $O/trace.log > $O/bench_under_trace.json || true
echo collected
