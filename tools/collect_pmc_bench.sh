#!/bin/bash
# The rocprofv3 --pmc passes of the bench command (level-0 launch as bench.py launches it: EA traffic by request size, SQ issue /
# wait / LDS counters).  Runs ON the GPU box, called by tools/collect_profiles.sh or alone:
#   gpurun --timeout 1500 -- 'bash tools/collect_pmc_bench.sh r06c'
# Counter passes only (--kernel-trace for the kernel names, no other trace domain), one counter group per run.  The legs beside
# the timed steps are switched off: under the counter tool every dispatch is serialised, and the thousands of small kernels
# of the frame generator in the configs legs took the tool down (SIGSEGV in the dispatch callback, round 6).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMCB="python $R/bench.py --distinct 4 --steps 3 --warmup 1 --no-cpu-baseline --no-find-boards --no-configs --no-chess-alone --no-sparse-leg --no-end-to-end"
# EA (fabric) traffic: read requests by size, write requests, and the derived KiB counters
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B --kernel-trace --output-format csv -d $OUT/pmc_rd -o p -- $PMCB > $OUT/pmc_rd.json 2> $OUT/pmc_rd.err
timeout 900 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --output-format csv -d $OUT/pmc_wr -o p -- $PMCB > $OUT/pmc_wr.json 2> $OUT/pmc_wr.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o p -- $PMCB > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o p -- $PMCB > $OUT/pmc_write.json 2> $OUT/pmc_write.err
# the SQ counters on the kernels exactly as bench.py launches them (chess_v1_pyr_kernel: 64 frames, hot list +
#     level images; chess_v1_multi_kernel), separate passes of the bench command
PMCQ="$PMCB --prime 2"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sqp1 -o p -- $PMCQ > /dev/null 2> $OUT/pmc_sqp1.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_sqp2 -o p -- $PMCQ > /dev/null 2> $OUT/pmc_sqp2.err
for d in pmc_rd pmc_wr pmc_fetch pmc_write pmc_sqp1 pmc_sqp2; do
    python $R/tools/pmc_summary.py $OUT/$d/p_counter_collection.csv > $OUT/$d.txt 2>> $OUT/$d.err
    rm -rf $OUT/$d
done
ls -la $OUT | grep pmc_
