#!/bin/bash
# SQ / LDS counter passes of the kernels exactly as bench.py launches them (chess_v1_pyr_kernel: 64 frames, hot list +
# level images; chess_v1_multi_kernel).  Runs ON the GPU box: gpurun -- 'bash tools/collect_sq_prod.sh r03a'
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMCB="python $R/bench.py --distinct 4 --steps 3 --warmup 1 --prime 2 --no-cpu-baseline --no-end-to-end"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq1 -o p -- $PMCB > $OUT/sq1.json 2> $OUT/sq1.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/sq2 -o p -- $PMCB > $OUT/sq2.json 2> $OUT/sq2.err
for d in sq1 sq2; do
    python $R/tools/pmc_summary.py $OUT/$d/p_counter_collection.csv > $OUT/$d.txt 2>> $OUT/$d.err
    rm -rf $OUT/$d
done
ls -la $OUT
