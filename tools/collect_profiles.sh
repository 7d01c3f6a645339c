#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's roofline numbers.  Runs ON the GPU box:
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh r01'
# Writes under gpurun_out/<tag>/ (scratch, merged back by gpurun); tools/make_profiles.py turns that
# into the committed summaries under profiles/.  PMC passes are their own runs (no trace domains).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"                              # defaults: 200 steps, 20 warmup
TRACEB="python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-find-boards"   # short run under the tracer (the find_boards leg has its own tool)
# 1. the bench line itself
timeout 600 $BENCH > $OUT/bench.json 2> $OUT/bench.err
# 2. kernel trace of the same command (no CPU baseline: it only adds host time)
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $TRACEB > $OUT/trace_bench.json 2> $OUT/trace.err
# 3. + 4b. the counter passes of the bench command itself (their own script: they can be repeated alone)
bash $R/tools/collect_pmc_bench.sh $TAG
# 4. issue / wait / LDS counters of the level-0 response kernel alone
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_sq1 -o p -- python $R/tools/chess_l0_alone.py > /dev/null 2> $OUT/pmc_sq1.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o p -- python $R/tools/chess_l0_alone.py > /dev/null 2> $OUT/pmc_sq2.err
# 4c. kernel trace of the textured-background workload
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace_clut -o t -- python $R/bench.py --workload c3_cluttered --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/trace_clut.err
timeout 600 python $R/bench.py --workload c3_cluttered --no-cpu-baseline --no-end-to-end > $OUT/bench_cluttered.json 2> $OUT/bench_cluttered.err
timeout 900 python $R/bench.py --workload c4_4096x3072_shard256 --force-gather --bind-numa --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
timeout 600 python $R/bench.py --workload c3_4096x3072_14x14_chain --no-cpu-baseline --no-end-to-end > $OUT/bench_14x14.json 2> $OUT/bench_14x14.err
# 4d. option sparse_refine as the timed schedule: kernel trace at 64 frames, bench lines at 64 and 256 frames
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace_sparse -o t -- python $R/bench.py --sparse-refine --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/trace_sparse.err
timeout 600 python $R/bench.py --sparse-refine --no-cpu-baseline --no-end-to-end > $OUT/bench_sparse.json 2> $OUT/bench_sparse.err
timeout 900 python $R/bench.py --sparse-refine --workload c4_4096x3072_shard256 --steps 50 --warmup 5 --no-cpu-baseline --no-end-to-end > $OUT/bench_sparse_c4.json 2> $OUT/bench_sparse_c4.err
# 4e. round 5: the plain ChESS pass alone (kernel trace + the script's own hipEvent line), BASELINE config 2 as a bench line +
#     kernel trace, the two-rank rehearsal of the N > 1 flow, the sixteen-pixels-per-lane kernel's counters beside chess_v1's
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_alone -o t -- python $R/tools/chess_pass_alone.py > $OUT/chess_alone.json 2> $OUT/trace_alone.err
timeout 300 python $R/tools/chess_pass_alone.py > $OUT/chess_alone_untraced.json 2> $OUT/chess_alone_untraced.err
timeout 600 python $R/bench.py --workload c2_1920x1080_level0 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_c2 -o t -- python $R/bench.py --workload c2_1920x1080_level0 --steps 40 --warmup 5 --no-cpu-baseline --no-find-boards --no-end-to-end > /dev/null 2> $OUT/trace_c2.err
timeout 300 python $R/bench.py --gpus 2 --rehearse --workload c1_640x480_chain --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_rehearsal.json 2> $OUT/bench_rehearsal.err
for v in 16 1; do
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/pmc_a$v -o p -- python $R/tools/chess16_pmc.py $v > /dev/null 2> $OUT/pmc_a$v.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_b$v -o p -- python $R/tools/chess16_pmc.py $v > /dev/null 2> $OUT/pmc_b$v.err
done
# EA (fabric) traffic of the plain ChESS pass (chess_v16_kernel, 32 frames of 4096x3072): reads by size, writes
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B --kernel-trace --output-format csv -d $OUT/pmc_ard -o p -- python $R/tools/chess16_pmc.py 16 > /dev/null 2> $OUT/pmc_ard.err
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --output-format csv -d $OUT/pmc_awr -o p -- python $R/tools/chess16_pmc.py 16 > /dev/null 2> $OUT/pmc_awr.err
timeout 300 python $R/tools/chess16_sweep.py > $OUT/chess16_sweep.txt 2> $OUT/chess16_sweep.err
timeout 300 python $R/tools/sparse_subsets_ab.py > $OUT/sparse_subsets_ab.txt 2> $OUT/sparse_subsets_ab.err
# 4f. round 6: balanced segment counts, every k against the automatic choice, at the sizes of configs 2 and 5
timeout 900 python $R/tools/seg_rounds_sweep.py 1920x1080 1280x800 2560x1440 4096x2160 > $OUT/seg_rounds_sweep.txt 2> $OUT/seg_rounds_sweep.err
# 5. preprocessing kernels (row (f)-2)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/pre -o t -- python $R/tools/preprocess_bench.py > $OUT/prebench.txt 2> $OUT/pre.err
# 5b. round 6: EA (fabric) traffic of the preprocessing kernels (3 B/px algorithmic: histogram read + fused read + write)
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B --kernel-trace --output-format csv -d $OUT/pmc_prd -o p -- python $R/tools/preprocess_bench.py 3 > /dev/null 2> $OUT/pmc_prd.err
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ TCC_EA0_WRREQ_64B --kernel-trace --output-format csv -d $OUT/pmc_pwr -o p -- python $R/tools/preprocess_bench.py 3 > /dev/null 2> $OUT/pmc_pwr.err
# summaries on the box (the raw rocprofv3 output is too big to travel back), then drop the raw files
python $R/tools/rocprof_summary.py $OUT/trace/t_results.db > $OUT/bench_kernel_trace.txt 2>> $OUT/trace.err
python $R/tools/rocprof_summary.py $OUT/pre/t_results.db > $OUT/preprocess_kernel_trace.txt 2>> $OUT/pre.err
python $R/tools/rocprof_summary.py $OUT/trace_clut/t_results.db > $OUT/cluttered_kernel_trace.txt 2>> $OUT/trace_clut.err
python $R/tools/rocprof_summary.py $OUT/trace_sparse/t_results.db > $OUT/sparse_kernel_trace.txt 2>> $OUT/trace_sparse.err
python $R/tools/rocprof_summary.py $OUT/trace_alone/t_results.db > $OUT/chess_alone_kernel_trace.txt 2>> $OUT/trace_alone.err
python $R/tools/rocprof_summary.py $OUT/trace_c2/t_results.db > $OUT/c2_kernel_trace.txt 2>> $OUT/trace_c2.err
for d in pmc_prd pmc_pwr pmc_sq1 pmc_sq2 pmc_a16 pmc_b16 pmc_a1 pmc_b1 pmc_ard pmc_awr; do
    python $R/tools/pmc_summary.py $OUT/$d/p_counter_collection.csv > $OUT/$d.txt 2>> $OUT/$d.err
done
rm -rf $OUT/pmc_prd $OUT/pmc_pwr $OUT/pmc_ard $OUT/pmc_awr $OUT/trace_alone $OUT/trace_c2 $OUT/pmc_a16 $OUT/pmc_b16 $OUT/pmc_a1 $OUT/pmc_b1 $OUT/trace $OUT/pre $OUT/trace_clut $OUT/trace_sparse $OUT/pmc_sq1 $OUT/pmc_sq2
ls -la $OUT
