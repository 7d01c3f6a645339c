#!/bin/bash
# Round-5 early evidence (runs ON the GPU box): the bench line with the chess_pass_alone leg, a kernel trace of the plain
# ChESS pass alone, BASELINE config 2 (64 x 1920x1080, level-0 detect) as a bench line + kernel trace, and the
# two-rank rehearsal of the N > 1 bench flow.   gpurun --timeout 1500 -- 'bash tools/collect_r05_early.sh r05a'
set -u
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_shape.json 2> $OUT/bench_driver_shape.err
# the plain ChESS pass alone: hipEvents (the script's own line) and the tracer's durations of the same process
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_alone -o t -- python $R/tools/chess_pass_alone.py > $OUT/chess_alone.json 2> $OUT/trace_alone.err
python $R/tools/rocprof_summary.py $OUT/trace_alone/t_results.db > $OUT/chess_alone_kernel_trace.txt 2>> $OUT/trace_alone.err
timeout 300 python $R/tools/chess_pass_alone.py > $OUT/chess_alone_untraced.json 2> $OUT/chess_alone_untraced.err
timeout 300 python $R/tools/chess_pass_alone.py 1920 1080 64 > $OUT/chess_alone_c2_untraced.json 2>> $OUT/chess_alone_untraced.err
# config 2
timeout 600 python $R/bench.py --workload c2_1920x1080_level0 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/bench_c2.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_c2 -o t -- python $R/bench.py --workload c2_1920x1080_level0 --steps 40 --warmup 5 --no-cpu-baseline --no-find-boards --no-end-to-end > $OUT/trace_c2_bench.json 2> $OUT/trace_c2.err
python $R/tools/rocprof_summary.py $OUT/trace_c2/t_results.db > $OUT/c2_kernel_trace.txt 2>> $OUT/trace_c2.err
rm -rf $OUT/trace_alone $OUT/trace_c2
# the rehearsal
cd $R && timeout 900 python -m pytest tests/test_gpu_parallel.py -x -q > $OUT/pytest_parallel.log 2>&1
timeout 300 python $R/bench.py --gpus 2 --rehearse --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_rehearsal.json 2> $OUT/bench_rehearsal.err
ls -la $OUT
