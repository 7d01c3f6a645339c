R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trsp; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t -o t -- python $R/bench.py --sparse-refine --steps 40 --warmup 5 --no-cpu-baseline --no-end-to-end --no-find-boards --no-chess-alone > $OUT/out.txt 2> $OUT/err.txt
python $R/tools/rocprof_summary.py $OUT/t/t_results.db > $OUT/trace.txt 2>> $OUT/err.txt
rm -rf $OUT/t
grep -v "^#" $OUT/trace.txt | head -14
