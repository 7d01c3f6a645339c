set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc16; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for v in 16 0; do
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/a$v -o p -- python $R/tools/chess16_pmc.py $v > /dev/null 2> $OUT/a$v.err
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/b$v -o p -- python $R/tools/chess16_pmc.py $v > /dev/null 2> $OUT/b$v.err
python $R/tools/pmc_summary.py $OUT/a$v/p_counter_collection.csv > $OUT/a$v.txt 2>> $OUT/a$v.err
python $R/tools/pmc_summary.py $OUT/b$v/p_counter_collection.csv > $OUT/b$v.txt 2>> $OUT/b$v.err
rm -rf $OUT/a$v $OUT/b$v
done
grep -A12 "chess_v" $OUT/a16.txt $OUT/b16.txt $OUT/a0.txt $OUT/b0.txt | head -120
