#!/bin/bash
# Long randomised validation of the final library of a round (runs ON the GPU box, ~20 minutes):
#   gpurun --timeout 2400 -- 'bash tools/validate_long.sh r05v'
TAG=${1:-val}
S=${2:-1}   # scale: 1 = ~20 minutes, 3 = ~1 hour (iterations and time limits go with it)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
{
echo "== sparse_fuzz $((2500*S)) (seed 9001)"; timeout $((900*S)) python tools/sparse_fuzz.py $((2500*S)) 9001 2>&1 | grep -v "component tables overflowed" | tail -2
echo "== find_boards_fuzz $((2000*S)) (seed 9002)"; timeout $((900*S)) python tools/find_boards_fuzz.py $((2000*S)) 9002 2>&1 | grep -v "component tables overflowed" | tail -2
echo "== soak 6000 dense"; timeout 600 python tools/soak.py 6000 2>&1 | tail -2
echo "== soak 8000 sparse"; timeout 600 python tools/soak.py 8000 sparse 2>&1 | tail -2
echo "== stress_threads 150 s"; timeout 400 python tools/stress_threads.py 150 2>&1 | tail -2
echo "== tests/test_gpu_fuzz.py, $((1500*S)) iterations"; MRG_FUZZ_ITERS=$((1500*S)) timeout $((1200*S)) python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
echo "== chess16 against chess_v1 and sizes"; timeout 300 python tools/chess16_ab.py --no-time 2>&1 | grep -c "^same"
} > $OUT/validate.txt 2>&1
cat $OUT/validate.txt
