cd $GRAFT_REPO_ROOT
gcc -O2 -std=c99 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lmrgingham_amd -lmrgingham_amd -Wl,-rpath,$PWD/mrgingham_amd -Wl,-rpath,/opt/rocm/lib
python - <<'PY'
import sys; sys.path.insert(0,'.')
from mrgingham_amd import synth
for (W,H) in ((640,480),(1920,1080),(4096,3072)):
    img=synth.board_frame(W,H,10,3).numpy()
    open(f'/tmp/b_{W}.pgm','wb').write(b"P5\n%d %d\n255\n"%(W,H)+img.tobytes())
PY
for W in 640 1920 4096; do /tmp/latency_c /tmp/b_$W.pgm 300 2>&1 | grep -v amdgpu | cut -c60-; done
python -m pytest tests/test_gpu_board.py tests/test_gpu_fuzz.py tests/test_c_client.py tests/test_cvmat_shim.py tests/test_gpu_robustness.py tests/test_cli.py -q -m gpu 2>&1 | tail -2
timeout 300 python tools/find_boards_fuzz.py 300 78 2>&1 | tail -1
for i in 1 2; do python tools/find_boards_bench.py --one 4096 3072 64 3 0 2>/dev/null | tail -1 | cut -c1-230; done
