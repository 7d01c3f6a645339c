cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_board.py tests/test_gpu_fuzz.py tests/test_c_client.py tests/test_cvmat_shim.py tests/test_gpu_robustness.py -q -m gpu 2>&1 | tail -2
timeout 300 python tools/find_boards_fuzz.py 300 77 2>&1 | tail -1
