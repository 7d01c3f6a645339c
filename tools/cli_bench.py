"""End-to-end rate of the command-line tool on PGM files in a RAM-backed directory (decode + upload +
preprocessing + detection + refinement + vnlog): python tools/cli_bench.py [W H N]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from mrgingham_amd import synth
W, H, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 3072, 32)
cli = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mrgingham_amd", "bin", "mrgingham-amd-from-image")
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
frames = synth.board_batch(N, W, H, 10, 0, device="cuda" if torch.cuda.is_available() else "cpu").cpu().numpy()
for i in range(N):
    with open(os.path.join(d, f"f{i:03d}.pgm"), "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (W, H)); f.write(frames[i].tobytes())
def run(jobs, n, extra=()):
    t0 = time.perf_counter()
    r = subprocess.run([cli, "--jobs", str(jobs), *extra] + [os.path.join(d, f"f{i:03d}.pgm") for i in range(n)], capture_output=True, text=True)
    dt = time.perf_counter() - t0
    found = len({l.split()[0] for l in r.stdout.splitlines() if not l.startswith("#") and l.split()[1] != "-"})
    return dt, found


for jobs in (1, 2, 4, 8, 16):
    small = max(N // 8, 1)
    t_small, _ = run(jobs, small)
    dt, found = run(jobs, N)
    # (process start + HIP initialisation are in both runs: the difference prices the images alone)
    marginal = (N - small) / max(dt - t_small, 1e-9)
    print(f"{W}x{H}, {N} files, --jobs {jobs:2d}: {dt:6.2f} s wall ({t_small:.2f} s for {small} files) -> {N/dt:7.1f} images/s with process start "
          f"+ HIP init, {marginal:7.1f} images/s in the steady state, boards found {found}/{N}", flush=True)
