"""Segment height of the level-0 response kernel (option "chess_seg") for a batch shape:
python tools/seg_ab.py W H [B]   -- pipelined level-0 detect calls, ms per step and us per ChESS launch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import mrgingham_amd
from mrgingham_amd import synth
W, H = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
frames = synth.board_batch(8, W, H, 10, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
for seg in (0, 256, 128, 64, 32):
    det = mrgingham_amd.Detector(0)
    det.set_option("chess_seg", seg)
    for i in range(40): det.detect(frames, 0, capacity=256, sync=False)
    det.sync(); det.set_kernel_timing(True); det.chess_kernel_ms()
    t0 = time.perf_counter()
    for i in range(200): det.detect(frames, 0, capacity=256, sync=False)
    det.sync(); dt = time.perf_counter() - t0
    ms, n = det.chess_kernel_ms()
    print(f"{W}x{H} x{B} chess_seg {seg if seg else 'auto':>4}: step {dt / 200 * 1e3:.4f} ms, ChESS launch {ms * 1e3:.1f} us", flush=True)
    det.close()
