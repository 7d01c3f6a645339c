"""Single-image latencies through the reference-symbol wrappers (host image in, host result out), the blob path,
the 16-bit preprocessing, and the staging paths on a ragged width.  python tools/latency_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import mrgingham_amd
from mrgingham_amd import synth


def med(f, n=15, warm=3):
    for _ in range(warm): f()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


for (w, h) in [(640, 480), (1920, 1080), (4096, 3072)]:
    img = synth.board_frame(w, h, 10, 1, device="cuda").cpu().numpy()
    pin = mrgingham_amd.api.PinnedArray(img.shape)          # the same frame in page-locked memory (mrgingham_amd_host_alloc)
    pin.array[...] = img
    pimg = pin.array
    print(f"{w}x{h} from page-locked host memory: ChESS_response_5 {med(lambda: mrgingham_amd.ChESS_response_5(pimg)):.2f} ms, "
          f"find_points L0 {med(lambda: mrgingham_amd.find_points(pimg, 0)):.2f} ms, "
          f"find_points L2 {med(lambda: mrgingham_amd.find_points(pimg, 2)):.2f} ms, "
          f"find_board (level search + refinement) {med(lambda: mrgingham_amd.find_board(pimg)):.2f} ms", flush=True)
    print(f"{w}x{h}: ChESS_response_5 {med(lambda: mrgingham_amd.ChESS_response_5(img)):.2f} ms, "
          f"find_points L0 {med(lambda: mrgingham_amd.find_points(img, 0)):.2f} ms, "
          f"find_points L2 {med(lambda: mrgingham_amd.find_points(img, 2)):.2f} ms, "
          f"find_board (level search + refinement) {med(lambda: mrgingham_amd.find_board(img)):.2f} ms", flush=True)
for (w, h) in [(1280, 960), (4096, 3072)]:
    dots = synth.dots_frame(w, h, 10, 2, device="cuda").cpu().numpy()
    n = len(mrgingham_amd.find_points(dots, 0, blobs=True))
    print(f"{w}x{h} circle grid: find_points(blobs=True) {med(lambda: mrgingham_amd.find_points(dots, 0, blobs=True), n=7):.2f} ms "
          f"({n} blobs), find_board(blobs=True) {med(lambda: mrgingham_amd.find_board(dots, 0, blobs=True), n=7):.2f} ms", flush=True)
noise = synth.noise_frame(1280, 960, 1, smooth=2, device="cuda").cpu().numpy()
print(f"1280x960 smoothed noise: find_points(blobs=True) {med(lambda: mrgingham_amd.find_points(noise, 0, blobs=True), n=5):.2f} ms", flush=True)
img16 = (synth.board_frame(4096, 3072, 10, 1, device="cuda").cpu().numpy().astype(np.uint16) * 120 + 9000)
print(f"4096x3072 16-bit: preprocess16 (normalize + CLAHE + convert + blur, host in / host out) "
      f"{med(lambda: mrgingham_amd.api.preprocess16(img16), n=7):.2f} ms; 8-bit preprocess "
      f"{med(lambda: mrgingham_amd.preprocess((img16 >> 8).astype(np.uint8)), n=7):.2f} ms", flush=True)
det = mrgingham_amd.Detector(0)
B = 32
frames = synth.board_batch(4, 4090, 3070, 10, 0, device="cuda").repeat(B // 4, 1, 1).contiguous()
out = torch.empty((B, 3070, 4090), dtype=torch.int16, device="cuda")
for st, name in [(0, "generic (divergent edge handling)"), (3, "typed P0 + alignbit P1"), (2, "typed P0 + P1")]:
    try:
        det.set_option("chess_stage", st)
    except ValueError:
        print("(staging variants: experiment builds only -- make -C mrgingham_amd/csrc EXPERIMENT=1, MRGINGHAM_AMD_LIB)")
        break
    for _ in range(10): det.chess_response(frames, 0, clamp=True, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): det.chess_response(frames, 0, clamp=True, out=out)
    torch.cuda.synchronize()
    print(f"4090x3070 (width not a multiple of 16), {B} frames, staging {name}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per launch", flush=True)
