"""A/B of the ChESS staging paths on the GPU: python tools/stage_ab.py [stage ...]
(stages: 0 = automatic (v_perm), 2 = typed P0+P1, 3 = typed P0 + alignbit P1).  Per stage: equality of
the response with the automatic path, the level-0 launch alone (clamp, no hot list; interleaved rounds),
and the pipelined chain (ms per step, level-0 launch inside it)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import mrgingham_amd
from mrgingham_amd import synth
W, H, B = 4096, 3072, 64
stages = [int(a) for a in sys.argv[1:]] or [0, 2, 3]
frames = synth.board_batch(8, W, H, 10, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
det = mrgingham_amd.Detector(0)
out = torch.empty((B, H, W), dtype=torch.int16, device='cuda')
det.set_option("chess_stage", 0)
ref = det.chess_response(frames, 0, clamp=True).clone()
for st in stages:
    det.set_option("chess_stage", st)
    r = det.chess_response(frames, 0, clamp=True, out=out)
    print(f"stage {st}: response {'equal' if torch.equal(r, ref) else 'MISMATCH'}", flush=True)
    odd = frames[:2, :1001, :1003].contiguous()
    det.set_option("chess_stage", 0); a = det.chess_response(odd, 0, clamp=False).clone()
    det.set_option("chess_stage", st); b = det.chess_response(odd, 0, clamp=False)
    print(f"stage {st}: ragged 1003x1001 {'equal' if torch.equal(a, b) else 'MISMATCH'}", flush=True)
del ref
for st in stages:
    det.set_option("chess_stage", st)
    for _ in range(40): det.chess_response(frames, 0, clamp=True, out=out)
torch.cuda.synchronize()
times = {s: [] for s in stages}
for rnd in range(7):
    for st in stages:
        det.set_option("chess_stage", st)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): det.chess_response(frames, 0, clamp=True, out=out)
        e1.record(); torch.cuda.synchronize()
        times[st].append(e0.elapsed_time(e1) / 20)
for st, t in times.items():
    t = sorted(t)
    print(f"stage {st}: alone median {t[len(t)//2]*1e3:7.1f} us  min {t[0]*1e3:7.1f} us -> {B*W*H*3/t[len(t)//2]/1e6/80:5.1f} % of 8 TB/s", flush=True)
P = 256
outs = [(torch.empty((B, P, 2), dtype=torch.float64, device='cuda'), torch.empty((B, P), dtype=torch.int8, device='cuda'),
         torch.empty((B,), dtype=torch.int32, device='cuda')) for _ in range(3)]
for rnd in range(2):
    for st in stages:
        det.set_option("chess_stage", st)
        for i in range(30): det.chain(frames, 3, P, out=outs[i % 3], sync=False)
        det.sync(); det.set_kernel_timing(True); det.chess_kernel_ms()
        t0 = time.perf_counter()
        for i in range(150): det.chain(frames, 3, P, out=outs[i % 3], sync=False)
        det.sync(); dt = time.perf_counter() - t0
        ms, n = det.chess_kernel_ms(); det.set_kernel_timing(False)
        print(f"stage {st}: chain {dt/150*1e3:.3f} ms/step, level-0 launch {ms*1e3:.1f} us ({B*W*H*3/ms/1e6/80:.1f} %), npts {int(outs[0][2][0])}", flush=True)
