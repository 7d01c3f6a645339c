"""Interleaved A/B of ChESS kernel variants: python scratch/ab.py a.so b.so ...  (non-hot clamp kernel, 64 frames 4096x3072)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrgingham_amd import synth, _lib
W, H, B = 4096, 3072, 64
frames = synth.board_batch(4, W, H, 10, 0, device='cuda').repeat(B // 4, 1, 1).contiguous()
out = torch.empty((B, H, W), dtype=torch.int16, device='cuda')
ref = None
libs = []
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.mrgingham_amd_create.restype = ctypes.c_void_p
    L.mrgingham_amd_create.argtypes = [ctypes.c_int]
    L.mrgingham_amd_chess_response_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_lib.Frames), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    ctx = L.mrgingham_amd_create(0)
    libs.append((path, L, ctx))
fr = _lib.Frames(frames.data_ptr(), H * W, B, W, H, W)
def run(L, ctx, clamp=1):
    rc = L.mrgingham_amd_chess_response_batch(ctx, ctypes.byref(fr), 0, clamp, out.data_ptr(), None)
    assert rc == 0
for path, L, ctx in libs:
    run(L, ctx); torch.cuda.synchronize()
    chk = int(out.to(torch.int64).sum().item()), int((out.to(torch.int64) * 3 % 1000003).sum().item())
    if ref is None: ref = chk
    print(os.path.basename(path), "checksum", chk, "OK" if chk == ref else "MISMATCH")
times = {p: [] for p, _, _ in libs}
for rnd in range(7):
    for path, L, ctx in libs:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run(L, ctx)
        e1.record(); torch.cuda.synchronize()
        times[path].append(e0.elapsed_time(e1) / 5)
for p, t in times.items():
    t = sorted(t)
    print(f"{os.path.basename(p):40s} median {t[len(t)//2]*1e3:7.1f} us  min {t[0]*1e3:7.1f} us  -> {B*W*H*3/t[len(t)//2]/1e6:6.0f} GB/s")
