"""Steady-state interleaved A/B of ChESS kernel variants: python tools/chess_ab.py a.so b.so[:seg] ...  (libraries built e.g. with tools/build_variant.sh)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mrgingham_amd import synth, _lib
W, H, B = 4096, 3072, 64
frames = synth.board_batch(4, W, H, 10, 0, device='cuda').repeat(B // 4, 1, 1).contiguous()
out = torch.empty((B, H, W), dtype=torch.int16, device='cuda')
libs = []
for spec in sys.argv[1:]:
    path, _, seg = spec.partition(":")
    L = ctypes.CDLL(os.path.abspath(path))
    L.mrgingham_amd_create.restype = ctypes.c_void_p
    L.mrgingham_amd_create.argtypes = [ctypes.c_int]
    L.mrgingham_amd_chess_response_batch.argtypes = [ctypes.c_void_p, ctypes.POINTER(_lib.Frames), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.mrgingham_amd_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]
    ctx = L.mrgingham_amd_create(0)
    libs.append((spec, L, ctx, int(seg) if seg else 0))
fr = _lib.Frames(frames.data_ptr(), H * W, B, W, H, W)
def run(L, ctx, seg):
    L.mrgingham_amd_set_option(ctx, b"chess_seg", seg)
    rc = L.mrgingham_amd_chess_response_batch(ctx, ctypes.byref(fr), 0, 1, out.data_ptr(), None)
    assert rc == 0
ref = None
for spec, L, ctx, seg in libs:
    run(L, ctx, seg); torch.cuda.synchronize()
    chk = int(out.to(torch.int64).sum().item())
    if ref is None: ref = chk
    print(spec, "OK" if chk == ref else "MISMATCH")
for spec, L, ctx, seg in libs:            # long warm-up
    for _ in range(60): run(L, ctx, seg)
torch.cuda.synchronize()
times = {s: [] for s, _, _, _ in libs}
for rnd in range(9):
    for spec, L, ctx, seg in libs:
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run(L, ctx, seg)
        e1.record(); torch.cuda.synchronize()
        times[spec].append(e0.elapsed_time(e1) / 20)
for p, t in times.items():
    t = sorted(t)
    print(f"{p:40s} median {t[len(t)//2]*1e3:7.1f} us  min {t[0]*1e3:7.1f} us  -> {B*W*H*3/t[len(t)//2]/1e6/80:5.1f} % of 8 TB/s")
