"""Pipelined chain under different component-stream placements / schedules (one process per setting:
the CU masks are read when the context is created).  python tools/interference_ab.py [filter]
Prints ms per step and the average level-0 ChESS launch inside the pipeline for every setting."""
import sys, os, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import mrgingham_amd
    from mrgingham_amd import synth
    W, H, B, P = 4096, 3072, 64, 256
    frames = synth.board_batch(8, W, H, 10, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
    det = mrgingham_amd.Detector(0)
    for kv in os.environ.get("OPTIONS", "").split(","):
        if kv:
            k, v = kv.split("=")
            det.set_option(k, int(v))
    outs = [(torch.empty((B, P, 2), dtype=torch.float64, device='cuda'), torch.empty((B, P), dtype=torch.int8, device='cuda'),
             torch.empty((B,), dtype=torch.int32, device='cuda')) for _ in range(3)]
    res = []
    for rnd in range(3):
        for i in range(40): det.chain(frames, 3, P, out=outs[i % 3], sync=False)
        det.sync(); det.set_kernel_timing(True); det.chess_kernel_ms()
        t0 = time.perf_counter()
        for i in range(200): det.chain(frames, 3, P, out=outs[i % 3], sync=False)
        det.sync(); dt = time.perf_counter() - t0
        ms, n = det.chess_kernel_ms(); det.set_kernel_timing(False)
        res.append((dt / 200 * 1e3, ms * 1e3))
    res.sort()
    print(f"{os.environ.get('LABEL', ''):46s} step {res[1][0]:.3f} ms   L0 launch {res[1][1]:.1f} us "
          f"({B*W*H*3/res[1][1]/1e3/80:.1f} %)  npts {int(outs[0][2][0])}", flush=True)
    sys.exit(0)
settings = [
    ("baseline", {}),
    ("cc on 1 CU/XCD", {"MRGINGHAM_AMD_CC_CUS": "1"}),
    ("cc on 2 CU/XCD", {"MRGINGHAM_AMD_CC_CUS": "2"}),
    ("cc on 4 CU/XCD", {"MRGINGHAM_AMD_CC_CUS": "4"}),
    ("cc on 8 CU/XCD", {"MRGINGHAM_AMD_CC_CUS": "8"}),
    ("cc on 2 CU/XCD, pix on the other 30", {"MRGINGHAM_AMD_CC_CUS": "2", "MRGINGHAM_AMD_PIX_COMPLEMENT": "1"}),
    ("cc on 1 CU/XCD, pix on the other 31", {"MRGINGHAM_AMD_CC_CUS": "1", "MRGINGHAM_AMD_PIX_COMPLEMENT": "1"}),
    ("schedule 1 (L1, L0 chains after ChESS L0)", {"OPTIONS": "cc_schedule=1"}),
    ("schedule 2 (all chains after ChESS L0)", {"OPTIONS": "cc_schedule=2"}),
    ("schedule 1 + cc on 2 CU/XCD", {"OPTIONS": "cc_schedule=1", "MRGINGHAM_AMD_CC_CUS": "2"}),
    ("schedule 2 + cc on 2 CU/XCD", {"OPTIONS": "cc_schedule=2", "MRGINGHAM_AMD_CC_CUS": "2"}),
    ("multi-level launch", {"OPTIONS": "multi_level_launch=1"}),
    ("multi-level launch + schedule 1", {"OPTIONS": "multi_level_launch=1,cc_schedule=1"}),
    ("multi-level + schedule 1 + cc 2 CU/XCD", {"OPTIONS": "multi_level_launch=1,cc_schedule=1", "MRGINGHAM_AMD_CC_CUS": "2"}),
    ("all four levels in one launch", {"OPTIONS": "multi_level_launch=2"}),
    ("all four levels in one launch + schedule 1", {"OPTIONS": "multi_level_launch=2,cc_schedule=1"}),
    ("merged small levels: min blocks 1024", {"OPTIONS": "chess_multi_min_blocks=1024"}),
    ("merged small levels: min blocks 512", {"OPTIONS": "chess_multi_min_blocks=512"}),
    ("merged small levels: min blocks 256", {"OPTIONS": "chess_multi_min_blocks=256"}),
    ("LDS search without s_setprio 3", {"OPTIONS": "cc_lds=17"}),
    ("dbg fused, nothing emitted", {"MRGINGHAM_AMD_PYR_SKIP": "7"}),
    ("dbg fused, level 1 only", {"MRGINGHAM_AMD_PYR_SKIP": "6"}),
    ("dbg fused, levels 2+3 only", {"MRGINGHAM_AMD_PYR_SKIP": "1"}),
    ("dbg fused, level 3 only", {"MRGINGHAM_AMD_PYR_SKIP": "3"}),
    ("pixel kernels only (no component kernels; results meaningless)", {"OPTIONS": "cc_lds=129"}),
    ("pixel kernels only, separate pyramid kernel", {"OPTIONS": "cc_lds=129,fuse_pyramid=0"}),
    ("cc LDS +896 B (40928: still one slot of 1280-B granules)", {"MRGINGHAM_AMD_CC_LDS_PAD": "896"}),
    ("cc LDS +1024 B (41056: 33 granules)", {"MRGINGHAM_AMD_CC_LDS_PAD": "1024"}),
    ("cc LDS +3900 B (43932: below 163840 - 3 * 39952)", {"MRGINGHAM_AMD_CC_LDS_PAD": "3900"}),
    ("three scratch sets", {"OPTIONS": "scratch_sets=3"}),
    ("separate pyramid kernel", {"OPTIONS": "fuse_pyramid=0"}),
    ("fused pyramid + schedule 2", {"OPTIONS": "cc_schedule=2"}),
    ("baseline again", {}),
]
import glob
for so in sorted(glob.glob(os.path.join(ROOT, "scratch", "lib_*.so"))):     # tools/build_variant.sh outputs
    settings.append(("variant: " + os.path.basename(so), {"MRGINGHAM_AMD_LIB": so}))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for label, env in settings:
    if flt and flt not in label:
        continue
    e = dict(os.environ); e.update(env); e["LABEL"] = label
    subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e)
