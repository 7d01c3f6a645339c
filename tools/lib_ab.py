"""A/B of whole-library builds on the bench workload: python tools/lib_ab.py a.so b.so ... [--rounds N] [--gridn 10]
One child process per library and round (MRGINGHAM_AMD_LIB), rounds interleaved; prints per library the median step
time of the pipelined chain (64 x 4096x3072), the level-0 launch inside it, the level-0 kernel alone (plain response
kernel, clamp, no hot list) and a checksum of the corner lists (must agree between builds)."""
import sys, os, subprocess, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import mrgingham_amd
    from mrgingham_amd import synth
    W, H, B, P = 4096, 3072, 64, 256
    gridn = int(os.environ.get("GRIDN", "10"))
    frames = synth.board_batch(8, W, H, gridn, 0, device='cuda').repeat(B // 8, 1, 1).contiguous()
    det = mrgingham_amd.Detector(0)
    try:
        det.set_option("sparse_refine", 0)      # the dense schedule (the library's default is 1 since round 4)
    except ValueError:
        pass
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
    n0 = int(outs[0][2][0])
    chk = float(outs[0][0][:, :n0].sum().item()) + float(outs[0][2].sum().item()) * 1e6
    # the plain response kernel alone
    out = torch.empty((B, H, W), dtype=torch.int16, device='cuda')
    for _ in range(30): det.chess_response(frames, 0, clamp=True, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): det.chess_response(frames, 0, clamp=True, out=out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    ts.sort()
    rchk = int(out[::7, ::5, ::3].to(torch.int64).sum().item())
    print(json.dumps({"step_ms": res[1][0], "l0_us": res[1][1], "alone_us": ts[2], "chk": chk, "rchk": rchk, "npts": n0}), flush=True)
    sys.exit(0)

args = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = 2
for i, a in enumerate(sys.argv):
    if a == "--rounds": rounds = int(sys.argv[i + 1]); args.remove(sys.argv[i + 1])
    if a == "--gridn": os.environ["GRIDN"] = sys.argv[i + 1]; args.remove(sys.argv[i + 1])
acc = {a: [] for a in args}
for r in range(rounds):
    for a in args:
        lib, _, opts = a.partition(":")
        e = dict(os.environ); e["MRGINGHAM_AMD_LIB"] = os.path.abspath(lib)
        if opts: e["OPTIONS"] = opts
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(a, "FAILED", p.stderr[-400:]); continue
        acc[a].append(json.loads(line[-1]))
for a, rs in acc.items():
    if not rs: continue
    med = lambda k: sorted(r[k] for r in rs)[len(rs) // 2]
    print(f"{a:44s} step {med('step_ms'):.4f} ms  L0 in chain {med('l0_us'):6.1f} us  plain alone {med('alone_us'):6.1f} us  "
          f"chk {rs[0]['chk']:.6f} rchk {rs[0]['rchk']} npts {rs[0]['npts']}", flush=True)
