"""A/B of whole-library builds on what mrgingham_amd_sync costs: python tools/sync_ab.py a.so b.so [--rounds N]
One child process per library and round (MRGINGHAM_AMD_LIB), interleaved.  Per library: median milliseconds of a
SYNCHRONOUS chain call (detect at level 3, refine to 0: four levels' status words come back in the sync) on one
12 MP frame, on 64 frames of 640x480, and of a 20-step region of the bench workload (64 x 4096x3072, dense schedule,
steps queued back to back, one sync at the end) -- and a checksum of the corner lists (must agree)."""
import sys, os, subprocess, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import mrgingham_amd
    from mrgingham_amd import synth
    P = 256
    det = mrgingham_amd.Detector(0)
    det.set_option("sparse_refine", 0)
    def med(f, n=41, warm=10):
        for _ in range(warm): f()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3
    one = synth.board_batch(1, 4096, 3072, 10, 3, device="cuda")
    small = synth.board_batch(64, 640, 480, 10, 0, device="cuda")
    big = synth.board_batch(8, 4096, 3072, 10, 0, device="cuda").repeat(8, 1, 1).contiguous()
    r = {}
    for name, fr in (("one_12mp", one), ("64x640x480", small)):
        B = fr.shape[0]
        out = (torch.empty((B, P, 2), dtype=torch.float64, device="cuda"), torch.empty((B, P), dtype=torch.int8, device="cuda"),
               torch.empty((B,), dtype=torch.int32, device="cuda"))
        r[name] = med(lambda: det.chain(fr, 3, P, out=out, sync=True))
        r["chk_" + name] = float(out[0][0, :int(out[2][0])].sum().item())
    outs = [(torch.empty((64, P, 2), dtype=torch.float64, device="cuda"), torch.empty((64, P), dtype=torch.int8, device="cuda"),
             torch.empty((64,), dtype=torch.int32, device="cuda")) for _ in range(3)]
    for i in range(60): det.chain(big, 3, P, out=outs[i % 3], sync=False)
    det.sync()
    def region():
        for i in range(20): det.chain(big, 3, P, out=outs[i % 3], sync=False)
        det.sync(); torch.cuda.synchronize()
    r["region20_ms"] = med(region, n=15, warm=3)
    print(json.dumps(r), flush=True)
    sys.exit(0)

args = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = 3
for i, a in enumerate(sys.argv):
    if a == "--rounds": rounds = int(sys.argv[i + 1]); args.remove(sys.argv[i + 1])
acc = {a: [] for a in args}
for rnd in range(rounds):
    for a in args:
        e = dict(os.environ); e["MRGINGHAM_AMD_LIB"] = os.path.abspath(a)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(a, "FAILED", p.stderr[-600:]); continue
        acc[a].append(json.loads(line[-1]))
for a, rs in acc.items():
    if not rs: continue
    m = lambda k: sorted(x[k] for x in rs)[len(rs) // 2]
    print(f"{a:14s} chain+sync one 12 MP frame {m('one_12mp'):.4f} ms | 64 x 640x480 {m('64x640x480'):.4f} ms | 20-step region {m('region20_ms'):.3f} ms "
          f"({64 * 20 / m('region20_ms'):.1f} k frames/s) | chk {rs[0]['chk_one_12mp']:.6f} {rs[0]['chk_64x640x480']:.6f}", flush=True)
