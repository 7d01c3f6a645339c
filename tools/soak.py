"""Soak test of the pipelined chain: many steps queued back to back (no host sync), every step's corner lists
compared ON THE DEVICE with a reference computed one call at a time.  python tools/soak.py [steps] [sparse]
(`sparse`: the same with option sparse_refine; the reference is the dense schedule's.)
A race between the pixel stream, the component streams and the scratch-set rotation would show as a mismatch."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import mrgingham_amd
from mrgingham_amd import synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
sparse = len(sys.argv) > 2 and sys.argv[2] == "sparse"
for (W, H, B, gridn, sets) in ((4096, 3072, 64, 10, 2), (4096, 3072, 64, 14, 3), (640, 480, 64, 10, 0), (1920, 1080, 32, 10, 3),
                              (4096, 3072, 32, -10, 2)):      # gridn < 0: textured background (windowed refinement)
    P = 1024
    render = synth.board_batch
    if gridn < 0:
        render, gridn = synth.cluttered_board_batch, -gridn
    batches = [render(8, W, H, gridn, 8 * k, device='cuda').repeat(B // 8, 1, 1).contiguous() for k in range(3)]
    det = mrgingham_amd.Detector(0)
    det.set_option("sparse_refine", 0)          # the reference outputs below: the dense schedule
    if sets:
        det.set_option("scratch_sets", sets)
    refs = []
    for fr in batches:
        p, l, n = det.chain(fr, 3, P)
        refs.append((p.clone(), l.clone(), n.clone()))
    if sparse:
        det.set_option("sparse_refine", 2)
    outs = [(torch.empty((B, P, 2), dtype=torch.float64, device='cuda'), torch.empty((B, P), dtype=torch.int8, device='cuda'),
             torch.empty((B,), dtype=torch.int32, device='cuda')) for _ in range(4)]
    bad = torch.zeros(1, dtype=torch.int32, device='cuda')
    idx = torch.arange(P, device='cuda')[None, :]
    t0 = time.perf_counter()
    for s in range(steps):
        k = s % 3
        o = outs[s % 4]
        det.after_stream()                      # the comparison of four steps ago has read this buffer
        det.chain(batches[k], 3, P, out=o, sync=False)
        det.stream_wait()                       # torch's stream: after this step, on the device
        rp, rl, rn = refs[k]
        live = idx < rn[:, None]
        bad += (o[2] != rn).any().int() + ((o[0] != rp).any(-1) & live).any().int() + ((o[1] != rl) & live).any().int()
    det.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'sparse ' if sparse else ''}{W}x{H} x{B} gridn {gridn} sets {sets or 'auto'}: {steps} steps, {int(bad.item())} mismatching steps, "
          f"{dt / steps * 1e3:.3f} ms per step incl. the comparison", flush=True)
    det.close()
    del batches, outs, refs
