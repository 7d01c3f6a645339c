"""Dense vs sparse refinement schedule of chain(): equality of every output and ms per step.
usage: python tools/sparse_ab.py [W H B gridn clutter]"""
import sys, time
import torch
sys.path.insert(0, ".")
from mrgingham_amd import Detector, synth


def run(W, H, B, gridn, clutter, steps=30):
    dev = torch.device("cuda:0")
    mk = synth.cluttered_board_batch if clutter else synth.board_batch
    frames = mk(B, W, H, gridn, 0, device=dev)
    res = {}
    for mode in (0, 1):
        det = Detector(0)
        det.set_option("sparse_refine", 2 * mode)
        try:
            out = det.chain(frames, 3, 1024, retry=False)
        except RuntimeError as e:
            print(f"{W}x{H} B={B} clutter={clutter} sparse={mode}: {e}")
            det.close()
            continue
        outs = [out] + [tuple(torch.empty_like(o) for o in out) for _ in range(2)]   # (one output set per step in flight)
        for i in range(3):
            det.chain(frames, 3, 1024, out=outs[i % 3], sync=False)
        det.sync()
        t0 = time.perf_counter()
        for i in range(steps):
            det.chain(frames, 3, 1024, out=outs[i % 3], sync=False)
        det.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        res[mode] = [o.clone() for o in out]
        det.close()          # (its streams hold hardware queues: a second live context would share one between its streams)
        print(f"{W}x{H} B={B} gridn={gridn} clutter={clutter} sparse={mode}: {ms:.3f} ms/step "
              f"npoints {out[2][:4].tolist()}", flush=True)
    if len(res) == 2:
        n = res[0][2]
        same = torch.equal(res[0][2], res[1][2])
        for b in range(B):
            k = int(n[b])
            same &= torch.equal(res[0][0][b, :k], res[1][0][b, :k]) and torch.equal(res[0][1][b, :k], res[1][1][b, :k])
        print("  identical:", bool(same), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        a = [int(x) for x in sys.argv[1:]]
        run(*a)
    else:
        run(1024, 768, 64, 10, 0)
        run(4096, 3072, 64, 10, 0)
        run(4096, 3072, 64, 14, 0)
        run(4096, 3072, 16, 10, 1)
        run(640, 480, 64, 10, 0)
