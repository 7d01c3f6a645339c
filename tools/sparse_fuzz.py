"""Randomised check of option sparse_refine: random frame sizes (odd ones included), boards, noise overlays, textured
backgrounds and start levels; whatever the sparse schedule ACCEPTS must equal the dense schedule's output on every
frame, and what it does not accept must be reported (the in-library dense repeat), never answered differently.
python tools/sparse_fuzz.py [iterations] [seed]"""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mrgingham_amd
from mrgingham_amd import synth


def make_frames(rng, dev):
    W = rng.choice([640, 801, 1024, 1283, 1920, 2048, 3000, 4096]) + rng.choice([0, 0, 1, 7, 16])
    H = max(240, int(W * rng.choice([0.5625, 0.75, 1.0])) + rng.choice([0, 0, 3, 8]))
    B = rng.choice([1, 2, 3, 5])
    gridn = rng.choice([6, 8, 10, 12, 14])
    kind = rng.choice(["clean", "clean", "clutter", "noise", "noise_smooth", "mixed"])
    seed = rng.randrange(1 << 20)
    if kind == "clutter":
        fr = synth.cluttered_board_batch(B, W, H, gridn, seed, device=dev, smooth=rng.choice([1, 2, 3]), amp=rng.choice([64, 128, 200]))
    else:
        fr = synth.board_batch(B, W, H, gridn, seed, device=dev)
        if kind in ("noise", "noise_smooth", "mixed"):
            sm = 0 if kind == "noise" else rng.choice([1, 2])
            amp = rng.choice([20, 40, 80, 120])
            nz = torch.stack([synth.noise_frame(W, H, seed=seed + 7 + b, smooth=sm, device=dev) for b in range(B)]).to(torch.int64)
            fr = (fr.to(torch.int64) + (nz - 128) * amp // 255).clamp(0, 255).to(torch.uint8)
            if kind == "mixed":
                fr[0] = synth.board_frame(W, H, gridn, seed + 99, device=dev)
    return fr, (W, H, B, gridn, kind)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dev = torch.device("cuda:0")
    dense, sparse = mrgingham_amd.Detector(0), mrgingham_amd.Detector(0)
    dense.set_option("sparse_refine", 0)
    sparse.set_option("sparse_refine", 2)
    accepted = reported = frames = bad = 0
    for it in range(iters):
        fr, desc = make_frames(rng, dev)
        start = rng.choice([1, 2, 3, 3, 3, 4])
        P = rng.choice([256, 1024, 2048])
        try:
            want = dense.chain(fr, start, P)
        except RuntimeError as e:
            print("dense failed", desc, start, e)
            continue
        try:
            got = sparse.chain(fr, start, P)
        except RuntimeError as e:
            print("sparse failed", desc, start, e)
            bad += 1
            continue
        reported += sparse.sparse_fallbacks()
        accepted += 1
        same = torch.equal(want[2], got[2])
        n = want[2].clamp(max=P).tolist()
        for f in range(fr.shape[0]):
            same = same and torch.equal(want[0][f, :n[f]], got[0][f, :n[f]]) and torch.equal(want[1][f, :n[f]], got[1][f, :n[f]])
        frames += fr.shape[0]
        if not same:
            bad += 1
            print("MISMATCH", desc, "start", start, "P", P, want[2].tolist(), got[2].tolist(), flush=True)
    print(f"{iters} calls: {accepted} compared ({frames} frames), {reported} frames repeated densely by the library, {bad} mismatching")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
