"""Row (f)-2: the CLI's contrast preprocessing on the device against the oracle's restatement of
OpenCV's published normalize / CLAHE / blur arithmetic (parity unpinned: OpenCV itself is absent).
Bit-exact, through the C-ABI (mrgingham_amd_preprocess_batch)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(kind, H, W, n, seed):
    from mrgingham_amd import synth
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        if kind == "board":
            f = synth.board_frame(W, H, gridn=10, seed=seed + i).numpy().astype(np.float64)
            f = f * rng.uniform(0.3, 0.8) + rng.uniform(0, 50)          # reduced dynamic range
        elif kind == "noise":
            f = rng.rand(H, W) * rng.uniform(60, 255)
        elif kind == "ramp":
            f = np.add.outer(np.arange(H) * 0.05, np.arange(W) * 0.1) + 20
        else:
            f = np.full((H, W), 77.0)
        out.append(np.clip(f, 0, 255).astype(np.uint8))
    return np.stack(out)


@pytest.mark.parametrize("kind,H,W", [
    ("board", 480, 640), ("noise", 96, 128), ("ramp", 200, 320), ("flat", 64, 64),
    ("board", 487, 643),     # neither side divides into the 8x8 tile grid: both sides get padded
    ("noise", 96, 131),      # only the width is ragged (OpenCV still pads the height by 8)
    ("noise", 101, 128),
    ("board", 1080, 1920),
    ("noise", 600, 1296),    # the fused blend + blur kernel: two 1024-column workgroups, the second partial; 16 rows per wave
    ("ramp", 272, 2064),     # ... three of them, 8 rows per wave, cells 162 px wide (nine across a workgroup)
])
@pytest.mark.parametrize("clahe,blur", [(True, 1), (True, 0), (False, 1), (False, 2), (True, 3)])
def test_preprocess_matches_oracle(kind, H, W, clahe, blur):
    import torch
    import mrgingham_amd
    from oracle import oracle
    frames = _frames(kind, H, W, 3 if H * W < 1000000 else 1, seed=H + W)   # (the oracle takes 7 s per 2 MP frame)
    det = mrgingham_amd.Detector(0)
    got = det.preprocess(torch.from_numpy(frames).cuda(), clahe=clahe, blur_radius=blur)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    for i in range(len(frames)):
        want = oracle.preprocess(frames[i], clahe=clahe, blur_radius=blur)
        assert np.array_equal(got[i], want), (kind, H, W, clahe, blur, i, int(np.abs(got[i].astype(int) - want).max()))


@pytest.mark.parametrize("H,W", [(3072, 4096), (1080, 1920), (1536, 2048), (1440, 2560), (800, 1280), (2160, 4096), (1042, 1936), (274, 48)])
def test_fused_blend_and_blur_equals_the_two_kernels(H, W):
    """mrgingham-from-image.cc:71-111 as ONE pass (clahe_blur3_kernel) against the blend and the blur as two kernels
    (option preprocess_fused 0), which the oracle pins at the small sizes above: identical bytes at full sizes, on
    noise (every blend weight and every rounding case gets used) and on a board; batch of 3 with a strided view."""
    import torch
    import mrgingham_amd
    from mrgingham_amd import synth
    g = torch.Generator().manual_seed(H * 7 + W)
    noise = (torch.rand((H, W + 16), generator=g) * torch.rand((H, 1), generator=g) * 255).to(torch.uint8)
    frames = torch.stack([noise, torch.roll(noise, 5, 1), torch.zeros_like(noise)]).cuda()
    frames[2, :, :W] = synth.board_frame(W, H, gridn=10, seed=3).cuda()
    view = frames[:, :, :W]                                              # row stride W + 16
    det = mrgingham_amd.Detector(0)
    one = det.preprocess(view, clahe=True, blur_radius=1)
    det.set_option("preprocess_fused", 0)
    two = det.preprocess(view, clahe=True, blur_radius=1)
    torch.cuda.synchronize()
    assert torch.equal(one, two), (H, W, int((one.int() - two.int()).abs().max()), int((one != two).sum()))
    det.close()


def test_preprocess_strided_input_and_no_ops():
    import torch
    import mrgingham_amd
    from oracle import oracle
    base = _frames("board", 240, 352, 2, seed=5)
    wide = np.zeros((2, 240, 400), dtype=np.uint8)
    wide[:, :, :352] = base
    det = mrgingham_amd.Detector(0)
    t = torch.from_numpy(wide).cuda()[:, :, :352]                        # row stride 400
    got = det.preprocess(t, clahe=True, blur_radius=1).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], oracle.preprocess(base[i], clahe=True, blur_radius=1))
    same = det.preprocess(t, clahe=False, blur_radius=0).cpu().numpy()   # dense copy
    assert np.array_equal(same, base)


def test_preprocessed_frames_give_the_same_board():
    """End to end like the CLI: preprocess on the device, then the level driver."""
    import torch
    import mrgingham_amd
    from mrgingham_amd import synth
    from oracle import oracle
    frame = (synth.board_frame(1280, 960, gridn=10, seed=3).numpy().astype(np.float64) * 0.5 + 40).astype(np.uint8)
    det = mrgingham_amd.Detector(0)
    pre = det.preprocess(torch.from_numpy(frame[None]).cuda(), clahe=True, blur_radius=1)
    pts, lv = det.find_boards(pre, gridn=10)
    want_img = oracle.preprocess(frame, clahe=True, blur_radius=1)
    ref = mrgingham_amd.find_board(want_img, gridn=10)
    assert lv[0] >= 0 and ref is not None
    assert np.array_equal(np.asarray(pts[0]), ref)


def test_host_image_preprocess_and_the_find_board_recipe():
    """mrgingham_amd.preprocess (host image in / out) + find_board = the cv2 recipe of
    find_board.docstring:8-10, and equals what the command-line tool does per image."""
    import mrgingham_amd
    from mrgingham_amd import synth
    from oracle import oracle
    frame = (synth.board_frame(800, 600, gridn=10, seed=9).numpy().astype(np.float64) * 0.55 + 35).astype(np.uint8)
    wide = np.zeros((600, 832), np.uint8)
    wide[:, :800] = frame
    for clahe, blur in [(True, 1), (False, 2), (True, 0)]:
        want = oracle.preprocess(frame, clahe=clahe, blur_radius=blur)
        assert np.array_equal(mrgingham_amd.preprocess(frame, clahe=clahe, blur_radius=blur), want)
        assert np.array_equal(mrgingham_amd.preprocess(wide[:, :800], clahe=clahe, blur_radius=blur), want)   # strided rows
    pre = mrgingham_amd.preprocess(frame)
    board = mrgingham_amd.find_board(pre, gridn=10)
    assert board is not None and board.shape == (100, 2)
    with pytest.raises(RuntimeError, match="8-bit"):
        mrgingham_amd.preprocess(frame.astype(np.uint16))
