"""GPU parity tests proper: the HIP path, called through the C-ABI, against
 * the known-answer vectors of the REAL upstream ChESS.c (tests/golden/chess_kat.npz),
 * the CPU oracle on the same seeded inputs (bit-exact: everything on this path
   is integer / index work; the refined doubles are compared for equality and
   the 1e-4 px tolerance of the north star is asserted as the outer bound),
 * the committed golden corner lists,
 * size-independent properties at BASELINE's full frame size.
Nothing here reads /root/reference.
"""
import os

import numpy as np
import pytest
import torch

import mrgingham_amd
from mrgingham_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu

FILL = -32768
TOL_PX = 1e-4   # north_star: sub-pixel refined coordinates within 1e-4 px


@pytest.fixture(scope="module")
def det():
    d = mrgingham_amd.Detector(0)
    yield d
    d.close()


def _cuda(frames_np):
    return torch.from_numpy(np.ascontiguousarray(frames_np)).cuda()


# ----------------------------------------------------------------------------- ChESS

def test_chess_kats_through_reference_symbol(golden_dir):
    """mrgingham_ChESS_response_5 (host pointers) vs vectors made by the upstream ChESS.c."""
    z = np.load(os.path.join(golden_dir, "chess_kat.npz"))
    names = [k[3:] for k in z.files if k.startswith("in_") and not k.endswith("_buf")]
    L = mrgingham_amd._lib.lib()
    for n in names:
        img = np.ascontiguousarray(z["in_" + n])
        h, w = img.shape
        out = np.full((h, w), FILL, np.int16)
        L.mrgingham_ChESS_response_5(out.ctypes.data, img.ctypes.data, w, h, w)
        assert np.array_equal(out, z["out_" + n]), n   # interior bit-exact AND frame left untouched
    buf = np.ascontiguousarray(z["in_strided77_64x48_buf"])
    out = np.full((48, 64), FILL, np.int16)
    L.mrgingham_ChESS_response_5(out.ctypes.data, buf.ctypes.data, 64, 48, 77)
    assert np.array_equal(out, z["out_strided77_64x48"])


def test_chess_python_mirror_broadcasts():
    rng = np.random.RandomState(3)
    imgs = rng.randint(0, 256, size=(2, 3, 40, 52)).astype(np.uint8)
    r = mrgingham_amd.ChESS_response_5(imgs)
    assert r.shape == imgs.shape and r.dtype == np.int16
    for i in range(2):
        for j in range(3):
            assert np.array_equal(r[i, j], oracle.chess_response_5(imgs[i, j], fill=0))
    with pytest.raises(RuntimeError):
        mrgingham_amd.ChESS_response_5(imgs.astype(np.float32))
    with pytest.raises(RuntimeError):
        mrgingham_amd.ChESS_response_5(np.zeros(5, np.uint8))


@pytest.mark.parametrize("shape", [(15, 15), (16, 80), (97, 130), (240, 333), (480, 640), (1080, 1920)])
def test_chess_batch_raw_and_clamped(det, shape):
    h, w = shape
    rng = np.random.RandomState(h * 7 + w)
    frames = rng.randint(0, 256, size=(3, h, w)).astype(np.uint8)
    frames[1] = synth.board_frame(w, h, 10, 1).numpy() if h >= 240 else frames[1]
    d = _cuda(frames)
    raw = det.chess_response(d, 0, clamp=False).cpu().numpy()
    cl = det.chess_response(d, 0, clamp=True).cpu().numpy()
    for f in range(3):
        ref = oracle.chess_response_5(frames[f], fill=0)
        assert np.array_equal(raw[f], ref), (shape, f)
        assert np.array_equal(cl[f], np.maximum(ref, 0)), (shape, f)


@pytest.mark.parametrize("variant", [1, 16])
@pytest.mark.parametrize("shape", [(16, 80), (40, 16), (48, 64), (300, 272), (480, 640), (777, 1008), (1080, 1920)])
def test_chess_both_response_kernels(det, shape, variant):
    """The response without a hot list has two kernels: chess_v1_kernel (8 pixels per lane, chess.hip) and, for widths
    that are multiples of 16, chess_v16_kernel (16 pixels per lane, chess16.hip: the default where the batch is large
    enough).  Each forced in turn (option chess_variant 1 / 16), raw and clamped, against the oracle; segment heights
    from one iteration to more rows than the frame has."""
    h, w = shape
    rng = np.random.RandomState(h * 11 + w + variant)
    frames = rng.randint(0, 256, size=(3, h, w)).astype(np.uint8)
    if h >= 240:
        frames[1] = synth.board_frame(w, h, 10, 1).numpy()
    d = _cuda(frames)
    ref = [oracle.chess_response_5(frames[f], fill=0) for f in range(3)]
    det.set_option("chess_variant", variant)
    try:
        for seg in ((0,) if variant == 1 else (0, 16, 64, 2048)):
            if variant == 16:
                det.set_option("chess16_seg", seg)
            raw = det.chess_response(d, 0, clamp=False).cpu().numpy()
            cl = det.chess_response(d, 0, clamp=True).cpu().numpy()
            for f in range(3):
                assert np.array_equal(raw[f], ref[f]), (shape, variant, seg, f)
                assert np.array_equal(cl[f], np.maximum(ref[f], 0)), (shape, variant, seg, f)
    finally:
        det.set_option("chess_variant", 0)
        det.set_option("chess16_seg", 0)


def test_chess_v16_is_the_default_on_a_large_batch_and_matches_v1(det):
    """64 frames of 1920x1080 take chess_v16_kernel by default; its output equals chess_v1_kernel's on every pixel (and
    the oracle's on a sample of frames); strided frames (rows and frames not dense) go through it as well."""
    frames = synth.board_batch(4, 1920, 1080, 10, 0, device="cuda").repeat(16, 1, 1).contiguous()
    frames[5] = synth.noise_frame(1920, 1080, seed=2, device="cuda")
    a = det.chess_response(frames, 0)
    det.set_option("chess_variant", 1)
    try:
        b = det.chess_response(frames, 0)
    finally:
        det.set_option("chess_variant", 0)
    assert torch.equal(a, b)
    for f in (0, 5, 63):
        assert np.array_equal(a[f].cpu().numpy(), oracle.chess_response_5(frames[f].cpu().numpy(), fill=0))
    big = torch.from_numpy(np.random.RandomState(9).randint(0, 256, size=(40, 300, 400)).astype(np.uint8)).cuda()
    view = big[:, 20:280, 16:336]                       # 320 wide: rows 400 apart, frames 300 * 400 apart
    det.set_option("chess_variant", 16)
    try:
        r = det.chess_response(view, 0).cpu().numpy()
    finally:
        det.set_option("chess_variant", 0)
    for f in (0, 17, 39):
        assert np.array_equal(r[f], oracle.chess_response_5(np.ascontiguousarray(big[f, 20:280, 16:336].cpu().numpy()), fill=0))
    # a view whose first pixel is not 4-byte aligned (its 16-byte staging loads would not be dword loads): the library
    # takes chess_v1 for it whatever the option says, and the answer is the same
    odd = big[:, 20:280, 3:323]
    det.set_option("chess_variant", 16)
    try:
        r = det.chess_response(odd, 0).cpu().numpy()
    finally:
        det.set_option("chess_variant", 0)
    for f in (1, 38):
        assert np.array_equal(r[f], oracle.chess_response_5(np.ascontiguousarray(big[f, 20:280, 3:323].cpu().numpy()), fill=0))


def test_chess_v16_fused_variants_give_the_same_chain(det):
    """The sixteen-pixels-per-lane kernels WITH the hot list, the level images and several levels per launch (option
    chess_variant_hot 16 | 32; experiment builds only: in the chain they are not faster than chess_v1's):
    the chain's outputs are the shipped kernels'."""
    try:
        det.set_option("chess_variant_hot", 0)
    except ValueError:
        pytest.skip("the fused chess_v16 kernels exist in experiment builds only")
    frames = synth.board_batch(3, 1024, 768, 10, 5, device="cuda")
    frames[1] = synth.cluttered_board_frame(1024, 768, 10, seed=2, device="cuda")
    det.set_option("sparse_refine", 0)
    try:
        want = det.chain(frames, 3, 1024)
        for v in (16, 32, 48):
            det.set_option("chess_variant_hot", v)
            got = det.chain(frames, 3, 1024)
            assert torch.equal(want[2], got[2]), v
            for f in range(3):
                n = int(want[2][f])
                assert torch.equal(want[0][f, :n], got[0][f, :n]) and torch.equal(want[1][f, :n], got[1][f, :n]), (v, f)
    finally:
        det.set_option("chess_variant_hot", 0)
        det.set_option("sparse_refine", 1)


def test_chess_v16_paired_half_strip_matches_the_oracle(det):
    """Widths with w % 256 = 128: the last strip's workgroups taking two row segments at once (chess_v16_pair_kernel,
    option chess16_pair; experiment builds only -- +1.1 % at 64 x 1920x1080).  Raw and clamped against the
    oracle on noise, for even and odd segment counts and a ragged last granule."""
    try:
        det.set_option("chess16_pair", 1)
    except ValueError:
        pytest.skip("the paired half-strip kernel exists in experiment builds only")
    det.set_option("chess_variant", 16)
    try:
        for (h, w) in ((200, 384), (137, 640), (480, 1152)):
            rng = np.random.RandomState(h + w)
            frames = rng.randint(0, 256, size=(3, h, w)).astype(np.uint8)
            d = _cuda(frames)
            ref = [oracle.chess_response_5(frames[f], fill=0) for f in range(3)]
            for seg in (0, 16, 48, 64, 4096):
                det.set_option("chess16_seg", seg)
                raw = det.chess_response(d, 0, clamp=False).cpu().numpy()
                cl = det.chess_response(d, 0, clamp=True).cpu().numpy()
                for f in range(3):
                    assert np.array_equal(raw[f], ref[f]), (h, w, seg, f)
                    assert np.array_equal(cl[f], np.maximum(ref[f], 0)), (h, w, seg, f)
    finally:
        det.set_option("chess16_pair", 0)
        det.set_option("chess16_seg", 0)
        det.set_option("chess_variant", 0)


def test_chess_strided_device_frames(det):
    rng = np.random.RandomState(5)
    big = rng.randint(0, 256, size=(2, 100, 200)).astype(np.uint8)
    d = _cuda(big)[:, 10:90, 16:150]          # non-dense rows and frames
    r = det.chess_response(d, 0).cpu().numpy()
    for f in range(2):
        assert np.array_equal(r[f], oracle.chess_response_5(np.ascontiguousarray(big[f, 10:90, 16:150]), fill=0))


def test_chess_v0_cross_check(det):
    """The plain reference-shaped kernel and the tuned kernel agree on the device (experiment builds only:
    make -C mrgingham_amd/csrc EXPERIMENT=1, MRGINGHAM_AMD_LIB; the shipped library does not carry the second kernel)."""
    frames = _cuda(np.stack([synth.board_frame(640, 480, 10, s).numpy() for s in range(2)] +
                            [synth.noise_frame(640, 480, 3).numpy()]))
    a = det.chess_response(frames, 0).cpu()
    try:
        det.set_option("chess_v0", 1)
    except ValueError:
        pytest.skip("the reference-shaped kernel exists in experiment builds only")
    try:
        b = det.chess_response(frames, 0).cpu()
    finally:
        det.set_option("chess_v0", 0)
    assert torch.equal(a, b)


# ----------------------------------------------------------------------------- decimate / blur

@pytest.mark.parametrize("shape", [(48, 64), (480, 640), (37, 53), (41, 66), (50, 51), (1001, 999)])
def test_decimate_levels(det, shape):
    h, w = shape
    rng = np.random.RandomState(h + w)
    frames = rng.randint(0, 256, size=(2, h, w)).astype(np.uint8)
    d = _cuda(frames)
    for level in (0, 1, 2, 3):
        got = det.decimate(d, level).cpu().numpy()
        for f in range(2):
            assert np.array_equal(got[f], oracle.decimate(frames[f], level)), (shape, level)


def test_box_blur(det):
    rng = np.random.RandomState(9)
    frames = rng.randint(0, 256, size=(2, 61, 83)).astype(np.uint8)
    for r in (1, 2):
        got = det.box_blur(_cuda(frames), r).cpu().numpy()
        for f in range(2):
            assert np.array_equal(got[f], oracle.box_blur(frames[f], r))


# ----------------------------------------------------------------------------- detect

def _frames_small():
    return {
        "board10_640x480_s0": synth.board_frame(640, 480, 10, 0).numpy(),
        "board10_640x480_s5": synth.board_frame(640, 480, 10, 5).numpy(),
        "board14_800x600_s1": synth.board_frame(800, 600, 14, 1).numpy(),
        "board10_clean_648x486": synth.board_frame(648, 486, 10, 2, noise=False).numpy(),
        "noise_320x240_s0": synth.noise_frame(320, 240, 0).numpy(),
        "noise_320x240_s1_smooth1": synth.noise_frame(320, 240, 1, smooth=1).numpy(),
        "noise_333x251_s2_smooth2": synth.noise_frame(333, 251, 2, smooth=2).numpy(),
    }


def _as_int(pts):
    return np.round(pts * 1000).astype(np.int64)


def test_find_points_matches_oracle_and_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "corners_golden.npz"))
    for name, img in _frames_small().items():
        for level in range(4):
            got = mrgingham_amd.find_points(img, image_pyramid_level=level)
            want = oracle.find_corners(img, level)
            assert got.dtype == np.float64 and got.shape == (len(want), 2), (name, level)
            assert np.array_equal(_as_int(got), want.astype(np.int64)), (name, level)    # values AND order
            assert np.array_equal(want, z[f"detect_L{level}_{name}"]), (name, level)
            # the callback scale is exactly 1/1000. (bridge.cc:66-69)
            assert np.array_equal(got, want.astype(np.float64) * (1. / 1000.))


def test_find_points_error_and_empty_paths():
    img = synth.board_frame(160, 120, 10, 0).numpy()
    assert mrgingham_amd.find_points(img, image_pyramid_level=-1).shape == (0, 2)   # :433-441 -> no points
    assert mrgingham_amd.find_points(img, image_pyramid_level=11).shape == (0, 2)
    wide = np.zeros((120, 200), np.uint8)
    wide[:, :160] = img
    assert mrgingham_amd.find_points(wide[:, :160], 0).shape == (0, 2)              # :461-466 non-continuous
    a = mrgingham_amd.find_points(wide[:, :160], 1)
    assert np.array_equal(_as_int(a), oracle.find_corners(wide[:, :160], 1))
    assert mrgingham_amd.find_points(np.zeros((10, 10), np.uint8)).shape == (0, 2)  # no interior at all
    assert mrgingham_amd.find_points(np.zeros((64, 64), np.uint8)).shape == (0, 2)  # nothing found
    assert np.array_equal(np.round(mrgingham_amd.find_points(img, blobs=True) * 1000).astype(np.int64),
                          oracle.find_blobs(img).astype(np.int64))                   # the blob path (find_blobs.cc)
    with pytest.raises(RuntimeError):
        mrgingham_amd.find_points(img, image_pyramid_level=1, blobs=True)           # mrgingham_pywrap.c:153-157
    with pytest.raises(RuntimeError):
        mrgingham_amd.find_points(np.zeros((4, 64, 64), np.uint8))
    with pytest.raises(RuntimeError):
        mrgingham_amd.find_points(np.zeros((64, 64), np.uint16))


def test_detect_adversarial_textures():
    """Dense corner textures overflow the default component tables; the
    reference-symbol wrapper retries with full tables and must still be exact."""
    yy, xx = np.arange(240).reshape(-1, 1), np.arange(320).reshape(1, -1)
    for cell in (3, 7):
        img = (((yy // cell + xx // cell) & 1) * 255).astype(np.uint8)
        got = mrgingham_amd.find_points(img, 0)
        want = oracle.find_corners(img, 0)
        assert len(want) > 1000
        assert np.array_equal(_as_int(got), want.astype(np.int64)), cell
    # ramps / constant images: nothing
    assert mrgingham_amd.find_points(np.tile(np.arange(256, dtype=np.uint8), (200, 1))).shape == (0, 2)


def test_detect_batch_mixed_content(det):
    frames = np.stack([synth.board_frame(640, 480, 10, 0).numpy(), synth.noise_frame(640, 480, 1).numpy(),
                       synth.board_frame(640, 480, 14, 2).numpy(), np.zeros((480, 640), np.uint8),
                       synth.noise_frame(640, 480, 4, smooth=1).numpy(), synth.board_frame(640, 480, 10, 9).numpy(),
                       synth.noise_frame(640, 480, 6, smooth=2).numpy()])
    d = _cuda(frames)
    for level in (0, 1, 3):
        xy, counts = det.detect(d, level, capacity=8192)
        xy, counts = xy.cpu().numpy(), counts.cpu().numpy()
        for f in range(len(frames)):
            want = oracle.find_corners(frames[f], level)
            assert counts[f] == len(want), (level, f)
            assert np.array_equal(xy[f, :counts[f]], want), (level, f)
    # capacity smaller than the count: count is still reported, the first `capacity` stored in order
    xy, counts = det.detect(d, 0, capacity=16)
    want = oracle.find_corners(frames[1], 0)
    assert counts[1].item() == len(want) and np.array_equal(xy[1].cpu().numpy(), want[:16])


# ----------------------------------------------------------------------------- refine / chain

def test_refine_symbol_matches_oracle():
    for name, img in _frames_small().items():
        for start in (3, 2, 1):
            cand = oracle.find_corners(img, start)
            pts = cand.astype(np.float64) / 1000.0
            lv = np.full(len(pts), start, np.int8)
            for level in range(start - 1, -1, -1):
                wp, wl, wn = oracle.refine_corners(pts, lv, img, level)
                gp, gl, gn = mrgingham_amd.refine_points(pts, lv, img, level)
                assert gn == wn, (name, start, level)
                assert np.array_equal(gl, wl), (name, start, level)
                assert np.abs(gp - wp).max(initial=0.) <= TOL_PX
                assert np.array_equal(gp, wp), (name, start, level)       # in fact bit-identical doubles
                pts, lv = wp, wl


def test_refine_shared_blobs_and_odd_points():
    """Points that seed from the same blob must be processed in index order on the
    mutating response (find_chessboard_corners.cc:358-396); junk points are ignored."""
    img = synth.board_frame(640, 480, 10, 0).numpy()
    cand = oracle.find_corners(img, 1).astype(np.float64) / 1000.0
    pts = np.concatenate([cand[:20], cand[:20] + 0.4, cand[5:10] - 0.6, [[-50., -50.], [1e6, 1e6], [3., 3.]], cand[20:40]])
    lv = np.full(len(pts), 1, np.int8)
    lv[7] = 2
    lv[33] = 0
    wp, wl, wn = oracle.refine_corners(pts, lv, img, 0)
    gp, gl, gn = mrgingham_amd.refine_points(pts, lv, img, 0)
    assert gn == wn and np.array_equal(gl, wl) and np.array_equal(gp, wp)
    assert 20 <= wn < len(pts)


def test_chain_batch_matches_oracle(det, golden_dir):
    z = np.load(os.path.join(golden_dir, "corners_golden.npz"))
    frames = np.stack([synth.board_frame(640, 480, 10, s).numpy() for s in (0, 5, 7)] +
                      [synth.noise_frame(640, 480, 1, smooth=1).numpy(), synth.board_frame(640, 480, 14, 3).numpy()])
    d = _cuda(frames)
    for start in (3, 2, 0):
        pts, lv, npts = det.chain(d, start_level=start, max_points=2048)
        pts, lv, npts = pts.cpu().numpy(), lv.cpu().numpy(), npts.cpu().numpy()
        for f in range(len(frames)):
            wp, wl = oracle.chain(frames[f], start)
            n = npts[f]
            assert n == len(wp), (start, f)
            assert np.array_equal(lv[f, :n], wl), (start, f)
            assert np.abs(pts[f, :n] - wp).max(initial=0.) <= TOL_PX
            assert np.array_equal(pts[f, :n], wp), (start, f)
    assert np.array_equal(oracle.chain(frames[0], 3)[0], z["chain3_pts_board10_640x480_s0"])


def test_global_memory_component_kernels_alone_give_the_same_chain():
    """option cc_lds = 0: every frame through the global-memory kernels (the fallback of the LDS path)."""
    d2 = mrgingham_amd.Detector(0)
    try:
        d2.set_option("cc_lds", 0)
        frames = np.stack([synth.board_frame(1280, 960, 10, s).numpy() for s in (0, 3)] +
                          [synth.noise_frame(1280, 960, 2, smooth=1).numpy()])
        d = _cuda(frames)
        for start in (3, 1):
            pts, lv, npts = d2.chain(d, start_level=start, max_points=8192)
            for f in range(len(frames)):
                wp, wl = oracle.chain(frames[f], start)
                n = int(npts[f])
                assert n == len(wp), (start, f)
                assert np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl)
    finally:
        d2.close()


def test_back_to_back_batches_without_sync(det):
    """Queued calls reuse the level scratch: the pixel stream of call N+1 must wait for the
    component stream of call N (results equal the synchronous ones)."""
    fa = _cuda(np.stack([synth.board_frame(640, 480, 10, s).numpy() for s in range(6)]))
    fb = _cuda(np.stack([synth.noise_frame(640, 480, s, smooth=1).numpy() for s in range(6)]))
    ref_a = [t.clone() for t in det.chain(fa, 3, 512)]
    ref_b = [t.clone() for t in det.chain(fb, 3, 512)]
    out_a = det.chain(fa, 3, 512, sync=False)
    out_b = det.chain(fb, 3, 512, sync=False)
    xy, counts = det.detect(fa, 0, capacity=512, sync=False)
    det.sync()
    for got, ref in ((out_a, ref_a), (out_b, ref_b)):
        assert torch.equal(got[2], ref[2])
        for f in range(6):
            k = int(ref[2][f])
            assert torch.equal(got[0][f, :k], ref[0][f, :k]) and torch.equal(got[1][f, :k], ref[1][f, :k])
    for f in range(6):
        want = oracle.find_corners(fa[f].cpu().numpy(), 0)
        assert int(counts[f]) == len(want) and np.array_equal(xy[f, :len(want)].cpu().numpy(), want)


# ----------------------------------------------------------------------------- full size

def test_full_size_frames_bit_exact_and_properties(det):
    """BASELINE size 4096x3072: two frames bit-exact against the oracle end to end,
    plus properties that need no oracle."""
    W, H = 4096, 3072
    frames = synth.board_batch(2, W, H, gridn=10, seed0=11, device="cuda")
    host = frames.cpu().numpy()
    r = det.chess_response(frames, 0, clamp=False)
    assert np.array_equal(r[0].cpu().numpy(), oracle.chess_response_5(host[0], fill=0))
    # frame is zero, interior range bound (ChESS.c:88-104)
    assert int(r[:, :7].abs().max()) == 0 and int(r[:, :, -7:].abs().max()) == 0
    assert int(r.min()) >= -6120 and int(r.max()) <= 2040
    # translation covariance: shifting the image shifts the response
    shifted = torch.roll(frames, shifts=(5, 9), dims=(1, 2))
    rs = det.chess_response(shifted, 0, clamp=False)
    assert torch.equal(rs[:, 20:-20, 20:-20], torch.roll(r, shifts=(5, 9), dims=(1, 2))[:, 20:-20, 20:-20])
    for level in (0, 3):
        xy, counts = det.detect(frames, level, capacity=4096)
        for f in range(2):
            want = oracle.find_corners(host[f], level)
            assert counts[f].item() == len(want) and np.array_equal(xy[f, :len(want)].cpu().numpy(), want)
    pts, lv, npts = det.chain(frames, 3, 1024)
    again = det.chain(frames, 3, 1024)
    for f in range(2):
        wp, wl = oracle.chain(host[f], 3)
        n = int(npts[f])
        assert n == len(wp) and np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl)
        assert torch.equal(again[0][f, :n], pts[f, :n])                  # deterministic re-run
        assert int((lv[f, :n] == 0).sum()) >= 100                        # the 10x10 grid reaches level 0


def test_dependent_calls_without_sync_are_ordered(det):
    """Consecutive calls finish on different component streams; a call that reads or overwrites a
    device buffer the previous call wrote must still run after it (no host sync in between)."""
    imgs = [synth.board_frame(1280, 960, 10, s).numpy() for s in range(4)]
    frames = _cuda(np.stack(imgs))
    # reference: the library's own chain (detect at level 2, refine 1, 0)
    rp, rl, rn = [t.clone() for t in det.chain(frames, 2, 256)]
    # the same thing as three dependent calls on shared buffers, queued back to back
    xy, counts = det.detect(frames, 2, capacity=256, sync=True)
    pts = (xy.to(torch.float64) / 1000.0).contiguous()
    lv = torch.full((4, 256), 2, dtype=torch.int8, device="cuda")
    npts = counts.clone()
    for rep in range(3):                                    # repeated: a race would not show every time
        p, l = pts.clone(), lv.clone()
        torch.cuda.synchronize()
        det.refine(frames, 1, p, l, npts, sync=False)
        det.refine(frames, 0, p, l, npts, sync=False)       # reads what the call before wrote
        out2 = det.chain(frames, 2, 256, out=(p.clone(), l.clone(), npts.clone()), sync=False)  # unrelated buffers
        det.sync()
        for f in range(4):
            k = int(rn[f])
            assert int(npts[f]) == k
            assert torch.equal(p[f, :k], rp[f, :k]) and torch.equal(l[f, :k], rl[f, :k]), (rep, f)
            assert torch.equal(out2[0][f, :k], rp[f, :k])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_multi_level_launch_option_gives_the_same_chain(mode):
    """set_option("multi_level_launch", ..): 0 = one ChESS launch per level; 1 (the default) = levels 3, 2, 1
    of a chain share one grid; 2 = level 0 as well."""
    d2 = mrgingham_amd.Detector(0)
    try:
        d2.set_option("multi_level_launch", mode)
        frames = np.stack([synth.board_frame(1280, 960, 10, s).numpy() for s in (0, 3)] +
                          [synth.noise_frame(1280, 960, 2, smooth=1).numpy()])
        d = _cuda(frames)
        for start in (3, 2):
            pts, lv, npts = d2.chain(d, start_level=start, max_points=4096)
            for f in range(len(frames)):
                wp, wl = oracle.chain(frames[f], start)
                n = int(npts[f])
                assert n == len(wp), (start, f)
                assert np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl)
        # a width that is not a multiple of 16 falls back to one launch per level
        odd = np.stack([synth.board_frame(1000, 760, 10, 1).numpy()])
        pts, lv, npts = d2.chain(_cuda(odd), start_level=3, max_points=2048)
        wp, wl = oracle.chain(odd[0], 3)
        assert int(npts[0]) == len(wp) and np.array_equal(pts[0, :len(wp)].cpu().numpy(), wp)
    finally:
        d2.close()


@pytest.mark.parametrize("fuse,multi", [(1, 1), (1, 0), (0, 1)])
def test_fused_pyramid_option_gives_the_same_chain(fuse, multi):
    """set_option("fuse_pyramid", 1) (the default): chain calls on frames of whole 16 x 8 blocks take the level
    images 1..3 out of the level-0 response kernel; 0 = the separate pyramid kernel.  Same corners either way
    (and the same as the oracle's), for every start level, strips that end inside a 256-pixel strip included."""
    d2 = mrgingham_amd.Detector(0)
    try:
        d2.set_option("fuse_pyramid", fuse)
        d2.set_option("multi_level_launch", multi)
        for (w, h) in ((1280, 960), (1328, 984), (272, 264)):
            frames = np.stack([synth.board_frame(w, h, 10, s).numpy() for s in (0, 3)] +
                              [synth.noise_frame(w, h, 2, smooth=1).numpy()])
            d = _cuda(frames)
            for start in (3, 2, 1, 4):
                if min(w, h) >> start < 32:
                    continue
                pts, lv, npts = d2.chain(d, start_level=start, max_points=4096)
                for f in range(len(frames)):
                    wp, wl = oracle.chain(frames[f], start)
                    n = int(npts[f])
                    assert n == len(wp), (w, h, start, f)
                    assert np.array_equal(pts[f, :n].cpu().numpy(), wp), (w, h, start, f)
                    assert np.array_equal(lv[f, :n].cpu().numpy(), wl), (w, h, start, f)
    finally:
        d2.close()


def test_fused_pyramid_random_shapes(det):
    """Frames of whole 16 x 8 blocks take the fused level-0 + pyramid kernel in chain(): random sizes (strips that
    end anywhere inside a 256-pixel strip, segments that end anywhere), random content, every start level that
    leaves a usable image; corners and levels equal the oracle's."""
    rng = np.random.default_rng(20260929)
    for case in range(14):
        w = 16 * int(rng.integers(4, 48))
        h = 8 * int(rng.integers(6, 70))
        kind = case % 3
        if kind == 0:
            frames = np.stack([synth.board_frame(w, h, 6 + case % 5, s).numpy() for s in (case, case + 1)])
        elif kind == 1:
            frames = np.stack([synth.noise_frame(w, h, s, smooth=case % 2).numpy() for s in (case, case + 1)])
        else:                                   # coarse random blocks: many strong corners at every level
            cell = int(rng.integers(5, 14))
            blocks = rng.integers(0, 2, size=(2, h // cell + 1, w // cell + 1), dtype=np.uint8) * 200 + 20
            frames = np.repeat(np.repeat(blocks, cell, axis=1), cell, axis=2)[:, :h, :w].copy()
            frames += rng.integers(0, 16, size=frames.shape, dtype=np.uint8)
        d = _cuda(frames)
        for start in (1, 2, 3):
            if min(w, h) >> start < 24:
                continue
            pts, lv, npts = det.chain(d, start_level=start, max_points=8192)
            assert det.chain_info()[0], (w, h)
            for f in range(2):
                wp, wl = oracle.chain(frames[f], start)
                n = int(npts[f])
                assert n == len(wp), (case, w, h, start, f, n, len(wp))
                assert np.array_equal(pts[f, :n].cpu().numpy(), wp), (case, w, h, start, f)
                assert np.array_equal(lv[f, :n].cpu().numpy(), wl), (case, w, h, start, f)


def test_fused_pyramid_on_strided_frames(det):
    """chain() on frames that are a window of a larger device buffer (row stride and frame pitch larger than
    the frame, 16-byte aligned): still the fused level-0 + pyramid kernel, same corners as the dense copy."""
    w, h = 1328, 984
    frames = np.stack([synth.board_frame(w, h, 10, s).numpy() for s in (0, 5, 6)])
    buf = torch.full((3, h + 8, w + 32), 77, dtype=torch.uint8, device="cuda")
    view = buf[:, 4:4 + h, 16:16 + w]
    view.copy_(torch.from_numpy(frames).cuda())
    assert not view.is_contiguous()
    pts, lv, npts = det.chain(view, start_level=3, max_points=2048)
    assert det.chain_info()[0]
    for f in range(3):
        wp, wl = oracle.chain(frames[f], 3)
        n = int(npts[f])
        assert n == len(wp) and np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl)
    # a window that starts at an odd byte is not 16-byte aligned: the separate pyramid kernel takes it
    view2 = buf[:, 4:4 + h, 3:3 + w]
    view2.copy_(torch.from_numpy(frames).cuda())
    pts, lv, npts = det.chain(view2, start_level=3, max_points=2048)
    assert not det.chain_info()[0]
    for f in range(3):
        wp, wl = oracle.chain(frames[f], 3)
        n = int(npts[f])
        assert n == len(wp) and np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl)


def test_three_scratch_sets_pipeline():
    """set_option("scratch_sets", 3): three calls' component searches in flight.  Calls queued back to back without
    a host sync -- distinct outputs, then dependent calls on shared buffers -- give the same results as one at a time."""
    d3 = mrgingham_amd.Detector(0)
    try:
        d3.set_option("scratch_sets", 3)
        imgs = [np.stack([synth.board_frame(1280, 960, 10, 7 * k + s).numpy() for s in range(3)]) for k in range(4)]
        devs = [_cuda(x) for x in imgs]
        outs = [d3.chain(devs[k % 4], start_level=3, max_points=1024, sync=False) for k in range(9)]
        d3.sync()
        for k, (pts, lv, npts) in enumerate(outs):
            for f in range(3):
                wp, wl = oracle.chain(imgs[k % 4][f], 3)
                n = int(npts[f])
                assert n == len(wp), (k, f)
                assert np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl), (k, f)
        # dependent calls on shared buffers (see test_dependent_calls_without_sync_are_ordered)
        frames = devs[0]
        rp, rl, rn = [t.clone() for t in d3.chain(frames, 2, 256)]
        xy, counts = d3.detect(frames, 2, capacity=256, sync=True)
        pts0 = (xy.to(torch.float64) / 1000.0).contiguous()
        for rep in range(3):
            p, l = pts0.clone(), torch.full((3, 256), 2, dtype=torch.int8, device="cuda")
            torch.cuda.synchronize()
            d3.refine(frames, 1, p, l, counts, sync=False)
            d3.chain(devs[1], 3, 256, sync=False)                 # an unrelated call in between
            d3.refine(frames, 0, p, l, counts, sync=False)        # reads what the call two before wrote
            d3.sync()
            for f in range(3):
                k = int(rn[f])
                assert torch.equal(p[f, :k], rp[f, :k]) and torch.equal(l[f, :k], rl[f, :k]), (rep, f)
        with pytest.raises(ValueError):
            d3.set_option("scratch_sets", 4)
    finally:
        d3.close()


@pytest.mark.parametrize("seg", [32, 64, 128, 256])
def test_segment_height_does_not_change_results(seg):
    """The segment height of the ChESS kernels (a cost model picks it per batch shape; option "chess_seg" fixes it)
    only changes how the frame is cut into workgroups: responses, hot lists and so the chain are the same."""
    d2 = mrgingham_amd.Detector(0)
    try:
        d2.set_option("chess_seg", seg)
        for (w, h) in ((1328, 984), (1000, 760)):
            frames = np.stack([synth.board_frame(w, h, 10, s).numpy() for s in (1, 4)])
            d = _cuda(frames)
            r = d2.chess_response(d, 0, clamp=True).cpu().numpy()
            for f in range(2):
                assert np.array_equal(r[f], oracle.clamped_response(frames[f], 0)[0]), (seg, w, h, f)
            pts, lv, npts = d2.chain(d, start_level=3, max_points=2048)
            for f in range(2):
                wp, wl = oracle.chain(frames[f], 3)
                n = int(npts[f])
                assert n == len(wp) and np.array_equal(pts[f, :n].cpu().numpy(), wp) and np.array_equal(lv[f, :n].cpu().numpy(), wl)
    finally:
        d2.set_option("chess_seg", 0)
        d2.close()


@pytest.mark.parametrize("sets", [2, 3])
def test_pipelined_chain_soak(sets):
    """400 chain calls queued back to back without a host sync, on three alternating batches and four rotating output
    buffers; every step's corner lists are compared ON THE DEVICE (torch's stream, ordered behind the step with
    stream_wait) with what the same batch gives one call at a time.  A race between the pixel stream, the component
    streams and the scratch-set rotation would show as a mismatching step (tools/soak.py is the long form)."""
    B, P, W, H = 16, 512, 1280, 960
    det = mrgingham_amd.Detector(0)
    try:
        det.set_option("scratch_sets", sets)
        batches = [synth.board_batch(8, W, H, 10, 8 * k, device="cuda").repeat(B // 8, 1, 1).contiguous() for k in range(3)]
        refs = []
        for k, fr in enumerate(batches):
            p, l, n = det.chain(fr, 3, P)
            refs.append((p.clone(), l.clone(), n.clone()))
            host = fr[0].cpu().numpy()
            wp, wl = oracle.chain(host, 3)
            assert int(n[0]) == len(wp) and np.array_equal(p[0, :len(wp)].cpu().numpy(), wp)
        outs = [(torch.empty((B, P, 2), dtype=torch.float64, device="cuda"), torch.empty((B, P), dtype=torch.int8, device="cuda"),
                 torch.empty((B,), dtype=torch.int32, device="cuda")) for _ in range(4)]
        bad = torch.zeros(1, dtype=torch.int32, device="cuda")
        idx = torch.arange(P, device="cuda")[None, :]
        for s in range(400):
            k, o = s % 3, outs[s % 4]
            det.after_stream()                       # the comparison of four steps ago has read this buffer
            det.chain(batches[k], 3, P, out=o, sync=False)
            det.stream_wait()
            rp, rl, rn = refs[k]
            live = idx < rn[:, None]
            bad += (o[2] != rn).any().int() + ((o[0] != rp).any(-1) & live).any().int() + ((o[1] != rl) & live).any().int()
        det.sync()
        torch.cuda.synchronize()
        assert int(bad.item()) == 0
    finally:
        det.close()
