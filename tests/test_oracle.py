"""CPU checks of the oracle itself (no GPU).

 * the ChESS restatement against the known-answer vectors produced by the REAL
   upstream ChESS.c (tests/golden/chess_kat.npz), and against oracle/_ref live
   when that build is present;
 * the connected-component / decimation restatement against its own committed
   regression vectors and against hand-derived expectations of the reference's
   rules (find_chessboard_corners.cc citations inline).
"""
import hashlib
import os

import numpy as np
import pytest

from mrgingham_amd import synth
from oracle import oracle

FILL = -32768


def _kat_cases(golden_dir):
    z = np.load(os.path.join(golden_dir, "chess_kat.npz"))
    names = [k[3:] for k in z.files if k.startswith("in_") and not k.endswith("_buf")]
    return z, names


def test_chess_restatement_matches_reference_kats(golden_dir):
    z, names = _kat_cases(golden_dir)
    assert len(names) >= 12
    for n in names:
        img = z["in_" + n]
        got = oracle.chess_response_5(img, fill=FILL)
        assert np.array_equal(got, z["out_" + n]), n
    buf = z["in_strided77_64x48_buf"]
    got = oracle.chess_response_5(buf[:, :64], fill=FILL)
    assert np.array_equal(got, z["out_strided77_64x48"])


def test_chess_kat_border_untouched_and_tiny_images(golden_dir):
    z, _ = _kat_cases(golden_dir)
    out = z["out_rand0_64x48"]
    assert (out[:7] == FILL).all() and (out[-7:] == FILL).all()
    assert (out[:, :7] == FILL).all() and (out[:, -7:] == FILL).all()
    assert (out[7:-7, 7:-7] != FILL).all()
    assert (z["out_none_14x14"] == FILL).all()                 # ChESS.c:62-63: no interior
    assert (z["out_single_15x15"] != FILL).sum() == 1           # exactly one interior pixel


@pytest.mark.skipif(not oracle.have_reference_build(), reason="oracle/_ref not built")
def test_chess_restatement_matches_live_reference_build():
    rng = np.random.RandomState(123)
    for (h, w) in [(15, 15), (31, 47), (97, 130), (240, 320)]:
        img = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        assert np.array_equal(oracle.chess_response_5(img, FILL), oracle.ref_chess_response_5(img, FILL))
    f = synth.board_frame(640, 480, 10, 3).numpy()
    assert np.array_equal(oracle.chess_response_5(f, FILL), oracle.ref_chess_response_5(f, FILL))


def test_chess_value_range():
    # ChESS.c:88-104: sum,diff <= 2040, mean,local_mean <= 4080 -> [-6120, 2040]
    img = (np.random.RandomState(5).randint(0, 2, size=(64, 64)) * 255).astype(np.uint8)
    r = oracle.chess_response_5(img)
    assert r.min() >= -6120 and r.max() <= 2040


def test_level_dims_and_decimate():
    assert oracle.level_dims(4096, 3072, 3) == (512, 384)
    assert oracle.level_dims(1001, 999, 1) == (500, 500)      # cvRound: half to even
    assert oracle.level_dims(1003, 1005, 1) == (502, 502)
    with pytest.raises(ValueError):
        oracle.level_dims(64, 64, 11)
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(48, 64)).astype(np.uint8)
    for L in (1, 2, 3):
        s = 1 << L
        o = s // 2 - 1
        a = img[o::s, o::s].astype(np.int32)[: 48 // s, : 64 // s]
        b = img[o::s, o + 1::s].astype(np.int32)[: 48 // s, : 64 // s]
        c = img[o + 1::s, o::s].astype(np.int32)[: 48 // s, : 64 // s]
        d = img[o + 1::s, o + 1::s].astype(np.int32)[: 48 // s, : 64 // s]
        assert np.array_equal(oracle.decimate(img, L), ((a + b + c + d + 2) >> 2).astype(np.uint8)), L
    assert np.array_equal(oracle.decimate(img, 0), img)
    # strided input honoured (cv::resize reads through the Mat step)
    buf = rng.randint(0, 256, size=(48, 80)).astype(np.uint8)
    assert np.array_equal(oracle.decimate(buf[:, :64], 2), oracle.decimate(np.ascontiguousarray(buf[:, :64]), 2))
    # ragged sizes run and give the rounded dims
    for (h, w) in [(37, 53), (41, 66), (50, 51)]:
        im = rng.randint(0, 256, size=(h, w)).astype(np.uint8)
        for L in (1, 2, 3):
            ow, oh = oracle.level_dims(w, h, L)
            assert oracle.decimate(im, L).shape == (oh, ow)


def test_box_blur():
    rng = np.random.RandomState(1)
    img = rng.randint(0, 256, size=(20, 31)).astype(np.uint8)
    p = np.pad(img.astype(np.int32), 1, mode="reflect")
    s = sum(p[dy:dy + 20, dx:dx + 31] for dy in range(3) for dx in range(3))
    assert np.array_equal(oracle.box_blur(img, 1), ((s + 4) // 9).astype(np.uint8))
    import torch
    assert np.array_equal(synth.box_blur3(torch.from_numpy(img.astype(np.int64))).numpy().astype(np.uint8),
                          oracle.box_blur(img, 1))


def test_synth_frames_are_stable(golden_dir):
    z = np.load(os.path.join(golden_dir, "corners_golden.npz"))
    f = synth.board_frame(640, 480, 10, 0).numpy()
    assert hashlib.sha256(f.tobytes()).digest() == z["sha_board10_640x480_s0"].tobytes()
    n = synth.noise_frame(320, 240, 0).numpy()
    assert hashlib.sha256(n.tobytes()).digest() == z["sha_noise_320x240_s0"].tobytes()


def test_detect_and_chain_regression(golden_dir):
    z = np.load(os.path.join(golden_dir, "corners_golden.npz"))
    frames = {
        "board10_640x480_s0": synth.board_frame(640, 480, 10, 0).numpy(),
        "board14_800x600_s1": synth.board_frame(800, 600, 14, 1).numpy(),
        "noise_320x240_s1_smooth1": synth.noise_frame(320, 240, 1, smooth=1).numpy(),
        "noise_333x251_s2_smooth2": synth.noise_frame(333, 251, 2, smooth=2).numpy(),
    }
    for name, img in frames.items():
        for L in range(4):
            assert np.array_equal(oracle.find_corners(img, L), z[f"detect_L{L}_{name}"]), (name, L)
        pts, lv = oracle.chain(img, 3)
        assert np.array_equal(pts, z[f"chain3_pts_{name}"]) and np.array_equal(lv, z[f"chain3_lv_{name}"])
    assert len(z["detect_L0_board10_640x480_s0"]) == 100
    assert len(z["detect_L1_board14_800x600_s1"]) == 196


def test_detect_error_paths():
    img = synth.board_frame(160, 120, 10, 0).numpy()
    assert oracle.find_corners(img, -1) is None            # find_chessboard_corners.cc:433-441
    assert oracle.find_corners(img, 11) is None
    wide = np.zeros((120, 200), np.uint8)
    wide[:, :160] = img
    assert oracle.find_corners(wide[:, :160], 0) is None   # :461-466 non-continuous at level 0
    assert oracle.find_corners(wide[:, :160], 1) is not None  # resize output is continuous
    assert len(oracle.find_corners(np.zeros((10, 10), np.uint8), 0)) == 0


# ---- hand-built responses exercising the fill rules (tests/cc_cases.py; the GPU suite runs the
# same buffers through the HIP kernels) ------------------------------------------------------

def test_cc_rules_on_handbuilt_responses():
    import cc_cases
    cases = cc_cases.detect_cases()
    assert len(cases) >= 25
    for name, d, img, expected in cases:
        out = oracle.cc_detect_on_response(d, img)
        if expected is not None:
            assert out.tolist() == expected, name


def test_refine_rules():
    import cc_cases
    for name, pts, lv, d, img, level, expected in cc_cases.refine_cases():
        p2, l2, n = oracle.cc_refine_on_response(pts, lv, d, img, level)
        if expected is not None:
            ep, el, en = expected
            assert n == en and l2.tolist() == el and np.array_equal(p2, ep), name
    # two points sharing one blob: exactly one of them is refined, and it is the first
    name, pts, lv, d, img, level, _ = [c for c in cc_cases.refine_cases() if c[0].startswith("two points share")][0]
    p2, l2, n = oracle.cc_refine_on_response(pts, lv, d, img, level)
    assert n == 1 and l2.tolist() == [0, 1]


def test_random_sparse_responses_are_nontrivial():
    """The random generator of the GPU rule tests produces accepted AND rejected blobs."""
    import cc_cases
    rng = np.random.RandomState(0)
    d = cc_cases.random_sparse_response(rng, 96, 131, 60, noise=0.002)
    out = oracle.cc_detect_on_response(d, cc_cases.flat_img(96, 131))
    assert 3 < len(out) < 60


# ----------------------------------------------------------------------------- preprocessing (row (f)-2)

def test_oracle_normalize_and_clahe_properties():
    """Hand-derived properties of the restated cv::normalize / CLAHE arithmetic (parity unpinned:
    OpenCV is absent, see the oracle's header)."""
    from oracle import oracle
    rng = np.random.RandomState(3)
    img = (rng.rand(96, 128) * 100 + 50).astype(np.uint8)
    n = oracle.normalize_minmax(img)
    assert n.min() == 0 and n.max() == 255                       # NORM_MINMAX to [0, 255]
    lo, hi = int(img.min()), int(img.max())
    scale = 255.0 / (hi - lo)
    want = np.clip(np.rint(np.float32(img.astype(np.float32) * np.float32(scale)) + np.float32(-lo * scale)), 0, 255)
    assert np.array_equal(n, want.astype(np.uint8))              # float multiply, float add, round half even
    order = np.argsort(img.ravel(), kind="stable")
    assert (np.diff(n.ravel()[order].astype(int)) >= 0).all()      # a per-frame value map: monotone
    flat = np.full((64, 64), 91, np.uint8)
    assert (oracle.normalize_minmax(flat) == 0).all()            # max == min: scale 0, everything maps to dmin
    # CLAHE of a constant 64x64 frame (tiles of 8x8 = 64 pixels, clip = int(8*64/256) = 2): bin 91 is
    # clipped to 2, the 62 clipped pixels go one each to bins 0, 4, ..., 244 (step 256/62 = 4), so the
    # cumulative count at 91 is 23 + 2 = 25 and the LUT value round(25 * 255/64) = 100
    c = oracle.clahe(flat)
    assert (c == 100).all()
    # every tile with the same content (a 32-column ramp per 32-column tile): all 64 LUTs are equal, the
    # blend returns LUT[v], and a LUT is a cumulative histogram -- monotone in v, identical in every tile
    ramp = np.tile((np.arange(32) * 8).astype(np.uint8), (64, 8))
    out = oracle.clahe(ramp)
    assert (np.diff(out[0, :32].astype(int)) >= 0).all() and out[0, 31] == 255
    assert all(np.array_equal(out[:, :32], out[:, 32 * k:32 * k + 32]) for k in range(8))
    # ragged size: both axes are padded as soon as one of them does not divide into 8 tiles
    odd = (rng.rand(61, 64) * 255).astype(np.uint8)
    assert oracle.clahe(odd).shape == (61, 64)
    # blur of a constant frame is the constant; preprocess composes the three steps
    assert (oracle.box_blur(flat, 2) == 91).all()
    p = oracle.preprocess(img, clahe=True, blur_radius=1)
    assert np.array_equal(p, oracle.box_blur(oracle.clahe(oracle.normalize_minmax(img)), 1))
    assert np.array_equal(oracle.preprocess(img, clahe=False, blur_radius=0), img)


def _blob_scenes():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return {"shapes_320x360": mg.blob_scene(320, 360), "dots_333x251": synth.dots_frame(333, 251, 4, 1).numpy(),
            "dots_640x480": synth.dots_frame(640, 480, 10, 0).numpy(),
            "noise_320x240_smooth2": synth.noise_frame(320, 240, 1, smooth=2).numpy()}


def test_blob_and_preprocess_restatements_regression(golden_dir):
    """oracle/blobs_oracle.c and the preprocessing restatement against the committed vectors (regression anchors of
    restatements of OpenCV's published algorithms -- parity with OpenCV itself stays unpinned)."""
    z = np.load(os.path.join(golden_dir, "blobs_golden.npz"))
    for name, img in _blob_scenes().items():
        assert hashlib.sha256(img.tobytes()).digest() == z["sha_" + name].tobytes(), name
        assert np.array_equal(oracle.find_blobs(img).astype(np.int64), z["blobs_" + name]), name
    pre = synth.board_frame(333, 251, 6, 3).numpy()
    assert np.array_equal(oracle.preprocess(pre, clahe=True, blur_radius=1), z["pre8_board_333x251"])
    pre16 = (pre.astype(np.uint16) * 120 + 9000).astype(np.uint16)
    assert np.array_equal(oracle.preprocess16(pre16, clahe=True, blur_radius=1), z["pre16_board_333x251"])
    assert len(z["blobs_dots_640x480"]) == 100
