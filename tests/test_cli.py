"""Row (f)-3: the command-line tool (mrgingham_amd/bin/mrgingham-amd-from-image), the reference's
mrgingham-from-image.cc over the library.  CPU tests: option / argument behaviour (exit codes and
messages of mrgingham-from-image.cc:222-330).  GPU tests: vnlog output for PGM / PNG files against the
oracle's preprocessing + detector + refinement composed with the grid finder."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "mrgingham_amd", "bin", "mrgingham-amd-from-image")


def _run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=600)


def _write_pgm(path, img, maxval=255):
    with open(path, "wb") as f:
        f.write(b"P5\n# a comment line\n%d %d\n%d\n" % (img.shape[1], img.shape[0], maxval))
        f.write(img.astype(">u2").tobytes() if maxval > 255 else img.astype(np.uint8).tobytes())


def _write_png(path, img, rgb=False):
    """Minimal PNG writer (8-bit grey or RGB), every row with a different filter type."""
    h, w = img.shape[:2]
    ch = 3 if rgb else 1
    rows = img.reshape(h, w * ch).astype(np.int32)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft = y % 5
        cur = rows[y]
        left = np.concatenate([np.zeros(ch, np.int32), cur[:-ch]])
        upleft = np.concatenate([np.zeros(ch, np.int32), prev[:-ch]])
        if ft == 0:
            pred = np.zeros_like(cur)
        elif ft == 1:
            pred = left
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (left + prev) >> 1
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
        raw.append(ft)
        raw += ((cur - pred) & 0xff).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    comp = zlib.compress(bytes(raw), 6)
    half = len(comp) // 2                                   # two IDAT chunks
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if rgb else 0, 0, 0, 0)) +
                chunk(b"IDAT", comp[:half]) + chunk(b"IDAT", comp[half:]) + chunk(b"IEND", b""))


def test_cli_is_built_and_prints_usage():
    assert os.path.exists(CLI), "run python __graft_entry__.py (make -C mrgingham_amd/csrc)"
    r = _run("--help")
    assert r.returncode == 0 and "imageglobs" in r.stdout and "--gridn" in r.stdout


def test_cli_argument_errors(tmp_path):
    r = _run()
    assert r.returncode == 1 and "Not enough arguments: need image globs" in r.stderr
    r = _run("--jobs", "0", "x*.pgm")
    assert r.returncode == 1 and "The job count must be a positive integer" in r.stderr
    r = _run("--gridn", "1", "x*.pgm")
    assert r.returncode == 1 and "--gridn value must be >= 2" in r.stderr
    r = _run("--blobs", "--level", "0", "x*.pgm")
    assert r.returncode == 1 and "'image_pyramid_level' only implemented for chessboards" in r.stderr   # :305-309
    r = _run("--debug-sequence", "nonsense", "x*.pgm")
    assert r.returncode != 0 and "could not parse 'x,y'" in r.stderr
    r = _run("--frobnicate", "x*.pgm")
    assert r.returncode == 1 and "Unknown option" in r.stderr
    r = _run(str(tmp_path / "nothing-here-*.pgm"))
    assert r.returncode == 1 and "matched no files!" in r.stderr
    a, b = tmp_path / "a.pgm", tmp_path / "b.pgm"
    for p in (a, b):
        _write_pgm(p, np.zeros((32, 32), np.uint8))
    r = _run("--debug", str(tmp_path / "*.pgm"))
    assert r.returncode == 1 and "When debugging, pass one image at a time. Got 2 instead" in r.stderr


def test_cli_without_a_gpu_fails_loudly(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = tmp_path / "a.pgm"
    _write_pgm(p, np.zeros((64, 64), np.uint8))
    r = _run(str(p))
    assert r.returncode == 2 and "no HIP device" in r.stderr and "# filename" not in r.stdout


def _expected(img, gridn, level, clahe, blur, refine=True):
    import mrgingham_amd
    from oracle import oracle
    pre = (oracle.preprocess16 if img.dtype == np.uint16 else oracle.preprocess)(img, clahe=clahe, blur_radius=blur)
    for L in ([level] if level >= 0 else [3, 2, 1, 0]):
        cand = oracle.find_corners(pre, L)
        if cand is None or len(cand) < gridn * gridn:
            continue
        grid = mrgingham_amd.find_grid_from_points(cand, gridn)
        if grid is None:
            continue
        pts, lv = grid.copy(), np.full(gridn * gridn, L, np.int8)
        if refine:
            for l in range(L - 1, -1, -1):
                pts, lv, n = oracle.refine_corners(pts, lv, pre, l)
                if n <= 0:
                    break
        return pts, lv
    return None, None


def _parse(stdout):
    out = {}
    lines = stdout.splitlines()
    assert lines[0].startswith("## generated with") and lines[1] == "# filename x y level"
    for ln in lines[2:]:
        if ln.startswith("#"):
            continue
        name, x, y, lv = ln.split()
        out.setdefault(name, []).append(None if x == "-" else (float(x), float(y), int(lv)))
    return out


@pytest.mark.gpu
def test_cli_vnlog_matches_composed_oracle(tmp_path):
    from mrgingham_amd import synth
    files = {}
    for i, (w, h, seed) in enumerate([(640, 480, 0), (1280, 960, 1), (800, 600, 2)]):
        img = (synth.board_frame(w, h, 10, seed).numpy().astype(np.float64) * 0.6 + 30).astype(np.uint8)
        p = str(tmp_path / f"board{i}.{'png' if i == 1 else 'pgm'}")
        (_write_png if i == 1 else _write_pgm)(p, img)
        files[p] = img
    noise = str(tmp_path / "board_none.pgm")
    files[noise] = synth.noise_frame(640, 480, 3, smooth=1).numpy()
    _write_pgm(noise, files[noise])
    for args, clahe, blur, level, refine in [((), True, 1, -1, True), (("--noclahe", "--blur", "0", "--level", "1"), False, 0, 1, True),
                                             (("--no-refine", "-j", "3"), True, 1, -1, False)]:
        r = _run(*args, str(tmp_path / "board*.p[gn][mg]"))
        assert r.returncode == 0, r.stderr
        got = _parse(r.stdout)
        assert set(got) == set(files)
        for name, img in files.items():
            want, lv = _expected(img, 10, level, clahe, blur, refine)
            if want is None:
                assert got[name] == [None], name
                continue
            assert len(got[name]) == 100, (name, args)
            g = np.array([(x, y) for x, y, _ in got[name]])
            assert np.abs(g - want).max() < 1e-6, (name, args)               # "%f": 6 decimals
            assert [l for _, _, l in got[name]] == list(lv), (name, args)


@pytest.mark.gpu
def test_cli_with_more_workers_than_device_slots_hands_its_images_to_device_threads(tmp_path):
    """--jobs above eight per GPU: the workers decode and print, eight device threads per GPU call the library
    (cli/mrgingham_from_image.cpp, kDeviceSlots).  Same lines per file as one worker gives, every file exactly once."""
    from mrgingham_amd import synth
    sizes = [(640, 480), (800, 600), (1280, 960)]
    for i in range(30):
        w, h = sizes[i % 3]
        img = synth.board_frame(w, h, 10, i).numpy() if i % 7 else synth.noise_frame(w, h, i, smooth=1).numpy()
        _write_pgm(str(tmp_path / f"img{i:02d}.pgm"), img)
    one = _run("--jobs", "1", str(tmp_path / "img*.pgm"))
    many = _run("--jobs", "40", "--gpus", "1", str(tmp_path / "img*.pgm"))
    assert one.returncode == 0 and many.returncode == 0, (one.stderr, many.stderr)
    a, b = _parse(one.stdout), _parse(many.stdout)
    assert len(a) == 30 and a == b


@pytest.mark.gpu
def test_cli_rgb_png_16bit_pgm_and_unreadable_file(tmp_path):
    from mrgingham_amd import synth
    img = synth.board_frame(640, 480, 10, 7).numpy()
    rgb = np.stack([img, img, img], axis=-1)                                 # grey stored as RGB: weights sum to 2^14
    p_rgb, p_16, p_bad = str(tmp_path / "a_rgb.png"), str(tmp_path / "b_16.pgm"), str(tmp_path / "c_bad.pgm")
    _write_png(p_rgb, rgb, rgb=True)
    _write_pgm(p_16, img.astype(np.uint16) * 257, maxval=65535)              # 8-bit values on the 16-bit scale
    open(p_bad, "wb").write(b"P5\nnot an image")
    want, lv = _expected(img, 10, -1, False, 1)
    r = _run("--noclahe", p_rgb, p_16)
    assert r.returncode == 0, r.stderr
    got = _parse(r.stdout)
    for name in (p_rgb, p_16):
        g = np.array([(x, y) for x, y, _ in got[name]])
        assert g.shape == (100, 2) and np.abs(g - want).max() < 1e-6, name
    # 16 bit WITH the contrast step: normalize to 0..65535, CLAHE on 16 bits, convertTo 8 bit (:85-92)
    rng = np.random.RandomState(4)
    img16 = (img.astype(np.float64) * 120 + 9000 + rng.randint(0, 120, img.shape)).astype(np.uint16)   # a dim 16-bit frame
    p_dim = str(tmp_path / "d_dim16.pgm")
    _write_pgm(p_dim, img16, maxval=65535)
    r = _run(p_dim)
    assert r.returncode == 0, r.stderr
    want16, lv16 = _expected(img16, 10, -1, True, 1)
    g = np.array([(x, y) for x, y, _ in _parse(r.stdout)[p_dim]])
    assert want16 is not None and g.shape == (100, 2) and np.abs(g - want16).max() < 1e-6
    r = _run(p_bad, p_rgb)                                                   # one worker: the bad file ends it (:58-68)
    assert r.returncode == 0 and "Couldn't open image" in r.stderr
    got = _parse(r.stdout)
    assert got[p_bad] == [None] and p_rgb not in got


@pytest.mark.gpu
def test_preprocess16_matches_the_oracle_on_ragged_sizes():
    """The 16-bit branch alone (mrgingham-from-image.cc:85-92) through process_image_ex's debug dump would be
    indirect; compare the preprocessed 8-bit image itself via the --debug PNG on three sizes."""
    import mrgingham_amd
    from oracle import oracle
    rng = np.random.RandomState(8)
    for (h, w) in [(64, 64), (61, 77), (240, 333)]:
        img16 = (rng.rand(h, w) * 30000 + 2000).astype(np.uint16)
        img16[h // 4:h // 2, w // 4:w // 2] += 20000
        for clahe, blur in [(True, 1), (True, 0), (False, 2)]:
            got = mrgingham_amd.api.preprocess16(img16, clahe=clahe, blur_radius=blur)
            assert np.array_equal(got, oracle.preprocess16(img16, clahe=clahe, blur_radius=blur)), (h, w, clahe, blur)


@pytest.mark.gpu
def test_cli_debug_dumps(tmp_path):
    """--debug: the preprocessed image, per pass the level image / normalised responses / corner vnlog, with
    the reference's file names and messages (mrgingham-from-image.cc:113-148; find_chessboard_corners.cc:282-315,
    :453-459, :513-541)."""
    import glob
    import mrgingham_amd
    from mrgingham_amd import synth
    from oracle import oracle
    for f in glob.glob("/tmp/mrgingham-*") + glob.glob("/tmp/dbgboard_preprocessed.png"):
        os.remove(f)
    img = synth.board_frame(640, 480, 10, 3).numpy()
    p = str(tmp_path / "dbgboard.pgm")
    _write_pgm(p, img)
    r = _run("--debug", "--level", "1", p)
    assert r.returncode == 0, r.stderr
    assert len(_parse(r.stdout)[p]) == 100
    for msg in ("Wrote preprocessed image to /tmp/dbgboard_preprocessed.png",
                "Wrote scaled,processed image to /tmp/mrgingham-scaled-processed-level1.png",
                "Wrote a normalized ChESS response to /tmp/mrgingham-chess-response-level1.png",
                "Wrote positive-only, normalized ChESS response to /tmp/mrgingham-chess-response-level1-positive.png",
                "Writing self-plotting corner dump to /tmp/mrgingham-1-corners.vnl",
                "Wrote a normalized ChESS response to /tmp/mrgingham-chess-response-refinement-level0.png",
                "Writing self-plotting corner dump to /tmp/mrgingham-1-corners-refinement-level0.vnl"):
        assert msg in r.stderr, msg
    pre = oracle.preprocess(img, clahe=True, blur_radius=1)
    assert np.array_equal(mrgingham_amd.read_image("/tmp/dbgboard_preprocessed.png"), pre)
    assert np.array_equal(mrgingham_amd.read_image("/tmp/mrgingham-scaled-processed-level1.png"), oracle.decimate(pre, 1))
    resp, _ = oracle.clamped_response(pre, 1)
    want = np.rint((resp.astype(np.float32) * np.float32(255.0 / resp.max()))).astype(np.uint8)   # min is 0 after the clamp
    assert np.array_equal(mrgingham_amd.read_image("/tmp/mrgingham-chess-response-level1-positive.png"), want)
    lines = open("/tmp/mrgingham-1-corners.vnl").read().splitlines()
    assert lines[0].startswith("#!/usr/bin/feedgnuplot") and p in lines[0] and lines[1] == "# x y"
    cand = oracle.find_corners(pre, 1)
    got = np.array([[float(t) for t in ln.split()] for ln in lines[2:]])
    assert np.array_equal(np.round(got * 1000).astype(np.int64), cand.astype(np.int64))
    ref = open("/tmp/mrgingham-1-corners-refinement-level0.vnl").read().splitlines()
    assert len(ref) == 2 + 100


@pytest.mark.gpu
def test_file_entry_points_match_the_array_functions(tmp_path):
    """find_chessboard_corners_from_image_file / find_chessboard_from_image_file (find_chessboard_corners.cc:623-648,
    mrgingham.cc:145-170) = decode the file + the array function."""
    import ctypes
    import mrgingham_amd
    from mrgingham_amd import synth, _lib
    L = _lib.lib()
    img = synth.board_frame(800, 600, 10, 4).numpy()
    p_pgm, p_png = str(tmp_path / "f.pgm"), str(tmp_path / "f.png")
    _write_pgm(p_pgm, img)
    _write_png(p_png, img)
    ADD_I = ctypes.CFUNCTYPE(ctypes.c_bool, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_double, ctypes.c_void_p)
    ADD_D = ctypes.CFUNCTYPE(ctypes.c_bool, ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_void_p)
    L.find_chessboard_corners_from_image_file_C.restype = ctypes.c_bool
    L.find_chessboard_corners_from_image_file_C.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_bool, ADD_I, ctypes.c_void_p]
    L.find_chessboard_from_image_file_C.restype = ctypes.c_bool
    L.find_chessboard_from_image_file_C.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_bool, ADD_D, ctypes.c_void_p]
    for path in (p_pgm, p_png):
        got = []
        cb = ADD_I(lambda xy, n, scale, _: got.append(np.ctypeslib.as_array(xy, (2 * n,)).reshape(n, 2).copy() * scale) or True)
        assert L.find_chessboard_corners_from_image_file_C(path.encode(), 1, False, cb, None)
        assert np.array_equal(got[0], mrgingham_amd.find_points(img, image_pyramid_level=1))
        board = []
        cbd = ADD_D(lambda xy, n, _: board.append(np.ctypeslib.as_array(xy, (2 * n,)).reshape(n, 2).copy()) or True)
        assert L.find_chessboard_from_image_file_C(path.encode(), 10, -1, False, cbd, None)
        assert np.array_equal(board[0], mrgingham_amd.find_board(img, gridn=10))
    nothing = ADD_I(lambda *a: True)
    assert not L.find_chessboard_corners_from_image_file_C(str(tmp_path / "missing.pgm").encode(), 0, False, nothing, None)
