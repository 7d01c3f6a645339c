"""BOUNDARY test (not an oracle, pins no arithmetic): include/find_chessboard_corners_amd.hh -- the cv::Mat shim a
maintainer of the reference compiles instead of find_chessboard_corners.cc -- builds against a structural stand-in for
cv::Mat (tests/boundary/), carries exactly the signatures of find_chessboard_corners.hh:12-30, :32-44, :51-72, links
against the library, and (GPU) returns what the C symbols return."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "mrgingham_amd")


REFERENCE = "/root/reference"   # present in the build container only: then the shim is compiled against the REAL point.hh


def _build(tmp_path):
    exe = str(tmp_path / "shim_main")
    if os.path.exists(os.path.join(REFERENCE, "point.hh")):
        points = REFERENCE
    else:                            # the GPU box: a stand-in with the two point types the shim names (two ints / two doubles)
        points = str(tmp_path)
        (tmp_path / "point.hh").write_text(
            "namespace mrgingham {\n"
            "struct PointInt { int x, y; PointInt(int a = 0, int b = 0) { x = a; y = b; } };\n"
            "struct PointDouble { double x, y; PointDouble(double a = 0, double b = 0) { x = a; y = b; } };\n"
            "}\n")
    cmd = ["g++", "-std=c++11", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "boundary"), "-I" + points,
           "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "boundary", "shim_main.cpp"), "-o", exe, "-L" + LIBDIR, "-lmrgingham_amd",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_cvmat_shim_compiles_with_the_reference_signatures_and_links(tmp_path):
    exe = _build(tmp_path)
    assert subprocess.run([exe], capture_output=True).returncode == 0     # (no arguments: nothing is computed)


@pytest.mark.gpu
def test_cvmat_shim_returns_what_the_c_symbols_return(tmp_path):
    import mrgingham_amd
    from mrgingham_amd import synth
    exe = _build(tmp_path)
    img = synth.board_frame(1280, 960, 10, 3).numpy()
    raw = tmp_path / "img.bin"
    raw.write_bytes(img.tobytes())
    r = subprocess.run([exe, str(raw), "1280", "960", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    pts = np.array([[int(t) for t in l.split()[1:]] for l in lines if l.startswith("p ")])
    want = mrgingham_amd.find_points(img, 2)
    assert lines[0] == f"found 1 n {len(want)}" and np.array_equal(pts, np.round(want * 1000).astype(np.int64))
    ref = np.array([[float(t) for t in l.split()[1:3]] for l in lines if l.startswith("r ")])
    lv = np.array([int(l.split()[3]) for l in lines if l.startswith("r ")])
    wp, wl, wn = mrgingham_amd.refine_points(want, np.full(len(want), 2, np.int8), img, 1)
    assert f"refined {wn}" in lines and np.array_equal(ref, wp) and np.array_equal(lv, wl)
    assert "bad_type 0" in lines                                           # not CV_8U: no points (find_chessboard_corners.cc:468-473)
