"""BOUNDARY test: a plain C99 caller of include/mrgingham_amd.h (tests/boundary/c_main.c).  CPU: the header compiles
as C (-std=c99 -pedantic -Werror) and every function it declares links.  GPU: the reference's symbols called from C
return what the Python mirror returns for the same file."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "mrgingham_amd")


def _build(tmp_path):
    from mrgingham_amd import _lib
    (tmp_path / "all_symbols.inc").write_text("".join(f"MRG_SYMBOL({s})\n" for s in _lib.EXPORTS))
    exe = str(tmp_path / "c_main")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + str(tmp_path),
           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "boundary", "c_main.c"), "-o", exe,
           "-L" + LIBDIR, "-lmrgingham_amd", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_header_is_c99_and_every_declared_function_links(tmp_path):
    from mrgingham_amd import _lib
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split()[:2] == ["symbols", str(len(_lib.EXPORTS))], (r.stdout, r.stderr)
    # the header alone, strictly: no C++-isms, no GNU extensions
    (tmp_path / "only_header.c").write_text('#include "mrgingham_amd.h"\nint mrg_unused_tu;\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic-errors", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                        "-I" + os.path.join(ROOT, "include"), str(tmp_path / "only_header.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_rccl_host_example_compiles_as_c99():
    """tests/boundary/rccl_host.c -- the rank of a one-process-per-GPU host of INTEGRATION.md 5d: shard, packed block,
    chain_batch, ONE ncclGather through mrgingham_amd_gather_rccl -- against the header, the HIP runtime API and rccl.h
    (syntax and types; ranks and GPUs are needed to run it)."""
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("no RCCL headers here")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__",
                        "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "boundary", "rccl_host.c")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.gpu
def test_c_caller_gets_what_the_python_mirror_gets(tmp_path):
    import mrgingham_amd
    from mrgingham_amd import synth
    exe = _build(tmp_path)
    W, H = 1280, 960
    img = synth.board_frame(W, H, 10, 7).numpy()
    pgm = tmp_path / "img.pgm"
    pgm.write_bytes(b"P5\n%d %d\n255\n" % (W, H) + img.tobytes())
    r = subprocess.run([exe, str(pgm)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    lines = r.stdout.splitlines()
    assert lines[0] == f"image {W} {H}"
    resp = mrgingham_amd.ChESS_response_5(img).astype(np.int64).ravel()
    assert lines[1] == f"response_checksum {int((resp * (1 + np.arange(resp.size) % 7)).sum())}"
    want = mrgingham_amd.find_points(img, 1)
    pts = np.array([[int(t) for t in l.split()[1:]] for l in lines if l.startswith("p ")])
    assert f"corners found 1 n {len(want)} scale 0.001000" in lines
    assert np.array_equal(pts, np.round(want * 1000).astype(np.int64))
    board = mrgingham_amd.find_board(img, gridn=10)
    got = np.array([[float(t) for t in l.split()[1:]] for l in lines if l.startswith("b ")])
    assert board is not None and "board found 1 n 100" in lines and np.array_equal(got, board)
    assert "bad_level 0 -1" in lines


@pytest.mark.gpu
def test_rccl_host_example_runs_one_rank(tmp_path):
    """tests/boundary/rccl_host.c with its main(): a ONE-rank communicator from ncclGetUniqueId / ncclCommInitRank, the
    rank's shard through chain_batch into the packed block, mrgingham_amd_gather_rccl -> the gathered block equals a
    plain chain_batch of the same frames (checked in C, byte for byte) AND what the Python mirror returns here."""
    import torch
    import mrgingham_amd
    from mrgingham_amd import parallel, synth
    if not os.path.exists("/opt/rocm/include/rccl/rccl.h") or not os.path.exists("/opt/rocm/lib/librccl.so"):
        pytest.skip("no RCCL development files here")
    exe = str(tmp_path / "rccl_host")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "boundary", "rccl_host.c"),
                        "-o", exe, "-L" + LIBDIR, "-lmrgingham_amd", "-L/opt/rocm/lib", "-lrccl", "-lamdhip64",
                        "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    B, W, H, P = 6, 1280, 960, 256
    frames = synth.board_batch(B, W, H, 10, 70, device="cuda:0")
    (tmp_path / "frames.raw").write_bytes(frames.cpu().numpy().tobytes())
    out = tmp_path / "gathered.bin"
    r = subprocess.run([exe, str(tmp_path / "frames.raw"), str(B), str(W), str(H), str(P), str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    assert "gathered_equals_chain 1" in r.stdout.splitlines()
    gathered = torch.from_numpy(np.frombuffer(out.read_bytes(), dtype=np.uint8).copy())[None]
    gp, gl, gn = parallel.unpack_outputs(gathered, B, P)
    det = mrgingham_amd.Detector(0)
    try:
        want = [t.cpu() for t in det.chain(frames, 3, P)]
    finally:
        det.close()
    assert torch.equal(gn[0], want[2]) and int(gn[0].min()) >= 50
    for f in range(B):
        n = int(want[2][f])
        assert torch.equal(gp[0][f, :n], want[0][f, :n]) and torch.equal(gl[0][f, :n], want[1][f, :n])
