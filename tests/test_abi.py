"""CPU checks of the boundary: the C-ABI library loads, exports every symbol
include/mrgingham_amd.h declares, and refuses to compute without a device."""
import ctypes
import os
import re

import pytest

from mrgingham_amd import _lib
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mrgingham_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", src)
    skip = {"defined", "add_points", "C", "bool", "int", "void", "double"}
    out = []
    for n in names:
        if n in skip or n.startswith("__") or n in out:
            continue
        out.append(n)
    return out


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert "mrgingham_ChESS_response_5" in declared
    assert "find_chessboard_corners_from_image_array_C" in declared
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/mrgingham_amd.h but not exported"
    assert sorted(declared) == sorted(_lib.EXPORTS)
    assert L.mrgingham_amd_abi_version() == 4          # (bumped with the boundary: the history is in the header)


def test_level_dims_match_oracle():
    L = _lib.lib()
    w, h = ctypes.c_int(), ctypes.c_int()
    for (W, H) in [(4096, 3072), (1920, 1080), (640, 480), (1001, 999), (1003, 1005), (37, 53), (15, 15), (6, 2)]:
        for level in range(0, 5):
            assert L.mrgingham_amd_level_dims(W, H, level, ctypes.byref(w), ctypes.byref(h)) == 0
            assert (w.value, h.value) == oracle.level_dims(W, H, level), (W, H, level)
    assert L.mrgingham_amd_level_dims(64, 64, 11, ctypes.byref(w), ctypes.byref(h)) != 0
    assert L.mrgingham_amd_level_dims(64, 64, -1, ctypes.byref(w), ctypes.byref(h)) != 0


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = _lib.lib()
    assert L.mrgingham_amd_create(0) is None            # fails loudly, no host path
    import numpy as np
    import mrgingham_amd
    assert L.mrgingham_amd_device_count() == 0
    with pytest.raises(RuntimeError):
        mrgingham_amd.Detector()
    img = np.zeros((64, 64), np.uint8)
    for call in (lambda: mrgingham_amd.find_points(img), lambda: mrgingham_amd.ChESS_response_5(img),
                 lambda: mrgingham_amd.find_board(img),
                 lambda: mrgingham_amd.refine_points(np.zeros((1, 2)), np.zeros(1, np.int8), img, 0)):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()
    # the raw C symbol can only say "nothing found" (and prints why); it must not write results
    out = np.full((64, 64), -7, np.int16)
    L.mrgingham_ChESS_response_5(out.ctypes.data, img.ctypes.data, 64, 64, 64)
    assert (out == -7).all()


def test_python_mirror_argument_checks_match_reference_messages():
    import numpy as np
    import mrgingham_amd
    with pytest.raises(RuntimeError, match="exactly 2 dims"):            # mrgingham_pywrap.c:163-168
        mrgingham_amd.find_points(np.zeros((2, 8, 8), np.uint8))
    with pytest.raises(RuntimeError, match="8-bit unsigned"):            # :169-173
        mrgingham_amd.find_points(np.zeros((8, 8), np.uint16))
    with pytest.raises(RuntimeError, match="contiguous memory"):         # :174-178
        mrgingham_amd.find_points(np.zeros((8, 16), np.uint8)[:, ::2])
    with pytest.raises(RuntimeError, match="at least 2 dims"):           # :53-58
        mrgingham_amd.ChESS_response_5(np.zeros(8, np.uint8))
    with pytest.raises(RuntimeError, match="image_pyramid_level == 0"):  # :153-157
        mrgingham_amd.find_points(np.zeros((8, 8), np.uint8), image_pyramid_level=1, blobs=True)
    with pytest.raises(RuntimeError, match="gridn value must be >= 2"):    # :312-316
        mrgingham_amd.find_board(np.zeros((8, 8), np.uint8), gridn=1)
    with pytest.raises(RuntimeError, match="INTEGER,INTEGER"):            # :275-282
        mrgingham_amd.find_board(np.zeros((8, 8), np.uint8), debug_sequence="x")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "mrgingham_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for pat in ("import oracle", "from oracle", "liboracle", "mrgingham_oracle", "oracle/", "oracle."):
                    assert pat not in text, f"{f} references the oracle ({pat})"


def test_shipped_library_has_no_experiment_hooks():
    """The timing ablations / placement experiments of tools/ exist in -DMRG_EXPERIMENT builds only: the shipped
    library must not even contain the names of their environment variables (a stray variable cannot change what
    it computes), and its source gates every one of them."""
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = open(os.path.join(here, "mrgingham_amd", "libmrgingham_amd.so"), "rb").read()
    for name in (b"MRGINGHAM_AMD_PYR_SKIP", b"MRGINGHAM_AMD_CC_LDS_PAD", b"MRGINGHAM_AMD_CC_CUS",
                 b"MRGINGHAM_AMD_PIX_COMPLEMENT", b"MRGINGHAM_AMD_CHESS_V0", b"MRG_DBG_FB", b"chess_variant_hot",
                 b"chess_v16_pyr_kernel", b"chess_v16_multi_kernel", b"chess_v16_pair_kernel", b"chess16_pair"):
        assert name not in blob, name
    assert b"MRGINGHAM_AMD_DEVICE" in blob                  # the one variable it does read
    for src in ("api.hip", "chess.hip", "cc.hip"):
        text = open(os.path.join(here, "mrgingham_amd", "csrc", src)).read()
        lines = text.split("\n")
        depth = 0
        for ln in lines:                                    # every getenv outside MRGINGHAM_AMD_DEVICE sits in #ifdef MRG_EXPERIMENT
            st = ln.strip()
            if st.startswith("#ifdef MRG_EXPERIMENT"):
                depth += 1
            elif st.startswith("#if") and depth:
                depth += 1
            elif st.startswith("#else") and depth == 1:
                depth = -1                                   # the release branch of an experiment block
            elif st.startswith("#endif") and depth:
                depth = 0 if depth in (1, -1) else depth - 1
            if "getenv(" in ln and "MRGINGHAM_AMD_DEVICE" not in ln:
                assert depth > 0, (src, ln)


def test_reference_module_name_is_importable():
    """`import mrgingham` gives the reference's five names (mrgingham_pywrap.c:357-368)."""
    import mrgingham
    import mrgingham_amd
    assert sorted(mrgingham.__all__) == ["ChESS_response_5", "find_board", "find_chessboard", "find_chessboard_corners",
                                         "find_points"]
    for name in mrgingham.__all__:
        assert getattr(mrgingham, name) is getattr(mrgingham_amd, name)
    assert mrgingham.find_chessboard is mrgingham.find_board and mrgingham.find_chessboard_corners is mrgingham.find_points


def test_thread_to_device_policy_and_shard_ranges():
    """Multi-GPU at the C boundary, the host-only parts: the k-th calling thread gets device k % devices unless
    MRGINGHAM_AMD_DEVICE names one (the reference's worker model -- image i on worker i % N,
    mrgingham-from-image.cc:50 -- mapped onto devices), and the contiguous shard split of mrgingham_amd_chain_multi's
    callers is the one parallel.py uses."""
    from mrgingham_amd import api, parallel
    assert [api.device_for_thread(k, 8) for k in range(18)] == [k % 8 for k in range(18)]
    assert [api.device_for_thread(k, 1) for k in range(4)] == [0, 0, 0, 0]
    assert [api.device_for_thread(k, 8, "5") for k in range(4)] == [5, 5, 5, 5]      # the variable wins, for every thread
    assert api.device_for_thread(3, 8, "") == 3                                        # (set but empty: not a choice)
    assert api.device_for_thread(7, 0) == 0
    for total in (0, 1, 7, 64, 2048, 2051):
        for n in (1, 2, 3, 8):
            got = [api.shard_range(total, k, n) for k in range(n)]
            assert [(a, a + c) for a, c in got] == [parallel.shard_range(total, k, n) for k in range(n)]
            assert sum(c for _, c in got) == total and got[0][0] == 0
            assert all(got[k][0] + got[k][1] == got[k + 1][0] for k in range(n - 1))
            assert max(c for _, c in got) - min(c for _, c in got) <= 1
    with pytest.raises(ValueError):
        api.shard_range(10, 3, 3)


def test_packed_layout_is_the_one_parallel_py_uses():
    """mrgingham_amd_packed_layout (the block mrgingham_amd_gather_rccl moves) against parallel.packed_outputs, on the CPU."""
    import ctypes
    from mrgingham_amd import parallel
    L = _lib.lib()
    for B, P in [(1, 1), (5, 256), (64, 256), (3, 7), (256, 1024), (0, 16)]:
        o_lv, o_np, nb = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        assert L.mrgingham_amd_packed_layout(B, P, ctypes.byref(o_lv), ctypes.byref(o_np), ctypes.byref(nb)) == 0
        pack, pts, lv, npts = parallel.packed_outputs(B, P, "cpu")
        assert nb.value == pack.numel()
        if B:
            assert lv.data_ptr() - pack.data_ptr() == o_lv.value and npts.data_ptr() - pack.data_ptr() == o_np.value
    assert L.mrgingham_amd_packed_layout(-1, 4, None, None, None) == -1 and L.mrgingham_amd_packed_layout(4, 0, None, None, None) == -1


def test_wait_policy_rejects_unknown_values_without_touching_the_device():
    from mrgingham_amd import _lib
    assert _lib.lib().mrgingham_amd_set_wait_policy(9) == -1          # MRGINGHAM_AMD_ERR_ARG
