"""The library's own image decoders (csrc/image_io.cpp behind mrgingham_amd_read_image): binary PGM and
non-interlaced PNG, 8 and 16 bit, and their behaviour on truncated / crafted files (host only, no GPU)."""
import struct
import zlib

import numpy as np

import mrgingham_amd
from test_cli import _write_pgm, _write_png


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def _png_bytes(w, h, bits, ctype, raw, ihdr_first=True, extra_ihdr=False):
    sig = b"\x89PNG\r\n\x1a\n"
    ihdr = _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bits, ctype, 0, 0, 0))
    idat = _chunk(b"IDAT", zlib.compress(raw))
    body = (ihdr + idat) if ihdr_first else (idat + ihdr)
    if extra_ihdr:
        body = ihdr + _chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 0)) + idat
    return sig + body + _chunk(b"IEND", b"")


def test_pgm_and_png_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(37, 53)).astype(np.uint8)
    p = str(tmp_path / "a.pgm")
    _write_pgm(p, img)
    assert np.array_equal(mrgingham_amd.read_image(p), img)
    q = str(tmp_path / "a.png")
    _write_png(q, img)
    assert np.array_equal(mrgingham_amd.read_image(q), img)
    rgb = rng.randint(0, 256, size=(20, 31, 3)).astype(np.uint8)
    r = str(tmp_path / "rgb.png")
    _write_png(r, rgb, rgb=True)
    g = (rgb[..., 0].astype(np.uint32) * 4899 + rgb[..., 1].astype(np.uint32) * 9617 +
         rgb[..., 2].astype(np.uint32) * 1868 + 8192) >> 14
    assert np.array_equal(mrgingham_amd.read_image(r), g.astype(np.uint8))


def test_16_bit_reduction_modes(tmp_path):
    """File entry points: cv::imread(IMREAD_GRAYSCALE) without ANYDEPTH keeps the high byte
    (find_chessboard_corners.cc:637-639); the CLI converts with 255/65535 and rounds
    (mrgingham-from-image.cc:85-92).  The two differ by one LSB on about half of the values."""
    v = np.arange(0, 65536, 37, dtype=np.uint16)
    img = np.resize(v, (40, 45))
    p = str(tmp_path / "w.pgm")
    _write_pgm(p, img, maxval=65535)
    hi = mrgingham_amd.read_image(p)
    cli = mrgingham_amd.read_image(p, cli_scaling=True)
    assert np.array_equal(hi, (img >> 8).astype(np.uint8))
    assert np.array_equal(cli, np.rint(img.astype(np.float64) * (255.0 / 65535.0)).astype(np.uint8))
    assert (hi != cli).any()
    # 16-bit grey PNG: same samples, big-endian, filter 0
    raw = b"".join(b"\x00" + img[y].astype(">u2").tobytes() for y in range(img.shape[0]))
    q = tmp_path / "w.png"
    q.write_bytes(_png_bytes(img.shape[1], img.shape[0], 16, 0, raw))
    assert np.array_equal(mrgingham_amd.read_image(str(q)), hi)


def test_malformed_files_are_rejected_not_trusted(tmp_path):
    good_raw = b"".join(b"\x00" + bytes([y] * 8) for y in range(8))

    def rd(name, data):
        f = tmp_path / name
        f.write_bytes(data)
        return mrgingham_amd.read_image(str(f))

    assert rd("ok.png", _png_bytes(8, 8, 8, 0, good_raw)) is not None
    # sides above the library's 32767 limit (and sizes that would wrap a size_t product) never allocate
    assert rd("huge.png", _png_bytes(0x7fffffff, 0x7fffffff, 8, 6, good_raw)) is None
    assert rd("wide.png", _png_bytes(40000, 8, 8, 0, good_raw)) is None
    assert rd("zero.png", _png_bytes(0, 8, 8, 0, good_raw)) is None
    # a header that claims more pixels than the data holds, or fewer
    assert rd("short.png", _png_bytes(8, 9, 8, 0, good_raw)) is None
    assert rd("long.png", _png_bytes(8, 7, 8, 0, good_raw)) is None
    # IDAT before IHDR, two IHDRs, bad filter byte, interlaced, odd bit depth, truncated file
    assert rd("order.png", _png_bytes(8, 8, 8, 0, good_raw, ihdr_first=False)) is None
    assert rd("two.png", _png_bytes(8, 8, 8, 0, good_raw, extra_ihdr=True)) is None
    assert rd("filter.png", _png_bytes(8, 8, 8, 0, b"".join(b"\x07" + bytes(8) for _ in range(8)))) is None
    data = bytearray(_png_bytes(8, 8, 8, 0, good_raw))
    data[8 + 8 + 12] = 1                                                      # interlace method 1
    assert rd("adam7.png", bytes(data)) is None
    assert rd("bits.png", _png_bytes(8, 8, 4, 0, good_raw)) is None
    assert rd("trunc.png", _png_bytes(8, 8, 8, 0, good_raw)[:40]) is None
    assert rd("garbage.png", b"\x89PNG\r\n\x1a\n" + bytes(64)) is None
    # PGM: oversize header, truncated data, nonsense
    assert rd("big.pgm", b"P5\n40000 2\n255\n" + bytes(16)) is None
    assert rd("trunc.pgm", b"P5\n8 8\n255\n" + bytes(63)) is None
    assert rd("max.pgm", b"P5\n8 8\n70000\n" + bytes(128)) is None
    assert rd("ascii.pgm", b"P2\n2 2\n255\n1 2 3 4\n") is None
    assert mrgingham_amd.read_image(str(tmp_path / "missing.pgm")) is None
