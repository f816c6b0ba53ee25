"""Film output: yart_film_tonemap / yart_film_write_bmp against the oracle's restatement of
the reference's tone map (src/bitmap.cpp:8-12) and against the BMP layout its writer emits
(src/bitmap.cpp:14-72, include/bitmap.h:10-33).  Parity unpinned: the reference's bitmap.cpp does not
compile with this g++, so there is no output of its own to compare with."""
import ctypes
import struct

import numpy as np
import pytest

from yart_amd import api


def oracle_tonemap(values):
    from oracle import oracle as orc
    L = ctypes.CDLL(orc.build_oracle())
    L.orc_tonemap.argtypes = [ctypes.c_float]
    L.orc_tonemap.restype = ctypes.c_uint8
    return np.array([L.orc_tonemap(float(v)) for v in values], np.uint8)


def test_tonemap_matches_restatement():
    rng = np.random.default_rng(11)
    rgb = np.concatenate([rng.exponential(0.7, (4000, 3)), [[0, 0, 0], [1e-8, 1e-3, 1], [50, 1e6, np.inf]],
                          np.linspace(0, 6, 999).reshape(-1, 3)]).astype(np.float32)
    got = api.film_tonemap(rgb)
    want = oracle_tonemap(rgb.ravel()).reshape(-1, 3)[:, ::-1]          # file order is b, g, r
    assert np.array_equal(got, want)
    assert got[4000].tolist() == [0, 0, 0] and got[4002].tolist() == [255, 255, 255]
    # where the reference is undefined (uint8 cast of a negative / NaN float) the engine writes 0
    assert api.film_tonemap(np.array([[-1.0, np.nan, -0.5]], np.float32)).tolist() == [[0, 0, 0]]


@pytest.mark.parametrize("width,height", [(1, 1), (2, 3), (3, 2), (4, 4), (5, 7), (33, 9)])
def test_bmp_layout(tmp_path, width, height):
    rng = np.random.default_rng(width * 100 + height)
    rgb = rng.exponential(0.5, (height, width, 3)).astype(np.float32)
    path = str(tmp_path / "out.bmp")
    api.film_write_bmp(path, rgb, width, height)
    raw = open(path, "rb").read()
    row = ((24 * width + 31) // 32) * 4
    assert len(raw) == 54 + row * height
    magic, size, r1, r2, offset = struct.unpack_from("<2sIHHI", raw, 0)
    assert (magic, size, r1, r2, offset) == (b"BM", len(raw), 0, 0, 54)
    (hsize, w, h, planes, bpp, comp, data, xres, yres, ncol, nimp) = struct.unpack_from("<IiiHHIIiiII", raw, 14)
    assert (hsize, w, h, planes, bpp, comp, data, xres, yres, ncol, nimp) == (40, width, height, 1, 24, 0, row * height, 0, 0, 0, 0)
    body = np.frombuffer(raw[54:], np.uint8).reshape(height, row)
    pixels = body[:, :width * 3].reshape(height, width, 3)
    assert np.array_equal(pixels[::-1], api.film_tonemap(rgb).reshape(height, width, 3))     # rows bottom-up
    assert not body[:, width * 3:].any()                                                      # zero padding


def test_bad_arguments():
    with pytest.raises(api.YartError):
        api.film_write_bmp("/nonexistent_dir/x.bmp", np.zeros((1, 1, 3), np.float32), 1, 1)
    assert api.lib().yart_film_write_bmp(b"x.bmp", None, 4, 4) == 1        # YART_ERR_INVALID_ARGUMENT
