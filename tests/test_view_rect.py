"""Host geometry of the views a pipeline may cut out of a warped image (stx_view_rect, include/stitching_amd.h): no GPU.

The column range is stx_strip_rect's (the strips of the sharded blender, exercised bit for bit by the GPU tests); the row
range is the same computation along y.  Transposing the whole problem must therefore turn one into the other."""
import ctypes as C

import numpy as np
import pytest

from stitching_amd import _lib
from stitching_amd.distributed import make_shard_blender


def _view(b, size, corner, bx, by):
    r = (C.c_int * 4)()
    _lib.check(_lib.lib().stx_view_rect(b._h, size[0], size[1], corner[0], corner[1], bx[0], bx[1], by[0], by[1], r))
    return tuple(int(v) for v in r)


@pytest.mark.parametrize("seed", range(200))
def test_rows_are_the_columns_of_the_transposed_problem(seed):
    rng = np.random.default_rng(seed)
    bands = int(rng.integers(1, 7))
    rw, rh = int(rng.integers(1500, 9000)), int(rng.integers(1500, 9000))
    rx, ry = int(rng.integers(-3000, 3000)), int(rng.integers(-3000, 3000))
    w, h = min(int(rng.integers(200, 4000)), rw), min(int(rng.integers(200, 4000)), rh)  # images lie inside the roi
    tlx, tly = rx + int(rng.integers(0, max(1, rw - w))), ry + int(rng.integers(0, max(1, rh - h)))
    align = max(8, 1 << bands)

    def band(lo, hi, n):
        a = int(rng.integers(lo, hi - 1))
        b = int(rng.integers(a + 1, hi))
        return max((a // align) * align, 0), min(-((-b) // align) * align, ((n + align - 1) // align) * align)

    bx = band(tlx - rx, tlx - rx + w, rw)
    by = band(tly - ry, tly - ry + h, rh)
    b = make_shard_blender(None, (rx, ry, rw, rh), bands)
    bt = make_shard_blender(None, (ry, rx, rh, rw), bands)
    x0, x1, y0, y1 = _view(b, (w, h), (tlx, tly), bx, by)
    tx0, tx1, ty0, ty1 = _view(bt, (h, w), (tly, tlx), by, bx)
    assert x1 > x0 and y1 > y0
    assert x0 % 8 == 0 and y0 % 2 == 0
    # the transposed problem's columns are this problem's rows up to the granularity they are cut with (8 columns, 2 rows)
    assert (y0 // 8) * 8 == tx0 and min(h, -((-y1) // 8) * 8) == tx1, ((x0, x1, y0, y1), (tx0, tx1, ty0, ty1))
    assert (ty0 // 8) * 8 == x0 and min(w, -((-ty1) // 8) * 8) == x1
    # the view holds the band's own pixels and the pyramids' reach around them (3 * 2^B + 2^B), unless the image ends first
    reach = 4 << bands
    assert x0 <= max(0, bx[0] - (tlx - rx) - reach) and x1 >= min(w, bx[1] - (tlx - rx) + reach)
    assert y0 <= max(0, by[0] - (tly - ry) - reach) and y1 >= min(h, by[1] - (tly - ry) + reach)


def test_bands_that_miss_the_image_give_nothing():
    b = make_shard_blender(None, (0, 0, 8000, 4000), 5)
    assert _view(b, (1000, 800), (100, 100), (4000, 4992), (0, 992)) == (0, 0, 0, 0)
    assert _view(b, (1000, 800), (100, 100), (0, 992), (2016, 3008)) == (0, 0, 0, 0)
