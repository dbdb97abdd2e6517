"""The level-0 gather with per-lane deferral (round 4): u8 images whose masks are not all 0 / 255 — the reference's default pipeline,
where SeamFinder.resize (stitching/seam_finder.py:37-43) leaves grey bytes along the seams — run the packed kernel on every lane and
queue the 8 x 2 patches that lie under a grey mask byte for a second, fp32-weight launch (csrc/stx_blend_fast.hip:
mb_level0_pk_kernel<.., DEFER>, mb_level0_deferred_kernel).  Every panorama must equal the oracle's bit for bit whatever the share of
queued patches: none, single bytes at lane / wavefront / tile borders, thin seams, everything; with the int16 result requested; and it
must equal what the wave-level kernel of round 3 produces (STITCHING_AMD_NO_DEFER=1, checked in a fresh interpreter)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import stitching_amd as S
from tests import helpers

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def grey_masks(kind, seed):
    rng = np.random.default_rng(seed)

    def fn(masks, corners, sizes):
        out = []
        for k, m in enumerate(masks):
            hh, ww = m.shape
            z = m.copy()
            if kind == "dense":  # every byte inside the warped mask grey: every patch is queued
                z = np.where(m > 0, rng.integers(1, 255, m.shape, dtype=np.uint8), 0).astype(np.uint8)
            elif kind == "single_bytes":  # lone grey bytes at the first / last pixel of lanes, wavefronts (512) and rows
                for (y, x) in ((0, 0), (1, 7), (2, 8), (hh // 2, 511), (hh // 2 + 1, 512), (hh - 1, ww - 1), (hh // 3, ww // 2), (hh - 2, 1)):
                    if m[y % hh, x % ww]:
                        z[y % hh, x % ww] = int(rng.integers(1, 255))
            elif kind == "seams":  # an 11-pixel ramp down the middle of every image, as a resized seam mask has
                x0 = ww // 2 + 13 * k
                ramp = np.linspace(255, 0, 11).astype(np.uint8)
                z[:, x0:x0 + 11] = np.minimum(z[:, x0:x0 + 11], ramp[None, :])
                z[:, x0 + 11:] = 0
            elif kind == "rows":  # grey horizontal lines: whole wavefronts of queued lanes next to untouched ones
                z[hh // 4, :] = np.minimum(z[hh // 4, :], 77)
                z[hh // 2:hh // 2 + 3, :] = np.minimum(z[hh // 2:hh // 2 + 3, :], 200)
            out.append(z)
        return out

    return fn


@pytest.mark.parametrize("kind", ["dense", "single_bytes", "seams", "rows"])
@pytest.mark.parametrize("strength", [6, 25])
def test_grey_masks_bit_exact(oracle, gpu_ctx, kind, strength):
    imgs, cams = helpers.small_ring(4, 1300, 410, span=140.0)
    o = helpers.run_pipeline(oracle.Warper, oracle.Blender, imgs, cams, blend_strength=strength, masks_fn=grey_masks(kind, 5))
    g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams, blend_strength=strength, masks_fn=grey_masks(kind, 5))
    assert g["blender"].blender.num_bands() == o["blender"].blender.num_bands() >= 3
    assert np.array_equal(g["pmask"], o["pmask"])
    assert np.array_equal(g["pano"], o["pano"]), int(np.count_nonzero(g["pano"] != o["pano"]))


def test_grey_masks_int16_result_and_device_resident_feed(oracle, gpu_ctx):
    """blender.blend()'s int16 (stitching/blender.py:46, before convertScaleAbs) through both launches of the deferral"""
    imgs, cams = helpers.small_ring(3, 1100, 380, span=110.0)
    ow = oracle.Warper("spherical")
    ow.set_scale(cams)
    sizes = [(1100, 380)] * 3
    wi = [ow.warp_image(i, c) for i, c in zip(imgs, cams)]
    wm = grey_masks("seams", 9)([ow.create_and_warp_mask(s, c) for s, c in zip(sizes, cams)], None, None)
    corners, wsz = ow.warp_rois(sizes, cams)
    ob, gb = oracle.Blender("multiband", 12), S.Blender("multiband", 12)
    ob.prepare(corners, wsz)
    gb.prepare(corners, wsz)
    for a, m, c in zip(wi, wm, corners):
        ob.blender.feed(a.astype(np.int16), m, c)
        gb.feed(S.DeviceImage.from_numpy(a, gpu_ctx), S.DeviceImage.from_numpy(m, gpu_ctx), c)  # u8 images: the packed path
    o16, omask = ob.blender.blend()
    pano, mask, p16 = gb.blender.blend(want_s16=True)
    assert np.array_equal(np.asarray(mask), omask)
    assert np.array_equal(np.asarray(p16), o16)
    assert np.array_equal(np.asarray(pano), oracle.convert_scale_abs(o16))


def test_deferral_equals_the_wave_level_kernel(gpu_ctx):
    """The same panorama with and without the deferral (a fresh interpreter per setting: the switch is read once)."""
    code = ("import hashlib, numpy as np, stitching_amd as S; from tests import helpers; from tests.test_gpu_defer import grey_masks\n"
            "imgs, cams = helpers.small_ring(4, 1300, 410, span=140.0)\n"
            "g = helpers.run_pipeline(S.Warper, S.Blender, imgs, cams, blend_strength=8, masks_fn=grey_masks('seams', 3))\n"
            "print(hashlib.sha256(g['pano'].tobytes() + g['pmask'].tobytes()).hexdigest())")
    outs = []
    for env in ({}, {"STITCHING_AMD_NO_DEFER": "1"}):
        e = dict(os.environ, **env)
        e.pop("STITCHING_AMD_NO_DEFER", None) if not env else None
        outs.append(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True, cwd=ROOT).stdout.split()[-1])
    assert outs[0] == outs[1] and len(outs[0]) == 64
