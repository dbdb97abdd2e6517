"""`stitching_amd.Images`: the reference's surface (stitching/images.py) on the host, the
INTER_LINEAR_EXACT resize and the decode -> page-locked buffer -> queued upload staging on the device."""
import os

import numpy as np
import pytest

import stitching_amd as S
from stitching_amd import synthetic
from stitching_amd.images import Images, MegapixDownscaler, MegapixScaler


def test_megapix_scalers():
    s = MegapixScaler(0.6)
    s.set_scale_by_img_size((4000, 3000))
    assert abs(s.scale - np.sqrt(0.6e6 / 12e6)) < 1e-12 and s.get_scaled_img_size((4000, 3000)) == (894, 671)
    d = MegapixDownscaler(50)  # would enlarge: clamped to 1
    d.set_scale_by_img_size((4000, 3000))
    assert d.scale == 1.0
    f = MegapixDownscaler(-1)  # FINAL default: original size
    f.set_scale_by_img_size((4000, 3000))
    assert f.scale == 1.0 and f.get_scaled_img_size((4000, 3000)) == (4000, 3000)


def test_images_of_arrays_host_logic():
    imgs = [synthetic.make_frame(i, 400, 300) for i in range(3)]
    im = Images.of(imgs, medium_megapix=0.06, low_megapix=0.01)
    assert im.sizes == [(400, 300)] * 3 and im.names == ["1", "2", "3"]
    assert im.get_scaled_img_sizes(Images.Resolution.FINAL) == [(400, 300)] * 3
    assert im.get_scaled_img_sizes(Images.Resolution.LOW) == [(115, 87)] * 3
    r = im.get_ratio(Images.Resolution.MEDIUM, Images.Resolution.LOW)
    assert abs(r - np.sqrt(0.01 / 0.06)) < 1e-12
    im.subset([2, 0])
    assert im.names == ["3", "1"] and list(im)[0] is imgs[2]
    # FINAL at the original size is the identity, without touching a device
    assert list(im.resize(Images.Resolution.FINAL))[1] is imgs[0]
    with pytest.raises(S.StitchingError):
        Images.of([])
    with pytest.raises(S.StitchingError):
        Images.of("a.jpg")
    with pytest.raises(S.StitchingError):
        Images.of([imgs[0]])
    with pytest.raises(S.StitchingError):
        Images.of(imgs, medium_megapix=0.01, low_megapix=0.06)
    with pytest.raises(S.StitchingError):
        Images.of([1, 2])


def test_to_binary_and_wildcards(tmp_path):
    g = np.array([[0, 1, 200]], np.uint8)
    assert Images.to_binary(g).tolist() == [[0, 255, 255]]
    c = np.zeros((1, 2, 3), np.uint8)
    c[0, 1] = (3, 0, 0)  # blue 3 -> gray (3 * 1868 + 8192) >> 14 = 0
    assert Images.to_binary(c).tolist() == [[0, 0]]
    for n in ("a.png", "b.png"):
        (tmp_path / n).write_bytes(b"x")
    os.mkdir(tmp_path / "d.png")
    assert sorted(os.path.basename(p) for p in Images.resolve_wildcards([str(tmp_path / "*.png")])) == ["a.png", "b.png"]
    assert Images.resolve_wildcards(["x", "y"]) == ["x", "y"]


def _write_pngs(tmp_path, frames):
    from PIL import Image

    names = []
    for i, f in enumerate(frames):
        p = str(tmp_path / f"f{i}.png")
        Image.fromarray(np.ascontiguousarray(f[:, :, ::-1])).save(p)  # BGR -> RGB on disk
        names.append(p)
    return names


def test_filename_images_decode(tmp_path):
    pytest.importorskip("PIL")
    frames = [synthetic.make_frame(i, 160, 120) for i in range(3)]
    names = _write_pngs(tmp_path, frames)
    im = Images.of(names)
    got = list(im)
    assert all(np.array_equal(a, b) for a, b in zip(got, frames))  # lossless file: cv.imread's BGR layout
    assert im.sizes == [(160, 120)] * 3 and im.names == names
    with pytest.raises(S.StitchingError):
        Images.read_image(str(tmp_path / "missing.png"))
    with pytest.raises(S.StitchingError):
        Images.of([names[0]])


def test_to_binary_zero_set():
    """Images.to_binary = cvtColor(BGR2GRAY) > 0.5: exactly seven BGR triples are black after the threshold — with the 15-bit
    coefficients of OpenCV 4.x / 5.x and with the 14-bit ones of 2.x / 3.x alike (the grey values themselves differ at 43 864
    triples, never across the 0 / 1 boundary)."""
    b, g, r = np.meshgrid(np.arange(8, dtype=np.uint8), np.arange(8, dtype=np.uint8), np.arange(8, dtype=np.uint8), indexing="ij")
    img = np.stack([b.ravel(), g.ravel(), r.ravel()], axis=1).reshape(1, -1, 3)
    out = Images.to_binary(img)
    zero = {tuple(int(v) for v in img[0, i]) for i in np.flatnonzero(out[0] == 0)}
    assert zero == {(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0), (4, 0, 0), (0, 0, 1), (1, 0, 1)}
    y14 = (r.astype(np.uint32) * 4899 + g.astype(np.uint32) * 9617 + b.astype(np.uint32) * 1868 + 8192) >> 14
    assert np.array_equal(out.reshape(b.shape) > 0, y14 > 0)
    assert np.array_equal(Images.to_binary(np.array([[0, 1, 200]], np.uint8)), [[0, 255, 255]])


def test_decode_applies_the_exif_orientation(tmp_path):
    """cv.imread applies the EXIF orientation tag; so does the Pillow fallback (a portrait phone JPEG keeps its swapped
    width and height: sizes, scales and camera estimates depend on it)."""
    Image = pytest.importorskip("PIL.Image")
    from stitching_amd.images import _decode

    a = np.zeros((40, 64, 3), np.uint8)
    a[:20, :32] = (255, 0, 0)  # a red top-left quadrant (RGB)
    plain, tagged = str(tmp_path / "plain.png"), str(tmp_path / "rot.jpg")
    Image.fromarray(a).save(plain)
    exif = Image.Exif()
    exif[0x0112] = 6  # the stored pixels must be turned by 90 degrees clockwise for display
    Image.fromarray(a).save(tagged, quality=100, subsampling=0, exif=exif)
    p = _decode(plain)
    assert p.shape == (40, 64, 3) and tuple(p[5, 5]) == (0, 0, 255)  # BGR
    t = _decode(tagged)
    assert t.shape == (64, 40, 3), "orientation 6 swaps width and height"
    # the red quadrant ends up top-right
    assert t[5, 35, 2] > 200 and t[5, 35, 0] < 60 and t[5, 5, 2] < 60 and t[40, 35, 2] < 60
    imgs = Images.of([tagged, tagged])
    assert [Images.get_image_size(i) for i in imgs] == [(40, 64)] * 2


@pytest.mark.gpu
def test_resize_generator_matches_oracle(oracle, gpu_ctx):
    frames = [synthetic.make_frame(i, 803, 601) for i in range(3)]
    im = Images.of(frames, medium_megapix=0.2, low_megapix=0.05, final_megapix=0.3)
    for res in (Images.Resolution.MEDIUM, Images.Resolution.LOW, Images.Resolution.FINAL):
        sizes = im.get_scaled_img_sizes(res)
        for out, f, sz in zip(im.resize(res), frames, sizes):
            assert isinstance(out, np.ndarray) and np.array_equal(out, oracle.resize_linear_exact(f, sz))
    S.set_device_resident(True)
    try:
        out = list(im.resize(Images.Resolution.LOW))
    finally:
        S.set_device_resident(False)
    assert all(isinstance(o, S.DeviceImage) for o in out)
    assert np.array_equal(np.asarray(out[2]), oracle.resize_linear_exact(frames[2], im.get_scaled_img_sizes(Images.Resolution.LOW)[2]))


@pytest.mark.gpu
@pytest.mark.parametrize("final_megapix", [-1, 0.25])
def test_staged_frames_equal_the_synchronous_path(oracle, gpu_ctx, tmp_path, final_megapix):
    pytest.importorskip("PIL")
    frames = [synthetic.make_frame(i, 640 + 16 * (i % 2), 480) for i in range(7)]  # two different shapes
    names = _write_pngs(tmp_path, frames)
    staged = Images.of(names, final_megapix=final_megapix)
    dev = list(staged.stage(depth=2, workers=3))
    assert all(isinstance(d, S.DeviceImage) for d in dev) and staged.sizes == [Images.get_image_size(f) for f in frames]
    sync = Images.of(names, final_megapix=final_megapix)
    ref = list(sync.resize(Images.Resolution.FINAL))
    for d, r, f, sz in zip(dev, ref, frames, sync.get_scaled_img_sizes(Images.Resolution.FINAL)):
        assert np.array_equal(np.asarray(d), r)
        assert np.array_equal(r, oracle.resize_linear_exact(f, sz) if sz != Images.get_image_size(f) else f)
    # ... and they feed the warper directly
    cams = synthetic.ring_cameras(7, 640, 480, span_deg=200.0)
    w = S.Warper("spherical")
    w.set_scale(cams)
    if final_megapix < 0:
        assert np.array_equal(np.asarray(w.warp_image(dev[0], cams[0])), np.asarray(w.warp_image(frames[0], cams[0])))
