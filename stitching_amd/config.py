"""Back-end switches.  The reference's settings dict (stitching/stitcher.py:22-48) rejects
unknown keys, so the back end is configured here / by environment, never through Stitcher."""
import os

_device_resident = os.environ.get("STITCHING_AMD_DEVICE_RESIDENT", "0") not in ("0", "", "false", "False")


def set_device_resident(on=True):
    """True: Warper / Blender return DeviceImage objects (pixels stay in HBM, host copy only on
    demand).  False (default): they return numpy arrays, exactly like the reference."""
    global _device_resident
    _device_resident = bool(on)


def device_resident():
    return _device_resident


def set_trig_mode(mode):
    """Which sinf / cosf the projectors follow (include/stitching_amd.h STX_TRIG_*): "exact" (correctly rounded, default),
    "glibc" (glibc >= 2.28 on an x86-64-v3 host, bit for bit: what cv.PyRotationWarper gets from libm there,
    stitching/warper.py:44-51) or "glibc-nofma".  Process-wide, like the libm it stands for; STITCHING_AMD_TRIG sets the start-up
    value.  Returns the previous mode."""
    from . import _lib

    L = _lib.lib()
    names = {v: k for k, v in _lib.TRIG_MODES.items()}
    prev = names[L.stx_get_trig_mode()]
    if mode not in _lib.TRIG_MODES:
        from .stitching_error import StitchingError

        raise StitchingError(f"unknown trig mode {mode!r}: one of {sorted(_lib.TRIG_MODES)}")
    _lib.check(L.stx_set_trig_mode(_lib.TRIG_MODES[mode]))
    return prev


def trig_mode():
    from . import _lib

    return {v: k for k, v in _lib.TRIG_MODES.items()}[_lib.lib().stx_get_trig_mode()]


def set_remap_mode(mode):
    """Interpolation model of the image samples of a warp (include/stitching_amd.h STX_REMAP_*): "q15" (OpenCV 4.x's fixed-point
    remap, default), "float" or "float-fma" (fp32 bilinear on the unquantised position: a model of a cv.remap that interpolates in
    floating point, stitching/warper.py:46-51 with the pinned opencv-python 5.x — unverified).  Process-wide; STITCHING_AMD_REMAP sets
    the start-up value.  Returns the previous mode."""
    from . import _lib

    L = _lib.lib()
    names = {v: k for k, v in _lib.REMAP_MODES.items()}
    prev = names[L.stx_get_remap_mode()]
    if mode not in _lib.REMAP_MODES:
        from .stitching_error import StitchingError

        raise StitchingError(f"unknown remap mode {mode!r}: one of {sorted(_lib.REMAP_MODES)}")
    _lib.check(L.stx_set_remap_mode(_lib.REMAP_MODES[mode]))
    return prev


def remap_mode():
    from . import _lib

    return {v: k for k, v in _lib.REMAP_MODES.items()}[_lib.lib().stx_get_remap_mode()]


def set_pyrdown_mode(mode, lanes=4):
    """Order of the fp32 sums of the weight pyramids (include/stitching_amd.h STX_PYRDOWN_*): "scalar" (default: the tuned kernels) or
    the order of OpenCV's vector code — "simd-v", "simd-hv", "simd-v-fma", "simd-hv-fma" with `lanes` (4, 8, 16) floats per vector
    (cv.detail_MultiBandBlender.feed -> pyrDown(CV_32F), stitching/blender.py:40-41).  Models of OpenCV builds, unverified; any mode but
    the default builds the pyramids with the plain kernels.  Process-wide, read when a blender builds its pyramids; STITCHING_AMD_PYRDOWN
    ("simd-hv:8") sets the start-up value.  Returns the previous (mode, lanes)."""
    from . import _lib

    L = _lib.lib()
    prev = pyrdown_mode()
    if mode not in _lib.PYRDOWN_MODES:
        from .stitching_error import StitchingError

        raise StitchingError(f"unknown pyrDown mode {mode!r}: one of {sorted(_lib.PYRDOWN_MODES)}")
    _lib.check(L.stx_set_pyrdown_mode(_lib.PYRDOWN_MODES[mode], int(lanes)))
    return prev


def pyrdown_mode():
    """-> (mode name, lanes)"""
    import ctypes as C

    from . import _lib

    lanes = C.c_int()
    m = _lib.lib().stx_get_pyrdown_mode(C.byref(lanes))
    return {v: k for k, v in _lib.PYRDOWN_MODES.items()}[m], int(lanes.value)
