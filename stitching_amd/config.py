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
