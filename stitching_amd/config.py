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
