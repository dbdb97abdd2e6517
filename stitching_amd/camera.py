"""CameraParams stand-in for cv.detail.CameraParams.

The hot path needs only `.focal`, `.R` (3x3 float32) and `.K()` (stitching/warper.py:36,48,86);
any object with those attributes works, including real cv.detail.CameraParams.
"""
import numpy as np


class CameraParams:
    def __init__(self, focal=1.0, aspect=1.0, ppx=0.0, ppy=0.0, R=None, t=None):
        self.focal = float(focal)
        self.aspect = float(aspect)
        self.ppx = float(ppx)
        self.ppy = float(ppy)
        self.R = np.eye(3, dtype=np.float32) if R is None else np.asarray(R)
        self.t = np.zeros((3, 1), np.float64) if t is None else np.asarray(t, np.float64)

    def K(self):
        """cv::detail::CameraParams::K(): float64 3x3 intrinsics."""
        k = np.eye(3, dtype=np.float64)
        k[0, 0] = self.focal
        k[0, 2] = self.ppx
        k[1, 1] = self.focal * self.aspect
        k[1, 2] = self.ppy
        return k
