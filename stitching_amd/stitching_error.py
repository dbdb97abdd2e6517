"""Error types of the reference (stitching/stitching_error.py:1-6), same names."""


class StitchingError(Exception):
    pass


class StitchingWarning(UserWarning):
    pass
