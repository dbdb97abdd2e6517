"""Exception types of the back end.

They carry the names the reference uses (stitching/stitching_error.py:1-6) so that user code written against
OpenStitching/stitching (`except StitchingError: ...`, `warnings.simplefilter("ignore", StitchingWarning)`) keeps
working when the Warper / Blender classes are swapped for the ones of this package.  Every non-zero status of the
C ABI (include/stitching_amd.h: STX_ERR_*) surfaces as a StitchingError carrying the library's message.
"""


class StitchingError(Exception):
    """A call could not be carried out: invalid arguments (the checks OpenCV does with CV_Assert), a call-order
    violation (feed after blend), an unsupported variant, or a HIP / allocation failure reported by the library."""


class StitchingWarning(UserWarning):
    """Non-fatal condition worth telling the user about (same role as in the reference)."""
