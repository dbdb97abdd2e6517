"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            text = open(os.path.join(ROOT, "include", fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names += re.findall(r"\b(stx_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ["stx_warp_roi", "stx_warp", "stx_blend_create", "stx_blend_feed", "stx_blend_finish",
                 "stx_buf_to_host", "stx_buf_free", "stx_last_error"]:
        assert must in names


def test_library_exports_every_declared_symbol():
    from stitching_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libstitching_amd.so not built (run __graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(L, n)]
    assert not missing, f"declared in include/ but not exported: {missing}"
    assert sorted(_lib.EXPORTS) == declared_functions()


def test_version_and_error_string():
    from stitching_amd import _lib

    L = _lib.lib()
    assert L.stx_version() == 100
    assert isinstance(L.stx_last_error(), bytes)


def test_no_gpu_fails_loudly_not_silently():
    """Without a GPU the product must raise, never fall back to a CPU path."""
    import stitching_amd as S

    n = ctypes.c_int(-1)
    rc = S._lib.lib().stx_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(S.StitchingError):
        S.Context(0)


def test_result_roi_is_host_only():
    import stitching_amd as S

    assert S.Blender.result_roi([(0, 0), (-5, 7)], [(10, 10), (3, 4)]) == (-5, 0, 15, 11)
    with pytest.raises(S.StitchingError):
        S.Blender.result_roi([], [])


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: no file of the package may reference it."""
    pkg = os.path.join(ROOT, "stitching_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "from oracle" not in text and "stx_oracle" not in text, fn
