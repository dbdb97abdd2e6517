"""ctypes loader for libstitching_amd.so (C ABI: include/stitching_amd.h).

There is NO CPU fallback: if the HIP library is missing or cannot be loaded the import of
the hot path fails loudly.  Nothing here (or anywhere in stitching_amd) touches oracle/.
"""
import ctypes as C
import os

from .stitching_error import StitchingError

_HERE = os.path.dirname(os.path.abspath(__file__))
# STITCHING_AMD_LIB: another build of the same library (A/B runs of kernel variants on one GPU box: tools/gpu_ab_lib.sh)
LIB_PATH = os.environ.get("STITCHING_AMD_LIB") or os.path.join(_HERE, "libstitching_amd.so")

# constants of include/stitching_amd.h
STX_OK = 0
WARP_PLANE, WARP_AFFINE, WARP_CYLINDRICAL, WARP_SPHERICAL = 0, 1, 2, 3
# cv.PyRotationWarper(name, scale): name -> STX_WARP_* id
WARP_TYPE_IDS = {"plane": 0, "affine": 1, "cylindrical": 2, "spherical": 3, "fisheye": 4, "stereographic": 5,
                 "compressedPlaneA2B1": 6, "compressedPlaneA1.5B1": 7, "compressedPlanePortraitA2B1": 8,
                 "compressedPlanePortraitA1.5B1": 9, "paniniA2B1": 10, "paniniA1.5B1": 11, "paniniPortraitA2B1": 12,
                 "paniniPortraitA1.5B1": 13, "mercator": 14, "transverseMercator": 15}
INTER_NEAREST, INTER_LINEAR = 0, 1
BORDER_CONSTANT, BORDER_REFLECT = 0, 2
BLEND_NO, BLEND_FEATHER, BLEND_MULTIBAND = 0, 1, 2
U8, S16, F32 = 0, 1, 2
TRIG_MODES = {"exact": 0, "glibc": 1, "glibc-nofma": 2}
REMAP_MODES = {"q15": 0, "float": 1, "float-fma": 2}
PYRDOWN_MODES = {"scalar": 0, "simd-v": 1, "simd-hv": 3, "simd-v-fma": 5, "simd-hv-fma": 7}
GAIN_MAP_BOUNDED = 1
CONTRIB_U8_BINARY = 1
STRIP_MASK_BITS = 2

EXPORTS = (
    "stx_version stx_last_error stx_set_trig_mode stx_get_trig_mode stx_set_remap_mode stx_get_remap_mode stx_set_pyrdown_mode stx_get_pyrdown_mode stx_device_count stx_ctx_create stx_ctx_destroy stx_ctx_sync "
    "stx_host_alloc stx_host_free stx_buf_from_host stx_buf_from_host_async stx_buf_alloc stx_buf_to_host stx_buf_to_host_async stx_buf_view stx_buf_info stx_buf_device_ptr stx_buf_free "
    "stx_warp_roi stx_warp_rois stx_warp stx_warp_image_and_mask stx_warp_batch stx_warp_batch_rects stx_warp_batch_gain stx_warp_batch_with_rois stx_warp_mask "
    "stx_gain_apply stx_block_gain_apply stx_block_gain_apply_batch stx_resize_linear_exact stx_seam_mask_resize stx_seam_mask_resize_batch stx_seam_mask_resize_batch_sub stx_timelapse_frame stx_result_roi stx_blend_create stx_blend_num_bands stx_blend_feed stx_blend_finish stx_blend_finish_ex "
    "stx_blend_destroy stx_blend_set_band stx_blend_feed_ex stx_blend_contrib_rect stx_blend_export_contrib stx_blend_export_contribs "
    "stx_blend_build stx_blend_feed_contrib stx_blend_feed_contrib_ex stx_buf_flags stx_strip_rect stx_view_rect stx_strip_pack stx_strip_pack_batch stx_strip_pack_batch_ex stx_strip_bytes stx_strip_unpack stx_blend_feed_strips stx_comm_unique_id stx_comm_create stx_comm_exchange stx_comm_exchange_begin stx_comm_exchange_end stx_comm_exchange_begin_on stx_comm_exchange_end_on stx_comm_info stx_comm_destroy stx_prof_enable stx_prof_reset stx_prof_count stx_prof_get stx_mark stx_mark_elapsed_ms "
    "stx_debug_warp_maps stx_debug_feather_dist_cap"  # include/stitching_amd_debug.h: test hooks, never called by the package's classes
).split()

_lib = None


def build_hint():
    return "build it with `make -C stitching_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`"


def lib():
    """Load the shared library once and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: the HIP back end is not built ({build_hint()}). "
                          "stitching_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, vpp = C.c_void_p, C.POINTER(C.c_void_p)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.stx_version.restype = C.c_int
    L.stx_last_error.restype = C.c_char_p
    L.stx_set_trig_mode.argtypes = [C.c_int]
    L.stx_get_trig_mode.restype = C.c_int
    L.stx_set_remap_mode.argtypes = [C.c_int]
    L.stx_get_remap_mode.restype = C.c_int
    L.stx_set_pyrdown_mode.argtypes = [C.c_int, C.c_int]
    L.stx_get_pyrdown_mode.argtypes = [C.POINTER(C.c_int)]
    L.stx_get_pyrdown_mode.restype = C.c_int
    L.stx_device_count.argtypes = [ip]
    L.stx_ctx_create.argtypes = [C.c_int, vpp]
    L.stx_ctx_destroy.argtypes = [vp]
    L.stx_ctx_sync.argtypes = [vp]
    L.stx_buf_from_host.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, vpp]
    L.stx_buf_from_host_async.argtypes = L.stx_buf_from_host.argtypes
    L.stx_buf_alloc.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vpp]
    L.stx_host_alloc.argtypes = [C.c_size_t, vpp]
    L.stx_host_free.argtypes = [vp]
    L.stx_buf_to_host.argtypes = [vp, vp, C.c_size_t]
    L.stx_buf_to_host_async.argtypes = [vp, vp, C.c_size_t]
    L.stx_buf_view.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vpp]
    L.stx_buf_info.argtypes = [vp, C.POINTER(C.c_int64)]
    L.stx_buf_device_ptr.argtypes = [vp, vpp]
    L.stx_buf_free.argtypes = [vp]
    L.stx_warp_roi.argtypes = [vp, C.c_int, C.c_float, fp, fp, C.c_int, C.c_int, ip]
    L.stx_warp_rois.argtypes = [vp, C.c_int, C.c_float, C.c_int, fp, fp, ip, ip]
    L.stx_warp.argtypes = [vp, C.c_int, C.c_float, fp, fp, vp, C.c_int, C.c_int, vpp, ip]
    L.stx_warp_image_and_mask.argtypes = [vp, C.c_int, C.c_float, fp, fp, vp, vpp, vpp, ip]
    L.stx_warp_batch.argtypes = [vp, C.c_int, C.c_float, C.c_int, fp, fp, vpp, vpp, vpp, ip]
    L.stx_warp_batch_rects.argtypes = [vp, C.c_int, C.c_float, C.c_int, fp, fp, vpp, ip, vpp, vpp]
    L.stx_warp_batch_gain.argtypes = [vp, C.c_int, C.c_float, C.c_int, fp, fp, vpp, ip, vpp, ip, vpp, vpp, ip]
    L.stx_warp_batch_with_rois.argtypes = [vp, C.c_int, C.c_float, C.c_int, fp, fp, vpp, vpp, ip, vpp, vpp, ip]
    L.stx_warp_mask.argtypes = [vp, C.c_int, C.c_float, fp, fp, C.c_int, C.c_int, vpp, ip]
    L.stx_gain_apply.argtypes = [vp, vp, fp]
    L.stx_block_gain_apply.argtypes = [vp, vp, vp]
    L.stx_block_gain_apply_batch.argtypes = [vp, C.c_int, vpp, vpp, ip, ip]
    L.stx_resize_linear_exact.argtypes = [vp, vp, C.c_int, C.c_int, vpp]
    L.stx_seam_mask_resize.argtypes = [vp, vp, vp, vpp]
    L.stx_seam_mask_resize_batch.argtypes = [vp, C.c_int, vpp, vpp, vpp]
    L.stx_seam_mask_resize_batch_sub.argtypes = [vp, C.c_int, vpp, vpp, ip, vpp]
    L.stx_timelapse_frame.argtypes = [vp, vp, C.c_int, C.c_int, ip, vpp]
    L.stx_result_roi.argtypes = [C.c_int, ip, ip, ip]
    L.stx_blend_create.argtypes = [vp, C.c_int, C.c_int, C.c_float, ip, vpp]
    L.stx_blend_num_bands.argtypes = [vp, ip]
    L.stx_blend_feed.argtypes = [vp, vp, vp, C.c_int, C.c_int]
    L.stx_blend_finish.argtypes = [vp, vpp, vpp]
    L.stx_blend_finish_ex.argtypes = [vp, vpp, vpp, vpp]
    L.stx_blend_destroy.argtypes = [vp]
    L.stx_blend_set_band.argtypes = [vp, C.c_int, C.c_int]
    L.stx_blend_feed_ex.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int]
    L.stx_blend_contrib_rect.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip,
                                         C.POINTER(C.c_size_t)]
    L.stx_blend_export_contrib.argtypes = [vp, C.c_int, C.c_int, C.c_int, vpp, ip]
    L.stx_blend_export_contribs.argtypes = [vp, C.c_int, ip, ip, ip, vpp, ip]
    L.stx_blend_build.argtypes = [vp]
    L.stx_blend_feed_contrib.argtypes = [vp, C.c_int, ip, vp]
    L.stx_blend_feed_contrib_ex.argtypes = [vp, C.c_int, ip, vp, C.c_int]
    L.stx_buf_flags.argtypes = [vp, ip]
    L.stx_strip_rect.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip, C.POINTER(C.c_size_t)]
    L.stx_view_rect.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip]
    L.stx_strip_pack.argtypes = [vp, vp, vp, C.c_int, C.c_int, vpp]
    L.stx_strip_unpack.argtypes = [vp, C.c_int, C.c_int, C.c_int, vpp, vpp]
    L.stx_strip_pack_batch.argtypes = [vp, C.c_int, vpp, vpp, ip, ip, vpp]
    L.stx_strip_pack_batch_ex.argtypes = [vp, C.c_int, vpp, vpp, ip, ip, C.c_int, vpp]
    L.stx_strip_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]
    L.stx_blend_feed_strips.argtypes = [vp, C.c_int, vpp, ip, ip, ip, ip, ip, C.c_int]
    L.stx_comm_unique_id.argtypes = [C.POINTER(C.c_ubyte)]
    L.stx_comm_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_ubyte), vpp]
    L.stx_comm_exchange.argtypes = [vp, C.c_int, ip, ip, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.stx_comm_exchange_begin.argtypes = [vp, C.c_int, ip, ip, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.stx_comm_exchange_end.argtypes = [vp]
    L.stx_comm_exchange_begin_on.argtypes = [vp, vp, C.c_int, ip, ip, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.stx_comm_exchange_end_on.argtypes = [vp, vp]
    L.stx_comm_info.argtypes = [vp, ip]
    L.stx_comm_destroy.argtypes = [vp]
    L.stx_prof_enable.argtypes = [vp, C.c_int]
    L.stx_prof_reset.argtypes = [vp]
    L.stx_prof_count.argtypes = [vp, ip]
    L.stx_prof_get.argtypes = [vp, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                               C.POINTER(C.c_double)]
    L.stx_debug_warp_maps.argtypes = [vp, C.c_int, C.c_float, fp, fp, C.c_int, C.c_int, C.c_int, ip, vpp, vpp, ip]
    L.stx_mark.argtypes = [vp, C.c_int]
    L.stx_mark_elapsed_ms.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_double)]
    for name in EXPORTS:
        if name not in ("stx_version", "stx_last_error"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc):
    """C status -> reference error convention (StitchingError; stitching/stitching_error.py:1-2)."""
    if rc != STX_OK:
        msg = lib().stx_last_error()
        raise StitchingError(f"stitching_amd [{rc}]: {msg.decode(errors='replace') if msg else 'unknown error'}")
