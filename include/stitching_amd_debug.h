/* stitching_amd_debug.h — TEST HOOKS of libstitching_amd.so.  Not part of the drop-in surface (include/stitching_amd.h): nothing in
 * stitching/warper.py or stitching/blender.py has a counterpart for these, and the Python classes never call them.  They exist so that
 * tests can look at intermediate values the fused kernels never store.
 */
#ifndef STITCHING_AMD_DEBUG_H
#define STITCHING_AMD_DEBUG_H

#include "stitching_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The fp32 backward map of a warp — what cv.PyRotationWarper(type, scale).buildMaps(size, K, R) hands to cv.remap in
 * stitching/warper.py:44-51 (RotationWarperBase::buildMaps: xmap, ymap = projector.mapBackward(u, v) for every pixel of the roi) — as
 * the DEVICE projector computes it: the warp kernels run up to the division x / z, y / z and store the two quotients instead of the
 * samples they select.  `which` = 1: the kernel stx_warp / stx_warp_batch would launch for this camera (the tuned kernel of the
 * spherical / cylindrical / plane / affine warpers: tabled trig, packed row pairs, shared-reciprocal division); 2: the generic
 * one-pixel-per-lane kernels (what the float remap modes run).  rect_xywh: NULL -> the warp roi (returned in out_xywh), else any
 * rectangle in warp coordinates.  Outputs: two f32x1 images of the rectangle's size.  The process-wide trig mode applies.
 * tests/test_gpu_maps.py compares them with the CPU checker's build_maps: 0 ULP. */
int stx_debug_warp_maps(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9], int w, int h, int which,
                        const int rect_xywh[4], stx_buf** out_xmap_f32, stx_buf** out_ymap_f32, int out_xywh[4]);

/* Saturation distance of the feather blender's distance transform (stitching/blender.py:34-36 -> FeatherBlender::createWeightMap ->
 * distanceTransform(DIST_L1, 3): OpenCV's 16.16 fixed point clamps at INT_MAX >> 2 = 8192.0f).  The sharded feather blender sizes the
 * halo of its strips by it (stitching_amd/distributed.py: FEATHER_DIST_CAP); tests/test_host_logic.py ties the two together. */
int stx_debug_feather_dist_cap(void);

#ifdef __cplusplus
}
#endif
#endif /* STITCHING_AMD_DEBUG_H */
