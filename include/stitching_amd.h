/* stitching_amd.h — C ABI of the MI355X (gfx950) warp + blend back end.
 *
 * Drop-in boundary for ONE hot path of OpenStitching/stitching: what
 * stitching/warper.py and stitching/blender.py reach through cv2.  Each entry point
 * below names the reference call site it replaces (file:line in /root/reference) and the
 * OpenCV routine behind it.  Plain C types only: the library is loaded with ctypes
 * (stitching_amd/_lib.py); INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - every function returns STX_OK (0) or a negative STX_ERR_* code; stx_last_error()
 *     gives the thread-local message.  The Python shim raises StitchingError
 *     (reference: stitching/stitching_error.py:1-2).
 *   - one stx_ctx per (process, GPU); it owns one HIP stream and a caching device
 *     allocator.  Calls on one ctx must be serialised by the caller (the reference is
 *     single threaded: stitching/stitcher.py:247-254).  All work is enqueued on the ctx
 *     stream; only stx_buf_to_host, stx_warp_roi(s), stx_ctx_sync and stx_prof_* wait.
 *   - inputs are borrowed for the duration of the call; outputs are library-owned
 *     opaque handles released with stx_buf_free / stx_blend_destroy / stx_ctx_destroy.
 *   - images are row-major HWC exactly as numpy/cv2 hand them over: u8x3 BGR images,
 *     u8x1 masks, optionally s16x3 (what stitching/blender.py:41 produces with astype).
 *   - K and R are 3x3 row-major fp32 (the reference casts: stitching/warper.py:86,
 *     stitching/camera_estimator.py:25-26).
 */
#ifndef STITCHING_AMD_H
#define STITCHING_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STX_VERSION 100 /* 0.1.0 */

/* status codes */
#define STX_OK 0
#define STX_ERR_INVALID (-1)     /* bad argument (mirrors CV_Assert failures)        */
#define STX_ERR_HIP (-2)         /* HIP runtime error                                 */
#define STX_ERR_OOM (-3)         /* device allocation failed                          */
#define STX_ERR_STATE (-4)       /* call order violated (e.g. feed after finish)      */
#define STX_ERR_UNSUPPORTED (-5) /* valid in the reference, not implemented here      */

/* trig modes.  stitching/warper.py:44-51 builds a cv.PyRotationWarper per call; its projectors evaluate sinf / cosf with the
 * HOST's libm, so the warp coordinates of the reference depend on the machine it runs on.  The back end offers:
 *   STX_TRIG_EXACT       correctly rounded sinf / cosf (default; equals any libm wherever that libm is correctly rounded);
 *   STX_TRIG_GLIBC       sinf / cosf as glibc >= 2.28 computes them on an x86-64-v3 host (__sinf_fma / __cosf_fma): bit-identical to
 *                        that libm on every float argument (tests/test_glibc_trig.py);
 *   STX_TRIG_GLIBC_NOFMA the same routines without fused multiply-add (__sinf_sse2 / __cosf_sse2).
 * The mode is process-wide, like the libm it stands for: STITCHING_AMD_TRIG = exact | glibc | glibc-nofma at first use, or
 * stx_set_trig_mode.  atan2f / acosf / tanf / ... (forward maps: warp_roi) stay correctly rounded in every mode. */
#define STX_TRIG_EXACT 0
#define STX_TRIG_GLIBC 1
#define STX_TRIG_GLIBC_NOFMA 2
int stx_set_trig_mode(int mode);
int stx_get_trig_mode(void);

/* remap modes.  stitching/warper.py:46-51 calls cv.remap(INTER_LINEAR, BORDER_REFLECT) on fp32 maps.  OpenCV 4.x quantises the
 * position to 1/32 pixel and blends the four taps with Q15 table weights (remapBilinear); the reference pins opencv-python 5.0.0.93
 * (requirements.txt:1), whose build is not available here — if it interpolates in fp32 on the unquantised position, the bytes differ by
 * 1-2 LSB on about a sixth of the pixels (profiles/r02_oracle_sensitivity.md).  The back end offers both arithmetic models:
 *   STX_REMAP_Q15        the classic fixed-point scheme (default; the tuned kernels);
 *   STX_REMAP_FLOAT      fp32 bilinear on the unquantised position: a = x - floor(x), t = a (p01 - p00) + p00, u likewise on the lower
 *                        row, cvRound(b (u - t) + t), every step rounded to fp32 (multiply and add separate);
 *   STX_REMAP_FLOAT_FMA  the same with each multiply-add fused.
 * The float modes are a MODEL of that build (oracle/stx_oracle.cpp: bilinear_px_float), unverified against it like everything here
 * that restates OpenCV.  Since round 5 they run on the tuned batched kernel like the default: wavefronts whose samples all lie inside the
 * source blend in fp32 straight from the same two 12-byte windows per row (warp_fast_kernel<.., RM>, blend_float_to_lds), wavefronts that
 * touch a border go one pixel at a time (sample_float); 0.90 x the default's rate on config 2.  Only what the tuned kernel does not
 * take at all (sources beyond 32767 px or 2 GB, a nearest-neighbour source image) runs on the plain one-pixel-per-lane kernels, in any
 * mode.  Masks (INTER_NEAREST) are the same in every mode.
 * Process-wide: STITCHING_AMD_REMAP = q15 | float | float-fma at first use, or stx_set_remap_mode. */
#define STX_REMAP_Q15 0
#define STX_REMAP_FLOAT 1
#define STX_REMAP_FLOAT_FMA 2
int stx_set_remap_mode(int mode);
int stx_get_remap_mode(void);

/* pyrDown order of the fp32 weight pyramids.  stitching/blender.py:40-41 -> MultiBandBlender::feed -> pyrDown(CV_32F) per level.  The
 * 5-tap sums are integers for the image planes (any order gives the same bits) but fp32 for the weights, where OpenCV's scalar loop
 * and its SIMD code associate differently:
 *   scalar      row:  s2*6 + (s1+s3)*4 + s0 + s4                 column: the same expression
 *   SIMD        row:  s2*6 + ((s1+s3)*4 + (s0+s4))               column: (r1+r3+r2)*4 + (r0+r4+(r2+r2))
 * and a build with FMA contracts each multiply-add.  The vector code covers the outputs 1 .. 1 + ((width0-1)/L)*L of a row
 * (width0 = min((w-3)/2+1, dw), L = lanes per vector) and the first (dw/L)*L outputs of the column pass; the rest is the scalar loop.
 * With 0 / 255 masks the levels 1..3 are exact in every order; coarser levels, and grey masks from level 1 on, move by an ULP and the
 * panorama by at most 1 LSB at a few bytes (profiles/r02_oracle_sensitivity.md).
 *   STX_PYRDOWN_SCALAR (default: the tuned kernels)   STX_PYRDOWN_SIMD_V: column pass vectorised   STX_PYRDOWN_SIMD_HV: both
 *   | STX_PYRDOWN_FMA: fused multiply-adds in the vector code;   lanes = 4 (SSE / NEON), 8 (AVX2), 16 (AVX-512)
 * Like the float remap these are MODELS of OpenCV builds (the CPU checker holds the same ones), unverified here; any mode but the default
 * builds the pyramids with the plain one-sample-per-lane kernels.  Process-wide: STITCHING_AMD_PYRDOWN = scalar | simd-v | simd-hv |
 * simd-v-fma | simd-hv-fma, optionally ":lanes" (simd-hv:8), at first use, or stx_set_pyrdown_mode; read when a blender builds its pyramids. */
#define STX_PYRDOWN_SCALAR 0
#define STX_PYRDOWN_SIMD_V 1
#define STX_PYRDOWN_SIMD_HV 3
#define STX_PYRDOWN_FMA 4
int stx_set_pyrdown_mode(int mode, int lanes);
int stx_get_pyrdown_mode(int* out_lanes);

/* warper types: the names of Warper.WARP_TYPE_CHOICES (stitching/warper.py:10-27);
 * cv.PyRotationWarper(type, scale) string -> id in the Python shim */
#define STX_WARP_PLANE 0
#define STX_WARP_AFFINE 1
#define STX_WARP_CYLINDRICAL 2
#define STX_WARP_SPHERICAL 3
/* the other twelve names cv.PyRotationWarper accepts: generic RotationWarperBase<P> warpers (roi from every source
 * pixel, per-pixel projector), (a, b) fixed by the name as in PyRotationWarper's constructor */
#define STX_WARP_FISHEYE 4
#define STX_WARP_STEREOGRAPHIC 5
#define STX_WARP_COMPRESSED_PLANE_A2B1 6
#define STX_WARP_COMPRESSED_PLANE_A15B1 7
#define STX_WARP_COMPRESSED_PLANE_PORTRAIT_A2B1 8
#define STX_WARP_COMPRESSED_PLANE_PORTRAIT_A15B1 9
#define STX_WARP_PANINI_A2B1 10
#define STX_WARP_PANINI_A15B1 11
#define STX_WARP_PANINI_PORTRAIT_A2B1 12
#define STX_WARP_PANINI_PORTRAIT_A15B1 13
#define STX_WARP_MERCATOR 14
#define STX_WARP_TRANSVERSE_MERCATOR 15
#define STX_WARP_TYPE_COUNT 16

/* cv.INTER_* / cv.BORDER_* values used by stitching/warper.py:49-50,65-66 */
#define STX_INTER_NEAREST 0
#define STX_INTER_LINEAR 1
#define STX_BORDER_CONSTANT 0
#define STX_BORDER_REFLECT 2

/* blender kinds (stitching/blender.py:8-12) */
#define STX_BLEND_NO 0
#define STX_BLEND_FEATHER 1
#define STX_BLEND_MULTIBAND 2

/* element types of stx_buf */
#define STX_U8 0
#define STX_S16 1
#define STX_F32 2

typedef struct stx_ctx stx_ctx;
typedef struct stx_buf stx_buf;
typedef struct stx_blender stx_blender;

/* ---- library / context ------------------------------------------------------------ */
int stx_version(void);
const char* stx_last_error(void);
int stx_device_count(int* out_n);
int stx_ctx_create(int device, stx_ctx** out);
int stx_ctx_destroy(stx_ctx* ctx);
int stx_ctx_sync(stx_ctx* ctx);

/* ---- device images ------------------------------------------------------------------
 * Replaces the numpy.ndarray / cv.UMat values that cross the reference's cv2 boundary
 * (stitching/blender.py:41 cv.UMat(img.astype(np.int16))). */
int stx_buf_from_host(stx_ctx* ctx, const void* host, size_t host_stride_bytes, int w, int h, int channels, int elem,
                      stx_buf** out);
int stx_buf_alloc(stx_ctx* ctx, int w, int h, int channels, int elem, stx_buf** out);
/* page-locked host memory (for decoded frames / read-backs): stx_buf_from_host / stx_buf_to_host on such memory run at
 * PCIe rate instead of the pageable-copy rate (the cv.imread -> UMat staging in front of stitching/images.py:111) */
int stx_host_alloc(size_t bytes, void** out);
int stx_host_free(void* p);
int stx_buf_to_host(const stx_buf* buf, void* host, size_t host_stride_bytes);
/* asynchronous forms for page-locked host memory (stx_host_alloc): the copy is queued on the context's stream and the
 * call returns; the host memory must stay untouched (upload) / unread (read-back) until stx_ctx_sync(ctx).  With pageable
 * memory they behave like the synchronous calls.  Two contexts fed alternately keep both PCIe directions busy. */
int stx_buf_from_host_async(stx_ctx* ctx, const void* host, size_t host_stride_bytes, int w, int h, int channels, int elem,
                            stx_buf** out);
int stx_buf_to_host_async(const stx_buf* buf, void* host, size_t host_stride_bytes);
/* rectangular sub-view sharing the parent's memory (numpy slicing in stitching/cropper.py:150-151) */
int stx_buf_view(const stx_buf* buf, int x, int y, int w, int h, stx_buf** out);
/* info = {w, h, channels, elem, stride_bytes, device} */
int stx_buf_info(const stx_buf* buf, int64_t info[6]);
/* device address of the first pixel (for zero-copy consumers, e.g. RCCL strip exchange) */
int stx_buf_device_ptr(const stx_buf* buf, void** out);
int stx_buf_free(stx_buf* buf);

/* ---- Warper -------------------------------------------------------------------------
 * stx_warp_roi  <- stitching/warper.py:79-82  cv.PyRotationWarper(type, scale).warpRoi(size, K, R)
 *                  (RotationWarperBase::warpRoi / detectResultRoi[ByBorder],
 *                   SphericalWarper::detectResultRoi, PlaneWarper/AffineWarper::warpRoi)
 * out_xywh = (tl.x, tl.y, br.x-tl.x+1, br.y-tl.y+1). */
int stx_warp_roi(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9], int w, int h,
                 int out_xywh[4]);
/* batched form of the loop in stitching/warper.py:70-77 (one device pass, one sync) */
int stx_warp_rois(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s, const int* sizes_wh,
                  int* out_xywh);
/* stx_warp <- stitching/warper.py:43-52 (interp=LINEAR, border=REFLECT, src u8x3) and
 *             stitching/warper.py:58-68 (interp=NEAREST, border=CONSTANT, src u8x1):
 *             cv.PyRotationWarper(type, scale).warp(src, K, R, interp, border)
 *             = RotationWarperBase::buildMaps + cv::remap, fused: maps are never stored.
 * out_tl receives the corner the reference discards (stitching/warper.py:45 `_`). */
int stx_warp(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9], const stx_buf* src, int interp,
             int border, stx_buf** out, int out_tl[2]);
/* fused form of warper.py:43-52 + 58-68 for one camera: one pass computes the backward map
 * once and writes both the bilinear image and the nearest-neighbour 255-mask.
 * out_img / out_mask may each be NULL. */
int stx_warp_image_and_mask(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9],
                            const stx_buf* src, stx_buf** out_img, stx_buf** out_mask, int out_xywh[4]);
/* batched form of the generator loops stitching/warper.py:39-41 (warp_images) and :54-56
 * (create_and_warp_masks) for n cameras of one warper: ROIs in one device pass, then ONE table launch and
 * ONE remap launch for all images (the per-image argument blocks travel as kernel arguments: no upload,
 * no host synchronisation).  out_imgs / out_masks are arrays of n handles (either may be NULL). */
int stx_warp_batch(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s,
                   const stx_buf* const* srcs, stx_buf** out_imgs, stx_buf** out_masks, int* out_xywh);
/* stx_warp_batch over caller-given destination rectangles (x, y, w, h in warp coordinates, normally sub-rectangles of the
 * ROIs of stx_warp_rois): the pixels are exactly those of the full warp inside the rectangle.  For pipelines that know
 * which part of a warped image the blender can see (seam masks: stitching/stitcher.py:124 cuts every image down to its
 * seam cell AFTER warping all of it, :119-121). */
int stx_warp_batch_rects(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s,
                         const stx_buf* const* srcs, const int* rects_xywh, stx_buf** out_imgs, stx_buf** out_masks);
/* stx_warp_batch[_rects] + the block gains of the exposure compensator in one call: the loops stitching/stitcher.py:119-123 (warp final
 * images, then compensator.apply on each) for the "gain_blocks" compensator.  gain_maps[i]: f32x1 (f32x3: channel_blocks) map of image i,
 * laid over its WHOLE warped image (BlocksCompensator::apply) also when rects_xywh names a rectangle of it; gain_flags as in
 * stx_block_gain_apply_batch.  The product happens in the warp kernel's epilogue when it can (tuned kernel, bounded f32x1 maps), else as
 * a second pass: out_imgs equal stx_block_gain_apply_batch(stx_warp_batch[_rects](...)) byte for byte either way.  rects_xywh may be NULL
 * (whole ROIs, returned in out_xywh). */
int stx_warp_batch_gain(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s, const stx_buf* const* srcs,
                        const int* rects_xywh, const stx_buf* const* gain_maps_f32, const int* gain_flags, stx_buf** out_imgs,
                        stx_buf** out_masks, int* out_xywh);
/* One panorama's warps, its ROI pass included: stitching/stitcher.py:188 (warp_rois) + :119-123 in ONE call.  The ROIs are computed by this
 * call (never taken from the cache of earlier calls: a panorama pays for its own ROI pass) and returned in out_xywh; the host waits for them
 * once, inside the call, by polling stamps the ROI kernel writes into pinned memory, and launches the warps right behind — the device idles
 * for ~25 us at the head of a panorama instead of the ~145 us of stx_warp_rois + Python + stx_warp_batch (profiles/r05_latency.md).
 * gain_maps / gain_flags: as in stx_warp_batch_gain, or NULL.  Results equal stx_warp_rois + stx_warp_batch[_gain] byte for byte. */
int stx_warp_batch_with_rois(stx_ctx* ctx, int type, float scale, int n, const float* K9s, const float* R9s, const stx_buf* const* srcs,
                             const stx_buf* const* gain_maps_f32_or_null, const int* gain_flags_or_null, stx_buf** out_imgs,
                             stx_buf** out_masks, int* out_xywh);
/* stitching/warper.py:58-68 without allocating the 255-filled source (size only) */
int stx_warp_mask(stx_ctx* ctx, int type, float scale, const float K[9], const float R[9], int w, int h,
                  stx_buf** out_mask, int out_xywh[4]);

/* ---- Blender ------------------------------------------------------------------------
 * stx_result_roi    <- stitching/blender.py:24    cv.detail.resultRoi(corners, sizes)
 * stx_blend_create  <- stitching/blender.py:27-38 Blender_createDefault(NO) | detail_MultiBandBlender()
 *                      + setNumBands | detail_FeatherBlender() + setSharpness, then .prepare(dst_sz)
 *                      (Blender/FeatherBlender/MultiBandBlender::prepare)
 * stx_blend_feed    <- stitching/blender.py:40-41 blender.feed(UMat(int16 img), mask, corner)
 *                      img: u8x3 (converted on load) or s16x3; mask u8x1.  DEFERRED for all three kinds: the blender
 *                      keeps a reference to img and mask and reads them in stx_blend_finish (one gather over the
 *                      panorama), so the two buffers must not be modified between feed and finish (OpenCV consumes
 *                      them inside feed)
 * stx_blend_finish  <- stitching/blender.py:43-48 blender.blend() + cv.convertScaleAbs(result)
 *                      returns the u8x3 panorama and the u8 mask; consumes the blender
 *                      (a second finish is STX_ERR_STATE; OpenCV releases dst_ in blend). */
int stx_result_roi(int n, const int* corners_xy, const int* sizes_wh, int out_xywh[4]);
int stx_blend_create(stx_ctx* ctx, int kind, int num_bands, float sharpness, const int roi_xywh[4],
                     stx_blender** out);
/* band count after MultiBandBlender::prepare's clamp (0 for the other kinds) */
int stx_blend_num_bands(const stx_blender* b, int* out_num_bands);
int stx_blend_feed(stx_blender* b, const stx_buf* img, const stx_buf* mask, int tlx, int tly);
int stx_blend_finish(stx_blender* b, stx_buf** out_pano_u8, stx_buf** out_mask_u8);
/* as stx_blend_finish, but also hands out the int16 result that blender.blend() returns
 * before convertScaleAbs (stitching/blender.py:46); any out pointer may be NULL */
int stx_blend_finish_ex(stx_blender* b, stx_buf** out_pano_u8, stx_buf** out_mask_u8, stx_buf** out_pano_s16);
int stx_blend_destroy(stx_blender* b);

/* ---- "next" rows either side of the path -----------------------------------------------------
 * stx_gain_apply      <- stitching/exposure_error_compensator.py:43-45 compensator.apply(idx, corner, img, mask) for the
 *                        "gain" / "channel" compensators (GainCompensator::apply, ChannelsCompensator::apply =
 *                        cv::multiply(image, gain): fp32 product, cvRound, saturate to u8), in place on the warped image
 *                        between warp and feed (stitching/stitcher.py:123,219-221).  The block compensators
 *                        ("gain_blocks", "channel_blocks") are stx_block_gain_apply below.
 * stx_timelapse_frame <- stitching/timelapser.py:36-52 timelapser.process(img, mask, corner) + getDst():
 *                        zero frame of the roi given to initialize(), the image pasted at its corner. */
int stx_gain_apply(stx_ctx* ctx, stx_buf* img_u8x3, const float gains_bgr[3]);
/* the reference's default compensator "gain_blocks": BlocksCompensator::apply = cv::resize(gain map, image size,
 * INTER_LINEAR) [fp32] + cv::multiply, fused (the full-size gain map is never stored); gain_map is f32x1, or f32x3 (BGR
 * maps, interleaved) for "channel_blocks" (BlocksChannelsCompensator) */
int stx_block_gain_apply(stx_ctx* ctx, stx_buf* img_u8x3, const stx_buf* gain_map_f32);
/* the same for all n images of a panorama (the generator loop stitching/stitcher.py:219-221): two launches per 16 images, gain maps
 * resident on the device.  full_wh_xy0 (or NULL) = {full_w, full_h, x0, y0} per image: imgs[i] is the rectangle at (x0, y0) of a warped
 * image of size full_w x full_h (a seam-cell crop, stx_warp_batch_rects) and the gain map is laid over the FULL image, as
 * BlocksCompensator::apply does — the rectangle's bytes equal those of the whole image compensated and then cut.
 * flags (or NULL): STX_GAIN_MAP_BOUNDED per image — the caller has checked that every gain of the map is finite and below 2^31 / 255. */
#define STX_GAIN_MAP_BOUNDED 1
int stx_block_gain_apply_batch(stx_ctx* ctx, int n, stx_buf* const* imgs_u8x3, const stx_buf* const* gain_maps_f32,
                               const int* full_wh_xy0, const int* flags);
/* stx_resize_linear_exact <- stitching/images.py:122-124 cv.resize(img, size, interpolation=cv.INTER_LINEAR_EXACT) (u8x1 / u8x3:
 *                            the final-resolution resize of Images.resize, next row N3)
 * stx_seam_mask_resize    <- stitching/seam_finder.py:37-43 SeamFinder.resize: cv.dilate(seam_mask, None), cv.resize(...,
 *                            INTER_LINEAR_EXACT) to the size of the final warped mask, cv.bitwise_and with it (next row N2);
 *                            one fused kernel, the result is what Blender.feed receives (stitching/stitcher.py:124,127) */
int stx_resize_linear_exact(stx_ctx* ctx, const stx_buf* src_u8, int dst_w, int dst_h, stx_buf** out);
int stx_seam_mask_resize(stx_ctx* ctx, const stx_buf* seam_mask_u8x1, const stx_buf* final_mask_u8x1, stx_buf** out);
/* the same for all n images of a panorama (the generator loop stitching/stitcher.py:223-225): one table upload, one dilate
 * and one resize launch per 16 images */
int stx_seam_mask_resize_batch(stx_ctx* ctx, int n, const stx_buf* const* seam_masks, const stx_buf* const* final_masks,
                               stx_buf** outs);
/* the same for rectangles of the final masks: final_masks[i] is the w x h rectangle at (x0, y0) of image i's warped mask of
 * size full_w x full_h; full_wh_xy0 = {full_w, full_h, x0, y0} per image (x0 a multiple of 4).  Output i equals that
 * rectangle of stx_seam_mask_resize_batch's output for the whole mask. */
int stx_seam_mask_resize_batch_sub(stx_ctx* ctx, int n, const stx_buf* const* seam_masks, const stx_buf* const* final_masks,
                                   const int* full_wh_xy0, stx_buf** outs);
int stx_timelapse_frame(stx_ctx* ctx, const stx_buf* img, int tlx, int tly, const int dst_roi_xywh[4], stx_buf** out_frame);

/* ---- sharded multi-band blending: one process per GPU, one stx_blender per rank ------------------
 * No counterpart in the reference (it is single process, stitching/stitcher.py:247-254).  Every rank
 * prepares the same roi; rank g produces the columns [x0, x1) of the panorama (stx_blend_set_band).
 * An image fed on rank o whose 2^bands-aligned feed rectangle reaches another rank's columns is
 * exported there as a CONTRIBUTION: per pyramid level the products (short)(L*W) and the weights W
 * over a strip (stx_blend_export_contrib -> RCCL send/recv -> stx_blend_feed_contrib).  Integer sums
 * are order independent and `order` (the global feed index) fixes the fp32 weight-sum order, so the
 * assembled panorama is bit-identical to the single-GPU result.  DESIGN.md §6. */
int stx_blend_set_band(stx_blender* b, int x0, int x1);
int stx_blend_feed_ex(stx_blender* b, const stx_buf* img, const stx_buf* mask, int tlx, int tly, int order);
/* strip rect (relative to the roi origin, level 0) and packed size of the contribution that an image
 * of size (img_w, img_h) at corner (tlx, tly) owes to the owner of columns [band_x0, band_x1);
 * w = 0 when it owes nothing.  Pure geometry: sender and receiver compute the same answer. */
int stx_blend_contrib_rect(const stx_blender* b, int img_w, int img_h, int tlx, int tly, int band_x0, int band_x1,
                           int out_rect_xywh[4], size_t* out_bytes);
int stx_blend_export_contrib(stx_blender* b, int order, int band_x0, int band_x1, stx_buf** out_packed,
                             int out_rect_xywh[4]);
/* all the strips a rank owes in one call (n images / destination bands): same results, one kernel launch per kernel
 * instantiation instead of one per strip and level; out_packed[n], out_rects_xywh[4 n] */
int stx_blend_export_contribs(stx_blender* b, int n, const int* orders, const int* band_x0s, const int* band_x1s,
                              stx_buf** out_packed, int* out_rects_xywh);
/* build the pyramids of every image fed so far now (they are otherwise built at the first export / blend()):
 * a sharded rank calls it for its interior images while its strips are still in flight */
int stx_blend_build(stx_blender* b);
int stx_blend_feed_contrib(stx_blender* b, int order, const int rect_xywh[4], const stx_buf* packed);
/* flags travel with a strip over the caller's control plane.  STX_CONTRIB_U8_BINARY: the strip was exported from
 * a u8 image whose mask holds only 0 / 255 (then its products are L or 0 and its weights 0.f or 1.f, and the
 * receiver may keep the packed 16-bit level-0 kernel).  stx_buf_flags reads it off an exported strip (or a mask:
 * the same bit says "only 0 / 255"). */
#define STX_CONTRIB_U8_BINARY 1
/* STX_STRIP_MASK_BITS (strips only): the mask rows of the flat buffer hold one bit per pixel instead of a byte (a 0 / 255 mask:
 * 3.125 instead of 4 bytes per pixel on the link); stx_strip_unpack / stx_blend_feed_strips expand them into a mask buffer. */
#define STX_STRIP_MASK_BITS 2
int stx_blend_feed_contrib_ex(stx_blender* b, int order, const int rect_xywh[4], const stx_buf* packed, int flags);
int stx_buf_flags(const stx_buf* buf, int* out_flags);

/* Rectangle {x0, x1, y0, y1} of an image that the panorama rectangle [band_x0, band_x1) x [band_y0, band_y1) (roi-relative)
 * depends on through the pyramids: stx_strip_rect's column range and the same range along y.  A pipeline that knows where an
 * image's fed mask is non-zero (its seam cell) needs no more of the image than this; all zeros: nothing. */
int stx_view_rect(const stx_blender* b, int img_w, int img_h, int tlx, int tly, int band_x0, int band_x1, int band_y0, int band_y1,
                  int out_x0x1y0y1[4]);
/* image-strip form of the same exchange: the owner of an image ships the COLUMNS [x0, x1) of the warped
 * image and mask that another band depends on (4 bytes per pixel, no export pass) and the receiver feeds them with
 * stx_blend_feed_ex like an image of its own, at corner (tlx + x0, tly) and with the image's global `order`; its band
 * comes out bit-identical.  stx_strip_rect: pure geometry (x0 == x1: nothing owed), columns relative to the image;
 * stx_strip_pack: image rows then mask rows in one flat buffer (two 2-D device copies on the context stream);
 * stx_strip_unpack: the two views of a received flat buffer (flags: STX_CONTRIB_U8_BINARY when the mask is 0 / 255). */
int stx_strip_rect(const stx_blender* b, int img_w, int img_h, int tlx, int tly, int band_x0, int band_x1, int out_x0x1[2],
                   size_t* out_bytes);
int stx_strip_pack(stx_ctx* ctx, const stx_buf* img, const stx_buf* mask, int x0, int x1, stx_buf** out_packed);
/* all strips a rank owes in one call (one copy kernel per 16 strips); x0 multiples of 8, whole image buffers (no views) */
int stx_strip_pack_batch(stx_ctx* ctx, int n, const stx_buf* const* imgs, const stx_buf* const* masks, const int* x0s, const int* x1s,
                         stx_buf** out_packed);
/* the same with flags (STX_STRIP_MASK_BITS); stx_strip_bytes: size of the flat buffer of a w x h strip under these flags */
int stx_strip_pack_batch_ex(stx_ctx* ctx, int n, const stx_buf* const* imgs, const stx_buf* const* masks, const int* x0s, const int* x1s,
                            int flags, stx_buf** out_packed);
int stx_strip_bytes(int w, int h, int flags, size_t* out_bytes);
/* all received strips in one call: unpack + stx_blend_feed_ex(strip i at (tlxs[i], tlys[i]), orders[i]) */
int stx_blend_feed_strips(stx_blender* b, int n, const stx_buf* const* packed, const int* ws, const int* hs, const int* tlxs,
                          const int* tlys, const int* orders, int flags);
int stx_strip_unpack(const stx_buf* packed, int w, int h, int flags, stx_buf** out_img, stx_buf** out_mask);

/* ---- RCCL strip exchange over xGMI ---------------------------------------------------------------
 * One communicator per rank (one process per GPU).  stx_comm_exchange issues every send / receive
 * of one step as a single RCCL group on the context's HIP stream: it is ordered after the kernels
 * that filled the send buffers and before the kernels that read the receive buffers.  The unique id
 * (128 bytes, from rank 0) travels over the caller's control plane. */
typedef struct stx_comm stx_comm;
int stx_comm_unique_id(unsigned char out[128]);
int stx_comm_create(stx_ctx* ctx, int nranks, int rank, const unsigned char id[128], stx_comm** out);
int stx_comm_exchange(stx_comm* comm, int n_ops, const int* peers, const int* is_send, void* const* dev_ptrs,
                      const size_t* bytes);
/* split form: _begin issues the group on the communicator's own stream, ordered after everything queued so far
 * on the context stream; kernels queued on the context stream between _begin and _end (warps and pyramids of
 * images that send nothing) overlap with the transfer; _end orders the context stream after the transfer.
 * The buffers must stay alive until _end has been called. */
int stx_comm_exchange_begin(stx_comm* comm, int n_ops, const int* peers, const int* is_send, void* const* dev_ptrs,
                            const size_t* bytes);
int stx_comm_exchange_end(stx_comm* comm);
/* same, for a context other than the one the communicator was created on (same device): one communicator serves
 * several panoramas in flight on different streams; their exchanges are serialised on the communicator's stream */
int stx_comm_exchange_begin_on(stx_comm* comm, stx_ctx* on, int n_ops, const int* peers, const int* is_send,
                               void* const* dev_ptrs, const size_t* bytes);
int stx_comm_exchange_end_on(stx_comm* comm, stx_ctx* on);
/* what librccl itself reports: {ncclCommCount, ncclCommUserRank, ncclGetVersion, device}; -1 where the loaded library lacks the call */
int stx_comm_info(const stx_comm* comm, int out_info[4]);
int stx_comm_destroy(stx_comm* comm);

/* ---- measurement hooks (bench.py) -----------------------------------------------------
 * When enabled, every kernel launch on the ctx stream is bracketed by HIP events recorded
 * on that stream; stx_prof_get reports per-kernel call count, summed duration and the
 * algorithmic bytes the launch sites declared. */
int stx_prof_enable(stx_ctx* ctx, int on);
int stx_prof_reset(stx_ctx* ctx);
int stx_prof_count(stx_ctx* ctx, int* out_n);
int stx_prof_get(stx_ctx* ctx, int index, char* name, int name_cap, int64_t* calls, double* total_ms,
                 double* algo_bytes);
/* elapsed device time between two marks recorded on the ctx stream */
int stx_mark(stx_ctx* ctx, int slot);
int stx_mark_elapsed_ms(stx_ctx* ctx, int slot_begin, int slot_end, double* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* STITCHING_AMD_H */
