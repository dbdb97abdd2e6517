// stx_blend_kernels.h — kernel argument blocks shared by stx_blend.hip (generic kernels) and
// stx_blend_fast.hip (register-blocked kernels for the fine pyramid levels).
#pragma once
#include "stx_internal.h"

struct MbLevelK {
    const StxMbImage* images;
    int n_images, level, num_bands, pw, ph;  // pw, ph: padded panorama size at this level (pyrUp border rule)
    // region of this level that is computed: [x0, x1) x [y0, y1) in panorama coordinates of the level.
    // Single GPU: everything the final crop depends on.  Sharded: this rank's column band + pyrUp halo.
    int x0, x1, y0, y1;
    short* out; long long out_stride, out_plane; int out_x0, out_y0;   // planar int16 level (levels >= 1, or emit)
    const short* up; long long up_stride, up_plane; int up_x0, up_y0;  // finished level+1 (null at the coarsest)
    // emit mode (sharded blending): write the un-normalised sums (short)acc -> out, weight sum -> out_w
    int emit; float* out_w; long long out_w_stride;
    // level 0 outputs, origin (pano_x0, pano_y0)
    uint8_t* pano; long long pano_stride;
    uint8_t* pmask; long long pmask_stride;
    short* pano16; long long pano16_stride;
    int pano_x0, pano_y0;
    int has_contrib;  // the image table holds kind-1 entries (received contribution strips)
    int all_u8;  // every level-0 source is u8x3 (the fast level-0 kernel has no int16 loader)
    int pk_ok;   // every image is kind 0, u8x3, with a mask known to hold only 0 / 255 (packed 16-bit kernels)
};

// fast-path launchers (stx_blend_fast.hip); each returns false when its alignment / size
// preconditions do not hold and the generic kernel must be used instead.
bool stx_fast_mb_down_batch(stx_ctx* ctx, const StxMbImage* d_images, const StxMbImage* h_images, int n, int level);
bool stx_fast_mb_level(stx_ctx* ctx, const MbLevelK& K);
