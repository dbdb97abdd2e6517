// stx_blend_kernels.h — kernel argument blocks shared by stx_blend.hip (generic kernels) and
// stx_blend_fast.hip (register-blocked kernels for the fine pyramid levels).
#pragma once
#include "stx_internal.h"

struct MbLevelK {
    const StxMbImage* images;
    int n_images, level, num_bands, pw, ph;
    short* out; long long out_stride, out_plane;
    const short* up; long long up_stride, up_plane;
    uint8_t* pano; long long pano_stride;
    uint8_t* pmask; long long pmask_stride;
    short* pano16; long long pano16_stride;
    int final_w, final_h;
    int all_u8;  // every level-0 source is u8x3 (the fast level-0 kernel has no int16 loader)
};

// fast-path launchers (stx_blend_fast.hip); each returns false when its alignment / size
// preconditions do not hold and the generic kernel must be used instead.
bool stx_fast_mb_down_batch(stx_ctx* ctx, const StxMbImage* d_images, const StxMbImage* h_images, int n, int level);
bool stx_fast_mb_level(stx_ctx* ctx, const MbLevelK& K);
