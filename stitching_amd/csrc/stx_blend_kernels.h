// stx_blend_kernels.h — kernel argument blocks shared by stx_blend.hip (generic kernels) and
// stx_blend_fast.hip (register-blocked kernels for the fine pyramid levels).
#pragma once
#include "stx_internal.h"

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed, used for speed only) and the per-XCD
// L2s share nothing, so tiles are dealt to the XCDs in bands of `band_rows` tile rows: neighbouring tiles — which share
// halo rows and the cache lines at their common edge — meet in one L2.  Index math by reciprocal multiplication.
struct StxTileMap {
    int tiles_x, tiles_y, band_rows, band_tiles;
    unsigned magic_tx, magic_band;  // floor(2^32 / d) + 1 for d = tiles_x, band_tiles (unused when d == 1)
    int plain;  // 1: row-major order, tile = workgroup index (the hardware's round-robin over the XCDs)
};
inline StxTileMap stx_tile_map(int tiles_x, int tiles_y, int band_rows)
{
    StxTileMap m;
    m.tiles_x = tiles_x; m.tiles_y = tiles_y; m.band_rows = band_rows; m.band_tiles = band_rows * tiles_x;
    m.magic_tx = (unsigned)((1ull << 32) / (unsigned)tiles_x) + 1u;
    m.magic_band = (unsigned)((1ull << 32) / (unsigned)m.band_tiles) + 1u;
    m.plain = 0;
    return m;
}
// 1-D grid size (x) that covers every tile: 8 XCDs x whole bands
inline unsigned stx_tile_grid(const StxTileMap& m)
{
    if (m.plain) return (unsigned)(m.tiles_x * m.tiles_y);
    const int bands = (m.tiles_y + m.band_rows - 1) / m.band_rows;
    return 8u * (unsigned)(((bands + 7) / 8) * m.band_tiles);
}

constexpr int STX_DEFER_SEGS = 256;
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
    StxTileMap tiles;  // fast kernels: XCD-aware order of the 512 x 8 tiles
    int pk_ok;   // every image is kind 0, u8x3, with a mask known to hold only 0 / 255 (packed 16-bit kernels)
    // level 0 with masks that MAY hold grey bytes (resized seam masks: grey along the seams only): the packed kernel runs on every lane
    // and a lane that meets a grey byte under its 8 x 2 patch appends the patch origin (x | y << 32) here instead of storing; a second,
    // small launch computes those patches with fp32 weights (mb_level0_deferred_kernel).  null: no deferral.
    // The queue is cut into segments, one per COLUMN of 512-pixel tiles (tile column tx -> segment tx % STX_DEFER_SEGS; counter s at
    // defer_count[32 s], its own 128-byte line).  Why: (i) one word takes ~88 returning atomics per microsecond on this chip (the
    // guide's price) and a fifth of the wavefronts of the default pipeline queue something: 21 000 wavefront-aggregated atomics
    // would be a fifth of a millisecond if they all met on one counter — a precaution; measured, the single counter was NOT what
    // made the first version slow (232 us with one counter, 241 us with 256 of them: what hurt was the second pass); (ii) the 64
    // patches of a wavefront of the second pass then lie in one 512-pixel column, under the same two or three images — scattered over
    // the panorama every wavefront walked all images, one dependent memory round trip after the other, alone on its SIMD (80 us).
    // defer_cap: entries per segment (room for every patch of the tile columns that map to it); defer_segs: segments in use.
    int defer_segs;
    unsigned long long* defer_list; unsigned* defer_count; unsigned defer_cap;
};

// fast-path launchers (stx_blend_fast.hip); each returns false when its alignment / size
// preconditions do not hold and the generic kernel must be used instead.
bool stx_fast_mb_down_batch(stx_ctx* ctx, const StxMbImage* d_images, const StxMbImage* h_images, int n, int level);
bool stx_fast_mb_level(stx_ctx* ctx, const MbLevelK& K);

// batched strip export: see stx_blend_fast.hip
int stx_fast_mb_emit_class(const MbLevelK& K, MbLevelK* KT);
bool stx_fast_mb_emit_launch(stx_ctx* ctx, int cls, const MbLevelK* d_Ks, const MbLevelK* h_Ks, int count);
// all argument blocks of one batched strip export (any mix of strips and levels): grouped by kernel instantiation,
// one launch per group.  d_Ks mirrors h_Ks (sorted by class by the caller: classes[i] ascending); bytes: algorithmic bytes
int stx_launch_mb_emit_batch(stx_ctx* ctx, const MbLevelK* d_Ks, const MbLevelK* h_Ks, const int* classes, int n, double bytes);
