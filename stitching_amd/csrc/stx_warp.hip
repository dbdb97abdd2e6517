// stx_warp.hip — fused backward-projector + remap kernels and the ROI border reduction (gfx950).
//
// Replaces, for stitching/warper.py:43-52,58-68,79-82, OpenCV's
//   RotationWarperBase<P>::buildMaps  (serial trig loop, 2 fp32 maps materialised)
//   cv::remap INTER_LINEAR/BORDER_REFLECT (u8x3) and INTER_NEAREST/BORDER_CONSTANT (u8x1)
//   RotationWarperBase<P>::detectResultRoiByBorder
// with ONE pass that never stores the maps: every thread evaluates mapBackward for 4 adjacent
// destination pixels per row, quantises to 1/32 px exactly as remap does, and samples the source
// through the vector L1/L2 (aligned 12-byte loads + v_alignbyte for the unaligned BGR pairs).
//
// Trig is separable for the spherical / cylindrical projectors: sin/cos(u) depend on the column
// only and sin/cos(pi - v) on the row only, so a 256x16 tile needs 4 sincos per thread (kept in
// registers) + 16 per block (LDS) instead of 4 per pixel.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <hip/hip_ext.h>

#include "stx_device_math.h"
#include "stx_internal.h"

using namespace stxd;

namespace {

constexpr int WARP_TW = 256;  // tile width  (64 lanes x 4 px)
constexpr int WARP_TH = 4;    // tile height (4 waves x 1 row)
// fast kernel: a wavefront takes WARP_IT blocks of WARP_TH rows of its 64 columns one after the other (tile 64 x WARP_IT * WARP_TH): the
// per-image scalars, the tile index and the column table entry are then fetched once for all of them.  Measured: round 3, stores after
// every block: 2 blocks per wavefront 212 us against 180 us with 1 — gfx9 has one vmcnt for loads and stores, so the second block's
// sample loads could not be waited for without also waiting for the first block's stores.  Round 4 samples every block before the first
// result leaves (the stores follow the loop): 179.4 / 182.8 us with 2 blocks against 177.5 / 178.1 with 1 — the diagnosis was right and
// the prologue it saves is worth nothing measurable.  1 it stays (-DSTX_WARP_IT=2 builds the other).
#ifndef STX_WARP_GAIN_EARLY
#define STX_WARP_GAIN_EARLY 1
#endif
#ifndef STX_WARP_IT
#define STX_WARP_IT 1
#endif
constexpr int WARP_IT = STX_WARP_IT;
constexpr int WARP_FTH = WARP_TH * WARP_IT;  // tile height of the fast kernel
// wavefronts per workgroup of the fast kernel (they share nothing: no workgroup barrier, LDS per wavefront): its tile is 64 x this wide.
// Measured (round 4, one box, interleaved, config 2): 1: 175.8 / 174.2 us, 2: 177.0 / 177.0, 4: 178.4 / 178.8, 8: 193.9 / 191.2 — a
// wavefront that finishes frees its slot for the next workgroup at once instead of waiting for its three siblings.
#ifndef STX_WARP_WAVES
#define STX_WARP_WAVES 1
#endif
constexpr int WARP_FW = 64 * STX_WARP_WAVES;
#ifndef STX_WARP_BAND
#define STX_WARP_BAND 4
#endif
// STX_WARP_ZGRID = 1 (A/B only): the grid of rounds 1-5, (workgroups of the LARGEST image, 1, images) — see WarpBatchK::first_wg
#ifndef STX_WARP_ZGRID
#define STX_WARP_ZGRID 0
#endif
constexpr int WARP_BAND = STX_WARP_BAND;  // tile rows per XCD band (fast kernel); measured 1: 548, 2: 424, 4: 360, 8: 327, 16: 311 MB fetched
constexpr float PI_F = 3.14159274101257324f;  // static_cast<float>(CV_PI)

struct WarpK {
    float kr[9];
    float t[3];
    float scale;
    int family;    // STX_F_* (the general kernel switches on it; the typed kernels are instantiated per family)
    float pa, pb;  // compressed-rectilinear / panini parameters
    int tlx, tly, dw, dh, sw, sh;
    long long sstride;
    const uint8_t* src;   // u8x3 source of the bilinear image (IMG)
    const uint8_t* msrc;  // optional u8x1 source of the nearest image; null -> constant 255
    long long msstride;
    uint8_t* dimg;
    long long dimg_stride;
    uint8_t* dmask;
    long long dmask_stride;
    int band_rows;      // fast kernel: tile rows per XCD band
    int tiles_x, tiles_y, band_tiles;
    uint32_t magic_tx, magic_band;  // floor(2^32 / d) + 1 for d = tiles_x, band_tiles
    // Fast kernel, sample positions in 1/32-px units.  s = cvRound(32 v) is read off the bit pattern of
    // fl(32 v + 1.5 * 2^23) = RND_U0 + s (round-half-even of the fp add = cvRound, |32 v| < 2^22), so ranges of s are
    // unsigned ranges of that pattern and a NaN / infinite coordinate is above every upper bound.
    //   interior: s >> 5 in [0, n-2]                       <=>  pattern in [RND_U0, u*_int]
    //   zone    : s >> 5 in [-n, 2n-2] and inside int16    <=>  pattern in [u*_zlo, u*_zhi]  (one mirror image away)
    uint32_t ux_int, uy_int;
    uint32_t ux_zlo, ux_zhi, uy_zlo, uy_zhi;
    //   periodic: |s| < 2^21 (any number of mirror images away): BORDER_REFLECT has the period 2n pixels = 64n units, so
    //             s (after the int16 saturation of its pixel part) is reduced into [-32n, 32n) first: per_* = 64 n,
    //             inv_* = 1 / per_* (fp32 quotient estimate, corrected by one step either way), bias_* = 32 n + k per_* >= 2^20
    int per_x, per_y, bias_x, bias_y;
    float inv_x, inv_y;
    // nearest-neighbour inside test on 32 v: cvRound(v) in [0, n-1]  <=>  -16 <= 32 v < m32_hi (ties go to even)
    float mx32_hi, my32_hi;
    float c2, c5, c8;  // plane: kr2 (1 - t2), kr5 (1 - t2), kr8 (1 - t2), each rounded once (host fp32 = device fp32)
    int trig;    // STX_TRIG_*: which sinf / cosf the projector's trigonometry follows (stx_device_math.h: sincosf_m)
    int remap;   // STX_REMAP_*: the interpolation model of the image samples; anything but Q15 runs the one-pixel-per-lane kernels
    int num_ok;  // host-proved: |numerators| <= 2^60 and finite tables, the per-lane magnitude test is skipped
    int z_one;   // host-proved (plane / affine): z = 1.f for every pixel, the quotients are the numerators
    // Exposure gain in the epilogue (stx_warp_batch_gain: BlocksCompensator::apply fused, stitching/stitcher.py:123,219-221): the
    // horizontally interpolated rows of the gain map over this destination rectangle's columns (gain_rows_kernel: [g_gh][g_hstride] floats,
    // g_hstride = dw rounded up to 4) and the row table (source row, bits of the fraction) of its rows; null: no gain
    const float* g_H; long long g_hstride; const int2* g_yt; int g_gh;
};

// Up to WARP_BATCH images per launch: the per-image argument blocks travel in the kernel-argument segment
// (scalar loads indexed by blockIdx), so a batch needs no descriptor upload and no host synchronisation.
constexpr int WARP_BATCH = 8;
struct WarpBatchK {
    WarpK k[WARP_BATCH];
    float2* colT[WARP_BATCH];
    // row table, in blocks of 4 destination rows (one tile row): rowT[4 b + 0..3] = {ra x4}, {p1 x4}, {p4 x4}, {p7 x4} of rows
    // 4 b .. 4 b + 3 (rows beyond the image repeat the last one); rowT[dh4 + r / 4][r % 4] = rb of row r (dh4 = dh rounded up to 4)
    float4* rowT[WARP_BATCH];
    // fast kernel: ONE 1-D grid for the whole batch — image i owns the workgroups [first_wg[i], first_wg[i + 1]), every bound a multiple of
    // 8 (the XCD of a workgroup is its index modulo 8); entries beyond the batch hold 0xffffffff.  Round 6: a grid of (largest image, 1,
    // images) launched 28 % empty workgroups for a column of config 3 (ROIs of 7965 x 3024 next to 4122 x 2783) and the launch took 234 us
    // against 190 us for its four images one after the other (profiles/r06_warp_split.md).
    uint32_t first_wg[WARP_BATCH];
};
constexpr uint32_t RND_U0 = 0x4B400000u;  // bit pattern of 1.5 * 2^23

STX_DEV uint32_t ldg32(const uint8_t* p) { return *reinterpret_cast<const uint32_t*>(p); }

// 6 consecutive bytes starting at byte address `a` (any alignment) -> lo = bytes 0..3, hi = bytes 4..7
STX_DEV void load6(const uint8_t* base, long long a, uint32_t& lo, uint32_t& hi)
{
    const uint8_t* q = base + (a & ~3ll);
    uint32_t d0 = ldg32(q), d1 = ldg32(q + 4), d2 = ldg32(q + 8);
    uint32_t s = (uint32_t)a & 3u;
    lo = __builtin_amdgcn_alignbyte(d1, d0, s);
    hi = __builtin_amdgcn_alignbyte(d2, d1, s);
}

// remapBilinear, u8x3: weights (32-fx)(32-fy)*32 etc. are Q15 and exact, so
//   (sum_w p*w + 2^14) >> 15  ==  ((p00*(32-fx) + p01*fx)*(32-fy) + (p10*(32-fx) + p11*fx)*fy + 512) >> 10
// (the saturate_cast<short>(32768)->32767,+1 quirk of initInterTab2D at fx=fy=0 cannot change a u8 result:
//  |p11 - p00| <= 255 < 2^14).
STX_DEV uint32_t bil(uint32_t p00, uint32_t p01, uint32_t p10, uint32_t p11, uint32_t fx, uint32_t fy)
{
    uint32_t h0 = p00 * (32u - fx) + p01 * fx;
    uint32_t h1 = p10 * (32u - fx) + p11 * fx;
    return (h0 * (32u - fy) + h1 * fy + 512u) >> 10;
}

// 24-bit pixel j of a lane's 4 adjacent pixels into its 3 output dwords
STX_DEV void put_px(uint32_t (&out)[3], int j, uint32_t px)
{
    if (j == 0) out[0] = px;
    else if (j == 1) { out[0] |= px << 24; out[1] = px >> 8; }
    else if (j == 2) { out[1] |= px << 16; out[2] = px >> 16; }
    else out[2] |= px << 8;
}

// STX_REMAP_FLOAT / STX_REMAP_FLOAT_FMA: fp32 bilinear on the unquantised position with BORDER_REFLECT taps — the model of a remap that
// interpolates in floating point (include/stitching_amd.h spells out the sequence of fp32 operations; the tests compare it with the
// CPU checker's).  A position that is not a finite number of moderate size samples (-1, -1).  One pixel at a time, byte loads: this mode
// exists to be compared with, not to be fast.
STX_DEV uint32_t sample_float(const uint8_t* __restrict__ src, long long sstride, int sw, int sh, bool fused, float x, float y)
{
    if (!(x > -1.0e9f && x < 1.0e9f && y > -1.0e9f && y < 1.0e9f)) { x = -1.f; y = -1.f; }
    const float fxf = floorf(x), fyf = floorf(y);
    const int ix = (int)fxf, iy = (int)fyf;
    const float a = fsub(x, fxf), b = fsub(y, fyf);
    const int sx0 = reflect(ix, sw), sx1 = reflect(ix + 1, sw);
    const int sy0 = reflect(iy, sh), sy1 = reflect(iy + 1, sh);
    const uint8_t* r0 = src + (long long)sy0 * sstride;
    const uint8_t* r1 = src + (long long)sy1 * sstride;
    uint32_t px = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float p00 = (float)r0[sx0 * 3 + c], p01 = (float)r0[sx1 * 3 + c];
        const float p10 = (float)r1[sx0 * 3 + c], p11 = (float)r1[sx1 * 3 + c];
        float t, u, v;
        if (fused) {
            t = __fmaf_rn(a, fsub(p01, p00), p00);
            u = __fmaf_rn(a, fsub(p11, p10), p10);
            v = __fmaf_rn(b, fsub(u, t), t);
        } else {
            t = fadd(fmul(a, fsub(p01, p00)), p00);
            u = fadd(fmul(a, fsub(p11, p10)), p10);
            v = fadd(fmul(b, fsub(u, t)), t);
        }
        px |= (uint32_t)min(max(cv_round(v), 0), 255) << (8 * c);
    }
    return px;
}

// remap(): sx = cvRound(x*32); (ix, fx) = (sx >> 5 saturated to short, sx & 31); remapBilinear with BORDER_REFLECT taps
// (borderInterpolate on each of the 4 taps) -> 24-bit BGR.  The one-pixel-at-a-time form of the fast kernel's sampling.
STX_DEV uint32_t sample_q15(const WarpK& P, float x, float yy)
{
    const int sx = cv_round(fmul(x, 32.f)), sy = cv_round(fmul(yy, 32.f));
    const uint32_t fx = (uint32_t)sx & 31u, fy = (uint32_t)sy & 31u;
    const int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
    uint32_t b, g, rr;
    if ((unsigned)ix < (unsigned)(P.sw - 1) && (unsigned)iy < (unsigned)(P.sh - 1)) {
        const long long a = (long long)iy * P.sstride + (long long)ix * 3;
        uint32_t l0, h0, l1, h1;
        load6(P.src, a, l0, h0);
        load6(P.src, a + P.sstride, l1, h1);
        b = bil(l0 & 255u, l0 >> 24, l1 & 255u, l1 >> 24, fx, fy);
        g = bil((l0 >> 8) & 255u, h0 & 255u, (l1 >> 8) & 255u, h1 & 255u, fx, fy);
        rr = bil((l0 >> 16) & 255u, (h0 >> 8) & 255u, (l1 >> 16) & 255u, (h1 >> 8) & 255u, fx, fy);
    } else {
        const int sx0 = reflect(ix, P.sw) * 3, sx1 = reflect(ix + 1, P.sw) * 3;
        const int sy0 = reflect(iy, P.sh), sy1 = reflect(iy + 1, P.sh);
        const uint8_t* r0 = P.src + (long long)sy0 * P.sstride;
        const uint8_t* r1 = P.src + (long long)sy1 * P.sstride;
        b = bil(r0[sx0], r0[sx1], r1[sx0], r1[sx1], fx, fy);
        g = bil(r0[sx0 + 1], r0[sx1 + 1], r1[sx0 + 1], r1[sx1 + 1], fx, fy);
        rr = bil(r0[sx0 + 2], r0[sx1 + 2], r1[sx0 + 2], r1[sx1 + 2], fx, fy);
    }
    return b | (g << 8) | (rr << 16);
}

// Separable part of mapBackward, evaluated once per destination column / row (fp64 "exact" trig):
//   spherical  : col = (sin u', cos u'),           row: ra = sin(pi - v'), rb = cos(pi - v')
//   mercator   : col = (sin u', cos u'),           row: ra = cos v_, rb = sin v_ with v_ = atan(sinh v')   (typed as spherical)
//   cylindrical: col = (sin u', cos u'),           row: ra = v'
//   plane      : col = (u'/scale - t0, -),         row: ra = v'/scale - t1
// plus the products of the projector that depend on the row only, each rounded once exactly as the per-pixel
// evaluation rounds it (dot3 = (k0 x_ + k1 y_) + k2 z_ with y_ = rb / ra / ra):
//   p1 = kr1 y_, p4 = kr4 y_, p7 = kr7 y_;  plane: c2 = kr2 (1 - t2), c5 = kr5 (1 - t2), c8 = kr8 (1 - t2)
STX_DEV int round_up4(int v) { return (v + 3) & ~3; }

template <int TYPE>
__global__ __launch_bounds__(256) void warp_tables_kernel(WarpBatchK B)
{
    const WarpK& P = B.k[blockIdx.y];
    float2* __restrict__ colT = B.colT[blockIdx.y];
    float* __restrict__ rowT = reinterpret_cast<float*>(B.rowT[blockIdx.y]);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int dh4 = round_up4(P.dh);
    if (i < P.dw) {
        const float uu = (float)(P.tlx + i);
        float2 o = make_float2(0.f, 0.f);
        if (TYPE == STX_WARP_SPHERICAL || TYPE == STX_WARP_CYLINDRICAL) sincosf_m(fdiv(uu, P.scale), P.trig, &o.x, &o.y);
        else o.x = fsub(fdiv(uu, P.scale), P.t[0]);
        colT[i] = o;
    } else if (i < P.dw + dh4) {
        const int slot = i - P.dw, r = min(slot, P.dh - 1);  // the padding rows repeat the last row
        const float vv = (float)(P.tly + r);
        float ra = 0.f, rb = 0.f;
        if (TYPE == STX_WARP_SPHERICAL && P.family == STX_F_MERCATOR) {
            // MercatorProjector::mapBackward is the sphere's with another latitude: v_ = atan(sinh(v')), x_ = cos v_ sin u',
            // y_ = sin v_, z_ = cos v_ cos u' — the same per-pixel arithmetic from a different row table
            sincosf_m(atanf_x(sinhf_x(fdiv(vv, P.scale))), P.trig, &rb, &ra);
        } else if (TYPE == STX_WARP_SPHERICAL) sincosf_m(fsub(PI_F, fdiv(vv, P.scale)), P.trig, &ra, &rb);
        else if (TYPE == STX_WARP_CYLINDRICAL) ra = fdiv(vv, P.scale);
        else ra = fsub(fdiv(vv, P.scale), P.t[1]);
        const float y_ = TYPE == STX_WARP_SPHERICAL ? rb : ra;
        float* blk = rowT + 16 * (slot >> 2) + (slot & 3);
        blk[0] = ra;
        blk[4] = fmul(P.kr[1], y_);
        blk[8] = fmul(P.kr[4], y_);
        blk[12] = fmul(P.kr[7], y_);
        rowT[4 * dh4 + slot] = rb;
    }
}

// One lane = 4 adjacent destination pixels of one row; one wave = 256 px of a row; block = 4 rows.
// DBG (stx_debug_warp_maps, a test hook): instead of sampling, the fp32 (x, y) of every destination pixel — what buildMaps would have
// stored — go to two float images (P.dimg = x map, P.dmask = y map; strides in bytes).
template <int TYPE, bool IMG, bool MASK, bool DBG = false>
__global__ __launch_bounds__(256) void warp_kernel(WarpK P, const float2* __restrict__ colT, const float4* __restrict__ rowT)
{
    const int lane = threadIdx.x & 63;
    const int x0 = blockIdx.x * WARP_TW + lane * 4;
    const int y = blockIdx.y * WARP_TH + (threadIdx.x >> 6);
    if (x0 >= P.dw || y >= P.dh) return;
    float ca[4], cb[4];
    {
        // colT is padded to a multiple of 4 entries, 32-byte aligned per 4 columns
        const float4 c01 = *reinterpret_cast<const float4*>(colT + x0);
        const float4 c23 = *reinterpret_cast<const float4*>(colT + x0 + 2);
        ca[0] = c01.x; cb[0] = c01.y; ca[1] = c01.z; cb[1] = c01.w;
        ca[2] = c23.x; cb[2] = c23.y; ca[3] = c23.z; cb[3] = c23.w;
    }
    const float* rowF = reinterpret_cast<const float*>(rowT);
    const float ra = rowF[16 * (y >> 2) + (y & 3)], rb = rowF[4 * round_up4(P.dh) + y];
    const float omt = fsub(1.f, P.t[2]);
    uint32_t out[3] = {0, 0, 0};
    uint32_t mout = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float x, yy;
        if (TYPE == STX_WARP_PLANE || TYPE == STX_WARP_AFFINE) {
            float z;
            x = fadd(fadd(fmul(P.kr[0], ca[j]), fmul(P.kr[1], ra)), fmul(P.kr[2], omt));
            yy = fadd(fadd(fmul(P.kr[3], ca[j]), fmul(P.kr[4], ra)), fmul(P.kr[5], omt));
            z = fadd(fadd(fmul(P.kr[6], ca[j]), fmul(P.kr[7], ra)), fmul(P.kr[8], omt));
            x = fdiv(x, z);
            yy = fdiv(yy, z);
        } else {
            float x_, y_, z_;
            if (TYPE == STX_WARP_SPHERICAL) {
                x_ = fmul(ra, ca[j]);
                y_ = rb;
                z_ = fmul(ra, cb[j]);
            } else {
                x_ = ca[j];
                y_ = ra;
                z_ = cb[j];
            }
            x = dot3(P.kr[0], x_, P.kr[1], y_, P.kr[2], z_);
            yy = dot3(P.kr[3], x_, P.kr[4], y_, P.kr[5], z_);
            float z = dot3(P.kr[6], x_, P.kr[7], y_, P.kr[8], z_);
            if (z > 0) {
                x = fdiv(x, z);
                yy = fdiv(yy, z);
            } else {
                x = yy = -1.f;
            }
        }
        if (DBG) {
            if (x0 + j < P.dw) {
                reinterpret_cast<float*>(P.dimg + (long long)y * P.dimg_stride)[x0 + j] = x;
                reinterpret_cast<float*>(P.dmask + (long long)y * P.dmask_stride)[x0 + j] = yy;
            }
            continue;
        }
        if (IMG && P.remap != STX_REMAP_Q15) {
            put_px(out, j, sample_float(P.src, P.sstride, P.sw, P.sh, P.remap == STX_REMAP_FLOAT_FMA, x, yy));
        } else if (IMG) {
            put_px(out, j, sample_q15(P, x, yy));
        }
        if (MASK) {
            // remapNearest: saturate_cast<short>(cvRound(x)); inside -> source (255), else 0
            int nx = sat_s16(cv_round(x)), ny = sat_s16(cv_round(yy));
            uint32_t m = 0;
            if ((unsigned)nx < (unsigned)P.sw && (unsigned)ny < (unsigned)P.sh)
                m = P.msrc ? (uint32_t)P.msrc[(long long)ny * P.msstride + nx] : 255u;
            mout |= m << (8 * j);
        }
    }
    if (IMG) {
        uint32_t* d = reinterpret_cast<uint32_t*>(P.dimg + (long long)y * P.dimg_stride + (long long)x0 * 3);
        d[0] = out[0];
        d[1] = out[1];
        d[2] = out[2];
    }
    if (MASK) *reinterpret_cast<uint32_t*>(P.dmask + (long long)y * P.dmask_stride + x0) = mout;
}


// ---------------------------------------------------------------------------------------------
// Fast kernel.  Same results as warp_kernel (bit for bit), well under half the VALU work:
//   * the products of the projector that depend on the row only come from the row table, those that depend on the
//     column only are formed once per lane (a lane owns one column, its 4 pixels one below the other);
//   * x/z and y/z share one Newton-refined reciprocal and finish with the fma sequence of the IEEE
//     division expansion (correctly rounded whenever no rescaling is needed: |z| in [2^-60, 2^60],
//     |x|, |y| <= 2^60; anything else takes __fdiv_rn);
//   * cvRound(32 v) is one fp add (RND_U0 trick, see WarpK); the range tests are unsigned min / max of the patterns;
//   * the sampling path is chosen PER WAVEFRONT (ballots), so no wavefront runs two of them:
//       interior : every sample of the wavefront has its 4 taps inside the source — packed 16-bit blend, per channel
//                  2 v_perm + v_pk_mul/mad_u16 (vertical lerp of both taps) + v_dot2_u32_u16 (horizontal lerp + rounding);
//       mirror   : every sample is at most one mirror image away from the source (BORDER_REFLECT).  Mirroring the
//                  position (s -> -s - 32 below, s -> 64 n - 32 - s above, clamped at the edges where both taps fall on
//                  the edge pixel) turns the two reflected taps into an adjacent pixel pair again, weights in [0, 32]:
//                  the interior code with 7 more integer ops per axis.  Masks by the nearest-neighbour range test;
//       periodic : |s| < 2^21, any number of mirror images away: s is reduced modulo the period of BORDER_REFLECT (2 n
//                  pixels) into the mirror zone first (13 more ops per axis) — the pitched rows of a multi-row panorama
//                  have large parts of their ROI there;
//       generic  : anything else (|32 v| >= 2^21, NaN, divisions that need rescaling): per-tap borderInterpolate.
// Preconditions (host): source < 2^31 bytes, 2 <= sw, sh <= 32767, no nearest-neighbour source image.
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short v2h __attribute__((ext_vector_type(2)));
#define STX_GAS __attribute__((address_space(1)))
#define STX_CAS __attribute__((address_space(4)))

STX_DEV v2h as_v2h(uint32_t v) { return __builtin_bit_cast(v2h, v); }

// x/z and y/z with one shared reciprocal; caller guarantees |z| in [2^-60, 2^60], |x|, |y| <= 2^60
STX_DEV void div2_fast(float d, float n0, float n1, float& q0, float& q1)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __fmaf_rn(-d, r, 1.0f);
    r = __fmaf_rn(e, r, r);
    v2f n = {n0, n1}, rr = {r, r}, nd = {-d, -d};
    v2f t = n * rr;
    v2f u = __builtin_elementwise_fma(nd, t, n);
    t = __builtin_elementwise_fma(u, rr, t);
    u = __builtin_elementwise_fma(nd, t, n);
    t = __builtin_elementwise_fma(u, rr, t);
    q0 = t.x;
    q1 = t.y;
}

// One BORDER_REFLECT tap: the 3 bytes of source pixel (sx, sy) in the low 24 bits (32-bit offsets, global loads)
STX_DEV uint32_t tap24(const STX_GAS uint8_t* src, uint32_t stride, int sx, int sy)
{
    const uint32_t off = (uint32_t)sy * stride + (uint32_t)sx * 3u;
    const STX_GAS uint32_t* q = reinterpret_cast<const STX_GAS uint32_t*>(src + (off & ~3u));
    return __builtin_amdgcn_alignbyte(q[1], q[0], off & 3u);
}

// remapBilinear with BORDER_REFLECT on every tap (borderInterpolate per tap, cvRound / short saturation emulated
// exactly); x32, y32 = 32 x, 32 y.  Same packed blend as the interior path.  Fast-kernel preconditions apply.
STX_DEV uint32_t sample_border(const STX_GAS uint8_t* src, uint32_t stride, int sw, int sh, float x32, float y32)
{
    const int sx = cv_round(x32), sy = cv_round(y32);
    const uint32_t fx = (uint32_t)sx & 31u, fy = (uint32_t)sy & 31u;
    const int ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5);
    const int sx0 = reflect(ix, sw), sx1 = reflect(ix + 1, sw);
    const int sy0 = reflect(iy, sh), sy1 = reflect(iy + 1, sh);
    const uint32_t t00 = tap24(src, stride, sx0, sy0), t01 = tap24(src, stride, sx1, sy0);
    const uint32_t t10 = tap24(src, stride, sx0, sy1), t11 = tap24(src, stride, sx1, sy1);
    const uint32_t wy1 = fy * 0x10001u, wy0 = 0x200020u - wy1;
    const uint32_t wx = fx * 0xffffu + 32u;
    uint32_t o[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const uint32_t sel = c == 0 ? 0x0c040c00u : (c == 1 ? 0x0c050c01u : 0x0c060c02u);  // byte c of both taps
        const v2h a = as_v2h(__builtin_amdgcn_perm(t01, t00, sel)), b = as_v2h(__builtin_amdgcn_perm(t11, t10, sel));
        const v2h v = a * as_v2h(wy0) + b * as_v2h(wy1);
        o[c] = __builtin_amdgcn_udot2(v, as_v2h(wx), 512u, false) >> 10;
    }
    return o[0] | (o[1] << 8) | (o[2] << 16);
}

// The adjacent pixel pair (pixel ix and ix + 1 of row iy and of row iy + 1), weights fx, fy in [0, 32] for the right /
// lower one: three channel bytes to LDS at p[0..2].  Every channel value comes out of its dot product already in byte 2
// of the register:  64 (h0 (32 - fx) + h1 fx + 512) = ((h0 (32 - fx) + h1 fx + 512) >> 10) << 16 + a remainder below
// bit 16 (weights scaled by 64: <= 2048, sums < 2^25); the byte is stored as it is (ds_write_b8_d16_hi).
// Issue costs on gfx950 (tools/ubench/valu_rate2.hip, inline asm, profiles/r04_valu_issue_cycles.txt): a wave64 instruction takes ~4.5 cycles
// of its SIMD for almost everything this kernel uses (v_perm, v_alignbyte, v_dot2, v_pk_*_u16, v_bfe, v_mad_u32_u24, v_lshl_add, v_min3,
// v_pk_fma_f32 — 5.0 for its two elements), ~2.8 for v_fma_f32 / v_mul_f32 / v_add_u32 / v_and_b32 and 8.4 for v_rcp_f32.  Measured A/Bs of
// round 4 (one box, interleaved):
//   STX_WARP_MUL = 1: 3 ix = (ix << 1) + ix, (fy, fy) = fy << 16 | fy, the horizontal weight pair by shifts instead of 24-bit multiplies
//                     — 185.4 / 186.8 us against 185.6 / 185.6: the multiplies cost what the shifts cost.  Default 0 (the round-3 code).
//   STX_WARP_UNALIGNED = 1: the 6-byte groups as byte-exact 8-byte loads (HSA unaligned-access mode) instead of aligned 12-byte windows +
//                     v_alignbyte: 268 us against 185 — the texture-address path splits every unaligned lane access.  Default 0.
#ifndef STX_WARP_MUL
#define STX_WARP_MUL 0
#endif
#ifndef STX_WARP_UNALIGNED
#define STX_WARP_UNALIGNED 0
#endif

STX_DEV uint32_t lshl_add_u32(uint32_t a, uint32_t b)  // (a << 1) + b, as ONE v_lshl_add_u32 (LLVM folds (x << 1) + x back into a multiply)
{
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
STX_DEV uint32_t lshl16_or_u32(uint32_t a, uint32_t b)  // (a << 16) | b
{
    uint32_t d;
    asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// STX_WARP_NOAND = 1 (round 5): v_alignbyte_b32 reads bits 1:0 of its shift operand only (gfx9 ISA), so the `a & 3` in front of it is a
//                     VALU instruction per pixel that LLVM cannot drop (the builtin's operand is a plain i32): 121 -> 117 per 4 pixels.
#ifndef STX_WARP_NOAND
#define STX_WARP_NOAND 1
#endif
// The adjacent pixel pair (ix, ix + 1) of rows iy and iy + 1, all four taps inside the source: l = first 4 bytes, h = next 4 of the 6-byte
// BGRBGR group of either row
STX_DEV void load_taps(const STX_GAS uint8_t* src, uint32_t stride, uint32_t ix, uint32_t iy, uint32_t& l0, uint32_t& h0, uint32_t& l1,
                       uint32_t& h1)
{
    // row < 2^15 and stride < 2^24 (fast_ok): 24-bit multiply
#if STX_WARP_MUL
    const uint32_t a = __umul24(iy, stride) + lshl_add_u32(ix, ix);
#else
    const uint32_t a = __umul24(iy, stride) + ix * 3u;
#endif
    // the lower row through its own scalar base (src + stride: one scalar add per wavefront) and the SAME lane offset — not
    // src + (offset + stride), a vector add per pixel
#if STX_WARP_UNALIGNED
    typedef uint32_t u32x2_u __attribute__((ext_vector_type(2), aligned(1)));
    const u32x2_u r0 = *reinterpret_cast<const STX_GAS u32x2_u*>(src + a);
    const u32x2_u r1 = *reinterpret_cast<const STX_GAS u32x2_u*>((src + stride) + a);
    l0 = r0.x; h0 = r0.y; l1 = r1.x; h1 = r1.y;
#else
    const STX_GAS uint32_t* q0 = reinterpret_cast<const STX_GAS uint32_t*>(src + (a & ~3u));
    const STX_GAS uint32_t* q1 = reinterpret_cast<const STX_GAS uint32_t*>((src + stride) + (a & ~3u));
    const uint32_t d0 = q0[0], d1 = q0[1], d2 = q0[2], e0 = q1[0], e1 = q1[1], e2 = q1[2];
#if STX_WARP_NOAND
    const uint32_t sh = a;
#else
    const uint32_t sh = a & 3u;
#endif
    l0 = __builtin_amdgcn_alignbyte(d1, d0, sh); h0 = __builtin_amdgcn_alignbyte(d2, d1, sh);
    l1 = __builtin_amdgcn_alignbyte(e1, e0, sh); h1 = __builtin_amdgcn_alignbyte(e2, e1, sh);
#endif
}

STX_DEV void blend_pair_to_lds(const STX_GAS uint8_t* src, uint32_t stride, uint32_t ix, uint32_t iy, uint32_t fx, uint32_t fy,
                               uint8_t* p)
{
    uint32_t l0, h0, l1, h1;
    load_taps(src, stride, ix, iy, l0, h0, l1, h1);
#if STX_WARP_MUL
    const uint32_t wy1 = lshl16_or_u32(fy, fy), wy0 = 0x200020u - wy1;  // (fy, fy), (32 - fy, 32 - fy)
    const uint32_t g = fx << 6;
    const uint32_t wx = lshl16_or_u32(g, 2048u - g);              // (64 (32 - fx), 64 fx)
#else
    const uint32_t wy1 = fy * 0x10001u, wy0 = 0x200020u - wy1;  // (fy, fy), (32 - fy, 32 - fy)
    const uint32_t wx = __umul24(fx, 0x3fffc0u) + 2048u;        // (64 (32 - fx), 64 fx)
#endif
#pragma unroll
    for (int c = 0; c < 3; c++) {
        // (left tap, right tap) of channel c as two u16: bytes c and c + 3 of the 6-byte BGRBGR group
        const uint32_t sel = c == 0 ? 0x0c030c00u : (c == 1 ? 0x0c040c01u : 0x0c050c02u);
        const v2h t0 = as_v2h(__builtin_amdgcn_perm(h0, l0, sel)), t1 = as_v2h(__builtin_amdgcn_perm(h1, l1, sel));
        const v2h v = t0 * as_v2h(wy0) + t1 * as_v2h(wy1);  // vertical lerp of both taps, <= 255 * 32
        p[c] = (uint8_t)(__builtin_amdgcn_udot2(v, as_v2h(wx), 32768u, false) >> 16);
    }
}

// STX_REMAP_FLOAT / STX_REMAP_FLOAT_FMA on the fast kernel (round 5): the fp32 bilinear model of sample_float for a position whose four
// taps are inside the source (x, y >= 0, so x - floor(x) is exact): the same two loads per row as the Q15 blend, the taps converted
// straight out of their bytes (v_cvt_f32_ubyteN: no v_perm), every step rounded to fp32 as include/stitching_amd.h spells it out.
template <bool FUSED>
STX_DEV void blend_float_to_lds(const STX_GAS uint8_t* src, uint32_t stride, float x, float y, uint8_t* p)
{
    const float fxf = floorf(x), fyf = floorf(y);
    const float a = fsub(x, fxf), b = fsub(y, fyf);
    uint32_t l0, h0, l1, h1;
    load_taps(src, stride, (uint32_t)(int)fxf, (uint32_t)(int)fyf, l0, h0, l1, h1);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        // left tap: byte c of l; right tap: byte c + 3 of the 6-byte group (c = 0: byte 3 of l, else byte c - 1 of h)
        const int rs = c == 0 ? 24 : 8 * (c - 1);
        const float p00 = (float)((l0 >> (8 * c)) & 255u), p10 = (float)((l1 >> (8 * c)) & 255u);
        const float p01 = (float)(((c > 0 ? h0 : l0) >> rs) & 255u), p11 = (float)(((c > 0 ? h1 : l1) >> rs) & 255u);
        float t, u, v;
        if (FUSED) {
            t = __fmaf_rn(a, fsub(p01, p00), p00);
            u = __fmaf_rn(a, fsub(p11, p10), p10);
            v = __fmaf_rn(b, fsub(u, t), t);
        } else {
            t = fadd(fmul(a, fsub(p01, p00)), p00);
            u = fadd(fmul(a, fsub(p11, p10)), p10);
            v = fadd(fmul(b, fsub(u, t)), t);
        }
        // cvRound + clamp in ONE instruction: v_cvt_pk_u8_f32 rounds to nearest even and saturates by itself (equal to the v_rndne_f32 +
        // conversion pair on every one of the 2^32 floats, NaN -> 0: tools/ubench/cvt_pk_u8.hip)
        p[c] = (uint8_t)__builtin_amdgcn_cvt_pk_u8_f32(v, 0u, 0u);
    }
}

// BORDER_REFLECT within one mirror image: position s (1/32 px, any value of the zone) -> base pixel and right weight
STX_DEV void mirror_axis(int s, int n, uint32_t& ip, uint32_t& fp)
{
    int m = max(max(s, -32 - s), 0);                    // below the image: s -> -s - 32; both taps on pixel 0 -> 0
    m = min(min(m, 64 * n - 32 - m), 32 * (n - 1));     // above: s -> 64 n - 32 - s; both taps on pixel n-1 -> 32 (n-1)
    const int i = min(m >> 5, n - 2);
    ip = (uint32_t)i;
    fp = (uint32_t)(m - (i << 5));                      // 0..32
}

// any position with |s| < 2^21 -> the position inside [-32 n, 32 n) that BORDER_REFLECT maps to the same two taps:
// the pixel part saturates to int16 as remap's saturate_cast<short> does, then s is reduced modulo the period 64 n
STX_DEV int periodic_axis(int s, int n, int period, float inv_period, int bias)
{
    const int se = (min(max(s >> 5, -32768), 32767) << 5) | (s & 31);
    const uint32_t a = (uint32_t)(se + bias);                          // >= 0, < 2^23: exact in fp32
    const uint32_t q = (uint32_t)(int)__fmul_rn((float)a, inv_period);  // floor(a / period), or one off
    uint32_t r = a - __umul24(q, (uint32_t)period);
    r = min(r, r + (uint32_t)period);  // the estimate was one too high: r < 0 wrapped around
    r = min(r, r - (uint32_t)period);  // ... or one too low
    return (int)r - 32 * n;
}

// RM: STX_REMAP_* of the image samples (Q15: the fixed-point blend; FLOAT / FLOAT_FMA: the fp32 model — interior wavefronts through
// blend_float_to_lds, every other wavefront one pixel at a time through sample_float: borders are a minority)
// GAIN: the warped image leaves multiplied by the block gain of its position (cvRound(p g) saturated, cv::multiply's arithmetic) — applied
// to the finished bytes on their way from LDS to memory, whatever sampling path made them
// FLAT: the batch's images share ONE 1-D grid (WarpBatchK::first_wg) instead of a grid of (largest image, 1, images): chosen by the host
// when the images' tile counts differ (see launch_typed)
template <int TYPE, bool IMG, bool MASK, bool DBG = false, int RM = STX_REMAP_Q15, bool GAIN = false, bool FLAT = false>
__global__ __launch_bounds__(WARP_FW) __attribute__((amdgpu_waves_per_eu(8, 8))) void warp_fast_kernel(WarpBatchK B)
{
    // which image of the batch this workgroup belongs to: seven scalar compares against the images' first workgroups
    uint32_t wg = blockIdx.x;
    int zi = blockIdx.z;
    if (FLAT) {
        zi = 0;
        uint32_t first = 0u;
#pragma unroll
        for (int i = 1; i < WARP_BATCH; i++) {
            const uint32_t f = B.first_wg[i];
            const bool past = wg >= f;
            zi = past ? i : zi;
            first = past ? f : first;
        }
        wg -= first;
    }
    const WarpK& P = B.k[zi];
    // the two table pointers ride in the same batch of scalar loads as the per-image scalars below (left to the compiler they are
    // fetched after the early exit: one more dependent scalar-cache round trip in front of the table reads)
    unsigned long long colT_a = (unsigned long long)B.colT[zi], rowT_a = (unsigned long long)B.rowT[zi];
    const int lane = threadIdx.x & 63;
    // The per-image scalars of the tile-index math and of the tests below are fetched up front, as a few wide scalar
    // loads with one wait: a wavefront lives for 256 pixels only, and the compiler otherwise sinks every one of these
    // kernel-argument loads to its first use (about fifteen dependent scalar-cache round trips per wavefront).
    int tiles_x = P.tiles_x, tiles_y = P.tiles_y, band_tiles = P.band_tiles, band_rows = P.band_rows;
    uint32_t magic_tx = P.magic_tx, magic_band = P.magic_band;
    int dw = P.dw, dh = P.dh;
    uint32_t ux_int = P.ux_int, uy_int = P.uy_int;
    int num_ok = P.num_ok;
    unsigned long long src_a = (unsigned long long)P.src, dimg_a = (unsigned long long)P.dimg, dmask_a = (unsigned long long)P.dmask;
    long long dimg_stride = P.dimg_stride, dmask_stride = P.dmask_stride;
    uint32_t sstride = (uint32_t)P.sstride;
    // Round 6: the scalars of the wavefronts that are NOT interior ride in the same batch.  Left in their branch they were four dependent
    // scalar-cache round trips (the range test of the mirror zone compiled into a chain of short-circuit branches, one s_load + wait
    // each) in front of a third of the wavefronts of a pitched frame: measured in column slices of the +-55 degree frames of config 3, a
    // wavefront off the interior path cost 2.7 x an interior one (profiles/r06_warp_split.md section 6).
    int sw_s = P.sw, sh_s = P.sh;
    uint32_t mxhi_b = __float_as_uint(P.mx32_hi), myhi_b = __float_as_uint(P.my32_hi);
    uint32_t ux_zlo = P.ux_zlo, ux_zhi = P.ux_zhi, uy_zlo = P.uy_zlo, uy_zhi = P.uy_zhi;
    asm volatile("" : "+s"(tiles_x), "+s"(tiles_y), "+s"(band_tiles), "+s"(band_rows), "+s"(magic_tx), "+s"(magic_band), "+s"(dw),
                 "+s"(dh), "+s"(ux_int), "+s"(uy_int), "+s"(num_ok), "+s"(sw_s), "+s"(sh_s), "+s"(mxhi_b), "+s"(myhi_b), "+s"(ux_zlo),
                 "+s"(ux_zhi), "+s"(uy_zlo), "+s"(uy_zhi));
    asm volatile("" : "+s"(src_a), "+s"(dimg_a), "+s"(dmask_a), "+s"(dimg_stride), "+s"(dmask_stride), "+s"(sstride), "+s"(colT_a),
                 "+s"(rowT_a));
    const STX_GAS v2f* colT = (const STX_GAS v2f*)colT_a;   // scalar base + 32-bit lane offset
    const STX_CAS v4f* rowT = (const STX_CAS v4f*)rowT_a;  // wave-uniform reads: scalar loads
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch rule, used for speed only), and the
    // per-XCD L2s do not share lines.  Bands of WARP_BAND tile rows go round-robin to the XCDs: vertically adjacent
    // tiles — which read the same source rows — mostly meet in one L2 instead of fetching those rows once per
    // XCD (measured for 8 frames of 36 MB: 605 MB fetched with the plain row-major order, 288 MB with whole-image
    // eighths — but then the curved-border bands unbalance the XCDs — and 360 MB with bands of 4 tile rows; the
    // kernel time is the same for all three).
    // divisions by the per-image constants use host-made reciprocals (n * m >> 32, exact for n * d < 2^32)
    const uint32_t local = wg >> 3;
    const uint32_t band_i = band_tiles == 1 ? local : __umulhi(local, magic_band);  // local / (band_rows * tiles_x)
    const uint32_t within = local - band_i * (uint32_t)band_tiles;
    const uint32_t wy = tiles_x == 1 ? within : __umulhi(within, magic_tx);  // within / tiles_x (2^32 / 1 has no 32-bit magic)
    const int tile_x = (int)(within - wy * (uint32_t)tiles_x);
    const int tile_y = (int)((band_i * 8u + (wg & 7u)) * (uint32_t)band_rows + wy);
    if (tile_y >= tiles_y) return;
    // Lane layout: a wavefront covers 64 columns x WARP_TH rows, one lane = one column, its WARP_TH pixels one below
    // the other (the workgroup's four wavefronts sit side by side: a 256 x 4 tile).  For every row the 64 lanes then
    // gather from ADJACENT source positions — about 3 cache lines per load instruction instead of the 8 that four
    // horizontal pixels per lane touched (one-line-gathers experiment: the scattered form cost 50 of 250 us) — and the
    // row constants are wave-uniform.  The 3-byte results go through LDS to leave as whole dwords: 768 + 256 bytes per
    // wavefront, written bytewise, read back as the 192 + 64 dwords of the wavefront's 4 rows.  Only the wavefront
    // itself reads what it wrote (LDS operations of one wavefront execute in order): no workgroup barrier.
    __shared__ uint32_t s_px[STX_WARP_WAVES][WARP_FTH][48];  // [wavefront][row][dword]: 64 px x 3 B
    __shared__ uint32_t s_mk[STX_WARP_WAVES][WARP_FTH][16];  // [wavefront][row][dword]: 64 px x 1 B
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform, and known to be
    const int xw = tile_x * WARP_FW + wv * 64;  // first column of this wavefront
    // columns beyond the image are computed on the clamped table entry, rows beyond it on the repeated last row
    // (harmless) and never stored
    const v2f ct = *(const STX_GAS v2f*)((const STX_GAS char*)colT + ((uint32_t)min(xw + lane, dw - 1) << 3));  // dw < 2^29
    // column-only products (cylinder, plane), once per lane
    float q0 = 0.f, q3 = 0.f, q6 = 0.f, r2 = 0.f, r5 = 0.f, r8 = 0.f;
    if (TYPE != STX_WARP_SPHERICAL) {
        q0 = fmul(P.kr[0], ct.x); q3 = fmul(P.kr[3], ct.x); q6 = fmul(P.kr[6], ct.x);
        if (TYPE == STX_WARP_CYLINDRICAL) { r2 = fmul(P.kr[2], ct.y); r5 = fmul(P.kr[5], ct.y); r8 = fmul(P.kr[8], ct.y); }
        else { r2 = P.c2; r5 = P.c5; r8 = P.c8; }
    }
    // GAIN (round 6): the two gain rows of the position this lane STORES in the epilogue (row y0 + (lane >> 4), columns xw + 4 (lane & 15) ..)
    // are fetched here, behind the tables, and wait under the projector and the gathers.  Fetched in the epilogue — row table entry, then
    // the two rows it names — they were two more dependent memory round trips at the end of every wavefront: the fused gain cost +58 us
    // on a 170 us kernel for arithmetic worth a fifth of that (STX_WARP_GAIN_EARLY=0 builds that form).
    float4 g_u = make_float4(0.f, 0.f, 0.f, 0.f), g_v = g_u;
    float g_b1 = 0.f;
    if (GAIN && WARP_IT == 1 && STX_WARP_GAIN_EARLY) {
        const int er = lane >> 4, ec = lane & 15;
        const int2 ty = P.g_yt[min(tile_y * WARP_TH + er, dh - 1)];
        g_b1 = __int_as_float(ty.y);
        const int r0 = min(max(ty.x, 0), P.g_gh - 1), r1 = min(max(ty.x + 1, 0), P.g_gh - 1);
        const int colq = min(xw + 4 * ec, (int)P.g_hstride - 4);  // (columns in the row pitch beyond the image: any gain will do)
        g_u = *reinterpret_cast<const float4*>(P.g_H + (long long)r0 * P.g_hstride + colq);
        g_v = *reinterpret_cast<const float4*>(P.g_H + (long long)r1 * P.g_hstride + colq);
    }
    const STX_GAS uint8_t* src = (const STX_GAS uint8_t*)src_a;
    uint8_t* const lpx0 = reinterpret_cast<uint8_t*>(&s_px[wv][0][0]) + lane * 3;  // + 192 per row, + channel
    uint8_t* const lmk0 = reinterpret_cast<uint8_t*>(&s_mk[wv][0][0]) + lane;      // + 64 per row
    uint32_t int_blocks = 0u;  // bit `it`: block `it` of this wavefront took the interior path (its mask is all 255)
    int n_blocks = 0;
#pragma unroll
  for (int it = 0; it < WARP_IT; it++) {
    const int row_blk = tile_y * WARP_IT + it;  // block of WARP_TH rows
    const int y0 = row_blk * WARP_TH;
    if (it > 0 && y0 >= dh) break;
    n_blocks = it + 1;
    uint8_t* const lpx = lpx0 + 768 * it;  // this block's 4 staging rows
    uint8_t* const lmk = lmk0 + 256 * it;
    // The wavefront's 4 rows as two row pairs: every step below is a packed fp32 operation on (row 2h, row 2h + 1).
    // Row constants: one 64-byte block {ra x4}, {p1 x4}, {p4 x4}, {p7 x4} per block of rows, a single scalar load.
    const v4f RA = rowT[4 * row_blk], P1 = rowT[4 * row_blk + 1], P4 = rowT[4 * row_blk + 2], P7 = rowT[4 * row_blk + 3];
    v2f X[2], Y[2], Z[2];
    {
        const v2f ra[2] = {{RA.x, RA.y}, {RA.z, RA.w}}, p1[2] = {{P1.x, P1.y}, {P1.z, P1.w}};
        const v2f p4[2] = {{P4.x, P4.y}, {P4.z, P4.w}}, p7[2] = {{P7.x, P7.y}, {P7.z, P7.w}};
        if (TYPE == STX_WARP_SPHERICAL) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const v2f x_ = ra[h] * ct.x, z_ = ra[h] * ct.y;
                X[h] = (x_ * P.kr[0] + p1[h]) + z_ * P.kr[2];
                Y[h] = (x_ * P.kr[3] + p4[h]) + z_ * P.kr[5];
                Z[h] = (x_ * P.kr[6] + p7[h]) + z_ * P.kr[8];
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                X[h] = (p1[h] + q0) + r2;
                Y[h] = (p4[h] + q3) + r5;
                Z[h] = (p7[h] + q6) + r8;
            }
        }
    }
    // every division of this lane may use the shared-reciprocal sequence: all |z| (z for the rotation warpers,
    // which also need z > 0) in [2^-60, 2^60] and all |x|, |y| <= 2^60.  Evaluated as integer min / max of the
    // float bit patterns of z (a NaN z is "too big" or negative there, so it cannot slip through a NaN-dropping
    // fp min) and one fp max over |x|, |y| (a NaN numerator gives NaN on both division paths) — the latter only when
    // the host could not bound the numerators from the camera (num_ok: a wave-uniform branch).
    // An affine map (AffineWarper: third row of K R^-1 = (0 0 1), t2 = 0) has z = (0 v' + 0 u') + 1 = 1.f exactly and x / 1.f = x:
    // no division at all (host-proved together with the finiteness of the tables, WarpK::z_one).
    if (!((TYPE == STX_WARP_PLANE || TYPE == STX_WARP_AFFINE) && P.z_one)) {
        int zb[4] = {__float_as_int(Z[0].x), __float_as_int(Z[0].y), __float_as_int(Z[1].x), __float_as_int(Z[1].y)};
        if (TYPE == STX_WARP_PLANE || TYPE == STX_WARP_AFFINE) {
    #pragma unroll
            for (int j = 0; j < 4; j++) zb[j] &= 0x7fffffff;
        }
        const int zlo = min(min(zb[0], zb[1]), min(zb[2], zb[3])), zhi = max(max(zb[0], zb[1]), max(zb[2], zb[3]));
        bool easy = zlo >= 0x21800000 /* 2^-60 */ && zhi <= 0x5d800000 /* 2^60 */;
        if (!num_ok) {
            const float nmax = fmaxf(fmaxf(fmaxf(fabsf(X[0].x), fabsf(Y[0].x)), fmaxf(fabsf(X[0].y), fabsf(Y[0].y))),
                                     fmaxf(fmaxf(fabsf(X[1].x), fabsf(Y[1].x)), fmaxf(fabsf(X[1].y), fabsf(Y[1].y))));
            easy = easy && nmax <= 0x1p60f;
        }
        if (easy) {
            // x/z and y/z of a row pair: Newton-refined reciprocals, then the fma sequence of the IEEE division expansion
    #pragma unroll
            for (int h = 0; h < 2; h++) {
                const v2f d = Z[h], nd = -d;
                v2f r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
                const v2f one = {1.0f, 1.0f};
                const v2f e = __builtin_elementwise_fma(nd, r, one);
                r = __builtin_elementwise_fma(e, r, r);
                v2f t = X[h] * r;
                v2f u = __builtin_elementwise_fma(nd, t, X[h]);
                t = __builtin_elementwise_fma(u, r, t);
                u = __builtin_elementwise_fma(nd, t, X[h]);
                X[h] = __builtin_elementwise_fma(u, r, t);
                t = Y[h] * r;
                u = __builtin_elementwise_fma(nd, t, Y[h]);
                t = __builtin_elementwise_fma(u, r, t);
                u = __builtin_elementwise_fma(nd, t, Y[h]);
                Y[h] = __builtin_elementwise_fma(u, r, t);
            }
        } else {
    #pragma unroll
            for (int h = 0; h < 2; h++) {
    #pragma unroll
                for (int e = 0; e < 2; e++) {
                    const float z = e ? Z[h].y : Z[h].x, x = e ? X[h].y : X[h].x, y = e ? Y[h].y : Y[h].x;
                    float qx = -1.f, qy = -1.f;
                    if (TYPE == STX_WARP_PLANE || TYPE == STX_WARP_AFFINE || z > 0) {
                        qx = fdiv(x, z);
                        qy = fdiv(y, z);
                    }
                    if (e) { X[h].y = qx; Y[h].y = qy; } else { X[h].x = qx; Y[h].x = qy; }
                }
            }
        }
    }
    if (DBG) {
        // test hook: the quotients as they stand — the (x, y) mapBackward returns — instead of the samples they select
        const int col = xw + lane;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (col < dw && y0 + j < dh) {
                reinterpret_cast<float*>((uint8_t*)dimg_a + (long long)(y0 + j) * dimg_stride)[col] = (j & 1) ? X[j >> 1].y : X[j >> 1].x;
                reinterpret_cast<float*>((uint8_t*)dmask_a + (long long)(y0 + j) * dmask_stride)[col] = (j & 1) ? Y[j >> 1].y : Y[j >> 1].x;
            }
        }
    } else {
    // 32 x, 32 y (exact) and their cvRound as bit patterns
    uint32_t ux[4], uy[4];
    float xs[4], ys[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        // fl(32 x + 1.5 * 2^23) as ONE fused multiply-add: 32 x is exact (a power of two), so the fma rounds the same real number
        // the multiply + add pair rounds; the products themselves (xs, ys) are only needed off the interior path
        const v2f x32 = X[h] * 32.f, y32 = Y[h] * 32.f;
        const v2f k32 = {32.f, 32.f}, kmg = {12582912.f, 12582912.f};
        const v2f tx = __builtin_elementwise_fma(X[h], k32, kmg), ty = __builtin_elementwise_fma(Y[h], k32, kmg);
        xs[2 * h] = x32.x; xs[2 * h + 1] = x32.y; ys[2 * h] = y32.x; ys[2 * h + 1] = y32.y;
        ux[2 * h] = __float_as_uint(tx.x); ux[2 * h + 1] = __float_as_uint(tx.y);
        uy[2 * h] = __float_as_uint(ty.x); uy[2 * h + 1] = __float_as_uint(ty.y);
    }
    const uint32_t uxmn = min(min(ux[0], ux[1]), min(ux[2], ux[3])), uxmx = max(max(ux[0], ux[1]), max(ux[2], ux[3]));
    const uint32_t uymn = min(min(uy[0], uy[1]), min(uy[2], uy[3])), uymx = max(max(uy[0], uy[1]), max(uy[2], uy[3]));
    // interior: valid for the image samples; the nearest-neighbour mask sample of an interior position is inside too
    // the fp32 model samples at floor(x), which is s >> 5 or one less (|32 x - s| <= 1/2): its interior starts one pixel further in
    const uint32_t u_lo = RM == STX_REMAP_Q15 ? RND_U0 : RND_U0 + 32u;
    const bool lane_int = uxmn >= u_lo && uxmx <= ux_int && uymn >= u_lo && uymx <= uy_int;
    const bool wave_int = __builtin_amdgcn_ballot_w64(!lane_int) == 0;
    if (wave_int) {
        if (IMG) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (RM == STX_REMAP_Q15)
                    blend_pair_to_lds(src, sstride, __builtin_amdgcn_ubfe(ux[j], 5, 17), __builtin_amdgcn_ubfe(uy[j], 5, 17), ux[j] & 31u,
                                           uy[j] & 31u, lpx + 192 * j);
                else
                    blend_float_to_lds<RM == STX_REMAP_FLOAT_FMA>(src, sstride, (j & 1) ? X[j >> 1].y : X[j >> 1].x,
                                                                      (j & 1) ? Y[j >> 1].y : Y[j >> 1].x, lpx + 192 * j);
            }
        }
    } else if (RM != STX_REMAP_Q15) {
        const int sw = sw_s, sh = sh_s;
        if (MASK) {
            const float mx32_hi = __uint_as_float(mxhi_b), my32_hi = __uint_as_float(myhi_b);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool in = xs[j] >= -16.f && xs[j] < mx32_hi && ys[j] >= -16.f && ys[j] < my32_hi;
                lmk[64 * j] = in ? 255 : 0;
            }
        }
        if (IMG) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t px = sample_float((const uint8_t*)src_a, (long long)sstride, sw, sh, RM == STX_REMAP_FLOAT_FMA,
                                                 (j & 1) ? X[j >> 1].y : X[j >> 1].x, (j & 1) ? Y[j >> 1].y : Y[j >> 1].x);
                lpx[192 * j] = (uint8_t)px;
                lpx[192 * j + 1] = (uint8_t)(px >> 8);
                lpx[192 * j + 2] = (uint8_t)(px >> 16);
            }
        }
    } else {
        const int sw = sw_s, sh = sh_s;
        const float mx32_hi = __uint_as_float(mxhi_b), my32_hi = __uint_as_float(myhi_b);
        if (MASK) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool in = xs[j] >= -16.f && xs[j] < mx32_hi && ys[j] >= -16.f && ys[j] < my32_hi;
                lmk[64 * j] = in ? 255 : 0;
            }
        }
        if (IMG) {
            const bool lane_zone = (uxmn >= ux_zlo) & (uxmx <= ux_zhi) & (uymn >= uy_zlo) & (uymx <= uy_zhi);  // (no short circuit: see the prologue)
            if (__builtin_amdgcn_ballot_w64(!lane_zone) == 0) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t ix, iy, fx, fy;
                    mirror_axis((int)(ux[j] - RND_U0), sw, ix, fx);
                    mirror_axis((int)(uy[j] - RND_U0), sh, iy, fy);
                    blend_pair_to_lds(src, sstride, ix, iy, fx, fy, lpx + 192 * j);
                }
            } else if (__builtin_amdgcn_ballot_w64(min(uxmn, uymn) < RND_U0 - (1u << 21) || max(uxmx, uymx) >= RND_U0 + (1u << 21)) == 0) {
                // several mirror images away somewhere in the wavefront: reduce by the period, then mirror as above
                const int per_x = P.per_x, per_y = P.per_y, bias_x = P.bias_x, bias_y = P.bias_y;
                const float inv_x = P.inv_x, inv_y = P.inv_y;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t ix, iy, fx, fy;
                    mirror_axis(periodic_axis((int)(ux[j] - RND_U0), sw, per_x, inv_x, bias_x), sw, ix, fx);
                    mirror_axis(periodic_axis((int)(uy[j] - RND_U0), sh, per_y, inv_y, bias_y), sh, iy, fy);
                    blend_pair_to_lds(src, sstride, ix, iy, fx, fy, lpx + 192 * j);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t px = sample_border(src, sstride, sw, sh, xs[j], ys[j]);
                    lpx[192 * j] = (uint8_t)px;
                    lpx[192 * j + 1] = (uint8_t)(px >> 8);
                    lpx[192 * j + 2] = (uint8_t)(px >> 16);
                }
            }
        }
    }
    int_blocks |= (wave_int ? 1u : 0u) << it;
    }  // !DBG
  }
    if (DBG) return;
    // Every block of rows is sampled before the first result leaves: gfx9 counts loads and stores in ONE vmcnt, so a store issued
    // between two blocks would sit in front of the second block's sample loads and the wavefront would wait out a store round trip.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // A block's 4 x 192 bytes (4 x 64 of the mask) sit in LDS in the order lane * 12 (lane * 4): lane = 16 r + c
    // stores the 12 (4) bytes at column offset 12 c (4 c) of row r — one 12-byte and one 4-byte store per lane and block.
    const int r = lane >> 4, c = lane & 15;
#pragma unroll
  for (int it = 0; it < WARP_IT; it++) {
    if (it >= n_blocks) break;
    const int y0 = (tile_y * WARP_IT + it) * WARP_TH;
    const bool wave_int = (int_blocks >> it) & 1u;
    const bool full = y0 + WARP_TH <= dh && (long long)xw * 3 + 192 <= dimg_stride && dimg_stride < (1ll << 24) &&
                      (long long)xw + 64 <= dmask_stride && dmask_stride < (1ll << 24);
    if (full) {
        if (IMG) {
            const uint32_t* sp = &s_px[wv][4 * it][0] + lane * 3;
            STX_GAS uint8_t* base = (STX_GAS uint8_t*)(dimg_a + (unsigned long long)y0 * (unsigned long long)dimg_stride + (unsigned long long)xw * 3ull);
            STX_GAS uint32_t* d = reinterpret_cast<STX_GAS uint32_t*>(base + (__umul24((uint32_t)r, (uint32_t)dimg_stride) + (uint32_t)c * 12u));
            uint32_t a0 = sp[0], a1 = sp[1], a2 = sp[2];
            if (GAIN) {
                // this lane's 4 pixels: columns xw + 4 c .. + 3 of row y0 + r.  g = H[r0][x] b0 + H[r1][x] b1 as cv::resize(INTER_LINEAR) rounds it
                float4 u, v;
                float b1;
                if (WARP_IT == 1 && STX_WARP_GAIN_EARLY) {  // fetched behind the tables (see there)
                    u = g_u; v = g_v; b1 = g_b1;
                } else {
                    const int2 ty = P.g_yt[y0 + r];
                    b1 = __int_as_float(ty.y);
                    const int r0 = min(max(ty.x, 0), P.g_gh - 1), r1 = min(max(ty.x + 1, 0), P.g_gh - 1);
                    const int colq = min(xw + 4 * c, (int)P.g_hstride - 4);  // (columns in the row pitch beyond the image: any gain will do)
                    u = *reinterpret_cast<const float4*>(P.g_H + (long long)r0 * P.g_hstride + colq);
                    v = *reinterpret_cast<const float4*>(P.g_H + (long long)r1 * P.g_hstride + colq);
                }
                const float b0 = fsub(1.f, b1);
                const float g[4] = {fadd(fmul(u.x, b0), fmul(v.x, b1)), fadd(fmul(u.y, b0), fmul(v.y, b1)), fadd(fmul(u.z, b0), fmul(v.z, b1)),
                                    fadd(fmul(u.w, b0), fmul(v.w, b1))};
                const uint32_t in[3] = {a0, a1, a2};
                uint32_t o[3] = {0u, 0u, 0u};
#pragma unroll
                for (int k = 0; k < 12; k++)  // byte k of the 12: pixel k / 3
                    o[k >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(fmul((float)((in[k >> 2] >> (8 * (k & 3))) & 255u), g[k / 3]), (uint32_t)(k & 3), o[k >> 2]);  // rounds + saturates itself (see blend_float_to_lds)
                a0 = o[0]; a1 = o[1]; a2 = o[2];
            }
            d[0] = a0; d[1] = a1; d[2] = a2;  // (non-temporal stores, round 5: 183.7 / 184.5 us against 177.1 / 177.9 — dropped)
        }
        if (MASK) {
            STX_GAS uint8_t* base = (STX_GAS uint8_t*)(dmask_a + (unsigned long long)y0 * (unsigned long long)dmask_stride + (unsigned long long)xw);
            *reinterpret_cast<STX_GAS uint32_t*>(base + (__umul24((uint32_t)r, (uint32_t)dmask_stride) + (uint32_t)c * 4u)) =
                wave_int ? 0xffffffffu : s_mk[wv][4 * it + r][c];
        }
    } else {
        if (IMG) {
            // a dword is skipped when it would pass the end of the row pitch (last tile of a row) or of the image (last tile row)
            uint8_t* const drow = (uint8_t*)dimg_a + (long long)xw * 3;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int idx = lane + 64 * k, rr = idx / 48, cdw = idx - rr * 48;
                if (y0 + rr < dh && (long long)xw * 3 + cdw * 4 + 4 <= dimg_stride) {
                    uint32_t w = s_px[wv][4 * it + rr][cdw];
                    if (GAIN) {  // edge tiles: the dword's 4 bytes belong to two pixels
                        const int2 ty = P.g_yt[y0 + rr];
                        const float b1 = __int_as_float(ty.y), b0 = fsub(1.f, b1);
                        const float* h0 = P.g_H + (long long)min(max(ty.x, 0), P.g_gh - 1) * P.g_hstride;
                        const float* h1 = P.g_H + (long long)min(max(ty.x + 1, 0), P.g_gh - 1) * P.g_hstride;
                        const int pa = (4 * cdw) / 3, pb = (4 * cdw + 3) / 3;
                        const int xa = min(xw + pa, (int)P.g_hstride - 1), xb = min(xw + pb, (int)P.g_hstride - 1);
                        const float ga = fadd(fmul(h0[xa], b0), fmul(h1[xa], b1)), gb = fadd(fmul(h0[xb], b0), fmul(h1[xb], b1));
                        uint32_t o = 0u;
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            o = __builtin_amdgcn_cvt_pk_u8_f32(fmul((float)((w >> (8 * q)) & 255u), (4 * cdw + q) / 3 == pa ? ga : gb), (uint32_t)q, o);
                        w = o;
                    }
                    *reinterpret_cast<uint32_t*>(drow + (long long)(y0 + rr) * dimg_stride + cdw * 4) = w;
                }
            }
        }
        if (MASK) {
            if (y0 + r < dh && (long long)xw + c * 4 + 4 <= dmask_stride)
                *reinterpret_cast<uint32_t*>((uint8_t*)dmask_a + (long long)(y0 + r) * dmask_stride + xw + c * 4) =
                    wave_int ? 0xffffffffu : s_mk[wv][4 * it + r][c];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The twelve projectors without a separable backward map (fisheye, stereographic, compressed rectilinear,
// panini, mercator, transverse mercator): per-pixel mapBackward with the exact-trig routines, then the same
// fixed-point sampling as warp_kernel.  Formulas: OpenCV warpers_inl.hpp [OCV-MEM], every fp32 step rounded separately.
// ---------------------------------------------------------------------------------------------
// generic remapBilinear / BORDER_REFLECT sample of one pixel -> 24-bit BGR
STX_DEV uint32_t sample_generic(const WarpK& P, float x, float yy)
{
    if (P.remap != STX_REMAP_Q15) return sample_float(P.src, P.sstride, P.sw, P.sh, P.remap == STX_REMAP_FLOAT_FMA, x, yy);
    return sample_q15(P, x, yy);
}

// Family groups of the per-pixel kernel.  What depends on the destination column alone (u') or on the row alone (v') is
// evaluated once per workgroup tile (256 columns x 16 rows) into LDS; a pixel pays only for the inseparable rest:
//   fisheye / stereographic : nothing separable                      pixel: atan2, sqrt (+ atan), 2 sincos
//   compressed rectilinear  : col u_ = a atan(u'/a), sincos(u_)      pixel: atan, sincos
//   panini                  : col u_, sincos(u_), b a tan(u_/a)      pixel: atan, sincos
//   transverse mercator     : col sinh u', cosh u'; row sincos(v')   pixel: asin, atan2, 2 sincos
// Every fp32 step is OpenCV's mapBackward step (warpers_inl.hpp: same operands, same order, each rounded to fp32 separately):
//   u' = u / scale (portrait: / -scale), v' = v / scale, then per family as in gen_col / gen_row / gen_dir below; the portrait
//   variants swap x_ and y_ at the end.
enum { GEN_FISH = 0, GEN_CRECT, GEN_PANINI, GEN_TMERC };
constexpr int GEN_TH = 16;  // tile rows

template <int G>
STX_DEV float4 gen_col(const WarpK& P, float upx)
{
    const bool portrait = P.family == STX_F_CRECT_PORTRAIT || P.family == STX_F_PANINI_PORTRAIT;
    const float u = fdiv(upx, portrait ? -P.scale : P.scale);
    float4 o = make_float4(u, 0.f, 0.f, 0.f);
    if (G == GEN_CRECT || G == GEN_PANINI) {
        const float u_ = fmul(P.pa, atanf_x(fdiv(u, P.pa)));
        sincosf_m(u_, P.trig, &o.x, &o.y);  // sinf(u_) / cosf(u_) are these very values
        if (G == GEN_PANINI) {
            o.z = fmul(fmul(P.pb, P.pa), tanf_x(fdiv(u_, P.pa)));
            o.w = u_ == u_ ? 1.f : 0.f;
        }
    } else if (G == GEN_TMERC) {
        o.x = sinhf_x(u);
        o.y = coshf_x(u);
    }
    return o;
}

template <int G>
STX_DEV float2 gen_row(const WarpK& P, float vpx)
{
    const float v = fdiv(vpx, P.scale);
    float2 o = make_float2(v, 0.f);
    if (G == GEN_TMERC) sincosf_m(v, P.trig, &o.x, &o.y);
    return o;
}

template <int G>
STX_DEV void gen_dir(const WarpK& P, const float4 c, const float2 r, float& x_, float& y_, float& z_)
{
    if (G == GEN_FISH) {
        const float u = c.x, v = r.x;
        const float u_ = atan2f_x(v, u);
        const float rad = fsqrt(fadd(fmul(u, u), fmul(v, v)));
        const float v_ = P.family == STX_F_FISHEYE ? rad : fmul(2.f, atanf_x(fdiv(1.f, rad)));
        float sinv, cosv, su, cu;
        sincosf_m(fsub(PI_F, v_), P.trig, &sinv, &cosv);
        sincosf_m(u_, P.trig, &su, &cu);
        x_ = fmul(sinv, su);
        y_ = cosv;
        z_ = fmul(sinv, cu);
        return;
    }
    float su, cu, v_;
    if (G == GEN_CRECT) {
        su = c.x; cu = c.y;
        v_ = atanf_x(fdiv(fmul(r.x, cu), P.pb));
    } else if (G == GEN_PANINI) {
        su = c.x; cu = c.y;
        v_ = c.w != 0.f ? atanf_x(fdiv(fmul(r.x, su), c.z)) : 0.f;
    } else {
        v_ = asinf_x(fdiv(r.x, c.y));
        sincosf_m(atan2f_x(c.x, r.y), P.trig, &su, &cu);
    }
    float sv, cv;
    sincosf_m(v_, P.trig, &sv, &cv);
    x_ = fmul(cv, su);
    y_ = sv;
    z_ = fmul(cv, cu);
    if (P.family == STX_F_CRECT_PORTRAIT || P.family == STX_F_PANINI_PORTRAIT) { const float t = x_; x_ = y_; y_ = t; }
}

template <int G, bool IMG, bool MASK, bool DBG = false>
__global__ __launch_bounds__(256) void warp_general_kernel(WarpK P)
{
    __shared__ float4 s_col[WARP_TW];
    __shared__ float2 s_row[GEN_TH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tx0 = blockIdx.x * WARP_TW, ty0 = blockIdx.y * GEN_TH;
    s_col[tid] = gen_col<G>(P, (float)(P.tlx + min(tx0 + tid, P.dw - 1)));
    if (tid < GEN_TH) s_row[tid] = gen_row<G>(P, (float)(P.tly + min(ty0 + tid, P.dh - 1)));
    __syncthreads();
    const int x0 = tx0 + lane * 4;
    if (x0 >= P.dw) return;
    for (int it = 0; it < GEN_TH / 4; it++) {
        const int yl = it * 4 + wv, y = ty0 + yl;
        if (y >= P.dh) break;
        const float2 rw = s_row[yl];
        uint32_t out[3] = {0, 0, 0};
        uint32_t mout = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float d0, d1, d2;
            gen_dir<G>(P, s_col[lane * 4 + j], rw, d0, d1, d2);
            float x = dot3(P.kr[0], d0, P.kr[1], d1, P.kr[2], d2);
            float yy = dot3(P.kr[3], d0, P.kr[4], d1, P.kr[5], d2);
            const float z = dot3(P.kr[6], d0, P.kr[7], d1, P.kr[8], d2);
            if (z > 0) {
                x = fdiv(x, z);
                yy = fdiv(yy, z);
            } else {
                x = yy = -1.f;
            }
            if (DBG) {
                if (x0 + j < P.dw) {
                    reinterpret_cast<float*>(P.dimg + (long long)y * P.dimg_stride)[x0 + j] = x;
                    reinterpret_cast<float*>(P.dmask + (long long)y * P.dmask_stride)[x0 + j] = yy;
                }
                continue;
            }
            if (IMG) put_px(out, j, sample_generic(P, x, yy));
            if (MASK) {
                int nx = sat_s16(cv_round(x)), ny = sat_s16(cv_round(yy));
                uint32_t m = 0;
                if ((unsigned)nx < (unsigned)P.sw && (unsigned)ny < (unsigned)P.sh)
                    m = P.msrc ? (uint32_t)P.msrc[(long long)ny * P.msstride + nx] : 255u;
                mout |= m << (8 * j);
            }
        }
        if (IMG) {
            uint32_t* d = reinterpret_cast<uint32_t*>(P.dimg + (long long)y * P.dimg_stride + (long long)x0 * 3);
            d[0] = out[0];
            d[1] = out[1];
            d[2] = out[2];
        }
        if (MASK) *reinterpret_cast<uint32_t*>(P.dmask + (long long)y * P.dmask_stride + x0) = mout;
    }
}

// ---------------------------------------------------------------------------------------------
// ROI: forward-project the source border, NaN-ignoring float min/max
// ---------------------------------------------------------------------------------------------
struct RoiK {
    float rk[9];
    float scale;
    int type, w, h;
    int family;    // STX_F_*
    float pa, pb;
    int full;      // 1: every source pixel (RotationWarperBase::detectResultRoi), 0: the border (detectResultRoiByBorder)
    int trig;      // STX_TRIG_* (the forward maps of the per-pixel projector families call sinf / cosf)
};

// mapForward of the projectors without a detectResultRoi override; direction already multiplied by r_kinv
__device__ __noinline__ void forward_general(int family, float a, float b, float scale, int trig, float x_, float y_, float z_, float* out2)
{
    const bool portrait = family == STX_F_CRECT_PORTRAIT || family == STX_F_PANINI_PORTRAIT;
    if (portrait) { const float t = x_; x_ = y_; y_ = t; }
    const float u_ = atan2f_x(x_, z_);
    const float w = fdiv(y_, fsqrt(fadd(fadd(fmul(x_, x_), fmul(y_, y_)), fmul(z_, z_))));
    float u, v;
    if (family == STX_F_FISHEYE || family == STX_F_STEREOGRAPHIC) {
        const float v_ = fsub(PI_F, acosf_x(w));
        float su, cu;
        sincosf_m(u_, trig, &su, &cu);
        float r = v_;
        if (family == STX_F_STEREOGRAPHIC) {
            float sv, cv;
            sincosf_m(v_, trig, &sv, &cv);
            r = fdiv(sv, fsub(1.f, cv));
        }
        u = fmul(fmul(scale, r), cu);
        v = fmul(fmul(scale, r), su);
    } else {
        const float v_ = asinf_x(w);
        const float s = portrait ? -scale : scale;
        if (family == STX_F_CRECT || family == STX_F_CRECT_PORTRAIT) {
            u = fmul(fmul(s, a), tanf_x(fdiv(u_, a)));
            v = fdiv(fmul(fmul(scale, b), tanf_x(v_)), cosf_m(u_, trig));
        } else if (family == STX_F_PANINI || family == STX_F_PANINI_PORTRAIT) {
            const float tg = fmul(a, tanf_x(fdiv(u_, a)));
            u = fmul(s, tg);
            const float sinu = sinf_m(u_, trig);
            if ((double)fabsf(sinu) < 1E-7) v = fmul(fmul(scale, b), tanf_x(v_));
            else v = fdiv(fmul(fmul(fmul(scale, b), tg), tanf_x(v_)), sinu);
        } else if (family == STX_F_MERCATOR) {
            u = fmul(scale, u_);
            v = fmul(scale, logf_x(tanf_x(fadd((float)(3.14159265358979323846 / 4), fdiv(v_, 2.f)))));
        } else {  // transverse mercator
            const float B = fmul(cosf_m(v_, trig), sinf_m(u_, trig));
            u = fmul(fdiv(scale, 2.f), logf_x(fdiv(fadd(1.f, B), fsub(1.f, B))));
            v = fmul(scale, atan2f_x(tanf_x(v_), cosf_m(u_, trig)));
        }
    }
    out2[0] = u; out2[1] = v;
}

STX_DEV uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int ROI_BATCH = 16;
constexpr int ROI_SPIN_US = 1000;  // host polling of the ROI stamps before it blocks
struct RoiBatchK { RoiK k[ROI_BATCH]; };

template <bool GEN>  // GEN: the per-pixel projector families (every source pixel), else cylindrical / spherical borders
__global__ __launch_bounds__(256) void roi_kernel(RoiBatchK B, float* __restrict__ out, uint32_t* __restrict__ stamps, uint32_t seq)
{
    const RoiK& P = B.k[blockIdx.y];
    const int npts = GEN ? P.w * P.h : 2 * P.w + 2 * P.h;
    float mnu = 3.402823466e+38f, mnv = 3.402823466e+38f, mxu = -3.402823466e+38f, mxv = -3.402823466e+38f;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npts; p += gridDim.x * blockDim.x) {
        float x, y;
        if (GEN) { const int py = p / P.w; x = (float)(p - py * P.w); y = (float)py; }
        else if (p < P.w) { x = (float)p; y = 0.f; }
        else if (p < 2 * P.w) { x = (float)(p - P.w); y = (float)(P.h - 1); }
        else if (p < 2 * P.w + P.h) { x = 0.f; y = (float)(p - 2 * P.w); }
        else { x = (float)(P.w - 1); y = (float)(p - 2 * P.w - P.h); }
        // mapForward: r_kinv * (x, y, 1)
        float x_ = fadd(fadd(fmul(P.rk[0], x), fmul(P.rk[1], y)), P.rk[2]);
        float y_ = fadd(fadd(fmul(P.rk[3], x), fmul(P.rk[4], y)), P.rk[5]);
        float z_ = fadd(fadd(fmul(P.rk[6], x), fmul(P.rk[7], y)), P.rk[8]);
        float u, v;
        if (GEN) {
            float o[2];
            forward_general(P.family, P.pa, P.pb, P.scale, P.trig, x_, y_, z_, o);
            u = o[0]; v = o[1];
        } else if (P.type == STX_WARP_SPHERICAL) {
            u = fmul(P.scale, atan2f_x(x_, z_));
            float n = fsqrt(fadd(fadd(fmul(x_, x_), fmul(y_, y_)), fmul(z_, z_)));
            float w = fdiv(y_, n);
            v = fmul(P.scale, fsub(PI_F, acosf_x(w == w ? w : 0.f)));
        } else {  // cylindrical
            u = fmul(P.scale, atan2f_x(x_, z_));
            v = fdiv(fmul(P.scale, y_), fsqrt(fadd(fmul(x_, x_), fmul(z_, z_))));
        }
        if (u < mnu) mnu = u;
        if (v < mnv) mnv = v;
        if (u > mxu) mxu = u;
        if (v > mxv) mxv = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float a = __shfl_xor(mnu, o), b = __shfl_xor(mnv, o), c = __shfl_xor(mxu, o), d = __shfl_xor(mxv, o);
        if (a < mnu) mnu = a;
        if (b < mnv) mnv = b;
        if (c > mxu) mxu = c;
        if (d > mxv) mxv = d;
    }
    // one partial result per block (no atomics, no initialisation pass): the host folds the partials
    __shared__ float s_part[4][4];
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        s_part[wv][0] = mnu; s_part[wv][1] = mnv; s_part[wv][2] = mxu; s_part[wv][3] = mxv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int wv = 1; wv < 4; wv++) {
            if (s_part[wv][0] < mnu) mnu = s_part[wv][0];
            if (s_part[wv][1] < mnv) mnv = s_part[wv][1];
            if (s_part[wv][2] > mxu) mxu = s_part[wv][2];
            if (s_part[wv][3] > mxv) mxv = s_part[wv][3];
        }
        const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        float* o = out + 4 * blk;
        o[0] = mnu; o[1] = mnv; o[2] = mxu; o[3] = mxv;
        // `out` is pinned host memory: the block's stamp follows its result at system scope, the host reads stamps, then results
        __hip_atomic_store(stamps + blk, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

bool fast_ok(const WarpK& K)
{
    return !K.msrc && K.sw <= 32767 && K.sh <= 32767 && K.sw >= 2 && K.sh >= 2 && (long long)K.sstride * K.sh < (1ll << 31);
}

template <int TYPE>
int launch_typed(stx_ctx* ctx, const WarpK* Ks, int n, bool img, bool mask, const char* prof_name, const double* algo_bytes, int dbg)
{
    hipStream_t s = ctx->stream;
    // per-column / per-row trig tables (a few KB per image, L2 resident), one allocation for the batch, freed in stream order
    size_t total = 0;
    // in float2 units: dw4 column entries + dh4 / 4 row blocks of 16 floats + dh4 floats (rb), rounded up to 3 dh4 float2
    for (int i = 0; i < n; i++) total += (((size_t)Ks[i].dw + 3) & ~(size_t)3) + 3 * (((size_t)Ks[i].dh + 3) & ~(size_t)3) + 32;
    void* tab = nullptr;
    STX_TRY(stx_dev_alloc(ctx, total * sizeof(float2), &tab));
    float2* cursor = (float2*)tab;
    for (int base = 0, m = 0; base < n; base += m) {
        // a launch takes up to WARP_BATCH images of one remap model (the sampling block is compiled per model), all with or all without a gain
        const int rm = Ks[base].src ? Ks[base].remap : STX_REMAP_Q15;
        const bool gain = Ks[base].g_H != nullptr;
        for (m = 1; m < WARP_BATCH && base + m < n && (Ks[base + m].src ? Ks[base + m].remap : STX_REMAP_Q15) == rm && (Ks[base + m].g_H != nullptr) == gain; m++) {}
        WarpBatchK B;
        memset(&B, 0, sizeof(B));
        int max_tab = 0, gx = 0, gy = 0;
        bool fast = true;
        double tab_bytes = 0.0, bytes = 0.0;
        for (int i = 0; i < m; i++) {
            const WarpK& K = Ks[base + i];
            B.k[i] = K;
            B.k[i].band_rows = WARP_BAND;
            B.k[i].tiles_x = (K.dw + WARP_FW - 1) / WARP_FW;
            B.k[i].tiles_y = (K.dh + WARP_FTH - 1) / WARP_FTH;
            B.k[i].band_tiles = B.k[i].band_rows * B.k[i].tiles_x;
            B.k[i].magic_tx = (uint32_t)((1ull << 32) / (uint32_t)B.k[i].tiles_x) + 1u;
            B.k[i].magic_band = (uint32_t)((1ull << 32) / (uint32_t)B.k[i].band_tiles) + 1u;
            B.colT[i] = cursor;
            cursor += ((size_t)K.dw + 3) & ~(size_t)3;
            cursor = reinterpret_cast<float2*>(((uintptr_t)cursor + 63) & ~(uintptr_t)63);  // 64-byte aligned row blocks
            B.rowT[i] = reinterpret_cast<float4*>(cursor);
            cursor += 3 * (((size_t)K.dh + 3) & ~(size_t)3);
            max_tab = std::max(max_tab, K.dw + ((K.dh + 3) & ~3));
            gx = std::max(gx, (K.dw + WARP_TW - 1) / WARP_TW);
            gy = std::max(gy, (K.dh + WARP_TH - 1) / WARP_TH);
            fast = fast && fast_ok(K) && dbg != 2;
            if (gain && (!fast_ok(K) || dbg)) {
                stx_dev_free(ctx, tab);
                return stx_fail(STX_ERR_UNSUPPORTED, "a fused gain needs the tuned warp kernel (image %d does not qualify)", base + i);
            }
            tab_bytes += (double)K.dw * sizeof(float2) + (double)K.dh * 5 * sizeof(float);
            bytes += algo_bytes[base + i];
        }
        {
            StxProfScope prof(ctx, "warp_tables", tab_bytes);
            hipLaunchKernelGGL((warp_tables_kernel<TYPE>), dim3((max_tab + 255) / 256, m), dim3(256), 0, s, B);
        }
        if (fast) {
            // one launch in this bracket: with the profiler on its events are attached to the launch itself (the kernel's own begin / end
            // stamps, what rocprofv3 reports), not recorded around it
            StxProfScope prof(ctx, prof_name, bytes, nullptr, true);
            // workgroups an image needs: 8 x (its share of the bands per XCD, whole bands only) x tiles per row.  Two grids:
            //   z grid   : (the LARGEST image's workgroups, 1, images) — the image is blockIdx.z, nothing to look up; workgroups beyond an
            //              image's own count exit at once, which costs their dispatch: a column of config 3 (ROIs of 7965 x 3024 next to
            //              4122 x 2783) launches 28 % empty workgroups;
            //   flat grid: one 1-D grid, image i owns [first_wg[i], first_wg[i + 1]) — no empty workgroups, but every workgroup makes one
            //              more dependent scalar load before its per-image block.
            // Measured (round 6, one box, profiles/r06_warp_split.md): config 2 (equal ROIs) 180-181 us z against 183-188 flat; a column
            // of config 3 235 z against 223 flat; config 4's 8 frames 1190 z against 1129 flat.  The flat grid is taken when the z grid
            // would launch more than 5 % empty workgroups (STX_WARP_ZGRID=1 builds a library that never does: the A/B).
            unsigned long long wgs = 0, wg_max = 0;
            for (int i = 0; i < WARP_BATCH; i++) B.first_wg[i] = 0xffffffffu;
            for (int i = 0; i < m; i++) {
                const int tx = (B.k[i].dw + WARP_FW - 1) / WARP_FW, ty = (B.k[i].dh + WARP_FTH - 1) / WARP_FTH;
                const int wb = B.k[i].band_rows, bands = (ty + wb - 1) / wb;
                const unsigned long long own = 8ull * (unsigned long long)(((bands + 7) / 8) * wb) * (unsigned long long)tx;
                B.first_wg[i] = (uint32_t)wgs;
                wgs += own;
                wg_max = std::max(wg_max, own);
            }
            if (wgs >= (1ull << 31)) {
                stx_dev_free(ctx, tab);
                return stx_fail(STX_ERR_UNSUPPORTED, "warp batch of %llu tiles exceeds the grid", wgs);
            }
            const bool flat = !STX_WARP_ZGRID && !dbg && (double)wgs < 0.95 * (double)(wg_max * (unsigned long long)m);
            const dim3 gf(flat ? (unsigned)wgs : (unsigned)wg_max, 1, flat ? 1 : m);
            // STITCHING_AMD_WARP_LDS (diagnostic): bytes of dynamic LDS requested on top of the kernel's own — an occupancy limit
            // (160 KB per CU / request = workgroups per CU) for co-residency experiments with the other panorama's kernels
            static const unsigned pad_lds = getenv("STITCHING_AMD_WARP_LDS") ? (unsigned)atoi(getenv("STITCHING_AMD_WARP_LDS")) : 0u;
#define STX_FAST_LAUNCH_K(...)                                                                                                    \
    do {                                                                                                                          \
        if (prof.start()) hipExtLaunchKernelGGL((__VA_ARGS__), gf, dim3(WARP_FW), pad_lds, s, prof.start(), prof.stop(), 0, B);     \
        else hipLaunchKernelGGL((__VA_ARGS__), gf, dim3(WARP_FW), pad_lds, s, B);                                                   \
    } while (0)
#define STX_FAST_LAUNCH_G(I, M, R, G)                                                    \
    do {                                                                                 \
        if (flat) STX_FAST_LAUNCH_K(warp_fast_kernel<TYPE, I, M, false, R, G, true>);     \
        else STX_FAST_LAUNCH_K(warp_fast_kernel<TYPE, I, M, false, R, G, false>);         \
    } while (0)
#define STX_FAST_LAUNCH(I, M, R)                              \
    do {                                                      \
        if (gain) STX_FAST_LAUNCH_G(I, M, R, true);           \
        else STX_FAST_LAUNCH_G(I, M, R, false);               \
    } while (0)
#define STX_FAST_LAUNCH_RM(I, M)                                             \
    do {                                                                     \
        if (rm == STX_REMAP_FLOAT) STX_FAST_LAUNCH(I, M, STX_REMAP_FLOAT);   \
        else if (rm == STX_REMAP_FLOAT_FMA) STX_FAST_LAUNCH(I, M, STX_REMAP_FLOAT_FMA); \
        else STX_FAST_LAUNCH(I, M, STX_REMAP_Q15);                           \
    } while (0)
            if (dbg) STX_FAST_LAUNCH_K(warp_fast_kernel<TYPE, false, false, true>);
            else if (img && mask) STX_FAST_LAUNCH_RM(true, true);
            else if (img) STX_FAST_LAUNCH_RM(true, false);
            else STX_FAST_LAUNCH_G(false, true, STX_REMAP_Q15, false);
#undef STX_FAST_LAUNCH_RM
#undef STX_FAST_LAUNCH
#undef STX_FAST_LAUNCH_G
#undef STX_FAST_LAUNCH_K
        } else {
            for (int i = 0; i < m; i++) {
                StxProfScope prof(ctx, prof_name, algo_bytes[base + i]);
                const WarpK& K = B.k[i];
                const dim3 grid((K.dw + WARP_TW - 1) / WARP_TW, (K.dh + WARP_TH - 1) / WARP_TH);
                if (dbg) hipLaunchKernelGGL((warp_kernel<TYPE, false, false, true>), grid, dim3(256), 0, s, K, B.colT[i], B.rowT[i]);
                else if (img && mask) hipLaunchKernelGGL((warp_kernel<TYPE, true, true>), grid, dim3(256), 0, s, K, B.colT[i], B.rowT[i]);
                else if (img) hipLaunchKernelGGL((warp_kernel<TYPE, true, false>), grid, dim3(256), 0, s, K, B.colT[i], B.rowT[i]);
                else hipLaunchKernelGGL((warp_kernel<TYPE, false, true>), grid, dim3(256), 0, s, K, B.colT[i], B.rowT[i]);
            }
        }
    }
    stx_dev_free(ctx, tab);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return stx_fail(STX_ERR_HIP, "warp kernel launch failed: %s", hipGetErrorString(e));
    return STX_OK;
}

// per-pixel projector warpers: one launch per image
template <int G>
void launch_general_group(stx_ctx* ctx, const WarpK& K, bool img, bool mask, int dbg)
{
    const dim3 grid((K.dw + WARP_TW - 1) / WARP_TW, (K.dh + GEN_TH - 1) / GEN_TH);
    if (dbg) hipLaunchKernelGGL((warp_general_kernel<G, false, false, true>), grid, dim3(256), 0, ctx->stream, K);
    else if (img && mask) hipLaunchKernelGGL((warp_general_kernel<G, true, true>), grid, dim3(256), 0, ctx->stream, K);
    else if (img) hipLaunchKernelGGL((warp_general_kernel<G, true, false>), grid, dim3(256), 0, ctx->stream, K);
    else hipLaunchKernelGGL((warp_general_kernel<G, false, true>), grid, dim3(256), 0, ctx->stream, K);
}

int launch_general(stx_ctx* ctx, const WarpK* Ks, int n, bool img, bool mask, const char* prof_name, const double* algo_bytes, int dbg)
{
    for (int i = 0; i < n; i++)
        if (Ks[i].g_H) return stx_fail(STX_ERR_UNSUPPORTED, "a fused gain needs the tuned warp kernel (per-pixel projector families have none)");
    for (int i = 0; i < n; i++) {
        StxProfScope prof(ctx, prof_name, algo_bytes[i]);
        const WarpK& K = Ks[i];
        switch (K.family) {
        case STX_F_FISHEYE:
        case STX_F_STEREOGRAPHIC: launch_general_group<GEN_FISH>(ctx, K, img, mask, dbg); break;
        case STX_F_CRECT:
        case STX_F_CRECT_PORTRAIT: launch_general_group<GEN_CRECT>(ctx, K, img, mask, dbg); break;
        case STX_F_PANINI:
        case STX_F_PANINI_PORTRAIT: launch_general_group<GEN_PANINI>(ctx, K, img, mask, dbg); break;
        case STX_F_TRANSVERSE_MERCATOR: launch_general_group<GEN_TMERC>(ctx, K, img, mask, dbg); break;
        default: return stx_fail(STX_ERR_INVALID, "no per-pixel kernel for projector family %d", K.family);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return stx_fail(STX_ERR_HIP, "warp kernel launch failed: %s", hipGetErrorString(e));
    return STX_OK;
}

void fill_warpk(const StxWarpLaunch& L, WarpK* Kp, double* bytes)
{
    WarpK& K = *Kp;
    memset(&K, 0, sizeof(K));
    for (int i = 0; i < 9; i++) K.kr[i] = L.proj.k_rinv[i];
    for (int i = 0; i < 3; i++) K.t[i] = L.proj.t[i];
    K.scale = L.proj.scale;
    K.family = L.proj.family; K.pa = L.proj.a; K.pb = L.proj.b;
    K.trig = L.proj.trig;
    K.remap = L.proj.remap;
    K.tlx = L.tlx; K.tly = L.tly; K.dw = L.dw; K.dh = L.dh;
    K.sw = L.sw; K.sh = L.sh;
    const bool img = L.dimg != nullptr, mask = L.dmask != nullptr;
    K.src = img ? L.src : nullptr;
    K.sstride = (long long)L.sstride;
    K.msrc = L.nearest_src ? L.src : nullptr;
    K.msstride = (long long)L.sstride;
    K.dimg = L.dimg; K.dimg_stride = (long long)L.dimg_stride;
    K.g_H = img ? L.gain_H : nullptr; K.g_hstride = L.gain_hstride; K.g_yt = reinterpret_cast<const int2*>(L.gain_yt); K.g_gh = L.gain_gh;
    K.dmask = L.dmask; K.dmask_stride = (long long)L.dmask_stride;
    // fast kernel: ranges of s = cvRound(32 v) as bit patterns of fl(32 v + 1.5 * 2^23), see WarpK
    const int zx_lo = std::max(-32 * L.sw, -32768 * 32), zx_hi = std::min(64 * L.sw - 33, 32767 * 32 + 31);
    const int zy_lo = std::max(-32 * L.sh, -32768 * 32), zy_hi = std::min(64 * L.sh - 33, 32767 * 32 + 31);
    K.ux_int = RND_U0 + (uint32_t)(32 * (L.sw - 1) - 1);
    K.uy_int = RND_U0 + (uint32_t)(32 * (L.sh - 1) - 1);
    K.per_x = 64 * L.sw; K.per_y = 64 * L.sh;
    K.inv_x = 1.0f / (float)K.per_x; K.inv_y = 1.0f / (float)K.per_y;
    K.bias_x = 32 * L.sw + K.per_x * (((1 << 20) + K.per_x - 1) / K.per_x);
    K.bias_y = 32 * L.sh + K.per_y * (((1 << 20) + K.per_y - 1) / K.per_y);
    K.ux_zlo = RND_U0 + (uint32_t)zx_lo; K.ux_zhi = RND_U0 + (uint32_t)zx_hi;
    K.uy_zlo = RND_U0 + (uint32_t)zy_lo; K.uy_zhi = RND_U0 + (uint32_t)zy_hi;
    // cvRound(v) <= n - 1: v <= n - 0.5 when n - 1 is even (the tie rounds down to it), v < n - 0.5 otherwise; times 32 (exact)
    K.mx32_hi = 32.f * (((L.sw - 1) & 1) ? (float)(L.sw - 0.5) : std::nextafterf((float)(L.sw - 0.5), 3.0e38f));
    K.my32_hi = 32.f * (((L.sh - 1) & 1) ? (float)(L.sh - 0.5) : std::nextafterf((float)(L.sh - 0.5), 3.0e38f));
    // Can the numerators of x/z, y/z exceed 2^60, can a table entry be non-finite?  Bounds of |x_|, |y_|, |z_| from the
    // camera: unit vectors for the rotation warpers (y_ = v / scale for the cylinder), table magnitudes for the plane.
    {
        bool ok = std::isfinite(K.scale) && std::fabs(K.scale) > 1e-30f;
        for (int i = 0; i < 9; i++) ok = ok && std::isfinite(K.kr[i]);
        for (int i = 0; i < 3; i++) ok = ok && std::isfinite(K.t[i]);
        const double sc = std::fabs((double)K.scale);
        const double umax = std::max(std::fabs((double)L.tlx), std::fabs((double)L.tlx + L.dw)) / std::max(sc, 1e-30);
        const double vmax = std::max(std::fabs((double)L.tly), std::fabs((double)L.tly + L.dh)) / std::max(sc, 1e-30);
        double bx = 1.0001, by = 1.0001, bz = 1.0001;
        if (L.proj.family == STX_F_SPHERICAL || L.proj.family == STX_F_MERCATOR) ok = ok && umax < 5e5 && vmax < 5e5;
        else if (L.proj.family == STX_F_CYLINDRICAL) { ok = ok && umax < 5e5; by = vmax * 1.0001; }
        else { bx = (umax + std::fabs((double)K.t[0])) * 1.0001; by = (vmax + std::fabs((double)K.t[1])) * 1.0001; bz = (1.0 + std::fabs((double)K.t[2])) * 1.0001; }
        for (int r = 0; r < 2; r++)
            ok = ok && std::fabs((double)K.kr[3 * r]) * bx + std::fabs((double)K.kr[3 * r + 1]) * by + std::fabs((double)K.kr[3 * r + 2]) * bz <= 0x1p59;
        K.num_ok = ok ? 1 : 0;
    }
    {
        volatile float omt = 1.f - K.t[2];  // volatile: every step rounded to fp32, no contraction
        volatile float c2 = K.kr[2] * omt, c5 = K.kr[5] * omt, c8 = K.kr[8] * omt;
        K.c2 = c2; K.c5 = c5; K.c8 = c8;
    }
    // plane / affine: z = (kr7 v' + kr6 u') + c8; with kr6 = kr7 = 0 and finite tables (num_ok) that is c8 for every pixel
    K.z_one = K.num_ok && L.proj.family == STX_F_PLANE && K.kr[6] == 0.f && K.kr[7] == 0.f && K.c8 == 1.0f ? 1 : 0;
    // algorithmic bytes: read the source once, write the warped image + mask once
    *bytes = (img ? 3.0 * L.sw * L.sh + 3.0 * L.dw * L.dh : 0.0) + (mask ? 1.0 * L.dw * L.dh : 0.0);
}

}  // namespace

// All launches of one call share the projector type and the (image, mask) output selection.
int stx_launch_warp_batch(stx_ctx* ctx, const StxWarpLaunch* Ls, int n)
{
    if (n <= 0) return STX_OK;
    std::vector<WarpK> Ks(n);
    std::vector<double> bytes(n);
    for (int i = 0; i < n; i++) fill_warpk(Ls[i], &Ks[i], &bytes[i]);
    const bool img = Ls[0].dimg != nullptr, mask = Ls[0].dmask != nullptr;
    const int dbg = Ls[0].debug_maps;  // stx_debug_warp_maps: dimg / dmask are the two float maps
    const char* name = dbg ? "warp_debug_maps" : (img ? (mask ? "warp_img_mask" : "warp_img") : "warp_mask");
    switch (Ls[0].proj.family) {
    case STX_F_PLANE: return launch_typed<STX_WARP_PLANE>(ctx, Ks.data(), n, img, mask, name, bytes.data(), dbg);
    case STX_F_CYLINDRICAL: return launch_typed<STX_WARP_CYLINDRICAL>(ctx, Ks.data(), n, img, mask, name, bytes.data(), dbg);
    case STX_F_SPHERICAL:
    case STX_F_MERCATOR:  // separable like the sphere (row table: warp_tables_kernel)
        return launch_typed<STX_WARP_SPHERICAL>(ctx, Ks.data(), n, img, mask, name, bytes.data(), dbg);
    default: return launch_general(ctx, Ks.data(), n, img, mask, name, bytes.data(), dbg);
    }
}

int stx_launch_warp(stx_ctx* ctx, const StxWarpLaunch& L) { return stx_launch_warp_batch(ctx, &L, 1); }

bool stx_warp_fast_eligible(const StxWarpLaunch& L)
{
    if (L.proj.family != STX_F_PLANE && L.proj.family != STX_F_CYLINDRICAL && L.proj.family != STX_F_SPHERICAL && L.proj.family != STX_F_MERCATOR) return false;
    WarpK K;
    double bytes;
    fill_warpk(L, &K, &bytes);
    return fast_ok(K) && !L.debug_maps && L.dimg != nullptr;
}

// out_minmax4[i] = {min u, min v, max u, max v} over the border of image i (cyl / spherical)
int stx_launch_roi_minmax(stx_ctx* ctx, int n, const StxProjector* projs, const int* sizes_wh, float* out_minmax4)
{
    // Argument blocks travel as kernel arguments; the per-block partial results are written by the kernel STRAIGHT into the context's
    // pinned host scratch, each followed by a stamp (this pass's sequence number, system-scope release).  The host polls the stamps:
    // no copy dispatch, no interrupt-driven wait — the ROI pass is the one place where a panorama's latency waits for the host
    // (82 us of idle device per panorama with copy + hipStreamSynchronize, profiles/r05_latency.md).  After ROI_SPIN_US without
    // all stamps (device busy with queued work) the wait becomes hipStreamSynchronize, which also covers any failure.
    // The pass runs on the context's side stream, so work already queued on the main stream keeps going.
    // blocks per image: 32 for the border walk, 256 when every source pixel is projected (one family per call)
    const bool full = n > 0 && projs[0].family > STX_F_SPHERICAL;
    const int BX = full ? 256 : 32;
    // results in the first three quarters of the scratch, stamps in the last one — the same split for both block counts, so a result
    // of one pass can never be read as a stamp of another
    const size_t stamp_off = ctx->pinned_bytes / 4 * 3;
    const int cap = (int)(stamp_off / (16 * BX));
    static const bool no_spin = getenv("STITCHING_AMD_ROI_NO_SPIN") != nullptr;  // diagnostic: A/B against the blocking wait
    for (int start = 0; start < n; start += cap) {
        const int cnt = std::min(cap, n - start);
        float* dout = (float*)ctx->pinned;
        uint32_t* stamps = (uint32_t*)((uint8_t*)ctx->pinned + stamp_off);
        uint32_t seq = ++ctx->roi_seq;
        if (seq == 0) seq = ++ctx->roi_seq;  // 0 = "never written"
        for (int base = 0; base < cnt; base += ROI_BATCH) {
            const int m = std::min(ROI_BATCH, cnt - base);
            RoiBatchK B;
            memset(&B, 0, sizeof(B));
            for (int i = 0; i < m; i++) {
                RoiK& k = B.k[i];
                const int g = start + base + i;
                for (int q = 0; q < 9; q++) k.rk[q] = projs[g].r_kinv[q];
                k.scale = projs[g].scale;
                k.type = projs[g].type;
                k.w = sizes_wh[2 * g];
                k.h = sizes_wh[2 * g + 1];
                k.family = projs[g].family; k.pa = projs[g].a; k.pb = projs[g].b;
                k.trig = projs[g].trig;
                k.full = full ? 1 : 0;
                if (full && (long long)k.w * k.h > 0x7fffffffll) return stx_fail(STX_ERR_UNSUPPORTED, "image of %dx%d pixels", k.w, k.h);
            }
            // algorithmic bytes: the per-block partial results (nothing is read: the points come from the arguments)
            StxProfScope prof(ctx, "warp_roi", 16.0 * BX * m, ctx->aux_stream);
            if (full) hipLaunchKernelGGL(roi_kernel<true>, dim3(BX, m), dim3(256), 0, ctx->aux_stream, B, dout + 4 * (size_t)BX * base, stamps + (size_t)BX * base, seq);
            else hipLaunchKernelGGL(roi_kernel<false>, dim3(BX, m), dim3(256), 0, ctx->aux_stream, B, dout + 4 * (size_t)BX * base, stamps + (size_t)BX * base, seq);
        }
        STX_HIP(hipGetLastError());
        const float* res = (const float*)ctx->pinned;
        bool seen = false;
        if (!no_spin) {
            const auto t0 = std::chrono::steady_clock::now();
            const size_t nst = (size_t)BX * cnt;
            size_t done = 0;  // stamps [0, done) have been seen
            for (unsigned spins = 0;; spins++) {
                while (done < nst && __atomic_load_n(stamps + done, __ATOMIC_ACQUIRE) == seq) done++;
                if (done == nst) { seen = true; break; }
                if ((spins & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(ROI_SPIN_US)) break;
                // a polite spin: the x86 pause hint where it exists, a scheduler yield elsewhere (the library is host-portable)
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#elif defined(__aarch64__)
                __asm__ __volatile__("yield");
#else
                std::this_thread::yield();
#endif
            }
        }
        if (!seen) STX_HIP(hipStreamSynchronize(ctx->aux_stream));
        else {
            // every stamp was seen, so the kernel ran to its last block; a sticky asynchronous error (of it or of anything before it on
            // this device) is still reported HERE, by the call that caused it, and not by some later, unrelated one
            hipError_t qe = hipStreamQuery(ctx->aux_stream);
            if (qe != hipSuccess && qe != hipErrorNotReady) STX_HIP(qe);
        }
        for (int i = 0; i < cnt; i++) {
            // NaN-ignoring fold in the comparison form of the device loop
            float mnu = 3.402823466e+38f, mnv = mnu, mxu = -mnu, mxv = -mnu;
            for (int b = 0; b < BX; b++) {
                const float* r = res + 4 * ((size_t)i * BX + b);
                if (r[0] < mnu) mnu = r[0];
                if (r[1] < mnv) mnv = r[1];
                if (r[2] > mxu) mxu = r[2];
                if (r[3] > mxv) mxv = r[3];
            }
            float* o = out_minmax4 + 4 * (size_t)(start + i);
            o[0] = mnu; o[1] = mnv; o[2] = mxu; o[3] = mxv;
        }
    }
    return STX_OK;
}
