// stx_blend.hip — multi-band (Laplacian pyramid), feather and "no" blenders for gfx950.
//
// Replaces, for stitching/blender.py:23-48, OpenCV's MultiBandBlender::{prepare,feed,blend},
// createLaplacePyr, restoreImageFromLaplacePyr, normalizeUsingWeightMap, pyrDown/pyrUp,
// copyMakeBorder, FeatherBlender (+createWeightMap/distanceTransform) and the base Blender.
//
// Multi-band is restructured for a 288 GB HBM part as a DEFERRED GATHER:
//   feed()   builds, per image, only the int16 Gaussian pyramid G_1..G_B (planar) and the fp32
//            weight pyramid W_1..W_B.  copyMakeBorder is an index map on load, the int16
//            conversion happens in registers, no Laplacian level is ever stored.
//   finish() runs ONE kernel per level, coarse to fine.  Each panorama pixel loops over the
//            images whose (2^B-aligned) feed rectangle covers it, forms
//            L = sat(G_i - pyrUp(G_{i+1})) on the fly, accumulates (short)(L*W) and W in
//            registers in feed order, normalises, adds pyrUp of the already finished coarser
//            level (saturating) and writes the level once.  Level 0 writes the u8 panorama,
//            the mask and (optionally) the int16 result directly.
// The accumulator pyramids dst_pyr_laplace_/dst_band_weights_ of OpenCV (13.3 B per panorama
// pixel, read-modify-written once per image) never exist in HBM.  Integer sums are exact in
// any order (int16 wrap-around adds); fp32 weight sums are taken in ascending feed order, the
// order OpenCV's += sees.
#include <cstring>

#include "stx_blend_kernels.h"
#include "stx_device_math.h"
#include "stx_internal.h"

using namespace stxd;

namespace {

constexpr float WEIGHT_EPS = 1e-5f;
constexpr float INV255 = 0.0039215688593685627f;  // (float)(1./255.)
constexpr float INV256 = 0.00390625f;

// ---------------------------------------------------------------------------------------------
// level-0 accessors: bordered image (copyMakeBorder REFLECT) and bordered weight (CONSTANT 0)
// ---------------------------------------------------------------------------------------------
template <bool S16>
STX_DEV void load_px0(const StxMbImage& im, int sx, int sy, int& b, int& g, int& r)
{
    if (S16) {
        const short* p = reinterpret_cast<const short*>(im.img0 + (long long)sy * im.img0_stride) + sx * 3;
        b = p[0]; g = p[1]; r = p[2];
    } else {
        const uint8_t* p = im.img0 + (long long)sy * im.img0_stride + sx * 3;
        b = p[0]; g = p[1]; r = p[2];
    }
}

// horizontal 1-4-6-4-1 of the fp32 weights in pyrDown_'s scalar evaluation order
STX_DEV float h5f(float s0, float s1, float s2, float s3, float s4)
{
    return fadd(fadd(fadd(fmul(s2, 6.f), fmul(fadd(s1, s3), 4.f)), s0), s4);
}

// ... and in the order of OpenCV's vector code (include/stitching_amd.h STX_PYRDOWN_*): row sums, column sums, fused or not
STX_DEV float h5f_simd(float s0, float s1, float s2, float s3, float s4, bool fused)
{
    const float outer = fadd(s0, s4), inner = fadd(s1, s3);
    const float t = fused ? __fmaf_rn(inner, 4.f, outer) : fadd(fmul(inner, 4.f), outer);
    return fused ? __fmaf_rn(s2, 6.f, t) : fadd(fmul(s2, 6.f), t);
}
STX_DEV float v5f_simd(float r0, float r1, float r2, float r3, float r4, bool fused)
{
    const float a = fadd(fadd(r1, r3), r2), b = fadd(fadd(r0, r4), fadd(r2, r2));
    return fused ? __fmaf_rn(a, 4.f, b) : fadd(fmul(a, 4.f), b);
}
// which outputs of a level w samples wide (dw outputs) the vector code forms: rows 1 <= x < hx1, columns x < vx1
struct PyrOrder { int hx1, vx1; bool fused; };
STX_DEV PyrOrder pyr_order(int mode, int lanes, int w, int dw)
{
    PyrOrder o;
    const int width0 = min((w - 3) / 2 + 1, dw);
    o.hx1 = (mode & 2) && width0 > 1 ? 1 + ((width0 - 1) / lanes) * lanes : 1;
    o.vx1 = (mode & 1) ? (dw / lanes) * lanes : 0;
    o.fused = (mode & 4) != 0;
    return o;
}

// sample `idx` (row * stride + column, + channel * plane) of level lv of an image's Gaussian pyramid: bytes when the image was
// fed as u8 (StxMbImage::g_u8), else int16
// (pointers read from a descriptor in memory are generic to the compiler; all of ours are device global memory: global_load
// instead of flat_load)
#define STX_GAS __attribute__((address_space(1)))
STX_DEV int ld_g(const StxMbImage& im, int lv, long long idx)
{
    return im.g_u8 ? (int)((const STX_GAS uint8_t*)reinterpret_cast<const uint8_t*>(im.g[lv]))[idx] : (int)((const STX_GAS short*)im.g[lv])[idx];
}
STX_DEV void st_g(const StxMbImage& im, int lv, long long idx, int v)
{
    if (im.g_u8) reinterpret_cast<uint8_t*>(im.g[lv])[idx] = (uint8_t)v;
    else im.g[lv][idx] = (short)v;
}
// sample `idx` of level lv of an image's weight pyramid: fp32, or the halves of level 1 of an image with a 0 / 255 mask (StxMbImage::w1_f16:
// k / 256, exact either way)
STX_DEV float ld_w(const StxMbImage& im, int lv, long long idx)
{
    if (lv == 1 && im.w1_f16) return (float)((const STX_GAS _Float16*)reinterpret_cast<const _Float16*>(im.wt[1]))[idx];
    return ((const STX_GAS float*)im.wt[lv])[idx];
}
STX_DEV void st_w(const StxMbImage& im, int lv, long long idx, float v)
{
    if (lv == 1 && im.w1_f16) reinterpret_cast<_Float16*>(im.wt[1])[idx] = (_Float16)v;
    else im.wt[lv][idx] = v;
}

// pyrDown of (bordered level 0) -> level 1: G_1 planar (u8 / int16), W_1 fp32
template <bool S16>
__global__ __launch_bounds__(256) void mb_down0_kernel(StxMbImage im, int pyr_mode, int pyr_lanes)
{
    const int ow = im.fw >> 1, oh = im.fh >> 1;
    const PyrOrder po = pyr_order(pyr_mode, pyr_lanes, im.fw, ow);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    int cx[5], mx[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        int bx = reflect101(2 * x - 2 + j, im.fw) - im.left;  // bordered -> image coords
        mx[j] = ((unsigned)bx < (unsigned)im.iw) ? bx : -1;
        cx[j] = reflect(bx, im.iw);
    }
    int vb[5], vg[5], vr[5];
    float vw[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        int by = reflect101(2 * y - 2 + k, im.fh) - im.top;
        bool yin = (unsigned)by < (unsigned)im.ih;
        int sy = reflect(by, im.ih);
        int b[5], g[5], r[5];
        float w[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            load_px0<S16>(im, cx[j], sy, b[j], g[j], r[j]);
            w[j] = (yin && mx[j] >= 0) ? fmul((float)im.mask0[(long long)by * im.mask0_stride + mx[j]], INV255) : 0.f;
        }
        vb[k] = b[2] * 6 + (b[1] + b[3]) * 4 + b[0] + b[4];
        vg[k] = g[2] * 6 + (g[1] + g[3]) * 4 + g[0] + g[4];
        vr[k] = r[2] * 6 + (r[1] + r[3]) * 4 + r[0] + r[4];
        vw[k] = (x >= 1 && x < po.hx1) ? h5f_simd(w[0], w[1], w[2], w[3], w[4], po.fused) : h5f(w[0], w[1], w[2], w[3], w[4]);
    }
    const long long o = (long long)y * im.g_stride[1] + x;
    st_g(im, 1, o, (vb[2] * 6 + (vb[1] + vb[3]) * 4 + vb[0] + vb[4] + 128) >> 8);
    st_g(im, 1, o + im.g_plane[1], (vg[2] * 6 + (vg[1] + vg[3]) * 4 + vg[0] + vg[4] + 128) >> 8);
    st_g(im, 1, o + 2 * im.g_plane[1], (vr[2] * 6 + (vr[1] + vr[3]) * 4 + vr[0] + vr[4] + 128) >> 8);
    const float col = x < po.vx1 ? v5f_simd(vw[0], vw[1], vw[2], vw[3], vw[4], po.fused) : h5f(vw[0], vw[1], vw[2], vw[3], vw[4]);
    st_w(im, 1, (long long)y * im.wt_stride[1] + x, fmul(col, INV256));
}

// pyrDown level i -> i+1 (i >= 1), planar int16 x3 + fp32
__global__ __launch_bounds__(256) void mb_down_kernel(StxMbImage im, int lv, int pyr_mode, int pyr_lanes)
{
    const int iw = im.fw >> lv, ih = im.fh >> lv;
    const int ow = iw >> 1, oh = ih >> 1;
    const PyrOrder po = pyr_order(pyr_mode, pyr_lanes, iw, ow);
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= ow || y >= oh) return;
    int cx[5];
#pragma unroll
    for (int j = 0; j < 5; j++) cx[j] = reflect101(2 * x - 2 + j, iw);
    const long long gs = im.g_stride[lv], gp = im.g_plane[lv];
    const long long ws = im.wt_stride[lv];
    int v[3][5];
    float vw[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        int sy = reflect101(2 * y - 2 + k, ih);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const long long row = c * gp + (long long)sy * gs;
            v[c][k] = ld_g(im, lv, row + cx[2]) * 6 + (ld_g(im, lv, row + cx[1]) + ld_g(im, lv, row + cx[3])) * 4 + ld_g(im, lv, row + cx[0]) +
                      ld_g(im, lv, row + cx[4]);
        }
        const long long wrow = (long long)sy * ws;
        const float w0 = ld_w(im, lv, wrow + cx[0]), w1 = ld_w(im, lv, wrow + cx[1]), w2 = ld_w(im, lv, wrow + cx[2]), w3 = ld_w(im, lv, wrow + cx[3]),
                    w4 = ld_w(im, lv, wrow + cx[4]);
        vw[k] = (x >= 1 && x < po.hx1) ? h5f_simd(w0, w1, w2, w3, w4, po.fused) : h5f(w0, w1, w2, w3, w4);
    }
#pragma unroll
    for (int c = 0; c < 3; c++)
        st_g(im, lv + 1, c * im.g_plane[lv + 1] + (long long)y * im.g_stride[lv + 1] + x,
             (v[c][2] * 6 + (v[c][1] + v[c][3]) * 4 + v[c][0] + v[c][4] + 128) >> 8);
    const float col = x < po.vx1 ? v5f_simd(vw[0], vw[1], vw[2], vw[3], vw[4], po.fused) : h5f(vw[0], vw[1], vw[2], vw[3], vw[4]);
    st_w(im, lv + 1, (long long)y * im.wt_stride[lv + 1] + x, fmul(col, INV256));
}

// pyrUp_<FixPtCast<short,6>> sampled at one destination pixel (X, Y) of a planar image (int16, or the bytes of a u8 pyramid)
template <class T>
STX_DEV int pyr_up_at(const T* __restrict__ plane, long long stride, int cw, int ch, int X, int Y)
{
    // Branch-free: all nine taps of the 3 x 3 window are loaded whatever the parities of X and Y and the unused ones get weight 0 —
    // a load behind a divergent parity branch is a memory round trip of its own (the three coarse levels' kernel spent 236 dependent
    // loads per wavefront that way); the integers summed are the same.
    const int px = X >> 1, py = Y >> 1;
    const int xl = up_idx(px - 1, cw), xr = up_idx(px + 1, cw);
    const int yt = up_idx(py - 1, ch), yb = up_idx(py + 1, ch);
    const T* rt = plane + (long long)yt * stride;
    const T* rc = plane + (long long)py * stride;
    const T* rb = plane + (long long)yb * stride;
    const int t0 = rt[xl], t1 = rt[px], t2 = rt[xr];
    const int c0 = rc[xl], c1 = rc[px], c2 = rc[xr];
    const int b0 = rb[xl], b1 = rb[px], b2 = rb[xr];
    // The tap weights are ARITHMETIC in the parity bits, not selects: a select between 0 and a loaded value is turned back into a
    // branch around the load by the compiler.
    const int ox = X & 1, oy = Y & 1;
    const int wl = 1 - ox, wc = 6 - 2 * ox, wr = 1 + 3 * ox;   // odd column: (centre + right) * 4; even: left + 6 centre + right
    const int ht = __mul24(t0, wl) + __mul24(t1, wc) + __mul24(t2, wr);
    const int hc = __mul24(c0, wl) + __mul24(c1, wc) + __mul24(c2, wr);
    const int hb = __mul24(b0, wl) + __mul24(b1, wc) + __mul24(b2, wr);
    const int v = __mul24(ht, 1 - oy) + __mul24(hc, 6 - 2 * oy) + __mul24(hb, 1 + 3 * oy);   // odd row: (centre + below) * 4
    return (int)(short)((v + 32) >> 6);
}
// ... of channel c of level lv of an image's Gaussian pyramid
STX_DEV int pyr_up_g(const StxMbImage& im, int lv, int c, int cw, int ch, int X, int Y)
{
    if (im.g_u8) return pyr_up_at((const STX_GAS uint8_t*)reinterpret_cast<const uint8_t*>(im.g[lv]) + c * im.g_plane[lv], im.g_stride[lv], cw, ch, X, Y);
    return pyr_up_at((const STX_GAS short*)im.g[lv] + c * im.g_plane[lv], im.g_stride[lv], cw, ch, X, Y);
}


// gather + normalise + collapse of one level (generic, one pixel per lane; all levels, all kinds)
template <bool L0>
STX_DEV void mb_level_pixel(const MbLevelK& P, const int x, const int y);

template <bool L0>
STX_DEV void mb_level_body(const MbLevelK& P)
{
    const int x = P.x0 + blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = P.y0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.x1 || y >= P.y1) return;
    mb_level_pixel<L0>(P, x, y);
}

// one sample (x, y) of a level: every image in feed order, normalise, collapse, store
template <bool L0>
STX_DEV void mb_level_pixel(const MbLevelK& P, const int x, const int y)
{
    const int lv = P.level;
    int acc0 = 0, acc1 = 0, acc2 = 0;
    float ws = 0.f;
    for (int k = 0; k < P.n_images; k++) {
        const StxMbImage& im = P.images[k];
        if (im.kind == 1 || !L0) {
            const int lx = x - (im.fx >> lv), ly = y - (im.fy >> lv);
            const int lw = im.fw >> lv, lh = im.fh >> lv;
            if ((unsigned)lx >= (unsigned)lw || (unsigned)ly >= (unsigned)lh) continue;
            const float w = ld_w(im, lv, (long long)ly * im.wt_stride[lv] + lx);
            const long long gi = (long long)ly * im.g_stride[lv] + lx;
            int L[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                int gval = ld_g(im, lv, gi + c * im.g_plane[lv]);
                if (im.kind == 0 && lv < P.num_bands) gval = sat_s16(gval - pyr_up_g(im, lv + 1, c, lw >> 1, lh >> 1, lx, ly));
                L[c] = gval;
            }
            if (im.kind == 1) {  // already (short)(L * W)
                acc0 += L[0]; acc1 += L[1]; acc2 += L[2];
            } else {
                acc0 += trunc_s16(fmul((float)L[0], w));
                acc1 += trunc_s16(fmul((float)L[1], w));
                acc2 += trunc_s16(fmul((float)L[2], w));
            }
            ws = fadd(ws, w);
        } else {
            // level 0 of a fed image: outside the image itself the bordered weight is the constant 0:
            // (short)(L*0) = 0, w += 0
            const int lx = x - im.ix, ly = y - im.iy;
            if ((unsigned)lx >= (unsigned)im.iw || (unsigned)ly >= (unsigned)im.ih) continue;
            const float w = fmul((float)im.mask0[(long long)ly * im.mask0_stride + lx], INV255);
            int L[3];
            if (im.img0_is_s16) load_px0<true>(im, lx, ly, L[0], L[1], L[2]);
            else load_px0<false>(im, lx, ly, L[0], L[1], L[2]);
            if (P.num_bands > 0) {
                const int bx = x - im.fx, by = y - im.fy;
#pragma unroll
                for (int c = 0; c < 3; c++) L[c] = sat_s16(L[c] - pyr_up_g(im, 1, c, im.fw >> 1, im.fh >> 1, bx, by));
            }
            acc0 += trunc_s16(fmul((float)L[0], w));
            acc1 += trunc_s16(fmul((float)L[1], w));
            acc2 += trunc_s16(fmul((float)L[2], w));
            ws = fadd(ws, w);
        }
    }
    if (P.emit) {  // contribution strip for another rank: un-normalised
        short* O = P.out + (long long)(y - P.out_y0) * P.out_stride + (x - P.out_x0);
        O[0] = (short)acc0; O[P.out_plane] = (short)acc1; O[2 * P.out_plane] = (short)acc2;
        P.out_w[(long long)(y - P.out_y0) * P.out_w_stride + (x - P.out_x0)] = ws;
        return;
    }
    const float den = fadd(ws, WEIGHT_EPS);
    int v[3];
    v[0] = trunc_s16(fdiv((float)(short)acc0, den));
    v[1] = trunc_s16(fdiv((float)(short)acc1, den));
    v[2] = trunc_s16(fdiv((float)(short)acc2, den));
    if (P.up) {
        const short* U = P.up - ((long long)P.up_y0 * P.up_stride + P.up_x0);
#pragma unroll
        for (int c = 0; c < 3; c++)
            v[c] = sat_s16(pyr_up_at(U + c * P.up_plane, P.up_stride, P.pw >> 1, P.ph >> 1, x, y) + v[c]);
    }
    if (!L0) {
        short* O = P.out + (long long)(y - P.out_y0) * P.out_stride + (x - P.out_x0);
        O[0] = (short)v[0]; O[P.out_plane] = (short)v[1]; O[2 * P.out_plane] = (short)v[2];
        return;
    }
    const bool keep = ws > WEIGHT_EPS;  // compare(dst_band_weights_0, WEIGHT_EPS, CMP_GT); setTo(0, !mask)
    if (!keep) v[0] = v[1] = v[2] = 0;
    const int ox = x - P.pano_x0, oy = y - P.pano_y0;
    uint8_t* o = P.pano + (long long)oy * P.pano_stride + ox * 3;
    // convertScaleAbs: saturate_cast<uchar>(|x|)
    o[0] = (uint8_t)min(abs(v[0]), 255);
    o[1] = (uint8_t)min(abs(v[1]), 255);
    o[2] = (uint8_t)min(abs(v[2]), 255);
    P.pmask[(long long)oy * P.pmask_stride + ox] = keep ? 255 : 0;
    if (P.pano16) {
        short* o16 = reinterpret_cast<short*>(reinterpret_cast<uint8_t*>(P.pano16) + (long long)oy * P.pano16_stride) + ox * 3;
        o16[0] = (short)v[0]; o16[1] = (short)v[1]; o16[2] = (short)v[2];
    }
}


// ---------------------------------------------------------------------------------------------
// The three coarsest levels B, B-1, B-2 in ONE launch (round 3).  They are tiny (config 2: 52 k, 209 k and 832 k samples) and
// each is a single wave of workgroups whose time is memory latency, not bandwidth: as three dependent launches they cost
// 8 + 19 + 33 us of an otherwise idle GPU.  Here a workgroup owns a 32 x 16 tile of level B-2 and recomputes, in LDS, the
// finished samples of level B-1 (18 x 10) and of level B (at most 12 x 8) that its tile's pyrUp chain reaches — the halo is recomputed
// instead of being exchanged between workgroups, so the levels B and B-1 never exist in memory and nothing waits for a grid.
// Per sample this is mb_level_body's arithmetic (same loads, same fp32 operations in the same order); only the finished coarser
// level comes from LDS instead of HBM.
// ---------------------------------------------------------------------------------------------
constexpr int CO_TW = 32, CO_TH = 16;            // tile of level B-2
constexpr int CO_W1 = CO_TW / 2 + 2, CO_H1 = CO_TH / 2 + 2;   // level B-1 samples a tile can reach: 18 x 10
constexpr int CO_W0 = CO_W1 / 2 + 3, CO_H0 = CO_H1 / 2 + 3;   // level B: 12 x 8 (a window that starts on an odd sample reaches one more)

// the Laplacian (kind 0, below the top level) or the stored value of one sample of one image: every load of the sample — the value
// itself and the 3 x 9 pyrUp taps of the next level — is issued before the first use (the element type is decided once, not per channel)
template <class T>
STX_DEV void mb_sample_of(const StxMbImage& im, bool lap, int lv, int lx, int ly, int lw, int lh, int (&L)[3])
{
    const STX_GAS T* g = (const STX_GAS T*)reinterpret_cast<const T*>(im.g[lv]);
    const long long gi = (long long)ly * im.g_stride[lv] + lx;
    int gv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) gv[c] = (int)g[gi + c * im.g_plane[lv]];
    if (lap) {
        const STX_GAS T* g1 = (const STX_GAS T*)reinterpret_cast<const T*>(im.g[lv + 1]);
        int up[3];
#pragma unroll
        for (int c = 0; c < 3; c++) up[c] = pyr_up_at(g1 + c * im.g_plane[lv + 1], im.g_stride[lv + 1], lw >> 1, lh >> 1, lx, ly);
#pragma unroll
        for (int c = 0; c < 3; c++) gv[c] = sat_s16(gv[c] - up[c]);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) L[c] = gv[c];
}

// gather + normalise of ONE sample of level lv >= 1 (the loop body of mb_level_body<false> without the store)
STX_DEV void mb_gather_norm(const StxMbImage* __restrict__ images, int n_images, int num_bands, int lv, int x, int y, int (&v)[3])
{
    int acc0 = 0, acc1 = 0, acc2 = 0;
    float ws = 0.f;
    for (int k = 0; k < n_images; k++) {
        const StxMbImage& im = images[k];
        const int lx = x - (im.fx >> lv), ly = y - (im.fy >> lv);
        const int lw = im.fw >> lv, lh = im.fh >> lv;
        if ((unsigned)lx >= (unsigned)lw || (unsigned)ly >= (unsigned)lh) continue;
        const float w = ld_w(im, lv, (long long)ly * im.wt_stride[lv] + lx);
        const bool lap = im.kind == 0 && lv < num_bands;
        int L[3];
        if (im.g_u8) mb_sample_of<uint8_t>(im, lap, lv, lx, ly, lw, lh, L);
        else mb_sample_of<short>(im, lap, lv, lx, ly, lw, lh, L);
        if (im.kind == 1) {  // already (short)(L * W)
            acc0 += L[0]; acc1 += L[1]; acc2 += L[2];
        } else {
            acc0 += trunc_s16(fmul((float)L[0], w));
            acc1 += trunc_s16(fmul((float)L[1], w));
            acc2 += trunc_s16(fmul((float)L[2], w));
        }
        ws = fadd(ws, w);
    }
    const float den = fadd(ws, WEIGHT_EPS);
    v[0] = trunc_s16(fdiv((float)(short)acc0, den));
    v[1] = trunc_s16(fdiv((float)(short)acc1, den));
    v[2] = trunc_s16(fdiv((float)(short)acc2, den));
}

// pyr_up_at on a window of the coarser level held in LDS: plane[(row - oy) * pitch + (col - ox)], level size cw x ch
STX_DEV int pyr_up_at_lds(const short* __restrict__ plane, int pitch, int ox, int oy, int cw, int ch, int X, int Y)
{
    const int px = X >> 1, py = Y >> 1;
    const int xl = up_idx(px - 1, cw) - ox, xc = px - ox, xr = up_idx(px + 1, cw) - ox;
    const short* rc = plane + (py - oy) * pitch;
    const short* rb = plane + (up_idx(py + 1, ch) - oy) * pitch;
    int hc, hb, v;
    if (X & 1) {
        hc = (rc[xc] + rc[xr]) * 4;
        hb = (rb[xc] + rb[xr]) * 4;
    } else {
        hc = rc[xl] + rc[xc] * 6 + rc[xr];
        hb = rb[xl] + rb[xc] * 6 + rb[xr];
    }
    if (Y & 1) {
        v = (hc + hb) * 4;
    } else {
        const short* rt = plane + (up_idx(py - 1, ch) - oy) * pitch;
        const int ht = (X & 1) ? (rt[xc] + rt[xr]) * 4 : rt[xl] + rt[xc] * 6 + rt[xr];
        v = ht + hc * 6 + hb;
    }
    return (int)(short)((v + 32) >> 6);
}

struct MbCoarseK {
    const StxMbImage* images;
    int n_images, num_bands;
    int x0, x1, y1;          // region of level B-2 that is produced: [x0, x1) x [0, y1)
    short* out; long long out_stride, out_plane; int out_x0;  // finished level B-2 (planar int16), origin (out_x0, 0)
    int pw, ph;              // padded panorama size at level B-2 (a multiple of 4)
};

// ---- round 6: one memory round trip per covering image instead of four ------------------------------------------------------------
// The general path below walks, per sample, all images of the table one after the other: a scalar round trip for the descriptor, a
// branch, then the loads of that ONE image (weight, 3 values, 27 pyrUp taps) and their wait — for every covering image again, in three
// phases of one, one and two samples per lane: (1 + 1 + 2) x (covering images) dependent memory round trips per wavefront, 30 us of
// nothing but latency on config 2 (profiles/r05_latency.md).  The fast path
//   * lists the images that reach the workgroup's three windows ONCE (a ballot per 64 images, feed order kept) and parks the
//     descriptors of their three levels in LDS (scalar registers again by v_readfirstlane: no scalar-cache round trip per image);
//   * gives every lane samples that share ONE parent sample — wavefronts 0, 1: a 2 x 2 block of level B-2; wavefronts 2, 3: a horizontal
//     pair of level B-1 and one sample of level B — so that the pyrUp taps of a lane are one 3 x 3 window, read as ONE unaligned dword per
//     row and plane (a byte pyramid): 17 loads per lane and image where the general path issues up to 124;
//   * gathers inside one loop over the list, branch-free: addresses clamped into the image's rectangle, the contribution of a sample
//     outside it multiplied out (weight 0.f, Laplacian 0: acc += 0, ws + 0.f = ws exactly), every load of an image issued before the
//     first is waited for (__builtin_amdgcn_sched_barrier: left alone, the scheduler interleaves loads and uses to save registers).
// Only the two pyrUp additions still wait for each other (levels B and B-1 finished in LDS, as before).  Integers and fp32 operations
// per sample are those of mb_gather_norm, in the same order; taken whenever every listed image is a u8 image fed on this rank whose
// level B is at least 4 samples wide (anything else — int16 images, received contribution strips, tiny levels, more than CO_MAXC
// images over one tile — takes the general path, uniformly per workgroup).
constexpr int CO_MAXC = 16;

struct __attribute__((aligned(16))) CoLevel {  // level lv of one listed image, wave-uniform (12 dwords: three 16-byte LDS reads)
    int ox, oy, lw, lh;                 // its rectangle at this level (panorama coordinates of the level)
    uint32_t gs, gpl, ws, pad;          // sample pitch of a row / of a plane of G, of a row of W
    const STX_GAS uint8_t* g;           // G_lv, byte planes
    const STX_GAS float* w;             // W_lv
};
STX_DEV CoLevel co_level(const StxMbImage& im, int lv)
{
    CoLevel L;
    L.ox = im.fx >> lv; L.oy = im.fy >> lv; L.lw = im.fw >> lv; L.lh = im.fh >> lv;
    L.gs = (uint32_t)im.g_stride[lv]; L.gpl = (uint32_t)im.g_plane[lv]; L.ws = (uint32_t)im.wt_stride[lv]; L.pad = 0u;
    L.g = (const STX_GAS uint8_t*)reinterpret_cast<const uint8_t*>(im.g[lv]);
    L.w = (const STX_GAS float*)im.wt[lv];
    return L;
}
// a listed level out of LDS into scalar registers (the address is wave-uniform; the compiler does not know the values are)
STX_DEV CoLevel co_level_lds(const CoLevel* p)
{
    const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
    uint32_t d[12];
#pragma unroll
    for (int i = 0; i < 12; i++) d[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q[i]);
    CoLevel L;
    L.ox = (int)d[0]; L.oy = (int)d[1]; L.lw = (int)d[2]; L.lh = (int)d[3];
    L.gs = d[4]; L.gpl = d[5]; L.ws = d[6]; L.pad = 0u;
    L.g = (const STX_GAS uint8_t*)(uintptr_t)((unsigned long long)d[8] | ((unsigned long long)d[9] << 32));
    L.w = (const STX_GAS float*)(uintptr_t)((unsigned long long)d[10] | ((unsigned long long)d[11] << 32));
    return L;
}
typedef uint32_t co_u32_u __attribute__((aligned(1)));
typedef uint16_t co_u16_u __attribute__((aligned(1)));
typedef float co_f2_u __attribute__((ext_vector_type(2), aligned(4)));
// the three pyrUp tap rows of the samples whose parent is (px, py) of level U: one unaligned dword per row and plane, 4 bytes from column s
struct CoTaps { uint32_t t[3][3]; };
STX_DEV int co_tap_col(int px, int cw) { return min(max(px - 1, 0), cw - 4); }  // cw >= 4: the workgroup's test of the path
STX_DEV void co_issue_taps(const CoLevel& U, int px, int py, CoTaps& T)
{
    const uint32_t s = (uint32_t)co_tap_col(px, U.lw);
    const uint32_t r0 = (uint32_t)up_idx(py - 1, U.lh) * U.gs + s, r1 = (uint32_t)py * U.gs + s, r2 = (uint32_t)up_idx(py + 1, U.lh) * U.gs + s;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const STX_GAS uint8_t* pl = U.g + (uint32_t)c * U.gpl;
        T.t[c][0] = *reinterpret_cast<const STX_GAS co_u32_u*>(pl + r0);
        T.t[c][1] = *reinterpret_cast<const STX_GAS co_u32_u*>(pl + r1);
        T.t[c][2] = *reinterpret_cast<const STX_GAS co_u32_u*>(pl + r2);
    }
}
// (32-bit BYTE offsets from the scalar bases: an element index would be widened to a 64-bit lane address)
STX_DEV float co_w1(const CoLevel& L, int cx, int cy)
{
    return *reinterpret_cast<const STX_GAS float*>(reinterpret_cast<const STX_GAS uint8_t*>(L.w) + (((uint32_t)cy * L.ws + (uint32_t)cx) << 2));
}
STX_DEV co_f2_u co_w2(const CoLevel& L, int cx, int cy)  // W at (cx, cy), (cx + 1, cy)
{
    return *reinterpret_cast<const STX_GAS co_f2_u*>(reinterpret_cast<const STX_GAS uint8_t*>(L.w) + (((uint32_t)cy * L.ws + (uint32_t)cx) << 2));
}
STX_DEV uint32_t co_g1(const CoLevel& L, int c, int cx, int cy) { return L.g[(uint32_t)c * L.gpl + (uint32_t)cy * L.gs + (uint32_t)cx]; }
STX_DEV uint32_t co_g2(const CoLevel& L, int c, int cx, int cy)  // G_c at (cx, cy) in byte 0, at (cx + 1, cy) in byte 1
{
    return *reinterpret_cast<const STX_GAS co_u16_u*>(L.g + ((uint32_t)c * L.gpl + (uint32_t)cy * L.gs + (uint32_t)cx));
}
// pyrUp_ (pyr_up_at's integers) of channel c at a sample of column parity ox and row parity oy whose parent column is px
STX_DEV int co_up(const CoTaps& T, int c, int px, int ox, int oy, int cw)
{
    const int s = co_tap_col(px, cw);
    // byte positions of the three taps inside the window (pyrUp's border rule: -1 -> 1, cw -> cw - 1)
    const uint32_t bl = 8u * (uint32_t)(up_idx(px - 1, cw) - s), bc = 8u * (uint32_t)(px - s), br = 8u * (uint32_t)(up_idx(px + 1, cw) - s);
    const int wl = 1 - ox, wc = 6 - 2 * ox, wr = 1 + 3 * ox;
    int h[3];
#pragma unroll
    for (int r = 0; r < 3; r++)
        h[r] = __mul24((int)__builtin_amdgcn_ubfe(T.t[c][r], bl, 8u), wl) + __mul24((int)__builtin_amdgcn_ubfe(T.t[c][r], bc, 8u), wc) +
               __mul24((int)__builtin_amdgcn_ubfe(T.t[c][r], br, 8u), wr);
    const int v = __mul24(h[0], 1 - oy) + __mul24(h[1], 6 - 2 * oy) + __mul24(h[2], 1 + 3 * oy);
    return (int)(short)((v + 32) >> 6);
}
// the accumulation step of mb_gather_norm for one sample of one image: Lp = its Laplacian (its value at the top level), w its weight;
// a sample outside the image adds 0 / 0.f (acc + 0, ws + 0.f = ws exactly)
STX_DEV void co_acc(int Lp, float wv, bool in, int& acc) { acc += trunc_s16(fmul((float)(in ? Lp : 0), in ? wv : 0.f)); }
STX_DEV void co_normalise(const int (&acc)[3], float ws, int (&v)[3])
{
    const float den = fadd(ws, WEIGHT_EPS);
#pragma unroll
    for (int c = 0; c < 3; c++) v[c] = trunc_s16(fdiv((float)(short)acc[c], den));
}

// 7 wavefronts per SIMD (72 registers; the unconstrained build takes 78 -> 6): config 2's 1 680 workgroups are one resident set on
// 256 CUs x 7, and a second round of a latency-bound kernel doubles its time
#ifndef STX_COARSE_WAVES
#define STX_COARSE_WAVES 7
#endif
#ifndef STX_COARSE_FAST
#define STX_COARSE_FAST 1
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(STX_COARSE_WAVES, 8))) void mb_coarse_kernel(MbCoarseK P)
{
    __shared__ short s0[3][CO_H0 * CO_W0];  // finished level B
    __shared__ short s1[3][CO_H1 * CO_W1];  // finished level B-1
    __shared__ CoLevel s_ent[CO_MAXC][3];  // [listed image][B, B-1, B-2]
    __shared__ int s_cnt;
    const int tid = threadIdx.x;
    const int B = P.num_bands;
    const int tx0 = P.x0 + blockIdx.x * CO_TW, ty0 = blockIdx.y * CO_TH;       // tile origin, level B-2
    const int tx1 = min(tx0 + CO_TW, P.x1), ty1 = min(ty0 + CO_TH, P.y1);    // exclusive
    const int pw1 = P.pw >> 1, ph1 = P.ph >> 1, pw0 = P.pw >> 2, ph0 = P.ph >> 2;
    // level B-1 samples the tile's pyrUp reaches: columns (x >> 1) - 1 .. (x >> 1) + 1 of every x in the tile, clipped (the border
    // rule of pyrUp maps -1 -> 1 and n -> n - 1, both inside the clipped window)
    const int ax0 = max((tx0 >> 1) - 1, 0), ax1 = min(((tx1 - 1) >> 1) + 1, pw1 - 1);
    const int ay0 = max((ty0 >> 1) - 1, 0), ay1 = min(((ty1 - 1) >> 1) + 1, ph1 - 1);
    const int aw = ax1 - ax0 + 1, ah = ay1 - ay0 + 1;
    // ... and the level B samples those reach
    const int bx0 = max((ax0 >> 1) - 1, 0), bx1 = min((ax1 >> 1) + 1, pw0 - 1);
    const int by0 = max((ay0 >> 1) - 1, 0), by1 = min((ay1 >> 1) + 1, ph0 - 1);
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
#if STX_COARSE_FAST
    // the images that reach any of the three windows, in feed order, their three levels' descriptors in LDS; s_cnt = -1: the general path
    if (tid < 64) {
        int cnt = 0;
        bool slow = false;
        for (int base = 0; base < P.n_images; base += 64) {
            const int k = min(base + tid, P.n_images - 1);
            const StxMbImage& im = P.images[k];
            const int kind = im.kind, g_u8 = im.g_u8, w1h = im.w1_f16;  // every field is loaded, whatever the tests below say (no load behind a branch)
            const int fxB = im.fx >> B, fyB = im.fy >> B, fwB = im.fw >> B, fhB = im.fh >> B;
            const CoLevel L0 = co_level(im, B), L1 = co_level(im, B - 1), L2 = co_level(im, B - 2);
            const bool hitB = fxB <= bx1 && fxB + fwB > bx0 && fyB <= by1 && fyB + fhB > by0;
            const bool hitA = 2 * fxB <= ax1 && 2 * (fxB + fwB) > ax0 && 2 * fyB <= ay1 && 2 * (fyB + fhB) > ay0;
            const bool hitT = 4 * fxB < tx1 && 4 * (fxB + fwB) > tx0 && 4 * fyB < ty1 && 4 * (fyB + fhB) > ty0;
            const bool rel = base + tid < P.n_images && (hitB || hitA || hitT);
            const bool odd = rel && (kind != 0 || !g_u8 || fwB < 4 || fhB < 1 || (w1h && B <= 3));  // (level 1 as halves: the general path reads them)
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(rel);
            slow = slow || __builtin_amdgcn_ballot_w64(odd) != 0ull;
            const int pos = cnt + __builtin_popcountll(bal & ((1ull << tid) - 1ull));
            if (rel && pos < CO_MAXC) { s_ent[pos][0] = L0; s_ent[pos][1] = L1; s_ent[pos][2] = L2; }
            cnt += __builtin_popcountll(bal);
        }
        if (tid == 0) s_cnt = (slow || cnt > CO_MAXC || (P.x0 & 1)) ? -1 : cnt;
    }
    __syncthreads();
    const int cnt = __builtin_amdgcn_readfirstlane(s_cnt);
    if (cnt >= 0) {
        // Two roles, 17 loads per lane and listed image each, all in flight before the first is used:
        //   wavefronts 0, 1: lane t owns the 2 x 2 block of level B-2 at (tx0 + 2 (t & 15), ty0 + 2 (t >> 4)) — one parent sample, so its four
        //                    samples share nine tap dwords; their values are two bytes per plane and row, their weights one 8-byte load per row;
        //   wavefronts 2, 3: lane u owns a horizontal pair of level B-1 (again one parent: nine tap dwords) and one sample of level B.
        // Even coordinates stay even inside every image: feed rectangles start and end on multiples of 2^B.
        const int role = __builtin_amdgcn_readfirstlane(tid >> 7);
        int accT[2][2][3] = {{{0, 0, 0}, {0, 0, 0}}, {{0, 0, 0}, {0, 0, 0}}};
        float wsT[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        int accA[2][3] = {{0, 0, 0}, {0, 0, 0}}, accB[3] = {0, 0, 0};
        float wsA[2] = {0.f, 0.f}, wsB = 0.f;
        const int t = tid & 127;
        // role 0
        const int xT = tx0 + 2 * (t & 15), yT = ty0 + 2 * (t >> 4);
        const bool hasT = xT < tx1 && yT < ty1;
        // role 1
        const int pc0 = ax0 >> 1, npc = (ax1 >> 1) - pc0 + 1;  // pair columns of the level B-1 window
        const bool hasP = t < npc * ah;
        const int rowA = hasP ? t / npc : 0, pcA = hasP ? t - rowA * npc : 0;
        const int xA = 2 * (pc0 + pcA), yA = ay0 + rowA;  // samples (xA, yA), (xA + 1, yA); inside the window iff ax0 <= x <= ax1
        const bool hasB = t < bw * bh;
        const int yyB = hasB ? t / bw : 0, xxB = hasB ? t - yyB * bw : 0;
        const int xB = bx0 + xxB, yB = by0 + yyB;
        if (role == 0) {
            for (int e = 0; e < cnt; e++) {
                const CoLevel LA = co_level_lds(&s_ent[e][1]), LT = co_level_lds(&s_ent[e][2]);
                const int lx = xT - LT.ox, ly = yT - LT.oy;
                const bool in = hasT && (unsigned)lx < (unsigned)LT.lw && (unsigned)ly < (unsigned)LT.lh;
                const int cx = min(max(lx, 0), LT.lw - 2), cy = min(max(ly, 0), LT.lh - 2);  // even, the block inside the rectangle
                co_f2_u w[2];
                uint32_t g[3][2];
                CoTaps T;
                w[0] = co_w2(LT, cx, cy); w[1] = co_w2(LT, cx, cy + 1);
#pragma unroll
                for (int c = 0; c < 3; c++) { g[c][0] = co_g2(LT, c, cx, cy); g[c][1] = co_g2(LT, c, cx, cy + 1); }
                co_issue_taps(LA, cx >> 1, cy >> 1, T);
                __builtin_amdgcn_sched_barrier(0);  // every load of the image has left before the first is waited for
#pragma unroll
                for (int dy = 0; dy < 2; dy++) {
#pragma unroll
                    for (int dx = 0; dx < 2; dx++) {
                        const float wv = dx ? w[dy].y : w[dy].x;
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const int gv = (int)((g[c][dy] >> (8 * dx)) & 255u);
                            co_acc(sat_s16(gv - co_up(T, c, cx >> 1, dx, dy, LA.lw)), wv, in, accT[dy][dx][c]);
                        }
                        wsT[dy][dx] = fadd(wsT[dy][dx], in ? wv : 0.f);
                    }
                }
            }
        } else {
            for (int e = 0; e < cnt; e++) {
                const CoLevel LB = co_level_lds(&s_ent[e][0]), LA = co_level_lds(&s_ent[e][1]);
                const int lx = xA - LA.ox, ly = yA - LA.oy;
                const bool inA = hasP && (unsigned)lx < (unsigned)LA.lw && (unsigned)ly < (unsigned)LA.lh;
                const int cx = min(max(lx, 0), LA.lw - 2), cy = min(max(ly, 0), LA.lh - 1);  // cx even
                const int lxB = xB - LB.ox, lyB = yB - LB.oy;
                const bool inB = hasB && (unsigned)lxB < (unsigned)LB.lw && (unsigned)lyB < (unsigned)LB.lh;
                const int cxB = min(max(lxB, 0), LB.lw - 1), cyB = min(max(lyB, 0), LB.lh - 1);
                uint32_t g[3], gB[3];
                CoTaps T;
                const co_f2_u w = co_w2(LA, cx, cy);
#pragma unroll
                for (int c = 0; c < 3; c++) g[c] = co_g2(LA, c, cx, cy);
                co_issue_taps(LB, cx >> 1, cy >> 1, T);
                const float wB = co_w1(LB, cxB, cyB);
#pragma unroll
                for (int c = 0; c < 3; c++) gB[c] = co_g1(LB, c, cxB, cyB);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dx = 0; dx < 2; dx++) {
                    const float wv = dx ? w.y : w.x;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const int gv = (int)((g[c] >> (8 * dx)) & 255u);
                        co_acc(sat_s16(gv - co_up(T, c, cx >> 1, dx, cy & 1, LB.lw)), wv, inA, accA[dx][c]);
                    }
                    wsA[dx] = fadd(wsA[dx], inA ? wv : 0.f);
                }
#pragma unroll
                for (int c = 0; c < 3; c++) co_acc((int)gB[c], wB, inB, accB[c]);  // the top level: L_B = G_B
                wsB = fadd(wsB, inB ? wB : 0.f);
            }
        }
        int v[3];
        if (role == 1 && hasB) {
            co_normalise(accB, wsB, v);
#pragma unroll
            for (int c = 0; c < 3; c++) s0[c][yyB * CO_W0 + xxB] = (short)v[c];
        }
        __syncthreads();
        if (role == 1 && hasP) {
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                const int x = xA + dx;
                if (x < ax0 || x > ax1) continue;
                co_normalise(accA[dx], wsA[dx], v);
#pragma unroll
                for (int c = 0; c < 3; c++)
                    s1[c][rowA * CO_W1 + (x - ax0)] = (short)sat_s16(pyr_up_at_lds(s0[c], CO_W0, bx0, by0, pw0, ph0, x, yA) + v[c]);
            }
        }
        __syncthreads();
        if (role == 0 && hasT) {
#pragma unroll
            for (int dy = 0; dy < 2; dy++) {
#pragma unroll
                for (int dx = 0; dx < 2; dx++) {
                    const int x = xT + dx, y = yT + dy;
                    if (x >= tx1 || y >= ty1) continue;
                    co_normalise(accT[dy][dx], wsT[dy][dx], v);
                    short* O = P.out + (long long)y * P.out_stride + (x - P.out_x0);
#pragma unroll
                    for (int c = 0; c < 3; c++) O[c * P.out_plane] = (short)sat_s16(pyr_up_at_lds(s1[c], CO_W1, ax0, ay0, pw1, ph1, x, y) + v[c]);
                }
            }
        }
        return;
    }
#endif
    for (int i = tid; i < bw * bh; i += 256) {  // level B: L_B = G_B, nothing above it
        const int yy = i / bw, xx = i - yy * bw;
        int v[3];
        mb_gather_norm(P.images, P.n_images, B, B, bx0 + xx, by0 + yy, v);
#pragma unroll
        for (int c = 0; c < 3; c++) s0[c][yy * CO_W0 + xx] = (short)v[c];
    }
    __syncthreads();
    for (int i = tid; i < aw * ah; i += 256) {  // level B-1
        const int yy = i / aw, xx = i - yy * aw;
        int v[3];
        mb_gather_norm(P.images, P.n_images, B, B - 1, ax0 + xx, ay0 + yy, v);
#pragma unroll
        for (int c = 0; c < 3; c++)
            s1[c][yy * CO_W1 + xx] = (short)sat_s16(pyr_up_at_lds(s0[c], CO_W0, bx0, by0, pw0, ph0, ax0 + xx, ay0 + yy) + v[c]);
    }
    __syncthreads();
    for (int i = tid; i < CO_TW * CO_TH; i += 256) {  // level B-2 -> memory
        const int yy = i / CO_TW, xx = i - yy * CO_TW;
        const int x = tx0 + xx, y = ty0 + yy;
        if (x >= tx1 || y >= ty1) continue;
        int v[3];
        mb_gather_norm(P.images, P.n_images, B, B - 2, x, y, v);
        short* O = P.out + (long long)y * P.out_stride + (x - P.out_x0);
#pragma unroll
        for (int c = 0; c < 3; c++)
            O[c * P.out_plane] = (short)sat_s16(pyr_up_at_lds(s1[c], CO_W1, ax0, ay0, pw1, ph1, x, y) + v[c]);
    }
}

template <bool L0>
__global__ __launch_bounds__(256) void mb_level_kernel(MbLevelK P)
{
    mb_level_body<L0>(P);
}

// batched strip export through the generic kernel: blockIdx.z picks the argument block (see mb_emit_multi_kernel)
template <bool L0>
__global__ __launch_bounds__(256) void mb_level_multi_kernel(const MbLevelK* __restrict__ Ps)
{
    mb_level_body<L0>(Ps[blockIdx.z]);
}

// ---------------------------------------------------------------------------------------------
// "no" blender (base cv::detail::Blender) and feather blender
// ---------------------------------------------------------------------------------------------
// distanceTransform(mask, DIST_L1, 3): exact city-block distance to the nearest zero pixel, as two separable passes
// that are plain prefix / suffix minima and therefore parallel:
//   columns: g(x, y) = min(y - last zero row <= y, first zero row >= y - y), INF when the column has no zero.
//            Chunks of DT_RC = 64 rows.  Kernel 1 reads the mask once and keeps, per (chunk, column), the zero rows as a
//            64-bit set plus its first / last member.  Kernel 2 folds the summaries of the chunks above / below (<= h / 64
//            coalesced loads) and writes its 64 rows straight from the bit set: distance up = a running counter, distance
//            down = count-trailing-zeros of the shifted set.  Lanes run along x, 4 columns each.
//   rows:    f(x) = min_x' (|x - x'| + g(x')) = min( x + min_{x'<=x} (g(x') - x'),  -x + min_{x'>=x} (g(x') + x') ):
//            one wavefront per row, 256 pixels per step, wave-level min scans, a scalar carry between steps; the forward
//            sweep parks its result in LDS (lane-private slots), the backward sweep overwrites the row in place.
// Distances are stored as 16 bits, saturated at 8192: the weight is
//            weight = min(dist * sharpness, 1), dist = (L1 >= 8192 or no zero) ? 8192.f : (float)L1
//            (distanceTransform_3x3's 16.16 fixed point saturates at INT_MAX >> 2, i.e. 8192.0f), min commutes with the
//            saturation, and the gather kernel makes the weight from the distance on the fly.
// HBM traffic per mask pixel: 1 (mask) + 1/8 + 1/8 (bit sets) + 2 (g) + 2 + 2 (f in place) — it was 30 with fp32 maps, a
// second mask read and both sweeps of both passes through memory.
// Same integers as the serial recurrences cur = min(v, cur + 1) of the reference.
constexpr int DT_INF = 1 << 28;
constexpr int DT_RC = STX_DT_RC;
// min over the lanes below this one (INT_MAX for lane 0): a DPP scan — shifts inside the 16-lane rows, then the row_bcast steps
// of the wave64 scan idiom, then one wave_shr to make it exclusive.  (The shuffle form costs 7 LDS-crossbar permutes.)
STX_DEV int wave_excl_prefix_min(int v)
{
    constexpr int ID = 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
    v = min(v, __builtin_amdgcn_update_dpp(ID, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
    return __builtin_amdgcn_update_dpp(ID, v, 0x138, 0xf, 0xf, false);       // wave_shr:1
}

int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return stx_fail(STX_ERR_HIP, "%s launch failed: %s", what, hipGetErrorString(e));
    return STX_OK;
}

inline dim3 grid64x4(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }

}  // namespace

// Builds G_1..G_B / W_1..W_B of every fed image: one launch per level for all images
// (grid.z = image) on the LDS kernels; int16 sources and degenerate sizes use the generic kernels.
// (Round 5 built and measured the fused tail asked for since round 3 — one workgroup per 8 x 8 tile of the last level, the 85 x 85
// samples of level B - 3 under it staged in LDS, three pyrDowns from LDS to LDS, the cores of the passed levels written: 134 us against
// the 36 us of the three launches it replaces on config 2 (84 against ~60 on config 4's share; commit history: "fused pyramid tail").
// 1.7 x redundant halo work, 62 KB of LDS = 2 workgroups per CU = three rounds of a workgroup whose own dependent chain is longer than
// a whole small launch.  One launch per level stays.)
int stx_launch_mb_pyramids(stx_ctx* ctx, const StxMbImage* d_images, const StxMbImage* h_images, int n, int num_bands, int pyr_mode,
                           int pyr_lanes)
{
    for (int lv = 0; lv < num_bands; lv++) {
        double bytes = 0.0;
        bool any_s16 = false;
        for (int i = 0; i < n; i++) {
            const StxMbImage& im = h_images[i];
            const double ip = (double)(im.fw >> lv) * (im.fh >> lv), op = ip / 4.0;
            const double gw = (im.g_u8 ? 3.0 : 6.0) + 4.0;  // bytes per pyramid sample: 3 Gaussian planes (bytes / int16) + the fp32 weight
            const double gw1 = gw - (im.w1_f16 ? 2.0 : 0.0);  // ... of level 1: its weight may be a half (StxMbImage::w1_f16)
            if (lv == 0) bytes += ((im.img0_is_s16 ? 6.0 : 3.0) + 1.0) * im.iw * im.ih + gw1 * op;
            else bytes += (lv == 1 ? gw1 : gw) * ip + gw * op;
            any_s16 = any_s16 || im.img0_is_s16;
        }
        StxProfScope prof(ctx, lv == 0 ? "mb_down0" : "mb_down", bytes);
        // the LDS kernels sum the weights in the scalar order; every other order goes through the generic kernels
        const bool batched = pyr_mode == STX_PYRDOWN_SCALAR && stx_fast_mb_down_batch(ctx, d_images, h_images, n, lv);
        for (int i = 0; i < n; i++) {
            const StxMbImage& im = h_images[i];
            if (batched && !(lv == 0 && im.img0_is_s16)) continue;
            const int ow = (im.fw >> lv) >> 1, oh = (im.fh >> lv) >> 1;
            if (lv == 0) {
                if (im.img0_is_s16) hipLaunchKernelGGL(mb_down0_kernel<true>, grid64x4(ow, oh), dim3(256), 0, ctx->stream, im, pyr_mode, pyr_lanes);
                else hipLaunchKernelGGL(mb_down0_kernel<false>, grid64x4(ow, oh), dim3(256), 0, ctx->stream, im, pyr_mode, pyr_lanes);
            } else {
                hipLaunchKernelGGL(mb_down_kernel, grid64x4(ow, oh), dim3(256), 0, ctx->stream, im, lv, pyr_mode, pyr_lanes);
            }
        }
        STX_TRY(check_launch("mb_down"));
    }
    return STX_OK;
}

// levels B, B-1, B-2 of a blender with B >= 3 bands in one launch: K = the argument block of level B-2 (its `up` is ignored)
int stx_launch_mb_coarse(stx_ctx* ctx, const MbLevelK& K, double algo_bytes)
{
    if (K.x1 <= K.x0 || K.y1 <= K.y0) return STX_OK;
    if (K.level != K.num_bands - 2 || K.level < 1 || K.y0 != 0 || K.out_y0 != 0 || K.emit)
        return stx_fail(STX_ERR_INVALID, "mb_coarse: not the argument block of level B-2");
    StxProfScope prof(ctx, "mb_coarse", algo_bytes);
    MbCoarseK C;
    C.images = K.images; C.n_images = K.n_images; C.num_bands = K.num_bands;
    C.x0 = K.x0; C.x1 = K.x1; C.y1 = K.y1;
    C.out = K.out; C.out_stride = K.out_stride; C.out_plane = K.out_plane; C.out_x0 = K.out_x0;
    C.pw = K.pw; C.ph = K.ph;
    const dim3 grid((K.x1 - K.x0 + CO_TW - 1) / CO_TW, (K.y1 + CO_TH - 1) / CO_TH);
    hipLaunchKernelGGL(mb_coarse_kernel, grid, dim3(256), 0, ctx->stream, C);
    return check_launch("mb_coarse");
}

int stx_launch_mb_level(stx_ctx* ctx, const MbLevelK& K, double algo_bytes)
{
    if (K.x1 <= K.x0 || K.y1 <= K.y0) return STX_OK;
    StxProfScope prof(ctx, K.emit ? "mb_contrib" : (K.level == 0 ? "mb_level0" : "mb_level"), algo_bytes);
    if ((K.level > 0 || K.all_u8) && stx_fast_mb_level(ctx, K)) return STX_OK;
    const dim3 grid = grid64x4(K.x1 - K.x0, K.y1 - K.y0);
    if (K.level == 0) hipLaunchKernelGGL(mb_level_kernel<true>, grid, dim3(256), 0, ctx->stream, K);
    else hipLaunchKernelGGL(mb_level_kernel<false>, grid, dim3(256), 0, ctx->stream, K);
    return check_launch("mb_level");
}

int stx_launch_mb_emit_batch(stx_ctx* ctx, const MbLevelK* d_Ks, const MbLevelK* h_Ks, const int* classes, int n, double bytes)
{
    if (n <= 0) return STX_OK;
    StxProfScope prof(ctx, "mb_contrib", bytes);
    for (int start = 0; start < n;) {
        int end = start;
        // generic blocks split into level 0 (class -1 stays -1) and level >= 1 (-2) by the caller's sort key
        while (end < n && classes[end] == classes[start]) end++;
        const int cls = classes[start], count = end - start;
        if (cls >= 0) {
            if (!stx_fast_mb_emit_launch(ctx, cls, d_Ks + start, h_Ks + start, count)) return check_launch("mb_contrib");
        } else {
            int gx = 1, gy = 1;
            for (int i = start; i < end; i++) {
                gx = std::max(gx, (h_Ks[i].x1 - h_Ks[i].x0 + 63) / 64);
                gy = std::max(gy, (h_Ks[i].y1 - h_Ks[i].y0 + 3) / 4);
            }
            const dim3 grid(gx, gy, count);
            if (cls == -1) hipLaunchKernelGGL(mb_level_multi_kernel<true>, grid, dim3(256), 0, ctx->stream, d_Ks + start);
            else hipLaunchKernelGGL(mb_level_multi_kernel<false>, grid, dim3(256), 0, ctx->stream, d_Ks + start);
        }
        start = end;
    }
    return check_launch("mb_contrib");
}

// ---------------------------------------------------------------------------------------------
// Feather blender as a deferred gather.  FeatherBlender::feed: weight = min(distanceTransform(mask, L1, 3) * sharpness, 1);
// dst += (short)(src * weight); dst_weight += weight.  blend: dst = (short)(dst / (dst_weight + 1e-5)), mask = weight sum >
// 1e-5, zero outside, then convertScaleAbs.  Fed images stay resident; at blend() the distance transforms of ALL images run
// as three launches (blockIdx.z = image; the two column kernels and the row kernel above, same integers), the weight map
// replaces the distances in place, and one pass over the panorama adds the products of the covering images in feed order
// (int16 wrap-around adds, fp32 weight sums in OpenCV's += order), normalises and writes the u8 panorama once.  The int16 /
// fp32 accumulators of OpenCV (10 bytes per panorama pixel, read-modify-written per image) never exist.
// ---------------------------------------------------------------------------------------------
namespace {
STX_DEV uint32_t ld_u32_unaligned(const uint8_t* p)  // 4 bytes at any address, as two aligned dword loads
{
    const uint32_t* q = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)3);
    return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u));
}

// bytes 0xff where pixel j (0..3) of a lane's group starting at image column lx lies inside an image w columns wide.  A group
// partly left / right of a u8 image is read with the same unaligned dword loads as a whole one — what lies outside is memory of the
// allocation (STX_BUF_FRONT_PAD, the row pitch) — and masked with this (round 3; the per-pixel byte path cost a wavefront that held
// such a lane four dependent round trips).
STX_DEV uint32_t quad_valid_bytes(int lx, int w)
{
    const int lo = max(-lx, 0), hi = min(w - lx, 4);
    const uint32_t bits = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
    return ((bits * 0x00204081u) & 0x01010101u) * 0xffu;
}

// Column passes: a lane owns 4 adjacent columns (one mask dword per row, 16-byte stores of the distances), a workgroup
// of 64 lanes 256 columns of one chunk of DT_RC rows.  Columns >= w inside the last group are computed on whatever the
// row padding holds and land in the padding of the distance rows (dstride is a multiple of 16), where nothing reads them.
STX_DEV uint32_t dt_mask4(const FeatherImg& P, int x, int y)
{
    const uint8_t* p = P.mask + (long long)y * P.mstride + x;
    if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) return *reinterpret_cast<const uint32_t*>(p);
    const uint32_t* q = reinterpret_cast<const uint32_t*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)3);
    return __builtin_amdgcn_alignbyte(q[1], q[0], (uint32_t)(reinterpret_cast<uintptr_t>(p) & 3u));
}
__global__ __launch_bounds__(64) void dt_col_summary_batch_kernel(const FeatherImg* __restrict__ imgs)
{
    const FeatherImg& P = imgs[blockIdx.z];
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, c = blockIdx.y;
    if (x >= P.w || c >= P.n_chunks) return;
    const int y0 = c * DT_RC, y1 = min(P.h, y0 + DT_RC);
    // Zero rows of the chunk, 8 rows at a time: bit 7 of every byte of `z` says "this column's mask byte is 0" (the classic
    // has-zero-byte test), and eight of those, each shifted one further down, fill the four bytes of `acc8` with the 8-row
    // pattern of the four columns — 6 operations per row for all four columns.
    uint32_t zb[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};  // rows 0..31, 32..63 of each column
#pragma unroll
    for (int g = 0; g < 8; g++) {
        if (y0 + 8 * g >= y1) break;
        uint32_t acc8 = 0;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int y = y0 + 8 * g + r;
            const uint32_t m = y < y1 ? dt_mask4(P, x, y) : 0x01010101u;  // rows past the image: not zero
            const uint32_t nz = (((m & 0x7f7f7f7fu) + 0x7f7f7f7fu) | m) & 0x80808080u;  // bit 7 of a byte: the byte is not 0
            acc8 = (acc8 >> 1) | (nz ^ 0x80808080u);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) zb[g >> 2][j] |= ((acc8 >> (8 * j)) & 255u) << (8 * (g & 3));
    }
    int f[4], l[4];
    unsigned long long bits[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        bits[j] = (unsigned long long)zb[0][j] | ((unsigned long long)zb[1][j] << 32);
        f[j] = bits[j] ? y0 + (int)__builtin_ctzll(bits[j]) : DT_INF;
        l[j] = bits[j] ? y0 + 63 - (int)__builtin_clzll(bits[j]) : -DT_INF;
    }
    *reinterpret_cast<int4*>(P.first + (long long)c * P.dstride + x) = make_int4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<int4*>(P.last + (long long)c * P.dstride + x) = make_int4(l[0], l[1], l[2], l[3]);
    unsigned long long* zo = P.zbits + (long long)c * P.dstride + x;
    *reinterpret_cast<ulonglong2*>(zo) = make_ulonglong2(bits[0], bits[1]);
    *reinterpret_cast<ulonglong2*>(zo + 2) = make_ulonglong2(bits[2], bits[3]);
}
// first[c] <- first zero row below chunk c, last[c] <- last zero row above it (in place, one lane per 4 columns, chunks in turn)
__global__ __launch_bounds__(64) void dt_col_link_batch_kernel(const FeatherImg* __restrict__ imgs)
{
    const FeatherImg& P = imgs[blockIdx.y];
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (x >= P.w) return;
    int4 run = make_int4(-DT_INF, -DT_INF, -DT_INF, -DT_INF);
    for (int c = 0; c < P.n_chunks; c++) {
        int4* q = reinterpret_cast<int4*>(P.last + (long long)c * P.dstride + x);
        const int4 v = *q;
        *q = run;
        run = make_int4(max(run.x, v.x), max(run.y, v.y), max(run.z, v.z), max(run.w, v.w));
    }
    run = make_int4(DT_INF, DT_INF, DT_INF, DT_INF);
    for (int c = P.n_chunks - 1; c >= 0; c--) {
        int4* q = reinterpret_cast<int4*>(P.first + (long long)c * P.dstride + x);
        const int4 v = *q;
        *q = run;
        run = make_int4(min(run.x, v.x), min(run.y, v.y), min(run.z, v.z), min(run.w, v.w));
    }
}
__global__ __launch_bounds__(64) void dt_col_fill_batch_kernel(const FeatherImg* __restrict__ imgs)
{
    const FeatherImg& P = imgs[blockIdx.z];
    const int x = (blockIdx.x * 64 + threadIdx.x) * 4, c = blockIdx.y;
    if (x >= P.dstride || c >= P.n_chunks) return;
    if (x >= P.w) {  // row padding: "no zero anywhere near" for the 8-pixel groups of the row pass
        for (int y = c * DT_RC; y < min(P.h, c * DT_RC + DT_RC); y++)
            *reinterpret_cast<uint2*>(P.dist + (long long)y * P.dstride + x) = make_uint2(0x20002000u, 0x20002000u);
        return;
    }
    const int4 pv = *reinterpret_cast<const int4*>(P.last + (long long)c * P.dstride + x);
    const int4 nv = *reinterpret_cast<const int4*>(P.first + (long long)c * P.dstride + x);
    const int prev[4] = {pv.x, pv.y, pv.z, pv.w}, next[4] = {nv.x, nv.y, nv.z, nv.w};
    const int y0 = c * DT_RC, y1 = min(P.h, y0 + DT_RC);
    unsigned long long bits[4];
    {
        const unsigned long long* zi = P.zbits + (long long)c * P.dstride + x;
        const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(zi), b = *reinterpret_cast<const ulonglong2*>(zi + 2);
        bits[0] = a.x; bits[1] = a.y; bits[2] = b.x; bits[3] = b.y;
    }
    int up[4];  // distance to the last zero row above the chunk (as seen from row y0 - 1)
    uint32_t lo[4], hi[4];
    int far0[4], far1[4];  // distance from the chunk's row 0 to the first zero in rows >= 32 / below the chunk (32-bit work only)
#pragma unroll
    for (int j = 0; j < 4; j++) {
        up[j] = prev[j] == -DT_INF ? DT_INF : y0 - 1 - prev[j];
        lo[j] = (uint32_t)bits[j]; hi[j] = (uint32_t)(bits[j] >> 32);
        far1[j] = next[j] == DT_INF ? DT_INF : next[j] - y0;
        far0[j] = hi[j] ? 32 + (int)__builtin_ctz(hi[j]) : far1[j];
    }
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
#pragma unroll 8
        for (int rr = 0; rr < 32; rr++) {
            const int r = 32 * hf + rr, y = y0 + r;
            if (y >= y1) break;
            uint32_t o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t t = (hf ? hi[j] : lo[j]) >> rr;  // the zero rows of this half from this one on
                up[j] = (t & 1u) ? 0 : min(up[j] + 1, DT_INF);
                const int dn = t ? (int)__builtin_ctz(t) : (hf ? far1[j] : far0[j]) - r;
                o[j] = x + j < P.w ? (uint32_t)min(min(up[j], dn), STX_FEATHER_DIST_CAP) : (uint32_t)STX_FEATHER_DIST_CAP;  // the row pass reads whole 8-pixel groups
            }
            *reinterpret_cast<uint2*>(P.dist + (long long)y * P.dstride + x) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
        }
    }
}
// rows: one wavefront per row; s_mid: the forward sweep's result, one 8-byte slot per lane and step (lane-private: a lane
// reads back exactly what it wrote, no barrier)
// A lane owns 8 adjacent pixels (16-byte loads), a step 512.  Columns past the row's end hold 8192 (dt_col_fill) or are
// replaced by it here: a candidate of 8192 or more never survives the final saturation, so it stands for "no zero".
constexpr int DT_RW = 8;
STX_DEV void dt_unpack8(const uint4 g, int (&v)[8])
{
    v[0] = (int)(g.x & 0xffffu); v[1] = (int)(g.x >> 16); v[2] = (int)(g.y & 0xffffu); v[3] = (int)(g.y >> 16);
    v[4] = (int)(g.z & 0xffffu); v[5] = (int)(g.z >> 16); v[6] = (int)(g.w & 0xffffu); v[7] = (int)(g.w >> 16);
}
STX_DEV uint4 dt_pack8(const int (&v)[8])
{
    return make_uint4((uint32_t)v[0] | ((uint32_t)v[1] << 16), (uint32_t)v[2] | ((uint32_t)v[3] << 16),
                      (uint32_t)v[4] | ((uint32_t)v[5] << 16), (uint32_t)v[6] | ((uint32_t)v[7] << 16));
}
__global__ __launch_bounds__(256) void dt_rows_batch_kernel(const FeatherImg* __restrict__ imgs, int rows_per_block, int lds_pitch)
{
    extern __shared__ uint4 s_mid[];
    const FeatherImg& P = imgs[blockIdx.y];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int y = blockIdx.x * rows_per_block + wv;
    if (wv >= rows_per_block || y >= P.h) return;
    const int wpad = (int)P.dstride;  // multiple of 16: whole 8-pixel groups, the columns >= w hold 8192
    uint16_t* row = P.dist + (long long)y * P.dstride;
    uint4* mid = s_mid + (long long)wv * lds_pitch;
    const int nseg = (P.w + 64 * DT_RW - 1) / (64 * DT_RW);
    const uint4 FAR = make_uint4(0x20002000u, 0x20002000u, 0x20002000u, 0x20002000u);
    int carry = 1 << 29;
    uint4 g = lane * DT_RW < wpad ? *reinterpret_cast<const uint4*>(row + lane * DT_RW) : FAR;
    for (int s = 0; s < nseg; s++) {
        const int x0 = (s * 64 + lane) * DT_RW;
        // the next step's distances are on their way while this step is scanned
        const uint4 gn = x0 + 64 * DT_RW < wpad ? *reinterpret_cast<const uint4*>(row + x0 + 64 * DT_RW) : FAR;
        int v[8];
        dt_unpack8(g, v);
        // p_j = g_j - (x0 + j) as q_j = g_j - j relative to x0; running minimum along the lane, then across the lanes below
#pragma unroll
        for (int j = 1; j < 8; j++) v[j] = min(v[j] - j, v[j - 1]);
        const int before = min(carry, wave_excl_prefix_min(v[7] - x0)) + x0;  // everything to the left, in this lane's frame
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = min(min(v[j], before) + j, 65535);  // back to a distance: <= g_j <= 8192 inside the row
        mid[s * 64 + lane] = dt_pack8(v);
        carry = __builtin_amdgcn_readlane(v[7] - 7 - x0, 63);
        g = gn;
    }
    // Backward sweep with the lanes mirrored (lane L owns the group of lane 63 - L): "every pixel to the right" is then
    // "every lane below", the same prefix scan.  The slots are read by another lane of the same wavefront than wrote them.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    carry = 1 << 29;
    for (int s = nseg - 1; s >= 0; s--) {
        const int x0 = (s * 64 + 63 - lane) * DT_RW;
        int v[8];
        dt_unpack8(mid[s * 64 + 63 - lane], v);
        // p_j = v_j + (x0 + j); running minimum from the right
        v[7] += 7;
#pragma unroll
        for (int j = 6; j >= 0; j--) v[j] = min(v[j] + j, v[j + 1]);
        const int after = min(carry, wave_excl_prefix_min(v[0] + x0)) - x0;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = min(min(v[j], after) - j, STX_FEATHER_DIST_CAP);
        carry = __builtin_amdgcn_readlane(v[0] + x0, 63);  // the leftmost group of the step (before the saturation matters: see below)
        if (x0 < wpad) *reinterpret_cast<uint4*>(row + x0) = dt_pack8(v);
    }
}

// FeatherBlender's weight from the stored distance: min(dist * sharpness, 1)
STX_DEV float feather_weight(uint32_t dist, float sharpness)
{
    const float t = fmul((float)dist, sharpness);
    return t > 1.f ? 1.f : t;
}

template <bool WITH16>
__global__ __launch_bounds__(256) void feather_gather_kernel(FeatherGatherK P)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = blockIdx.x * 256;
    if (y >= P.h || xw0 >= P.w) return;
    int acc[4][3];
    float ws[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; j++) acc[j][0] = acc[j][1] = acc[j][2] = 0;
    for (int i = 0; i < P.n; i++) {  // ascending feed order: the order of OpenCV's += on the fp32 weights
        const FeatherImg& im = P.imgs[i];
        if (y < im.y || y >= im.y + im.h || xw0 + 256 <= im.x || xw0 >= im.x + im.w) continue;  // wave-uniform
        const int lx = x4 - im.x, ly = y - im.y;
        if (lx + 3 < 0 || lx >= im.w) continue;
        const uint16_t* wrow = im.dist + (long long)ly * im.dstride;
        const uint8_t* irow = im.img + (long long)ly * im.istride;
        if (!im.is_s16) {
            // a u8 image: the four weights and the twelve image bytes as wide loads — also for a group partly left / right of the
            // image, whose outside pixels get the weight 0.f (nothing is added: (short)(px * 0.f) = 0, w + 0.f = w)
            const uint32_t d01 = ld_u32_unaligned(reinterpret_cast<const uint8_t*>(wrow + lx)), d23 = ld_u32_unaligned(reinterpret_cast<const uint8_t*>(wrow + lx + 2));
            float w4[4] = {feather_weight(d01 & 0xffffu, P.sharpness), feather_weight(d01 >> 16, P.sharpness),
                           feather_weight(d23 & 0xffffu, P.sharpness), feather_weight(d23 >> 16, P.sharpness)};
            if (lx < 0 || lx + 3 >= im.w) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((unsigned)(lx + j) >= (unsigned)im.w) w4[j] = 0.f;
            }
            const uint32_t b3[3] = {ld_u32_unaligned(irow + lx * 3), ld_u32_unaligned(irow + lx * 3 + 4), ld_u32_unaligned(irow + lx * 3 + 8)};
            // away from the mask's edge every weight is exactly 1.f (distance * sharpness >= 1) and (short)(px * 1.f) = px:
            // a wavefront in which that holds for all lanes adds the bytes as integers
            const bool ones = __builtin_amdgcn_ballot_w64(!(w4[0] == 1.f && w4[1] == 1.f && w4[2] == 1.f && w4[3] == 1.f)) == 0;
            if (ones) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        acc[j][c] = (short)(acc[j][c] + (int)((b3[(3 * j + c) >> 2] >> (8 * ((3 * j + c) & 3))) & 255u));
                    ws[j] = fadd(ws[j], 1.f);
                }
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int px = (int)((b3[(3 * j + c) >> 2] >> (8 * ((3 * j + c) & 3))) & 255u);
                    acc[j][c] = (short)(acc[j][c] + trunc_s16(fmul((float)px, w4[j])));
                }
                ws[j] = fadd(ws[j], w4[j]);
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (lx + j < 0 || lx + j >= im.w) continue;
            const float w = feather_weight(wrow[lx + j], P.sharpness);
            int b, g, r;
            if (im.is_s16) {
                const short* p = reinterpret_cast<const short*>(irow) + (lx + j) * 3;
                b = p[0]; g = p[1]; r = p[2];
            } else {
                const uint8_t* p = irow + (lx + j) * 3;
                b = p[0]; g = p[1]; r = p[2];
            }
            acc[j][0] = (short)(acc[j][0] + trunc_s16(fmul((float)b, w)));
            acc[j][1] = (short)(acc[j][1] + trunc_s16(fmul((float)g, w)));
            acc[j][2] = (short)(acc[j][2] + trunc_s16(fmul((float)r, w)));
            ws[j] = fadd(ws[j], w);
        }
    }
    if (x4 >= P.w) return;
    uint8_t* po = P.pano + (long long)y * P.pano_stride + (long long)x4 * 3;
    uint8_t* pm = P.pmask + (long long)y * P.pmask_stride + x4;
    uint32_t o[3] = {0, 0, 0}, mo = 0;
    int vv[4][3];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const float den = fadd(ws[j], WEIGHT_EPS);
        const bool in = ws[j] > WEIGHT_EPS;
        float q[3];  // the three quotients share one refined reciprocal (bit-identical to the IEEE division for these operands)
        div3_shared(den, (float)acc[j][0], (float)acc[j][1], (float)acc[j][2], q[0], q[1], q[2]);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            vv[j][c] = in ? trunc_s16(q[c]) : 0;
            o[(3 * j + c) >> 2] |= (uint32_t)min(abs(vv[j][c]), 255) << (8 * ((3 * j + c) & 3));
        }
        mo |= (in ? 255u : 0u) << (8 * j);
    }
    if (x4 + 4 <= P.w) {
        reinterpret_cast<uint32_t*>(po)[0] = o[0]; reinterpret_cast<uint32_t*>(po)[1] = o[1]; reinterpret_cast<uint32_t*>(po)[2] = o[2];
        *reinterpret_cast<uint32_t*>(pm) = mo;
    } else {
        for (int j = 0; x4 + j < P.w; j++) {
            for (int c = 0; c < 3; c++) po[3 * j + c] = (uint8_t)(o[(3 * j + c) >> 2] >> (8 * ((3 * j + c) & 3)));
            pm[j] = (uint8_t)(mo >> (8 * j));
        }
    }
    if (WITH16) {
        for (int j = 0; j < 4 && x4 + j < P.w; j++) {
            short* p16 = reinterpret_cast<short*>(reinterpret_cast<uint8_t*>(P.pano16) + (long long)y * P.pano16_stride) + (long long)(x4 + j) * 3;
            p16[0] = (short)vv[j][0]; p16[1] = (short)vv[j][1]; p16[2] = (short)vv[j][2];
        }
    }
}
}  // namespace

// distance transforms + weight maps of n fed images (device table d_imgs, host copy h_imgs for the grid sizes)
int stx_launch_feather_weights(stx_ctx* ctx, const FeatherImg* d_imgs, const FeatherImg* h_imgs, int n)
{
    if (n <= 0) return STX_OK;
    int max_w = 0, max_h = 0, max_chunks = 0;
    double px = 0.0;
    for (int i = 0; i < n; i++) {
        max_w = std::max(max_w, h_imgs[i].w); max_h = std::max(max_h, h_imgs[i].h); max_chunks = std::max(max_chunks, h_imgs[i].n_chunks);
        px += (double)h_imgs[i].w * h_imgs[i].h;
    }
    {
        StxProfScope prof(ctx, "feather_dt_cols", px * (1 + 0.25 + 2));
        hipLaunchKernelGGL(dt_col_summary_batch_kernel, dim3((max_w + 255) / 256, max_chunks, n), dim3(64), 0, ctx->stream, d_imgs);
        hipLaunchKernelGGL(dt_col_link_batch_kernel, dim3((max_w + 255) / 256, n), dim3(64), 0, ctx->stream, d_imgs);
        hipLaunchKernelGGL(dt_col_fill_batch_kernel, dim3((max_w + 255) / 256, max_chunks, n), dim3(64), 0, ctx->stream, d_imgs);
    }
    {
        StxProfScope prof(ctx, "feather_dt_rows", px * (2 + 2));
        // LDS: 2 bytes per pixel and row, at most 64 KB per workgroup (1 / 2 / 4 rows)
        const int lds_pitch = ((max_w + 511) / 512) * 64;  // 16-byte slots per row
        const int rows = lds_pitch * 16 * 4 <= 65536 ? 4 : (lds_pitch * 16 * 2 <= 65536 ? 2 : 1);
        hipLaunchKernelGGL(dt_rows_batch_kernel, dim3((max_h + rows - 1) / rows, n), dim3(256), (size_t)lds_pitch * 16 * rows, ctx->stream,
                           d_imgs, rows, lds_pitch);
    }
    return check_launch("feather_weights");
}

int stx_launch_feather_gather(stx_ctx* ctx, const FeatherGatherK& K, double algo_bytes)
{
    StxProfScope prof(ctx, "feather_gather", algo_bytes);
    const dim3 grid((K.w + 255) / 256, (K.h + 3) / 4);
    if (K.pano16) hipLaunchKernelGGL(feather_gather_kernel<true>, grid, dim3(256), 0, ctx->stream, K);
    else hipLaunchKernelGGL(feather_gather_kernel<false>, grid, dim3(256), 0, ctx->stream, K);
    return check_launch("feather_gather");
}

// ---------------------------------------------------------------------------------------------
// "no" blender as a deferred gather (Blender::feed: dst = src where mask != 0, dst_mask |= mask; Blender::blend: zero
// where dst_mask == 0; then convertScaleAbs).  OpenCV keeps an int16 panorama and overwrites it once per image — 10 bytes
// per panorama pixel zeroed, (3 + 1) read and 7 written per image pixel, 7 read + 4 written at the end.  Fed images
// stay resident here; one pass over the panorama finds, per pixel, the LAST fed image whose mask is set (walking the
// image table backwards), ORs the masks of all of them and writes the u8 panorama (+ mask, + int16 on request) once.
// A lane owns 4 adjacent pixels; inside an image the four mask bytes / the twelve image bytes come from aligned dword
// loads + v_alignbyte.
// ---------------------------------------------------------------------------------------------
namespace {
template <bool WITH16>
__global__ __launch_bounds__(256) void no_gather_kernel(NoGatherK P)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int xw0 = blockIdx.x * 256;  // first column of the wavefront
    if (y >= P.h || xw0 >= P.w) return;
    int v[4][3];
    uint32_t macc = 0, decided = 0;  // per pixel byte: OR of the masks / 0xff once a value is chosen
#pragma unroll
    for (int j = 0; j < 4; j++) v[j][0] = v[j][1] = v[j][2] = 0;
    for (int i = P.n - 1; i >= 0; i--) {
        const NoImg& im = P.imgs[i];
        // wave-uniform rejection
        if (y < im.y || y >= im.y + im.h || xw0 + 256 <= im.x || xw0 >= im.x + im.w) continue;
        if (P.all_binary && __builtin_amdgcn_ballot_w64(decided != 0xffffffffu && x4 < P.w) == 0) break;  // every pixel of the wavefront has its value and 255
        const int lx = x4 - im.x, ly = y - im.y;
        if (lx + 3 < 0 || lx >= im.w) continue;
        const uint8_t* mrow = im.mask + (long long)ly * im.mstride;
        uint32_t m4;
        const bool whole = lx >= 0 && lx + 3 < im.w;
        const bool quad = whole || !im.is_s16;  // u8 images: partial groups take the wide loads too, masked
        if (quad) {
            m4 = ld_u32_unaligned(mrow + lx);
            if (!whole) m4 &= quad_valid_bytes(lx, im.w);
        } else {
            m4 = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (lx + j >= 0 && lx + j < im.w) m4 |= (uint32_t)mrow[lx + j] << (8 * j);
        }
        macc |= m4;
        // bytes of pixels this image gives their value: mask != 0 and not decided yet
        uint32_t nz = (m4 | (m4 >> 4)) & 0x0f0f0f0fu; nz = (nz | (nz >> 2)) & 0x03030303u; nz = (nz | (nz >> 1)) & 0x01010101u;
        const uint32_t take = (nz * 255u) & ~decided;
        if (take) {
            if (im.is_s16) {
                const short* irow = reinterpret_cast<const short*>(im.img + (long long)ly * im.istride);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((take >> (8 * j)) & 1u) { v[j][0] = irow[(lx + j) * 3]; v[j][1] = irow[(lx + j) * 3 + 1]; v[j][2] = irow[(lx + j) * 3 + 2]; }
            } else {
                const uint8_t* irow = im.img + (long long)ly * im.istride;
                if (quad) {
                    const uint32_t w0 = ld_u32_unaligned(irow + lx * 3), w1 = ld_u32_unaligned(irow + lx * 3 + 4), w2 = ld_u32_unaligned(irow + lx * 3 + 8);
                    const uint32_t w[3] = {w0, w1, w2};
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if ((take >> (8 * j)) & 1u) {
#pragma unroll
                            for (int c = 0; c < 3; c++) v[j][c] = (int)((w[(3 * j + c) >> 2] >> (8 * ((3 * j + c) & 3))) & 255u);
                        }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if ((take >> (8 * j)) & 1u) { v[j][0] = irow[(lx + j) * 3]; v[j][1] = irow[(lx + j) * 3 + 1]; v[j][2] = irow[(lx + j) * 3 + 2]; }
                }
            }
            decided |= take;
        }
    }
    if (x4 >= P.w) return;
    // Blender::blend zeroes where dst_mask == 0 (those pixels never took a value: v is 0 already); convertScaleAbs
    uint32_t o[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int c = 0; c < 3; c++) o[(3 * j + c) >> 2] |= (uint32_t)min(abs(v[j][c]), 255) << (8 * ((3 * j + c) & 3));
    uint8_t* po = P.pano + (long long)y * P.pano_stride + (long long)x4 * 3;
    uint8_t* pm = P.pmask + (long long)y * P.pmask_stride + x4;
    if (x4 + 4 <= P.w) {
        reinterpret_cast<uint32_t*>(po)[0] = o[0]; reinterpret_cast<uint32_t*>(po)[1] = o[1]; reinterpret_cast<uint32_t*>(po)[2] = o[2];
        *reinterpret_cast<uint32_t*>(pm) = macc;
    } else {
        for (int j = 0; x4 + j < P.w; j++) {
            for (int c = 0; c < 3; c++) po[3 * j + c] = (uint8_t)(o[(3 * j + c) >> 2] >> (8 * ((3 * j + c) & 3)));
            pm[j] = (uint8_t)(macc >> (8 * j));
        }
    }
    if (WITH16) {
        short* p16 = reinterpret_cast<short*>(reinterpret_cast<uint8_t*>(P.pano16) + (long long)y * P.pano16_stride) + (long long)x4 * 3;
        for (int j = 0; j < 4 && x4 + j < P.w; j++)
            for (int c = 0; c < 3; c++) p16[3 * j + c] = (short)v[j][c];
    }
}
}  // namespace

int stx_launch_no_gather(stx_ctx* ctx, const NoGatherK& K, double algo_bytes)
{
    StxProfScope prof(ctx, "no_gather", algo_bytes);
    const dim3 grid((K.w + 255) / 256, (K.h + 3) / 4);
    if (K.pano16) hipLaunchKernelGGL(no_gather_kernel<true>, grid, dim3(256), 0, ctx->stream, K);
    else hipLaunchKernelGGL(no_gather_kernel<false>, grid, dim3(256), 0, ctx->stream, K);
    return check_launch("no_gather");
}

// ---------------------------------------------------------------------------------------------
// "next" rows of the scope table: consumers / producers either side of the path
// ---------------------------------------------------------------------------------------------
namespace {
// GainCompensator::apply / ChannelsCompensator::apply: cv::multiply(u8x3 image, double scalar) evaluates in fp32
// (arithm_op demotes a CV_64F scalar to CV_32F for 8-bit sources) and stores saturate_cast<uchar>(float) = cvRound + clamp
struct GainK { uint8_t* img; long long stride; int w, h; float g[3]; };
__global__ __launch_bounds__(256) void gain_apply_kernel(GainK P)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;  // 4 pixels = 12 bytes = 3 dwords per lane
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x4 >= P.w || y >= P.h) return;
    uint8_t* row = P.img + (long long)y * P.stride;
    if (x4 + 4 <= P.w) {
        uint32_t* q = reinterpret_cast<uint32_t*>(row + (long long)x4 * 3);
        uint32_t d[3] = {q[0], q[1], q[2]}, o[3] = {0, 0, 0};
#pragma unroll
        for (int b = 0; b < 12; b++) {
            const float v = stxd::fmul((float)((d[b >> 2] >> (8 * (b & 3))) & 255u), P.g[b % 3]);
            const int r = min(max(stxd::cv_round(v), 0), 255);
            o[b >> 2] |= (uint32_t)r << (8 * (b & 3));
        }
        q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
    } else {
        for (int x = x4; x < P.w; x++)
            for (int c = 0; c < 3; c++) {
                const float v = stxd::fmul((float)row[x * 3 + c], P.g[c]);
                row[x * 3 + c] = (uint8_t)min(max(stxd::cv_round(v), 0), 255);
            }
    }
}
}  // namespace

int stx_launch_gain_apply(stx_ctx* ctx, stx_buf* img, const float g[3])
{
    GainK K;
    K.img = img->ptr; K.stride = (long long)img->stride; K.w = img->w; K.h = img->h;
    K.g[0] = g[0]; K.g[1] = g[1]; K.g[2] = g[2];
    StxProfScope prof(ctx, "gain_apply", 6.0 * img->w * img->h);
    hipLaunchKernelGGL(gain_apply_kernel, dim3((img->w + 255) / 256, (img->h + 3) / 4), dim3(256), 0, ctx->stream, K);
    return check_launch("gain_apply");
}

// ---------------------------------------------------------------------------------------------
// Strip packing (image-strip sharding): the columns [x0, x0 + w) of up to STRIP_BATCH warped images and of
// their masks -> flat buffers (image rows at pitch si, then mask rows at pitch sm), one launch.  8 bytes per lane; w is
// copied in whole 8-pixel groups (rows of every image buffer hold whole groups), x0 is a multiple of 8.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int STRIP_BATCH = 16;
struct StripK { const uint8_t* img; const uint8_t* mask; uint8_t* dst; long long istride, mstride, si, sm; int h, img_chunks, mask_chunks, bits; };
struct StripBatchK { StripK k[STRIP_BATCH]; };
constexpr int STRIP_ROWS = 4;  // rows per workgroup: four independent 8-byte copies in flight per lane
// bit j of the result = byte j of m is not 0
STX_DEV uint32_t strip_nz4(uint32_t m)
{
    const uint32_t t = ((((m & 0x7f7f7f7fu) + 0x7f7f7f7fu) | m) & 0x80808080u) >> 7;  // bit 8 j = byte j != 0
    return (t | (t >> 7) | (t >> 14) | (t >> 21)) & 15u;
}
// receiver side of STX_STRIP_MASK_BITS: one byte of bits -> 8 mask bytes 0 / 255 (a lane per 8 pixels, rows on blockIdx.y)
struct StripBitsK { const uint8_t* bits; long long sm; uint8_t* mask; long long mstride; int groups, h; };
struct StripBitsBatchK { StripBitsK k[STRIP_BATCH]; };
__global__ __launch_bounds__(256) void strip_bits_expand_kernel(StripBitsBatchK B)
{
    const StripBitsK& P = B.k[blockIdx.z];
    const int row0 = blockIdx.y * STRIP_ROWS;
    if (row0 >= P.h) return;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < P.groups; g += gridDim.x * 256) {
#pragma unroll
        for (int r = 0; r < STRIP_ROWS; r++)
            if (row0 + r < P.h) {
                const uint32_t b = P.bits[(long long)(row0 + r) * P.sm + g];
                // bit k of a nibble -> byte k: n * (1 + 2^7 + 2^14 + 2^21) puts bit k at 8 k (and elsewhere: masked off)
                const uint32_t lo = (((b & 15u) * 0x00204081u) & 0x01010101u) * 255u, hi = (((b >> 4) * 0x00204081u) & 0x01010101u) * 255u;
                *reinterpret_cast<uint2*>(P.mask + (long long)(row0 + r) * P.mstride + 8ll * g) = make_uint2(lo, hi);
            }
    }
}
__global__ __launch_bounds__(256) void strip_pack_kernel(StripBatchK B)
{
    const StripK& P = B.k[blockIdx.z];
    const int row0 = blockIdx.y * STRIP_ROWS;
    if (row0 >= P.h) return;
    const int per_row = P.img_chunks + P.mask_chunks;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < per_row; c += gridDim.x * 256) {
        const bool im = c < P.img_chunks;
        if (!im && P.bits) {  // STX_STRIP_MASK_BITS: 8 mask bytes -> one byte, bit j = pixel j is not 0
            const uint8_t* ms = P.mask + 8ll * (c - P.img_chunks);
            uint8_t* md = P.dst + P.si * P.h + (c - P.img_chunks);
#pragma unroll
            for (int r = 0; r < STRIP_ROWS; r++)
                if (row0 + r < P.h) {
                    const uint2 v = *reinterpret_cast<const uint2*>(ms + (long long)(row0 + r) * P.mstride);
                    md[(long long)(row0 + r) * P.sm] = (uint8_t)(strip_nz4(v.x) | (strip_nz4(v.y) << 4));
                }
            continue;
        }
        const uint8_t* src = im ? P.img + 8ll * c : P.mask + 8ll * (c - P.img_chunks);
        uint8_t* dst = im ? P.dst + 8ll * c : P.dst + P.si * P.h + 8ll * (c - P.img_chunks);
        const long long ss = im ? P.istride : P.mstride, ds = im ? P.si : P.sm;
        uint2 v[STRIP_ROWS];
#pragma unroll
        for (int r = 0; r < STRIP_ROWS; r++)
            if (row0 + r < P.h) v[r] = *reinterpret_cast<const uint2*>(src + (long long)(row0 + r) * ss);
#pragma unroll
        for (int r = 0; r < STRIP_ROWS; r++)
            if (row0 + r < P.h) *reinterpret_cast<uint2*>(dst + (long long)(row0 + r) * ds) = v[r];
    }
}
}  // namespace

int stx_launch_strip_pack(stx_ctx* ctx, int n, const stx_buf* const* imgs, const stx_buf* const* masks, const int* x0, const int* w,
                          stx_buf* const* dsts, const size_t* si, const size_t* sm, bool mask_bits)
{
    for (int base = 0; base < n; base += STRIP_BATCH) {
        const int m = std::min(STRIP_BATCH, n - base);
        StripBatchK B = {};
        int max_h = 0, max_chunks = 0;
        double bytes = 0.0;
        for (int i = 0; i < m; i++) {
            const int g = base + i, w8 = (w[g] + 7) & ~7;
            StripK& K = B.k[i];
            K.img = imgs[g]->ptr + (size_t)x0[g] * 3; K.mask = masks[g]->ptr + x0[g]; K.dst = dsts[g]->ptr;
            K.istride = (long long)imgs[g]->stride; K.mstride = (long long)masks[g]->stride; K.si = (long long)si[g]; K.sm = (long long)sm[g];
            K.h = imgs[g]->h; K.img_chunks = w8 * 3 / 8; K.mask_chunks = w8 / 8; K.bits = mask_bits ? 1 : 0;
            max_h = std::max(max_h, K.h);
            max_chunks = std::max(max_chunks, K.img_chunks + K.mask_chunks);
            bytes += (mask_bits ? 7.125 : 8.0) * (double)w[g] * K.h;
        }
        StxProfScope prof(ctx, "strip_pack", bytes);
        hipLaunchKernelGGL(strip_pack_kernel, dim3((max_chunks + 255) / 256, (max_h + STRIP_ROWS - 1) / STRIP_ROWS, m), dim3(256), 0, ctx->stream, B);
    }
    return check_launch("strip_pack");
}

// masks[i] (w x h, whole 8-pixel groups per row) <- the bit rows at bits[i] (pitch sm[i])
int stx_launch_strip_bits_expand(stx_ctx* ctx, int n, const uint8_t* const* bits, const size_t* sm, stx_buf* const* masks)
{
    for (int base = 0; base < n; base += STRIP_BATCH) {
        const int m = std::min(STRIP_BATCH, n - base);
        StripBitsBatchK B = {};
        int max_h = 0, max_groups = 0;
        double bytes = 0.0;
        for (int i = 0; i < m; i++) {
            const int g = base + i;
            StripBitsK& K = B.k[i];
            K.bits = bits[g]; K.sm = (long long)sm[g]; K.mask = masks[g]->ptr; K.mstride = (long long)masks[g]->stride;
            K.groups = (masks[g]->w + 7) / 8; K.h = masks[g]->h;
            max_h = std::max(max_h, K.h); max_groups = std::max(max_groups, K.groups);
            bytes += 1.125 * (double)masks[g]->w * K.h;
        }
        StxProfScope prof(ctx, "strip_unpack", bytes);
        hipLaunchKernelGGL(strip_bits_expand_kernel, dim3((max_groups + 255) / 256, (max_h + STRIP_ROWS - 1) / STRIP_ROWS, m), dim3(256), 0, ctx->stream, B);
    }
    return check_launch("strip_unpack");
}

// ---------------------------------------------------------------------------------------------
// cv::resize(INTER_LINEAR_EXACT) for 8-bit images (next rows N2 / N3): 8.8 fixed-point coefficients made on the host in
// double precision exactly as interpolationLinear<ufixedpoint16>::getCoeffs does (tables: x = offset, y = coeff1 |
// interior << 16), horizontal sums p0 * c0 + p1 * c1 in 8.8, vertical (h0 * d0 + h1 * d1 + 2^15) >> 16, rows
// outside the source (h + 128) >> 8.  DILATE: the source is read through cv::dilate(3x3) on the fly and the result is
// ANDed with a mask of the destination size (SeamFinder.resize, stitching/seam_finder.py:37-43).
// ---------------------------------------------------------------------------------------------
namespace {
struct ResizeK {
    const uint8_t* src; long long sstride; int sw, sh;
    uint8_t* dst; long long dstride; int dw, dh;
    const int2* xt; const int2* yt;
    const uint8_t* andmask; long long amstride;
};
template <int C, bool DILATE>
STX_DEV uint32_t resize_src(const ResizeK& P, int x, int y, int ch)
{
    if (!DILATE) return P.src[(long long)y * P.sstride + x * C + ch];
    uint32_t m = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int xx = x + dx, yy = y + dy;
            if ((unsigned)xx < (unsigned)P.sw && (unsigned)yy < (unsigned)P.sh) m = max(m, (uint32_t)P.src[(long long)yy * P.sstride + xx]);
        }
    return m;
}
template <int C, bool DILATE>
__global__ __launch_bounds__(256) void resize_exact_kernel(ResizeK P)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.dw || y >= P.dh) return;
    const int2 tx = P.xt[x], ty = P.yt[y];
    const int ox = tx.x, oy = ty.x;
    const uint32_t cx1 = (uint32_t)tx.y & 0xffffu, cx0 = 256u - cx1, cy1 = (uint32_t)ty.y & 0xffffu, cy0 = 256u - cy1;
    const bool iy = (ty.y >> 16) != 0;
    const int ox1 = min(ox + 1, P.sw - 1), oy1 = min(oy + 1, P.sh - 1);
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
        const uint32_t h0 = resize_src<C, DILATE>(P, ox, oy, ch) * cx0 + resize_src<C, DILATE>(P, ox1, oy, ch) * cx1;
        uint32_t v;
        if (iy) {
            const uint32_t h1 = resize_src<C, DILATE>(P, ox, oy1, ch) * cx0 + resize_src<C, DILATE>(P, ox1, oy1, ch) * cx1;
            v = (h0 * cy0 + h1 * cy1 + 32768u) >> 16;
        } else {
            v = (h0 + 128u) >> 8;
        }
        v = min(v, 255u);
        if (P.andmask) v &= P.andmask[(long long)y * P.amstride + x];
        P.dst[(long long)y * P.dstride + x * C + ch] = (uint8_t)v;
    }
}
}  // namespace

namespace {
// SeamFinder.resize, fast form (the reference's default pipeline runs it once per image and panorama): the 3x3 dilation of
// the small low-resolution mask is done once into a scratch image (a few tens of KB, L2 resident) instead of nine
// bounds-checked byte loads per tap; a lane then produces 4 adjacent destination pixels (taps from the scratch, one
// dword of the final warped mask, one dword store).  Same integer arithmetic as resize_exact_kernel<1, true>.
__global__ __launch_bounds__(256) void dilate3x3_kernel(const uint8_t* __restrict__ src, long long sstride, int w, int h,
                                                        uint8_t* __restrict__ dst, long long dstride)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    uint32_t m = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int xx = x + dx, yy = y + dy;
            if ((unsigned)xx < (unsigned)w && (unsigned)yy < (unsigned)h) m = max(m, (uint32_t)src[(long long)yy * sstride + xx]);
        }
    dst[(long long)y * dstride + x] = (uint8_t)m;
}

constexpr int SEAM_ROWS = 16;  // destination rows per lane (seam_resize4_body)
STX_DEV void seam_resize4_body(const ResizeK& P);
__global__ __launch_bounds__(256) void seam_resize4_kernel(ResizeK P) { seam_resize4_body(P); }  // P.src: the dilated low-resolution mask
// all seam masks of a panorama in one launch each (blockIdx.z = image); the argument blocks travel as kernel arguments
constexpr int SEAM_BATCH = 16;
struct SeamBatchK { ResizeK k[SEAM_BATCH]; const uint8_t* raw[SEAM_BATCH]; long long raw_stride[SEAM_BATCH]; };
__global__ __launch_bounds__(256) void seam_resize4_batch_kernel(SeamBatchK B) { seam_resize4_body(B.k[blockIdx.z]); }
__global__ __launch_bounds__(256) void dilate3x3_batch_kernel(SeamBatchK B)
{
    const ResizeK& P = B.k[blockIdx.z];
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.sw || y >= P.sh) return;
    const uint8_t* src = B.raw[blockIdx.z];
    const long long ss = B.raw_stride[blockIdx.z];
    uint32_t m = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; dy++)
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            const int xx = x + dx, yy = y + dy;
            if ((unsigned)xx < (unsigned)P.sw && (unsigned)yy < (unsigned)P.sh) m = max(m, (uint32_t)src[(long long)yy * ss + xx]);
        }
    const_cast<uint8_t*>(P.src)[(long long)y * P.sstride + x] = (uint8_t)m;
}
// One lane = 4 adjacent columns of SEAM_ROWS consecutive destination rows.  The seam masks are enlarged (0.1 Mpix -> final
// size, about 11 x): consecutive destination rows interpolate between the same two source rows, so the horizontal sums
// (8.8 fixed point) are kept in registers and recomputed only when the source row changes (a wave-uniform test: the row is);
// a destination pixel then costs the vertical blend, the AND and a quarter of a dword store.
STX_DEV void seam_resize4_body(const ResizeK& P)
{
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int yb = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * SEAM_ROWS;
    if (x4 >= P.dw || yb >= P.dh) return;
    // the table holds dw entries rounded up to 4 (host), so the four reads are always inside it
    const int4 ta = *reinterpret_cast<const int4*>(P.xt + x4), tb = *reinterpret_cast<const int4*>(P.xt + x4 + 2);
    const int ox[4] = {ta.x, ta.z, tb.x, tb.z}, cf[4] = {ta.y, ta.w, tb.y, tb.w};
    uint32_t h0[4] = {0, 0, 0, 0}, h1[4] = {0, 0, 0, 0};
    int cur = -1;
    uint32_t am[SEAM_ROWS];  // the final-size masks of all rows first: SEAM_ROWS loads in flight instead of one per iteration
#pragma unroll
    for (int r = 0; r < SEAM_ROWS; r++)
        am[r] = yb + r < P.dh ? *reinterpret_cast<const uint32_t*>(P.andmask + (long long)(yb + r) * P.amstride + x4) : 0u;
#pragma unroll
    for (int r = 0; r < SEAM_ROWS; r++) {
        const int y = yb + r;
        if (y >= P.dh) break;
        const int2 ty = P.yt[y];
        if (ty.x != cur) {
            cur = ty.x;
            const uint8_t* r0 = P.src + (long long)cur * P.sstride;
            const uint8_t* r1 = P.src + (long long)min(cur + 1, P.sh - 1) * P.sstride;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int o0 = ox[j], o1 = min(o0 + 1, P.sw - 1);
                const uint32_t cx1 = (uint32_t)cf[j] & 0xffffu, cx0 = 256u - cx1;
                h0[j] = (uint32_t)r0[o0] * cx0 + (uint32_t)r0[o1] * cx1;
                h1[j] = (uint32_t)r1[o0] * cx0 + (uint32_t)r1[o1] * cx1;
            }
        }
        const uint32_t cy1 = (uint32_t)ty.y & 0xffffu, cy0 = 256u - cy1;
        const bool iy = (ty.y >> 16) != 0;
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t v = iy ? (h0[j] * cy0 + h1[j] * cy1 + 32768u) >> 16 : (h0[j] + 128u) >> 8;
            out |= min(v, 255u) << (8 * j);
        }
        out &= am[r];
        uint8_t* d = P.dst + (long long)y * P.dstride + x4;
        if (x4 + 4 <= P.dw) *reinterpret_cast<uint32_t*>(d) = out;
        else for (int j = 0; x4 + j < P.dw; j++) d[j] = (uint8_t)(out >> (8 * j));
    }
}
}  // namespace

// SeamFinder.resize for n images: tmp[i] = scratch of the dilated low-resolution mask (pitch tstride[i]), d_xt / d_yt per image
int stx_launch_seam_resize_batch(stx_ctx* ctx, int n, const stx_buf* const* seams, const stx_buf* const* masks, stx_buf* const* dsts,
                                 const int* const* d_xt, const int* const* d_yt, uint8_t* const* tmp, const size_t* tstride)
{
    for (int base = 0; base < n; base += SEAM_BATCH) {
        const int m = std::min(SEAM_BATCH, n - base);
        SeamBatchK B = {};
        int msw = 0, msh = 0, mdw = 0, mdh = 0;
        double bytes = 0.0;
        for (int i = 0; i < m; i++) {
            const int g = base + i;
            ResizeK& K = B.k[i];
            K.src = tmp[g]; K.sstride = (long long)tstride[g]; K.sw = seams[g]->w; K.sh = seams[g]->h;
            K.dst = dsts[g]->ptr; K.dstride = (long long)dsts[g]->stride; K.dw = dsts[g]->w; K.dh = dsts[g]->h;
            K.xt = (const int2*)d_xt[g]; K.yt = (const int2*)d_yt[g];
            K.andmask = masks[g]->ptr; K.amstride = (long long)masks[g]->stride;
            B.raw[i] = seams[g]->ptr; B.raw_stride[i] = (long long)seams[g]->stride;
            msw = std::max(msw, K.sw); msh = std::max(msh, K.sh); mdw = std::max(mdw, K.dw); mdh = std::max(mdh, K.dh);
            bytes += (double)K.sw * K.sh + 2.0 * K.dw * K.dh;
        }
        StxProfScope prof(ctx, "seam_mask_resize", bytes);
        hipLaunchKernelGGL(dilate3x3_batch_kernel, dim3((msw + 63) / 64, (msh + 3) / 4, m), dim3(256), 0, ctx->stream, B);
        hipLaunchKernelGGL(seam_resize4_batch_kernel, dim3((mdw + 255) / 256, (mdh + 4 * SEAM_ROWS - 1) / (4 * SEAM_ROWS), m), dim3(256), 0, ctx->stream, B);
    }
    return check_launch("seam_mask_resize");
}

// ---------------------------------------------------------------------------------------------
// SeamFinder.resize as ONE launch (round 6; stitching/seam_finder.py:37-43, stitching/stitcher.py:124,223-225).
// Until round 5 a panorama's seam masks took a table upload (448 KB of coefficients made by the host in double precision), a dilate
// launch into a global scratch and the resize launch with byte gathers from that scratch (81.6 us for config 2's eight masks, 0.24 of
// the HBM peak).  Here a workgroup owns a 512 x 32 tile of the destination:
//   * the coefficients of its 512 columns and 32 rows are made IN the kernel with the same IEEE double operations the host table used
//     (scale * (v + 0.5) - 0.5, floor, rint of the fraction * 256 — interpolationLinear<ufixedpoint16>::getCoeffs; this file is built
//     with -ffp-contract=off) and kept in LDS: nothing is uploaded;
//   * the low-resolution source window under the tile (+ 1 px) goes to LDS, is dilated 3 x 3 there, and every tap is an LDS byte;
//   * a lane makes 8 adjacent pixels of 8 rows (rows per lane, one box, interleaved: 4: 48.4-50.0 us, 6: 48.6-49.4, 8: 47.7-49.1,
//     16: 52.2-55.4, 32: 69.7-73.3 for config 2's eight masks): the final-mask dwords of all its rows are requested before the window is prepared,
//     the horizontal 8.8 sums live in registers and are redone only when the source row changes, a pixel costs two 24-bit
//     multiply-adds and 3/4 of a byte permute, a row leaves as one 8-byte store.
// Same integers as resize_exact_kernel<1, true> (the tests compare both with the CPU checker).
// ---------------------------------------------------------------------------------------------
namespace {
// rows per lane: -DSTX_SEAM1_ROWS=4 / 6 / 16 / 32 build the variants of the A/B (tools/gpu_r6g.sh)
#ifndef STX_SEAM1_ROWS
#define STX_SEAM1_ROWS 8
#endif
constexpr int SEAM1_COLS = 8, SEAM1_ROWS = STX_SEAM1_ROWS;
constexpr int SEAM1_TW = 64 * SEAM1_COLS, SEAM1_TH = 4 * SEAM1_ROWS;
struct Seam1K {
    const uint8_t* src; long long sstride; int sw, sh;  // the low-resolution seam mask as the seam finder made it
    uint8_t* dst; long long dstride; int dw, dh;         // the destination rectangle ...
    const uint8_t* am; long long amstride;               // ... and the same rectangle of the final warped mask
    int x0, y0;                                          // where the rectangle lies in the full final mask
    double xscale, yscale;                               // 1 / (full width / sw), 1 / (full height / sh): the host's doubles
};
struct Seam1BatchK { Seam1K k[SEAM_BATCH]; int pitch, raw_bytes; };

// interpolationLinear<ufixedpoint16>::getCoeffs for destination index v: (offset, coeff1 | interior << 16) — stx_api.cpp linear_exact_table
STX_DEV int2 seam1_coeff(int v, double scale, int src_n)
{
    const double fval = scale * ((double)v + 0.5) - 0.5;
    const int ival = (int)floor(fval);
    int ofs = 0, c1 = 0, interior = 0;
    if (ival >= 0 && src_n > 1) {
        if (ival < src_n - 1) { ofs = ival; c1 = (int)rint((fval - (double)ival) * 256.0); interior = 1; }
        else ofs = src_n - 1;
    }
    return make_int2(ofs, c1 | (interior << 16));
}

__global__ __launch_bounds__(256) void seam_resize_lds_kernel(Seam1BatchK B)
{
    extern __shared__ uint8_t s_win[];
    __shared__ int2 s_xt[SEAM1_TW];
    __shared__ int2 s_yt[SEAM1_TH];
    const Seam1K& P = B.k[blockIdx.z];
    const int X0 = blockIdx.x * SEAM1_TW, Y0 = blockIdx.y * SEAM1_TH;
    if (X0 >= P.dw || Y0 >= P.dh) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x8 = X0 + lane * SEAM1_COLS, yb = Y0 + wv * SEAM1_ROWS;
    const bool mine = x8 < P.dw && yb < P.dh;  // (the row pitch holds whole 8-pixel groups: host-checked)
    // the final-mask bytes of this lane's rows: in flight while the window is prepared
    uint2 am[SEAM1_ROWS];
#pragma unroll
    for (int r = 0; r < SEAM1_ROWS; r++)
        am[r] = mine && yb + r < P.dh ? *reinterpret_cast<const uint2*>(P.am + (long long)(yb + r) * P.amstride + x8) : make_uint2(0u, 0u);
    for (int i = tid; i < SEAM1_TW; i += 256) s_xt[i] = seam1_coeff(P.x0 + min(X0 + i, P.dw - 1), P.xscale, P.sw);
    if (tid < SEAM1_TH) s_yt[tid] = seam1_coeff(P.y0 + min(Y0 + tid, P.dh - 1), P.yscale, P.sh);
    __syncthreads();
    // the source window under this tile
    const int nx = min(SEAM1_TW, P.dw - X0), ny = min(SEAM1_TH, P.dh - Y0);
    const int wx0 = s_xt[0].x, wx1 = min(s_xt[nx - 1].x + 1, P.sw - 1);
    const int wy0 = s_yt[0].x, wy1 = min(s_yt[ny - 1].x + 1, P.sh - 1);
    const int cw = wx1 - wx0 + 1, rh = wy1 - wy0 + 1, cw2 = cw + 2, pitch = B.pitch;
    uint8_t* const s_raw = s_win;                 // (rh + 2) x (cw + 2): the window and one pixel around it, 0 outside the image
    uint8_t* const s_dil = s_win + B.raw_bytes;   // rh x cw: cv::dilate(3 x 3) of it (pixels outside the image do not take part: 0 = no-op for max)
    const float inv2 = 1.0f / (float)cw2, inv1 = 1.0f / (float)cw;
    for (int i = tid; i < (rh + 2) * cw2; i += 256) {
        int r = (int)((float)i * inv2);  // i / cw2: the estimate is one off at most (i < 2^24)
        int c = i - r * cw2;
        if (c < 0) { r--; c += cw2; } else if (c >= cw2) { r++; c -= cw2; }
        const int sy = wy0 - 1 + r, sx = wx0 - 1 + c;
        s_raw[r * pitch + c] = ((unsigned)sx < (unsigned)P.sw && (unsigned)sy < (unsigned)P.sh) ? P.src[(long long)sy * P.sstride + sx] : (uint8_t)0;
    }
    __syncthreads();
    for (int i = tid; i < rh * cw; i += 256) {
        int r = (int)((float)i * inv1);
        int c = i - r * cw;
        if (c < 0) { r--; c += cw; } else if (c >= cw) { r++; c -= cw; }
        uint32_t m = 0;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) m = max(m, (uint32_t)s_raw[(r + dy) * pitch + c + dx]);
        s_dil[r * pitch + c] = (uint8_t)m;
    }
    __syncthreads();
    if (!mine) return;
    int o0[SEAM1_COLS], o1[SEAM1_COLS];
    uint32_t c1[SEAM1_COLS];
#pragma unroll
    for (int j = 0; j < SEAM1_COLS; j++) {
        const int2 t = s_xt[lane * SEAM1_COLS + j];
        o0[j] = t.x - wx0;
        o1[j] = min(t.x + 1, P.sw - 1) - wx0;
        c1[j] = (uint32_t)t.y & 0xffffu;
    }
    uint32_t h0[SEAM1_COLS], h1[SEAM1_COLS];
    int cur = -(1 << 30);
    const uint32_t keep_lo = x8 + 4 <= P.dw ? 0xffffffffu : (x8 + 0 < P.dw ? 0xffffffffu >> (8 * (4 - (P.dw - x8))) : 0u);
    const uint32_t keep_hi = x8 + 8 <= P.dw ? 0xffffffffu : (x8 + 4 < P.dw ? 0xffffffffu >> (8 * (8 - (P.dw - x8))) : 0u);
#pragma unroll
    for (int r = 0; r < SEAM1_ROWS; r++) {
        const int y = yb + r;
        if (y >= P.dh) break;
        const int2 ty = s_yt[wv * SEAM1_ROWS + r];
        const int row = __builtin_amdgcn_readfirstlane(ty.x);
        if (row != cur) {
            cur = row;
            const uint8_t* d0 = s_dil + (cur - wy0) * pitch;
            const uint8_t* d1 = s_dil + (min(cur + 1, P.sh - 1) - wy0) * pitch;
#pragma unroll
            for (int j = 0; j < SEAM1_COLS; j++) {
                const uint32_t a0 = 256u - c1[j];
                h0[j] = (uint32_t)d0[o0[j]] * a0 + (uint32_t)d0[o1[j]] * c1[j];
                h1[j] = (uint32_t)d1[o0[j]] * a0 + (uint32_t)d1[o1[j]] * c1[j];
            }
        }
        const uint32_t cy1 = (uint32_t)__builtin_amdgcn_readfirstlane(ty.y) & 0xffffu, cy0 = 256u - cy1;
        const bool iy = (__builtin_amdgcn_readfirstlane(ty.y) >> 16) != 0;
        uint32_t sum[SEAM1_COLS];  // the result is byte 2 of every sum: (h0 cy0 + h1 cy1 + 2^15) >> 16 <= 255, ((h0 + 128) >> 8) << 16 likewise
#pragma unroll
        for (int j = 0; j < SEAM1_COLS; j++)
            sum[j] = iy ? __umul24(h1[j], cy1) + (__umul24(h0[j], cy0) + 32768u) : (h0[j] + 128u) << 8;
        uint2 out;
        out.x = __builtin_amdgcn_perm(sum[1], sum[0], 0x0c0c0602u) | __builtin_amdgcn_perm(sum[3], sum[2], 0x06020c0cu);
        out.y = __builtin_amdgcn_perm(sum[5], sum[4], 0x0c0c0602u) | __builtin_amdgcn_perm(sum[7], sum[6], 0x06020c0cu);
        out.x &= am[r].x & keep_lo;  // bytes beyond the image inside the row pitch: 0, whatever stands in the final mask's padding
        out.y &= am[r].y & keep_hi;
        *reinterpret_cast<uint2*>(P.dst + (long long)y * P.dstride + x8) = out;
    }
}
}  // namespace

// -> STX_OK and *done = true when the images qualified and the launch was made; *done = false: the caller takes the table path
int stx_launch_seam_resize_lds(stx_ctx* ctx, int n, const stx_buf* const* seams, const stx_buf* const* masks, stx_buf* const* dsts,
                               const int* full_wh_xy0, bool* done)
{
    *done = false;
    static const bool off = getenv("STITCHING_AMD_SEAM_LDS") && atoi(getenv("STITCHING_AMD_SEAM_LDS")) == 0;
    if (off) return STX_OK;
    int pitch = 0, rows = 0;
    for (int i = 0; i < n; i++) {
        const stx_buf *m = masks[i], *d = dsts[i];
        const size_t w8 = ((size_t)d->w + 7) & ~(size_t)7;
        if (((uintptr_t)m->ptr & 7) || (m->stride & 7) || w8 > m->stride || ((uintptr_t)d->ptr & 7) || (d->stride & 7) || w8 > d->stride) return STX_OK;
        const int fw = full_wh_xy0 ? full_wh_xy0[4 * i] : d->w, fh = full_wh_xy0 ? full_wh_xy0[4 * i + 1] : d->h;
        const double xs = 1.0 / ((double)fw / (double)seams[i]->w), ys = 1.0 / ((double)fh / (double)seams[i]->h);
        // source columns / rows under a tile: floor((n - 1) scale) + 2, and one more for the fraction the first one starts at
        pitch = std::max(pitch, std::min(seams[i]->w, (int)std::ceil(SEAM1_TW * xs) + 3) + 2);
        rows = std::max(rows, std::min(seams[i]->h, (int)std::ceil(SEAM1_TH * ys) + 3));
    }
    pitch = (pitch + 3) & ~3;
    const size_t raw_bytes = (size_t)pitch * (rows + 2), lds = raw_bytes + (size_t)pitch * rows;
    if (lds > (40u << 10)) return STX_OK;  // an enlargement factor near 1: the window would not leave room for a second workgroup per CU
    for (int base = 0; base < n; base += SEAM_BATCH) {
        const int m = std::min(SEAM_BATCH, n - base);
        Seam1BatchK B = {};
        B.pitch = pitch; B.raw_bytes = (int)raw_bytes;
        int mdw = 0, mdh = 0;
        double bytes = 0.0;
        for (int i = 0; i < m; i++) {
            const int g = base + i;
            Seam1K& K = B.k[i];
            K.src = seams[g]->ptr; K.sstride = (long long)seams[g]->stride; K.sw = seams[g]->w; K.sh = seams[g]->h;
            K.dst = dsts[g]->ptr; K.dstride = (long long)dsts[g]->stride; K.dw = dsts[g]->w; K.dh = dsts[g]->h;
            K.am = masks[g]->ptr; K.amstride = (long long)masks[g]->stride;
            const int fw = full_wh_xy0 ? full_wh_xy0[4 * g] : K.dw, fh = full_wh_xy0 ? full_wh_xy0[4 * g + 1] : K.dh;
            K.x0 = full_wh_xy0 ? full_wh_xy0[4 * g + 2] : 0; K.y0 = full_wh_xy0 ? full_wh_xy0[4 * g + 3] : 0;
            K.xscale = 1.0 / ((double)fw / (double)K.sw); K.yscale = 1.0 / ((double)fh / (double)K.sh);
            mdw = std::max(mdw, K.dw); mdh = std::max(mdh, K.dh);
            bytes += (double)K.sw * K.sh + 2.0 * K.dw * K.dh;
        }
        StxProfScope prof(ctx, "seam_mask_resize", bytes);
        hipLaunchKernelGGL(seam_resize_lds_kernel, dim3((mdw + SEAM1_TW - 1) / SEAM1_TW, (mdh + SEAM1_TH - 1) / SEAM1_TH, m), dim3(256), lds, ctx->stream, B);
    }
    *done = true;
    return check_launch("seam_mask_resize");
}

int stx_launch_resize_exact(stx_ctx* ctx, const stx_buf* src, stx_buf* dst, const int* d_xt, const int* d_yt, bool dilate,
                            const stx_buf* andmask)
{
    ResizeK K;
    K.src = src->ptr; K.sstride = (long long)src->stride; K.sw = src->w; K.sh = src->h;
    K.dst = dst->ptr; K.dstride = (long long)dst->stride; K.dw = dst->w; K.dh = dst->h;
    K.xt = (const int2*)d_xt; K.yt = (const int2*)d_yt;
    K.andmask = andmask ? andmask->ptr : nullptr; K.amstride = andmask ? (long long)andmask->stride : 0;
    const dim3 grid((dst->w + 63) / 64, (dst->h + 3) / 4);
    StxProfScope prof(ctx, dilate ? "seam_mask_resize" : "resize_linear_exact", (double)src->w * src->h * src->c + (double)dst->w * dst->h * dst->c * (andmask ? 2 : 1));
    // dword access to the destination and to the final mask: 4-byte aligned rows (every stx_buf_new image; views may not be),
    // and the mask rows must be readable up to the next multiple of 4 columns
    const bool fast = dilate && andmask && ((uintptr_t)dst->ptr & 3) == 0 && (dst->stride & 3) == 0 && ((uintptr_t)andmask->ptr & 3) == 0 &&
                      (andmask->stride & 3) == 0 && (size_t)((dst->w + 3) & ~3) <= andmask->stride && (size_t)((dst->w + 3) & ~3) <= dst->stride;
    if (fast) {
        void* tmp = nullptr;
        const size_t tstride = ((size_t)src->w + 63) & ~(size_t)63;
        STX_TRY(stx_dev_alloc(ctx, tstride * src->h, &tmp));
        hipLaunchKernelGGL(dilate3x3_kernel, dim3((src->w + 63) / 64, (src->h + 3) / 4), dim3(256), 0, ctx->stream, src->ptr,
                           (long long)src->stride, src->w, src->h, (uint8_t*)tmp, (long long)tstride);
        K.src = (const uint8_t*)tmp; K.sstride = (long long)tstride;
        hipLaunchKernelGGL(seam_resize4_kernel, dim3((dst->w + 255) / 256, (dst->h + 4 * SEAM_ROWS - 1) / (4 * SEAM_ROWS)), dim3(256), 0, ctx->stream, K);
        stx_dev_free(ctx, tmp);  // stream-ordered reuse
    } else if (dilate) hipLaunchKernelGGL((resize_exact_kernel<1, true>), grid, dim3(256), 0, ctx->stream, K);
    else if (src->c == 1) hipLaunchKernelGGL((resize_exact_kernel<1, false>), grid, dim3(256), 0, ctx->stream, K);
    else if (src->c == 3) hipLaunchKernelGGL((resize_exact_kernel<3, false>), grid, dim3(256), 0, ctx->stream, K);
    else return stx_fail(STX_ERR_UNSUPPORTED, "resize: 1 or 3 channels");
    return check_launch("resize_linear_exact");
}

// ---------------------------------------------------------------------------------------------
// BlocksCompensator::apply (next row N1, the reference's default "gain_blocks" compensator): the small fp32 gain map is
// interpolated to the image size exactly as cv::resize(INTER_LINEAR) does for CV_32F — fp32 throughout, horizontal
// S[s] * a0 + S[s+1] * a1 (S[s] alone at the right end), vertical R0 * b0 + R1 * b1 with the rows clamped, no FMA — and
// multiplied in with cv::multiply's fp32 product + cvRound + saturation.  Fused: the full-size gain map never exists.
// Tables (host, double -> float as upstream): x = (s, bits of f), y likewise (s may be -1).
// ---------------------------------------------------------------------------------------------
namespace {
struct BlockGainK {
    uint8_t* img; long long stride; int w, h;
    const float* gmap; long long gstride; int gw, gh, gc;  // gstride in floats; gc = 1 (one map) or 3 (BGR maps, interleaved)
    const int2* xt; const int2* yt;
};
STX_DEV float gain_hrow(const BlockGainK& P, int row, int sx, int ch, float a0, float a1)
{
    const float* r = P.gmap + (long long)row * P.gstride + ch;
    if (sx >= P.gw - 1) return r[sx * P.gc];
    return stxd::fadd(stxd::fmul(r[sx * P.gc], a0), stxd::fmul(r[(sx + 1) * P.gc], a1));
}
__global__ __launch_bounds__(256) void block_gain_kernel(BlockGainK P)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= P.w || y >= P.h) return;
    const int2 tx = P.xt[x], ty = P.yt[y];
    const float a1 = __int_as_float(tx.y), a0 = stxd::fsub(1.f, a1), b1 = __int_as_float(ty.y), b0 = stxd::fsub(1.f, b1);
    const int r0 = min(max(ty.x, 0), P.gh - 1), r1 = min(max(ty.x + 1, 0), P.gh - 1);
    uint8_t* p = P.img + (long long)y * P.stride + x * 3;
    float g = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (c == 0 || P.gc == 3)
            g = stxd::fadd(stxd::fmul(gain_hrow(P, r0, tx.x, c, a0, a1), b0), stxd::fmul(gain_hrow(P, r1, tx.x, c, a0, a1), b1));
        p[c] = (uint8_t)min(max(stxd::cv_round(stxd::fmul((float)p[c], g)), 0), 255);
    }
}
}  // namespace

int stx_launch_block_gain(stx_ctx* ctx, stx_buf* img, const stx_buf* gmap, const int* d_xt, const int* d_yt)
{
    BlockGainK K;
    K.img = img->ptr; K.stride = (long long)img->stride; K.w = img->w; K.h = img->h;
    K.gmap = (const float*)gmap->ptr; K.gstride = (long long)(gmap->stride / sizeof(float)); K.gw = gmap->w; K.gh = gmap->h;
    K.gc = gmap->c;
    K.xt = (const int2*)d_xt; K.yt = (const int2*)d_yt;
    StxProfScope prof(ctx, "block_gain_apply", 6.0 * img->w * img->h);
    hipLaunchKernelGGL(block_gain_kernel, dim3((img->w + 63) / 64, (img->h + 3) / 4), dim3(256), 0, ctx->stream, K);
    return check_launch("block_gain_apply");
}

// ---------------------------------------------------------------------------------------------
// The same product for all images of a panorama in two launches, at HBM rate (stx_block_gain_apply_batch).  cv::resize(INTER_LINEAR)
// is separable and rounds each pass to fp32, so the horizontal pass is made ONCE per gain-map row: H[r][x] = S[r][s] a0 + S[r][s+1] a1
// for every column x of the image (gain_rows_kernel: gh x w floats per image, a few hundred KB, L2 resident; the coefficient set-up
// f = (float)((x + 0.5) scale - 0.5) runs on the device in IEEE double, operation for operation the host loop of linear_f32_table).
// The product pass then needs two 16-byte reads of H per 4 pixels and row: g = H[r0][x] b0 + H[r1][x] b1, cvRound(p g) saturated.
// One lane = 4 pixels (three dwords in, three out), one wavefront = 256 pixels of a row, 4 rows one after the other.
// An image may be the rectangle at (x0, y0) of a larger warped image (full_w x full_h: StitchJob's seam-cell crops): the gain map is
// laid over the FULL image, as BlocksCompensator::apply lays it over the whole warped image.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int GAIN_BATCH = 16;
struct BlockGain4K {
    uint8_t* img; long long stride; int w, h;          // the (sub-)image, whole rows of the library's own buffers (4-pixel groups)
    int x0, y0, full_w, full_h;                        // its place in the full warped image
    const float* gmap; long long gstride; int gw, gh;  // gstride in floats
    double xscale, yscale;                             // 1 / (full / gain-map size), host double
    float* H; long long hstride;                       // [gh][hstride] floats x GC; hstride = w rounded up to 4 (x GC)
    int2* yt;                                          // [h]: (source row, bits of the fraction)
    int fast;                                          // host-proved: every gain finite and below 2^31 / 255 (no product leaves int range)
};
struct BlockGainBatchK { BlockGain4K k[GAIN_BATCH]; };

STX_DEV void gain_coeff(int d, double scale, int src_n, bool clamp, int& sidx, float& f)
{
    f = (float)(((double)d + 0.5) * scale - 0.5);
    sidx = (int)floorf(f);
    f = stxd::fsub(f, (float)sidx);
    if (clamp) {
        if (sidx < 0) { sidx = 0; f = 0.f; }
        else if (sidx >= src_n - 1) { sidx = src_n - 1; f = 0.f; }
    }
}

template <int GC>
__global__ __launch_bounds__(256) void gain_rows_kernel(BlockGainBatchK B)
{
    const BlockGain4K& P = B.k[blockIdx.z];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.y == 0) {  // row table
        if (i < P.h) {
            int sy; float f;
            gain_coeff(P.y0 + i, P.yscale, P.gh, false, sy, f);
            P.yt[i] = make_int2(sy, __float_as_int(f));
        }
        return;
    }
    const int r = blockIdx.y - 1;  // gain-map row
    // the columns w .. round_up4(w) - 1 of the row pitch are written too (the gain of the last column): the consumers read a row in
    // groups of 4 and multiply the image's ROW PADDING with what stands there — scratch left as it was made those padding bytes differ
    // from run to run
    const int wpad = (P.w + 3) & ~3;
    if (r >= P.gh || i >= wpad) return;
    int sx; float a1;
    gain_coeff(P.x0 + min(i, P.w - 1), P.xscale, P.gw, true, sx, a1);
    const float a0 = stxd::fsub(1.f, a1);
    const float* row = P.gmap + (long long)r * P.gstride;
#pragma unroll
    for (int c = 0; c < GC; c++) {
        float v;
        if (sx >= P.gw - 1) v = row[sx * GC + c];
        else v = stxd::fadd(stxd::fmul(row[sx * GC + c], a0), stxd::fmul(row[(sx + 1) * GC + c], a1));
        P.H[(long long)r * P.hstride + (long long)i * GC + c] = v;
    }
}

// cvRound(p g) saturated to u8 into byte `sel` of `old`: v_cvt_pk_u8_f32 is the round-half-even of cvRound AND the saturation
template <bool FAST>
STX_DEV uint32_t gain_px(uint32_t old, int sel, float p, float g)
{
    float v = stxd::fmul(p, g);
    if (!FAST) v = v < 2147483648.f ? v : 0.f;  // cvRound's INT_MIN for NaN / out-of-range products saturates to 0
    return __builtin_amdgcn_cvt_pk_u8_f32(v, (uint32_t)sel, old);  // rounds to nearest even and saturates itself: tools/ubench/cvt_pk_u8.hip
}

template <int GC, bool FAST>
__global__ __launch_bounds__(256) void block_gain4_kernel(BlockGainBatchK B)
{
    const BlockGain4K& P = B.k[blockIdx.z];
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (x4 >= P.w) return;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int y = blockIdx.y * 16 + wv * 4 + j;
        if (y >= P.h) break;
        const int2 ty = P.yt[y];
        const float b1 = __int_as_float(ty.y), b0 = stxd::fsub(1.f, b1);
        const int r0 = min(max(ty.x, 0), P.gh - 1), r1 = min(max(ty.x + 1, 0), P.gh - 1);
        const float* h0 = P.H + (long long)r0 * P.hstride + (long long)x4 * GC;
        const float* h1 = P.H + (long long)r1 * P.hstride + (long long)x4 * GC;
        float g[4 * GC];
#pragma unroll
        for (int q = 0; q < GC; q++) {
            const float4 u = reinterpret_cast<const float4*>(h0)[q], v = reinterpret_cast<const float4*>(h1)[q];
            g[4 * q + 0] = stxd::fadd(stxd::fmul(u.x, b0), stxd::fmul(v.x, b1));
            g[4 * q + 1] = stxd::fadd(stxd::fmul(u.y, b0), stxd::fmul(v.y, b1));
            g[4 * q + 2] = stxd::fadd(stxd::fmul(u.z, b0), stxd::fmul(v.z, b1));
            g[4 * q + 3] = stxd::fadd(stxd::fmul(u.w, b0), stxd::fmul(v.w, b1));
        }
        uint32_t* p = reinterpret_cast<uint32_t*>(P.img + (long long)y * P.stride + (long long)x4 * 3);
        uint32_t d[3] = {p[0], p[1], p[2]}, o[3] = {0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < 12; k++) {  // byte k of the 12: pixel k / 3, channel k % 3
            const float pv = (float)((d[k >> 2] >> (8 * (k & 3))) & 255u);
            o[k >> 2] = gain_px<FAST>(o[k >> 2], k & 3, pv, GC == 1 ? g[k / 3] : g[k]);
        }
        p[0] = o[0]; p[1] = o[1]; p[2] = o[2];
    }
}
}  // namespace

// H rows + row tables of n rectangles (see stx_launch_block_gain_batch); imgs may be null (a consumer that multiplies the gain in itself)
static void fill_gain_k(BlockGain4K& K, stx_buf* img, int w, int h, const stx_buf* gmap, const int* q, float* H, void* yt, int fast)
{
    K.img = img ? img->ptr : nullptr; K.stride = img ? (long long)img->stride : 0; K.w = w; K.h = h;
    K.full_w = q ? q[0] : w; K.full_h = q ? q[1] : h;
    K.x0 = q ? q[2] : 0; K.y0 = q ? q[3] : 0;
    K.gmap = (const float*)gmap->ptr; K.gstride = (long long)(gmap->stride / sizeof(float));
    K.gw = gmap->w; K.gh = gmap->h;
    K.xscale = 1.0 / ((double)K.full_w / (double)K.gw);
    K.yscale = 1.0 / ((double)K.full_h / (double)K.gh);
    K.H = H; K.hstride = (long long)((w + 3) & ~3) * gmap->c;
    K.yt = (int2*)yt;
    K.fast = fast;
}

static void launch_gain_rows(stx_ctx* ctx, const BlockGainBatchK& B, int m, int gc)
{
    int mt = 0, mgh = 0;
    double rows_bytes = 0.0;
    for (int i = 0; i < m; i++) {
        mt = std::max(mt, std::max(B.k[i].w, B.k[i].h));
        mgh = std::max(mgh, B.k[i].gh);
        rows_bytes += 4.0 * gc * B.k[i].gh * B.k[i].w + 8.0 * B.k[i].h;
    }
    StxProfScope prof(ctx, "block_gain_rows", rows_bytes);
    const dim3 grid((mt + 255) / 256, mgh + 1, m);
    if (gc == 1) hipLaunchKernelGGL(gain_rows_kernel<1>, grid, dim3(256), 0, ctx->stream, B);
    else hipLaunchKernelGGL(gain_rows_kernel<3>, grid, dim3(256), 0, ctx->stream, B);
}

int stx_launch_gain_rows(stx_ctx* ctx, int n, const int* wh, const stx_buf* const* gmaps, const int* full_wh_xy0, float* const* Hs,
                         void* const* yts)
{
    for (int base = 0; base < n; base += GAIN_BATCH) {
        const int m = std::min(GAIN_BATCH, n - base);
        BlockGainBatchK B;
        memset(&B, 0, sizeof(B));
        for (int i = 0; i < m; i++) {
            const int g = base + i;
            fill_gain_k(B.k[i], nullptr, wh[2 * g], wh[2 * g + 1], gmaps[g], full_wh_xy0 ? full_wh_xy0 + 4 * g : nullptr, Hs[g], yts[g], 1);
        }
        launch_gain_rows(ctx, B, m, gmaps[base]->c);
    }
    return check_launch("block_gain_rows");
}

// Hs / yts: per image device scratch (gh x hstride x gc floats, h int2), carved by the caller from one allocation
int stx_launch_block_gain_batch(stx_ctx* ctx, int n, stx_buf* const* imgs, const stx_buf* const* gmaps, const int* full_wh_xy0,
                                float* const* Hs, void* const* yts, const int* fast)
{
    for (int base = 0; base < n; base += GAIN_BATCH) {
        const int m = std::min(GAIN_BATCH, n - base);
        const int gc = gmaps[base]->c;
        BlockGainBatchK B;
        memset(&B, 0, sizeof(B));
        int mw = 0, mh = 0;
        bool all_fast = true;
        double bytes = 0.0;
        for (int i = 0; i < m; i++) {
            const int g = base + i;
            fill_gain_k(B.k[i], imgs[g], imgs[g]->w, imgs[g]->h, gmaps[g], full_wh_xy0 ? full_wh_xy0 + 4 * g : nullptr, Hs[g], yts[g], fast[g]);
            all_fast = all_fast && fast[g];
            mw = std::max(mw, B.k[i].w); mh = std::max(mh, B.k[i].h);
            bytes += 6.0 * B.k[i].w * B.k[i].h;
        }
        launch_gain_rows(ctx, B, m, gc);
        StxProfScope prof(ctx, "block_gain_apply", bytes);
        const dim3 grid((mw + 255) / 256, (mh + 15) / 16, m);
        if (gc == 1 && all_fast) hipLaunchKernelGGL((block_gain4_kernel<1, true>), grid, dim3(256), 0, ctx->stream, B);
        else if (gc == 1) hipLaunchKernelGGL((block_gain4_kernel<1, false>), grid, dim3(256), 0, ctx->stream, B);
        else if (all_fast) hipLaunchKernelGGL((block_gain4_kernel<3, true>), grid, dim3(256), 0, ctx->stream, B);
        else hipLaunchKernelGGL((block_gain4_kernel<3, false>), grid, dim3(256), 0, ctx->stream, B);
    }
    return check_launch("block_gain_apply");
}
