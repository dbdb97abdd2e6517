// stx_blend_fast.hip — register-blocked multi-band kernels for the fine pyramid levels (gfx950).
//
// Same arithmetic as the generic kernels in stx_blend.hip (bit-identical results), restructured
// for the memory system: every lane owns a strip of adjacent pixels so that global accesses are
// 8/16-byte vector loads and stores on 64-byte-aligned planar rows, the 5-tap pyrDown runs as a
// sliding window over rows (2 new input rows per output row instead of 5), the 3x3 pyrUp
// neighbourhood is shared by an 8x2 output patch, and the per-tile image list is compacted once
// per workgroup so that lanes never walk the whole image table.
//
//   mb_down0_fast : bordered level 0 (u8 BGR + mask, copyMakeBorder as an index map) -> G_1, W_1
//   mb_down_fast  : G_i, W_i -> G_{i+1}, W_{i+1}
//   mb_level_fast : gather + normalise + collapse for levels <= B-3 (2^(B-level) >= 8, so an
//                   8-pixel strip never straddles a feed-rectangle edge); level 0 writes the
//                   u8 panorama + mask (+ int16 result)
#include "stx_blend_kernels.h"
#include "stx_device_math.h"

using namespace stxd;

namespace {

constexpr float WEIGHT_EPS = 1e-5f;
constexpr float INV255 = 0.0039215688593685627f;  // (float)(1./255.)
constexpr float INV256 = 0.00390625f;

struct __attribute__((aligned(4))) U4a4 { uint32_t v[4]; };  // 16 bytes, only dword-aligned
struct __attribute__((aligned(4))) U2a4 { uint32_t v[2]; };

STX_DEV int s16lo(uint32_t v) { return (int)(short)(v & 0xffffu); }
STX_DEV int s16hi(uint32_t v) { return (int)(short)(v >> 16); }
STX_DEV uint32_t pack16(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
STX_DEV uint32_t byte_of(const uint32_t* w, int k) { return (w[k >> 2] >> (8 * (k & 3))) & 255u; }

STX_DEV float h5f(float s0, float s1, float s2, float s3, float s4)
{
    return fadd(fadd(fadd(fmul(s2, 6.f), fmul(fadd(s1, s3), 4.f)), s0), s4);
}
STX_DEV int h5i(int s0, int s1, int s2, int s3, int s4) { return s2 * 6 + (s1 + s3) * 4 + s0 + s4; }

// ---------------------------------------------------------------------------------------------
// pyrDown, sliding window.  One lane = 2 adjacent outputs x R output rows.
// ---------------------------------------------------------------------------------------------
struct HRow {  // horizontal 1-4-6-4-1 sums of one input row for the lane's 2 outputs
    int v[3][2];
    float w[2];
};

// level >= 1 source: planar int16 x3 + fp32
STX_DEV void hrow_planar(const short* __restrict__ G, long long gs, long long gp, const float* __restrict__ W,
                         long long ws, int iw, int ih, int row, int c0, bool fastx, HRow& o)
{
    const int sy = reflect101(row, ih);
    int idx[7];
    if (!fastx) {
#pragma unroll
        for (int j = 0; j < 7; j++) idx[j] = reflect101(c0 + j, iw);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const short* p = G + c * gp + (long long)sy * gs;
        int s[7];
        if (fastx) {
            uint32_t a = *reinterpret_cast<const uint32_t*>(p + c0);
            uint2 b = *reinterpret_cast<const uint2*>(p + c0 + 2);
            s[0] = s16lo(a); s[1] = s16hi(a);
            s[2] = s16lo(b.x); s[3] = s16hi(b.x); s[4] = s16lo(b.y); s[5] = s16hi(b.y);
            s[6] = p[c0 + 6];
        } else {
#pragma unroll
            for (int j = 0; j < 7; j++) s[j] = p[idx[j]];
        }
        o.v[c][0] = h5i(s[0], s[1], s[2], s[3], s[4]);
        o.v[c][1] = h5i(s[2], s[3], s[4], s[5], s[6]);
    }
    const float* q = W + (long long)sy * ws;
    float f[7];
    if (fastx) {
        float2 a = *reinterpret_cast<const float2*>(q + c0);
        float4 b = *reinterpret_cast<const float4*>(q + c0 + 2);
        f[0] = a.x; f[1] = a.y; f[2] = b.x; f[3] = b.y; f[4] = b.z; f[5] = b.w;
        f[6] = q[c0 + 6];
    } else {
#pragma unroll
        for (int j = 0; j < 7; j++) f[j] = q[idx[j]];
    }
    o.w[0] = h5f(f[0], f[1], f[2], f[3], f[4]);
    o.w[1] = h5f(f[2], f[3], f[4], f[5], f[6]);
}

// level 0 source: u8 BGR interleaved image seen through copyMakeBorder(REFLECT), u8 mask through
// copyMakeBorder(CONSTANT 0) and convertTo(32F, 1/255)
STX_DEV void hrow_level0(const StxMbImage& im, int row, int c0, bool fastx, HRow& o)
{
    const int by = reflect101(row, im.fh) - im.top;  // bordered row -> image row
    const bool yin = (unsigned)by < (unsigned)im.ih;
    const int sy = reflect(by, im.ih);
    const uint8_t* irow = im.img0 + (long long)sy * im.img0_stride;
    int px[7][3];
    float f[7];
    if (fastx) {
        const int a0 = c0 - im.left;  // 7 contiguous image columns a0 .. a0+6, all inside
        {
            const long long off = (long long)a0 * 3;
            const uint8_t* q = irow + (off & ~3ll);
            const uint32_t s = (uint32_t)off & 3u;
            U4a4 d0 = *reinterpret_cast<const U4a4*>(q);
            U2a4 d1 = *reinterpret_cast<const U2a4*>(q + 16);
            uint32_t w[6];
            w[0] = __builtin_amdgcn_alignbyte(d0.v[1], d0.v[0], s);
            w[1] = __builtin_amdgcn_alignbyte(d0.v[2], d0.v[1], s);
            w[2] = __builtin_amdgcn_alignbyte(d0.v[3], d0.v[2], s);
            w[3] = __builtin_amdgcn_alignbyte(d1.v[0], d0.v[3], s);
            w[4] = __builtin_amdgcn_alignbyte(d1.v[1], d1.v[0], s);
            w[5] = __builtin_amdgcn_alignbyte(0u, d1.v[1], s);
#pragma unroll
            for (int j = 0; j < 7; j++) {
                px[j][0] = (int)byte_of(w, 3 * j);
                px[j][1] = (int)byte_of(w, 3 * j + 1);
                px[j][2] = (int)byte_of(w, 3 * j + 2);
            }
        }
        if (yin) {
            const long long off = (long long)by * im.mask0_stride + a0;
            const uint8_t* q = im.mask0 + (off & ~3ll);
            const uint32_t s = (uint32_t)off & 3u;
            uint32_t d0 = *reinterpret_cast<const uint32_t*>(q), d1 = *reinterpret_cast<const uint32_t*>(q + 4),
                     d2 = *reinterpret_cast<const uint32_t*>(q + 8);
            uint32_t w[2];
            w[0] = __builtin_amdgcn_alignbyte(d1, d0, s);
            w[1] = __builtin_amdgcn_alignbyte(d2, d1, s);
#pragma unroll
            for (int j = 0; j < 7; j++) f[j] = fmul((float)byte_of(w, j), INV255);
        } else {
#pragma unroll
            for (int j = 0; j < 7; j++) f[j] = 0.f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const int bx = reflect101(c0 + j, im.fw) - im.left;
            const int sx = reflect(bx, im.iw);
            const uint8_t* p = irow + sx * 3;
            px[j][0] = p[0]; px[j][1] = p[1]; px[j][2] = p[2];
            f[j] = (yin && (unsigned)bx < (unsigned)im.iw)
                       ? fmul((float)im.mask0[(long long)by * im.mask0_stride + bx], INV255) : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        o.v[c][0] = h5i(px[0][c], px[1][c], px[2][c], px[3][c], px[4][c]);
        o.v[c][1] = h5i(px[2][c], px[3][c], px[4][c], px[5][c], px[6][c]);
    }
    o.w[0] = h5f(f[0], f[1], f[2], f[3], f[4]);
    o.w[1] = h5f(f[2], f[3], f[4], f[5], f[6]);
}

template <bool L0, int R>
__global__ __launch_bounds__(256) void mb_down_fast_kernel(StxMbImage im, int lv)
{
    const int iw = im.fw >> lv, ih = im.fh >> lv;
    const int ow = iw >> 1, oh = ih >> 1;
    const int t = blockIdx.x * 64 + (threadIdx.x & 63);
    const int xo = 2 * t;
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * R;
    if (xo >= ow || y0 >= oh) return;
    const int c0 = 4 * t - 2;
    bool fastx = c0 >= 0 && c0 + 6 < iw;
    if (L0) fastx = fastx && (c0 - im.left) >= 0 && (c0 - im.left) + 6 < im.iw;

    const short* G = L0 ? nullptr : im.g[lv];
    const long long gs = L0 ? 0 : im.g_stride[lv], gp = L0 ? 0 : im.g_plane[lv];
    const float* W = L0 ? nullptr : im.wt[lv];
    const long long ws = L0 ? 0 : im.wt_stride[lv];
    short* O = im.g[lv + 1];
    const long long os = im.g_stride[lv + 1], op = im.g_plane[lv + 1];
    float* OW = im.wt[lv + 1];
    const long long ows = im.wt_stride[lv + 1];

    HRow h0, h1, h2, h3, h4;
    if (L0) {
        hrow_level0(im, 2 * y0 - 2, c0, fastx, h0);
        hrow_level0(im, 2 * y0 - 1, c0, fastx, h1);
        hrow_level0(im, 2 * y0, c0, fastx, h2);
    } else {
        hrow_planar(G, gs, gp, W, ws, iw, ih, 2 * y0 - 2, c0, fastx, h0);
        hrow_planar(G, gs, gp, W, ws, iw, ih, 2 * y0 - 1, c0, fastx, h1);
        hrow_planar(G, gs, gp, W, ws, iw, ih, 2 * y0, c0, fastx, h2);
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int y = y0 + r;
        if (y >= oh) break;
        if (L0) {
            hrow_level0(im, 2 * y + 1, c0, fastx, h3);
            hrow_level0(im, 2 * y + 2, c0, fastx, h4);
        } else {
            hrow_planar(G, gs, gp, W, ws, iw, ih, 2 * y + 1, c0, fastx, h3);
            hrow_planar(G, gs, gp, W, ws, iw, ih, 2 * y + 2, c0, fastx, h4);
        }
        const bool two = xo + 1 < ow;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int a = (h5i(h0.v[c][0], h1.v[c][0], h2.v[c][0], h3.v[c][0], h4.v[c][0]) + 128) >> 8;
            int b = (h5i(h0.v[c][1], h1.v[c][1], h2.v[c][1], h3.v[c][1], h4.v[c][1]) + 128) >> 8;
            short* o = O + c * op + (long long)y * os + xo;
            if (two) *reinterpret_cast<uint32_t*>(o) = pack16(a, b);
            else o[0] = (short)a;
        }
        float wa = fmul(h5f(h0.w[0], h1.w[0], h2.w[0], h3.w[0], h4.w[0]), INV256);
        float wb = fmul(h5f(h0.w[1], h1.w[1], h2.w[1], h3.w[1], h4.w[1]), INV256);
        float* ow_ = OW + (long long)y * ows + xo;
        if (two) *reinterpret_cast<float2*>(ow_) = make_float2(wa, wb);
        else ow_[0] = wa;
        h0 = h2; h1 = h3; h2 = h4;
    }
}

// ---------------------------------------------------------------------------------------------
// gather + normalise + collapse.  One lane = 8 adjacent pixels x 2 rows; tile 512 x 8.
// ---------------------------------------------------------------------------------------------
STX_DEV int s6(int v) { return (int)(short)((v + 32) >> 6); }

// pyrUp_<FixPtCast<short,6>> of one plane for the 8x2 patch whose coarse origin is (cx, cy);
// cx is a multiple of 4 and cx+3 < cw
STX_DEV void up_patch(const short* __restrict__ plane, long long stride, int cw, int ch, int cx, int cy, int up[2][8])
{
    const int rr[3] = {up_idx(cy - 1, ch), cy, up_idx(cy + 1, ch)};
    const int cl = up_idx(cx - 1, cw), cr = up_idx(cx + 4, cw);
    int he[3][4], ho[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const short* p = plane + (long long)rr[r] * stride;
        int c[6];
        uint2 v = *reinterpret_cast<const uint2*>(p + cx);
        c[0] = p[cl];
        c[1] = s16lo(v.x); c[2] = s16hi(v.x); c[3] = s16lo(v.y); c[4] = s16hi(v.y);
        c[5] = p[cr];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            he[r][j] = c[j] + 6 * c[j + 1] + c[j + 2];
            ho[r][j] = 4 * (c[j + 1] + c[j + 2]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        up[0][2 * j] = s6(he[0][j] + 6 * he[1][j] + he[2][j]);
        up[0][2 * j + 1] = s6(ho[0][j] + 6 * ho[1][j] + ho[2][j]);
        up[1][2 * j] = s6(4 * (he[1][j] + he[2][j]));
        up[1][2 * j + 1] = s6(4 * (ho[1][j] + ho[2][j]));
    }
}

template <bool L0>
__global__ __launch_bounds__(256) void mb_level_fast_kernel(MbLevelK P)
{
    __shared__ int s_list[64];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int lv = P.level;
    const int tile_x = blockIdx.x * 512, tile_y = blockIdx.y * 8;
    const int X0 = tile_x + (tid & 63) * 8, Y0 = tile_y + (tid >> 6) * 2;
    const int lim_w = L0 ? P.final_w : P.pw, lim_h = L0 ? P.final_h : P.ph;
    const bool active = X0 < lim_w && Y0 < lim_h;

    int acc[2][8][3];
    float ws[2][8];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[r][j][0] = acc[r][j][1] = acc[r][j][2] = 0;
            ws[r][j] = 0.f;
        }

    for (int base = 0; base < P.n_images; base += 64) {
        __syncthreads();
        if (tid < 64) {
            const int k = base + tid;
            bool hit = false;
            if (k < P.n_images) {
                const StxMbImage& im = P.images[k];
                int rx, ry, rw, rh;
                if (L0) { rx = im.ix; ry = im.iy; rw = im.iw; rh = im.ih; }
                else { rx = im.fx >> lv; ry = im.fy >> lv; rw = im.fw >> lv; rh = im.fh >> lv; }
                hit = rx < tile_x + 512 && rx + rw > tile_x && ry < tile_y + 8 && ry + rh > tile_y;
            }
            const unsigned long long m = __ballot(hit);
            if (hit) s_list[__popcll(m & ((1ull << tid) - 1ull))] = k;
            if (tid == 0) s_n = __popcll(m);
        }
        __syncthreads();
        const int cnt = s_n;
        for (int i = 0; i < cnt; i++) {
            const int k = __builtin_amdgcn_readfirstlane(s_list[i]);
            const StxMbImage& im = P.images[k];
            if (!active) continue;
            if (!L0) {
                const int lx0 = X0 - (im.fx >> lv), ly0 = Y0 - (im.fy >> lv);
                const int lw = im.fw >> lv, lh = im.fh >> lv;
                if ((unsigned)lx0 >= (unsigned)lw || (unsigned)ly0 >= (unsigned)lh) continue;
                float w[2][8];
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const float* q = im.wt[lv] + (long long)(ly0 + r) * im.wt_stride[lv] + lx0;
                    float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
                    w[r][0] = a.x; w[r][1] = a.y; w[r][2] = a.z; w[r][3] = a.w;
                    w[r][4] = b.x; w[r][5] = b.y; w[r][6] = b.z; w[r][7] = b.w;
                }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    int up[2][8];
                    up_patch(im.g[lv + 1] + c * im.g_plane[lv + 1], im.g_stride[lv + 1], lw >> 1, lh >> 1, lx0 >> 1,
                             ly0 >> 1, up);
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const short* gp = im.g[lv] + c * im.g_plane[lv] + (long long)(ly0 + r) * im.g_stride[lv] + lx0;
                        uint4 gv = *reinterpret_cast<const uint4*>(gp);
                        const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            int g = (j & 1) ? s16hi(gw[j >> 1]) : s16lo(gw[j >> 1]);
                            int L = sat_s16(g - up[r][j]);
                            acc[r][j][c] += trunc_s16(fmul((float)L, w[r][j]));
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int j = 0; j < 8; j++) ws[r][j] = fadd(ws[r][j], w[r][j]);
            } else {
                const int lx0 = X0 - im.ix, ly0 = Y0 - im.iy;
                if (lx0 + 8 <= 0 || lx0 >= im.iw || ly0 + 2 <= 0 || ly0 >= im.ih) continue;
                int up[3][2][8];
                if (P.num_bands > 0) {
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        up_patch(im.g[1] + c * im.g_plane[1], im.g_stride[1], im.fw >> 1, im.fh >> 1, (X0 - im.fx) >> 1,
                                 (Y0 - im.fy) >> 1, up[c]);
                }
                const bool fastx = lx0 >= 0 && lx0 + 8 <= im.iw;
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int ly = ly0 + r;
                    if ((unsigned)ly >= (unsigned)im.ih) continue;
                    uint32_t pw_[6], mw[2];
                    if (fastx) {
                        const long long off = (long long)ly * im.img0_stride + (long long)lx0 * 3;
                        const uint8_t* q = im.img0 + (off & ~3ll);
                        const uint32_t s = (uint32_t)off & 3u;
                        U4a4 d0 = *reinterpret_cast<const U4a4*>(q);
                        U4a4 d1 = *reinterpret_cast<const U4a4*>(q + 16);
                        pw_[0] = __builtin_amdgcn_alignbyte(d0.v[1], d0.v[0], s);
                        pw_[1] = __builtin_amdgcn_alignbyte(d0.v[2], d0.v[1], s);
                        pw_[2] = __builtin_amdgcn_alignbyte(d0.v[3], d0.v[2], s);
                        pw_[3] = __builtin_amdgcn_alignbyte(d1.v[0], d0.v[3], s);
                        pw_[4] = __builtin_amdgcn_alignbyte(d1.v[1], d1.v[0], s);
                        pw_[5] = __builtin_amdgcn_alignbyte(d1.v[2], d1.v[1], s);
                        const long long moff = (long long)ly * im.mask0_stride + lx0;
                        const uint8_t* mq = im.mask0 + (moff & ~3ll);
                        const uint32_t ms = (uint32_t)moff & 3u;
                        uint32_t m0 = *reinterpret_cast<const uint32_t*>(mq), m1 = *reinterpret_cast<const uint32_t*>(mq + 4),
                                 m2 = *reinterpret_cast<const uint32_t*>(mq + 8);
                        mw[0] = __builtin_amdgcn_alignbyte(m1, m0, ms);
                        mw[1] = __builtin_amdgcn_alignbyte(m2, m1, ms);
                    } else {
#pragma unroll
                        for (int i = 0; i < 6; i++) pw_[i] = 0;
                        mw[0] = mw[1] = 0;
                        const uint8_t* irow = im.img0 + (long long)ly * im.img0_stride;
                        const uint8_t* mrow = im.mask0 + (long long)ly * im.mask0_stride;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const int lx = lx0 + j;
                            if ((unsigned)lx < (unsigned)im.iw) {
                                const uint8_t* p = irow + lx * 3;
                                const uint32_t v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
                                // place the 24-bit pixel at byte 3*j of the stream
                                const int bo = 3 * j;
                                pw_[bo >> 2] |= v << (8 * (bo & 3));
                                if ((bo & 3) > 1) pw_[(bo >> 2) + 1] |= v >> (32 - 8 * (bo & 3));
                                mw[j >> 2] |= (uint32_t)mrow[lx] << (8 * (j & 3));
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int lx = lx0 + j;
                        if (!fastx && (unsigned)lx >= (unsigned)im.iw) continue;
                        const float w = fmul((float)byte_of(mw, j), INV255);
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            int L = (int)byte_of(pw_, 3 * j + c);
                            if (P.num_bands > 0) L = sat_s16(L - up[c][r][j]);
                            acc[r][j][c] += trunc_s16(fmul((float)L, w));
                        }
                        ws[r][j] = fadd(ws[r][j], w);
                    }
                }
            }
        }
    }
    if (!active) return;

    // normalizeUsingWeightMap, then + pyrUp(finished coarser level), saturating
    int v[2][8][3];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float den = fadd(ws[r][j], WEIGHT_EPS);
#pragma unroll
            for (int c = 0; c < 3; c++) v[r][j][c] = trunc_s16(fdiv((float)(short)acc[r][j][c], den));
        }
    if (P.up) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int up[2][8];
            up_patch(P.up + c * P.up_plane, P.up_stride, P.pw >> 1, P.ph >> 1, X0 >> 1, Y0 >> 1, up);
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int j = 0; j < 8; j++) v[r][j][c] = sat_s16(up[r][j] + v[r][j][c]);
        }
    }
    if (!L0) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                uint4 o;
                o.x = pack16(v[r][0][c], v[r][1][c]);
                o.y = pack16(v[r][2][c], v[r][3][c]);
                o.z = pack16(v[r][4][c], v[r][5][c]);
                o.w = pack16(v[r][6][c], v[r][7][c]);
                *reinterpret_cast<uint4*>(P.out + c * P.out_plane + (long long)(Y0 + r) * P.out_stride + X0) = o;
            }
    } else {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int y = Y0 + r;
            if (y >= P.final_h) break;
            uint32_t ob[6] = {0, 0, 0, 0, 0, 0}, om[2] = {0, 0};
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const bool keep = ws[r][j] > WEIGHT_EPS;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (!keep) v[r][j][c] = 0;
                    const uint32_t u = (uint32_t)min(abs(v[r][j][c]), 255);  // convertScaleAbs
                    const int bo = 3 * j + c;
                    ob[bo >> 2] |= u << (8 * (bo & 3));
                }
                om[j >> 2] |= (keep ? 255u : 0u) << (8 * (j & 3));
            }
            uint32_t* po = reinterpret_cast<uint32_t*>(P.pano + (long long)y * P.pano_stride + (long long)X0 * 3);
            *reinterpret_cast<uint2*>(po) = make_uint2(ob[0], ob[1]);
            *reinterpret_cast<uint2*>(po + 2) = make_uint2(ob[2], ob[3]);
            *reinterpret_cast<uint2*>(po + 4) = make_uint2(ob[4], ob[5]);
            *reinterpret_cast<uint2*>(P.pmask + (long long)y * P.pmask_stride + X0) = make_uint2(om[0], om[1]);
            if (P.pano16) {
                uint32_t* p16 = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(P.pano16) + (long long)y * P.pano16_stride +
                                                            (long long)X0 * 6);
                uint32_t s[12];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {  // 2 pixels = 6 shorts = 3 dwords
                    s[(j >> 1) * 3 + 0] = pack16(v[r][j][0], v[r][j][1]);
                    s[(j >> 1) * 3 + 1] = pack16(v[r][j][2], v[r][j + 1][0]);
                    s[(j >> 1) * 3 + 2] = pack16(v[r][j + 1][1], v[r][j + 1][2]);
                }
                *reinterpret_cast<uint4*>(p16) = make_uint4(s[0], s[1], s[2], s[3]);
                *reinterpret_cast<uint4*>(p16 + 4) = make_uint4(s[4], s[5], s[6], s[7]);
                *reinterpret_cast<uint4*>(p16 + 8) = make_uint4(s[8], s[9], s[10], s[11]);
            }
        }
    }
}

bool launched_ok() { return hipGetLastError() == hipSuccess; }

}  // namespace

bool stx_fast_mb_down0(stx_ctx* ctx, const StxMbImage& im)
{
    if (im.img0_is_s16 || im.fw < 8 || im.fh < 2) return false;
    constexpr int R = 8;
    const int ow = im.fw >> 1, oh = im.fh >> 1;
    dim3 grid(((ow + 1) / 2 + 63) / 64, (oh + 4 * R - 1) / (4 * R));
    hipLaunchKernelGGL((mb_down_fast_kernel<true, R>), grid, dim3(256), 0, ctx->stream, im, 0);
    return launched_ok();
}

bool stx_fast_mb_down(stx_ctx* ctx, const StxMbImage& im, int level)
{
    const int iw = im.fw >> level, ih = im.fh >> level;
    if (iw < 8 || ih < 2) return false;
    constexpr int R = 4;
    const int ow = iw >> 1, oh = ih >> 1;
    dim3 grid(((ow + 1) / 2 + 63) / 64, (oh + 4 * R - 1) / (4 * R));
    hipLaunchKernelGGL((mb_down_fast_kernel<false, R>), grid, dim3(256), 0, ctx->stream, im, level);
    return launched_ok();
}

bool stx_fast_mb_level(stx_ctx* ctx, const MbLevelK& K)
{
    // 8-pixel strips must never straddle a feed-rectangle edge: 2^(B - level) >= 8
    if (K.num_bands - K.level < 3) return false;
    if (K.level == 0) {
        dim3 grid((K.final_w + 511) / 512, (K.final_h + 7) / 8);
        hipLaunchKernelGGL(mb_level_fast_kernel<true>, grid, dim3(256), 0, ctx->stream, K);
    } else {
        dim3 grid((K.pw + 511) / 512, (K.ph + 7) / 8);
        hipLaunchKernelGGL(mb_level_fast_kernel<false>, grid, dim3(256), 0, ctx->stream, K);
    }
    return launched_ok();
}
