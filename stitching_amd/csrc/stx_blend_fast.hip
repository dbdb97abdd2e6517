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
#include <algorithm>
#include <cstdlib>

#include "stx_blend_kernels.h"
#include "stx_device_math.h"

using namespace stxd;

namespace {

// waves per SIMD the level-0 gather is compiled for: 5 = 95 VGPRs without spills (A/B on one box: 218 us at 4, 206 us at 5,
// 283 us at 6: 80 VGPRs spill in the collapse)
#ifndef STX_L0_WAVES
#define STX_L0_WAVES 5
#endif
// STX_L0_FULL = 1 (round 6, shipped): the level-0 gather issues EVERY load of an image in one batch — both pixel rows (a row outside the
// image from the clamped row, its mask cleared), the pyrUp windows of the three planes — with the batched image search and the epilogue's
// windows ahead in every instantiation: 92 registers = 5 wavefronts per SIMD instead of 80 = 6, and still 159.2 -> 147.9 us on config 2
// (four interleaved runs, tools/gpu_r6w.sh), 176.7 -> 167.0 on the reference-default leg.  Each of the three alone, at 88-92 registers,
// had measured slower than the plane-by-plane form at 80 (profiles/r06_round_trips.md): it is the whole chain that pays for the wavefront.
// normalisation of the level-0 gather on integer counts (binary masks): bit 0 the packed shift form for counts <= 2, bit 1 the
// one-multiplication form for counts < 16 (level0_epilogue_pk); 0 builds the shared-reciprocal IEEE division for every count above 1
#ifndef STX_L0_NORM
#define STX_L0_NORM 3
#endif
#ifndef STX_ABLATE_MASK
#define STX_ABLATE_MASK 0
#endif
#ifndef STX_L0_FULL
#define STX_L0_FULL 1
#endif
constexpr float WEIGHT_EPS = 1e-5f;
constexpr float INV255 = 0.0039215688593685627f;  // (float)(1./255.)
constexpr float INV256 = 0.00390625f;

// Pointers read from a descriptor in memory are generic ("flat") to the compiler; all of ours are device
// global memory.  Saying so gives global_load with an SGPR base + 32-bit lane offset instead of flat_load
// with 64-bit lane address arithmetic (and keeps lgkmcnt free for scalar / LDS traffic).
#define STX_GAS __attribute__((address_space(1)))
template <class T> STX_DEV const STX_GAS T* gp(const T* p) { return (const STX_GAS T*)p; }
template <class T> STX_DEV STX_GAS T* gp(T* p) { return (STX_GAS T*)p; }

// tile (tx, ty) of workgroup b under the XCD-aware order; false when b is padding
STX_DEV bool xcd_tile(const StxTileMap& M, uint32_t b, int& tx, int& ty)
{
    if (M.plain) {
        const uint32_t wy = M.tiles_x == 1 ? b : __umulhi(b, M.magic_tx);
        tx = (int)(b - wy * (uint32_t)M.tiles_x);
        ty = (int)wy;
        return true;
    }
    const uint32_t local = b >> 3;
    const uint32_t band_i = M.band_tiles == 1 ? local : __umulhi(local, M.magic_band);
    const uint32_t within = local - band_i * (uint32_t)M.band_tiles;
    const uint32_t wy = M.tiles_x == 1 ? within : __umulhi(within, M.magic_tx);
    tx = (int)(within - wy * (uint32_t)M.tiles_x);
    ty = (int)((band_i * 8u + (b & 7u)) * (uint32_t)M.band_rows + wy);
    return ty < M.tiles_y;
}

typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((aligned(4))) v4u_a4;  // 16 bytes, only dword-aligned
typedef v2u __attribute__((aligned(4))) v2u_a4;

struct __attribute__((aligned(4))) U4a4 { uint32_t v[4]; };  // 16 bytes, only dword-aligned
struct __attribute__((aligned(4))) U2a4 { uint32_t v[2]; };

// packed 16-bit helpers (two pixels per VGPR) shared by the level kernels; see mb_level0_pk_kernel below
typedef unsigned short pk16 __attribute__((ext_vector_type(2)));
STX_DEV pk16 pk(uint32_t v) { return __builtin_bit_cast(pk16, v); }
STX_DEV uint32_t unpk(pk16 v) { return __builtin_bit_cast(uint32_t, v); }
STX_DEV pk16 pk_splat(unsigned short v) { pk16 r = {v, v}; return r; }

// The six coarse samples c0..c5 = plane[cx - 1 .. cx + 4] of one row, as the pairs the 8-wide pyrUp patch needs, from ONE
// dword-aligned 16-byte load of plane[cx - 2 .. cx + 5] (cx is a multiple of 4, so the window starts 4 bytes before an
// 8-byte boundary).  pyrUp's border rule — column -1 -> 1, column cw -> cw - 1 — only changes c0 (to c2, when cx == 0)
// and c5 (to c4, when cx + 4 == cw).  Both are byte permutations of the window, so the rule lives in the SELECTOR of the
// v_perm that builds the pair anyway (UpSel: two registers made once per image and lane) — no select, no second candidate:
// 3 VALU per window (round 2: 7).  The two shorts in front of a plane's first row are read and never used: every plane this
// is called on has >= 4 readable bytes in front.
struct UpRow { uint32_t A0, B0, A1, B1, A2; };  // (c0,c1) (c1,c2) (c2,c3) (c3,c4) (c4,c5)
struct UpSel { uint32_t a0, a2; };
// window dwords (x,c0) (c1,c2) (c3,c4) (c5,x); v_perm(hi, lo, sel): selector bytes 0-3 pick from `lo`, 4-7 from `hi`
STX_DEV UpSel up_sel(bool left_edge, bool right_edge)
{
    UpSel s;
    s.a0 = left_edge ? 0x05040706u : 0x05040302u;   // perm(w.y, w.x): (c2,c1) at the left edge, else (c0,c1)
    s.a2 = right_edge ? 0x03020302u : 0x05040302u;  // perm(w.w, w.z): (c4,c4) at the right edge, else (c4,c5)
    return s;
}
// row: first sample of the plane's row (wave-uniform); boff: byte offset of sample cx in the row (2 cx)
STX_DEV UpRow up_row_window(const STX_GAS short* __restrict__ row, uint32_t boff, UpSel sel)
{
    const STX_GAS char* q = reinterpret_cast<const STX_GAS char*>(row) + (size_t)boff;
    const v4u w = *reinterpret_cast<const STX_GAS v4u_a4*>(q - 4);  // (x,c0) (c1,c2) (c3,c4) (c5,x)
    UpRow r;
    r.B0 = w.y;
    r.B1 = w.z;
    r.A1 = __builtin_amdgcn_alignbit(w.z, w.y, 16);
    r.A0 = __builtin_amdgcn_perm(w.y, w.x, sel.a0);
    r.A2 = __builtin_amdgcn_perm(w.w, w.z, sel.a2);
    return r;
}

// pyrUp_'s row rule (up_idx: -1 -> 1, or 0 when there is one row; n -> n - 1) for i >= -1, as scalar-unit arithmetic
STX_DEV int up_idx_s(int i, int n) { return min(abs(i), n - 1); }
// first sample of row `r` of a plane, held in SGPRs: the loads below then take the scalar base + 32-bit lane offset form
// (the empty asm keeps the compiler from folding the lane offset into a 64-bit vector address first; it is not volatile: a
// side-effecting asm counts as a memory clobber and turns every scalar descriptor load after it into a vector load).
// readfirstlane of a value the compiler already knows to be uniform folds away; where it does not know — argument blocks read
// through a pointer — it is what makes the value scalar.
STX_DEV const STX_GAS short* row_ptr_s(const STX_GAS short* plane, int r, uint32_t stride)
{
    const unsigned long long a = (unsigned long long)(plane + (size_t)((uint32_t)r * stride));
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const STX_GAS short* row = (const STX_GAS short*)(((unsigned long long)hi << 32) | lo);
    asm("" : "+s"(row));
    return row;
}

// The same six samples from a plane of BYTES (the Gaussian levels of u8 images, StxMbImage::g_u8): one dword-aligned 12-byte load
// of plane[cx - 4 .. cx + 7] — (., ., ., c0) (c1, c2, c3, c4) (c5, ., ., .) — and five v_perm that widen the bytes to the 16-bit
// pairs; the border rule again lives in two selectors.  Half the bytes of the int16 window (and two registers less per load),
// two VALU more.  boff: byte offset of sample cx in the row (= cx).
typedef uint32_t v3u __attribute__((ext_vector_type(3)));
typedef v3u __attribute__((aligned(4))) v3u_a4;
STX_DEV UpSel up_sel_u8(bool left_edge, bool right_edge)
{
    UpSel s;
    s.a0 = left_edge ? 0x0c040c05u : 0x0c040c03u;   // perm(d1, d0): (c2, c1) at the left edge, else (c0, c1)
    s.a2 = right_edge ? 0x0c030c03u : 0x0c040c03u;  // perm(d2, d1): (c4, c4) at the right edge, else (c4, c5)
    return s;
}
STX_DEV UpRow up_row_window_u8(const STX_GAS uint8_t* __restrict__ row, uint32_t boff, UpSel sel)
{
    const STX_GAS uint8_t* q = row + (size_t)boff;
    const v3u w = *reinterpret_cast<const STX_GAS v3u_a4*>(q - 4);
    UpRow r;
    r.A0 = __builtin_amdgcn_perm(w.y, w.x, sel.a0);
    r.B0 = __builtin_amdgcn_perm(w.y, w.y, 0x0c010c00u);
    r.A1 = __builtin_amdgcn_perm(w.y, w.y, 0x0c020c01u);
    r.B1 = __builtin_amdgcn_perm(w.y, w.y, 0x0c030c02u);
    r.A2 = __builtin_amdgcn_perm(w.z, w.y, sel.a2);
    return r;
}
STX_DEV const STX_GAS uint8_t* row_ptr_s(const STX_GAS uint8_t* plane, int r, uint32_t stride)
{
    const unsigned long long a = (unsigned long long)(plane + (size_t)((uint32_t)r * stride));
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const STX_GAS uint8_t* row = (const STX_GAS uint8_t*)(((unsigned long long)hi << 32) | lo);
    asm("" : "+s"(row));
    return row;
}

// pyrUp_<FixPtCast<short,6>> of one byte plane (values 0..255) for the 8x2 patch with coarse origin (cx, cy).
// cy (and with it the three row pointers) is wave-uniform: a wavefront owns two panorama rows.  boff = cx, sel = up_sel_u8(...).
// In two halves (round 6): the three 12-byte window loads, and the arithmetic on what they returned — a caller with several planes
// in front of it issues all their loads before the first use (one memory round trip per image instead of one per plane).
STX_DEV void up_patch_pk_load(const STX_GAS uint8_t* __restrict__ plane, uint32_t stride, int ch, uint32_t boff, int cy, v3u (&w)[3])
{
    const int rr[3] = {up_idx_s(cy - 1, ch), cy, up_idx_s(cy + 1, ch)};
    asm("" : "+v"(boff));  // the zero-extension of the lane offset stays in this block: scalar base + 32-bit lane offset loads
#pragma unroll
    for (int r = 0; r < 3; r++) w[r] = *reinterpret_cast<const STX_GAS v3u_a4*>(row_ptr_s(plane, rr[r], stride) + (size_t)boff - 4);
}
STX_DEV UpRow up_row_of_u8(v3u w, UpSel sel)
{
    UpRow r;
    r.A0 = __builtin_amdgcn_perm(w.y, w.x, sel.a0);
    r.B0 = __builtin_amdgcn_perm(w.y, w.y, 0x0c010c00u);
    r.A1 = __builtin_amdgcn_perm(w.y, w.y, 0x0c020c01u);
    r.B1 = __builtin_amdgcn_perm(w.y, w.y, 0x0c030c02u);
    r.A2 = __builtin_amdgcn_perm(w.z, w.y, sel.a2);
    return r;
}
STX_DEV void up_patch_pk_math(const v3u (&w)[3], UpSel sel, pk16 up[2][4])
{
    pk16 HE[3][2], HO[3][2];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const UpRow t = up_row_of_u8(w[r], sel);
        HE[r][0] = pk(t.A0) + pk(t.B0) * pk_splat(6) + pk(t.A1);            // c[j] + 6 c[j+1] + c[j+2], j = 0,1
        HE[r][1] = pk(t.A1) + pk(t.B1) * pk_splat(6) + pk(t.A2);            // j = 2,3
        HO[r][0] = pk(t.B0) + pk(t.A1);                                     // c[j+1] + c[j+2] (the factor 4 is folded below)
        HO[r][1] = pk(t.B1) + pk(t.A2);
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
        up[0][k] = (HE[0][k] + HE[1][k] * pk_splat(6) + HE[2][k] + pk_splat(32)) >> pk_splat(6);
        up[0][2 + k] = (HO[0][k] + HO[1][k] * pk_splat(6) + HO[2][k] + pk_splat(8)) >> pk_splat(4);
        up[1][k] = (HE[1][k] + HE[2][k] + pk_splat(8)) >> pk_splat(4);
        up[1][2 + k] = (HO[1][k] + HO[2][k] + pk_splat(2)) >> pk_splat(2);
    }
}
STX_DEV void up_patch_pk(const STX_GAS uint8_t* __restrict__ plane, uint32_t stride, int ch, uint32_t boff, int cy, UpSel sel, pk16 up[2][4])
{
    v3u w[3];
    up_patch_pk_load(plane, stride, ch, boff, cy, w);
    up_patch_pk_math(w, sel, up);
}
// byte plane c of level lv of a u8 pyramid
STX_DEV const STX_GAS uint8_t* g8(const StxMbImage& im, int lv, int c)
{
    return gp(reinterpret_cast<const uint8_t*>(im.g[lv])) + c * im.g_plane[lv];
}
// the 8 samples at (row r, column x0) of a byte plane as the pyrUp pair order (0,2)(4,6)(1,3)(5,7), zero-extended to 16 bits
STX_DEV v2u g8_row_load(const STX_GAS uint8_t* plane, int r, uint32_t stride, uint32_t x0)
{
    return *reinterpret_cast<const STX_GAS v2u*>(row_ptr_s(plane, r, stride) + (size_t)x0);
}
STX_DEV void g8_pairs_of(v2u gv, uint32_t (&q)[4])
{
    q[0] = __builtin_amdgcn_perm(gv.x, gv.x, 0x0c020c00u);
    q[1] = __builtin_amdgcn_perm(gv.y, gv.y, 0x0c020c00u);
    q[2] = __builtin_amdgcn_perm(gv.x, gv.x, 0x0c030c01u);
    q[3] = __builtin_amdgcn_perm(gv.y, gv.y, 0x0c030c01u);
}
STX_DEV void g8_row_pairs(const STX_GAS uint8_t* plane, int r, uint32_t stride, uint32_t x0, uint32_t (&q)[4])
{
    g8_pairs_of(g8_row_load(plane, r, stride, x0), q);
}


// bytes o1, o2 of the little-endian byte stream held in w[] -> (w[o1], 0, w[o2], 0)
template <int O1, int O2>
STX_DEV uint32_t pair_u8(const uint32_t* w)
{
    constexpr uint32_t sel = (uint32_t)(O1 & 3) | (0x0cu << 8) | ((uint32_t)(4 + (O2 & 3)) << 16) | (0x0cu << 24);
    return __builtin_amdgcn_perm(w[O2 >> 2], w[O1 >> 2], sel);
}
// mask bytes a, b (0 or 255) of m[] -> (0xffff or 0, 0xffff or 0)
template <int A, int B>
STX_DEV uint32_t pair_mask(const uint32_t* m)
{
    constexpr uint32_t sel = (uint32_t)(A & 3) * 0x0101u | (uint32_t)(4 + (B & 3)) * 0x01010000u;
    return __builtin_amdgcn_perm(m[B >> 2], m[A >> 2], sel);
}




// bytes 0xff where pixel j (0..7) of a lane's strip starting at image column lx0 lies inside the image: all ones for the wavefronts
// that hold no partial lane (one ballot), else bit j -> byte j by multiplication
STX_DEV void lane_valid_bytes(int lx0, int iw, uint32_t& vm0, uint32_t& vm1)
{
    vm0 = vm1 = 0xffffffffu;
    if (__builtin_amdgcn_ballot_w64(lx0 < 0 || lx0 + 8 > iw) != 0ull) {
        const int lo = max(-lx0, 0), hi = min(iw - lx0, 8);                 // pixels lo .. hi - 1 are inside
        const uint32_t bits = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        vm0 = (((bits & 15u) * 0x00204081u) & 0x01010101u) * 0xffu;
        vm1 = (((bits >> 4) * 0x00204081u) & 0x01010101u) * 0xffu;
    }
}
// The 8 BGR pixels (24 bytes -> pw[6]) and 8 mask bytes (-> mw[2], ANDed with the validity bytes) at (lx0, ly) of a u8 image:
// two dword-aligned 16-byte loads and three dword loads whatever lx0 is (-7 .. iw - 1).  A lane whose pixels lie partly left / right
// of the image reads memory of the allocation there (STX_BUF_FRONT_PAD in front of row 0, the row pitch / the next row elsewhere)
// and never uses it: those mask bytes are cleared and every use of a pixel is multiplied by its mask.  Offsets are taken from 64
// bytes in front of the image so that they stay non-negative.  (Round 2 fetched the inside pixels of such a lane byte by byte
// behind per-pixel branches: 16 dependent round trips for every wavefront that holds one — two lanes per image and row pair, a
// quarter of all wavefront visits.)
STX_DEV void load_px8_u8(const uint8_t* img0, uint32_t img0_stride, const uint8_t* mask0, uint32_t mask0_stride, int lx0, int ly, uint32_t vm0,
                         uint32_t vm1, uint32_t (&pw)[6], uint32_t (&mw)[2])
{
    const uint32_t off = (uint32_t)ly * img0_stride + (uint32_t)(lx0 * 3 + 64);
    const STX_GAS uint8_t* q = gp(img0) - 64 + (off & ~3u);
    const uint32_t s = off & 3u;
    const v4u d0 = *reinterpret_cast<const STX_GAS v4u_a4*>(q);
    const v4u d1 = *reinterpret_cast<const STX_GAS v4u_a4*>(q + 16);
    pw[0] = __builtin_amdgcn_alignbyte(d0.y, d0.x, s);
    pw[1] = __builtin_amdgcn_alignbyte(d0.z, d0.y, s);
    pw[2] = __builtin_amdgcn_alignbyte(d0.w, d0.z, s);
    pw[3] = __builtin_amdgcn_alignbyte(d1.x, d0.w, s);
    pw[4] = __builtin_amdgcn_alignbyte(d1.y, d1.x, s);
    pw[5] = __builtin_amdgcn_alignbyte(d1.z, d1.y, s);
#if STX_ABLATE_MASK
    mw[0] = vm0; mw[1] = vm1;  // timing experiment only: every mask byte taken as 255
    return;
#endif
    const uint32_t moff = (uint32_t)ly * mask0_stride + (uint32_t)(lx0 + 64);
    const STX_GAS uint8_t* mq = gp(mask0) - 64 + (moff & ~3u);
    const uint32_t ms = moff & 3u;
    const uint32_t m0 = *reinterpret_cast<const STX_GAS uint32_t*>(mq), m1 = *reinterpret_cast<const STX_GAS uint32_t*>(mq + 4),
                   m2 = *reinterpret_cast<const STX_GAS uint32_t*>(mq + 8);
    mw[0] = __builtin_amdgcn_alignbyte(m1, m0, ms) & vm0;
    mw[1] = __builtin_amdgcn_alignbyte(m2, m1, ms) & vm1;
}
STX_DEV void load_px8_u8(const StxMbImage& im, int lx0, int ly, uint32_t vm0, uint32_t vm1, uint32_t (&pw)[6], uint32_t (&mw)[2])
{
    load_px8_u8(im.img0, (uint32_t)im.img0_stride, im.mask0, (uint32_t)im.mask0_stride, lx0, ly, vm0, vm1, pw, mw);
}

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
// pyrDown through LDS.  Workgroup = 64 x 14 outputs.
//   phase 1: the 31 input rows the tile needs are streamed from HBM with 16-byte loads; each lane
//            filters horizontally (1-4-6-4-1, stride 2) as it loads and parks the row sums in LDS
//            (the 5-tap stencil is staged in LDS, never re-read from memory);
//   phase 2: vertical 1-4-6-4-1 from LDS, (v + 128) >> 8, packed stores.
// copyMakeBorder (REFLECT image / CONSTANT weight) and pyrDown's REFLECT_101 are index maps that
// only the lanes at a border evaluate (per-wavefront slow path); interior lanes take vector loads.
// ---------------------------------------------------------------------------------------------
// 64 x 14 outputs: the 31 input rows give 31*8 = 248 (levels >= 1) / 31*16 = 496 (level 0) load tasks, i.e. one /
// two full passes of the 256 threads (16 rows would leave a third, 9 %-full pass)
constexpr int DN_TOW = 64, DN_TOH = 14, DN_ROWS = 2 * DN_TOH + 3;
// tile rows (of 14 output rows) per XCD band of the level-0 pyrDown.  Round 4, interleaved: 1: 127.8 / 127.0 us, 2: 131.9 / 132.1, 4: 144.2 / 142.2;
// the plain row-major order 129.2 / 126.8 at 1.6 x the fetched bytes
#ifndef STX_DN_BAND
#define STX_DN_BAND 1
#endif
constexpr int DN_BAND = STX_DN_BAND;
// The gather kernels of the levels: a wavefront owns 512 x 2 samples and shares nothing with its siblings (no LDS, no barrier), so a
// workgroup is LV_WAVES independent wavefronts and the tile 512 x LV_TH.  One wavefront per workgroup: a wavefront that finishes frees
// its slot at once instead of waiting for three siblings (level 0 of config 2: 176.5 / 176.4 us against 180.7 / 181.0 with four, 178.9 /
// 178.3 with two; interleaved A/B on one box, round 4).
#ifndef STX_LV_WG_WAVES
#define STX_LV_WG_WAVES 1
#endif
constexpr int LV_WAVES = STX_LV_WG_WAVES, LV_TH = 2 * LV_WAVES, LV_THREADS = 64 * LV_WAVES;
// sample rows per XCD band of the gather kernels.  Measured with one-wavefront workgroups (round 4, interleaved, level 0 of config 2):
// 64: 181.1 / 180.7 us, 32: 175.9 / 175.9, 16: 172.4 / 174.5, 8: 170.4 / 173.1, 4: 171.0 / 173.1
#ifndef STX_LV_BAND_ROWS
#define STX_LV_BAND_ROWS 8
#endif
constexpr int LV_BAND = STX_LV_BAND_ROWS / LV_TH;  // tile rows per XCD band

// 36 bytes of an image row from byte offset `off` on (any alignment) as a little-endian byte stream w[0..8]
STX_DEV void dn_load36(const STX_GAS uint8_t* img, uint32_t off, uint32_t (&w)[9])
{
    const STX_GAS uint8_t* q = img + (off & ~3u);
    const uint32_t s = off & 3u;
    const v4u d0 = *reinterpret_cast<const STX_GAS v4u_a4*>(q);
    const v4u d1 = *reinterpret_cast<const STX_GAS v4u_a4*>(q + 16);
    const v2u d2 = *reinterpret_cast<const STX_GAS v2u_a4*>(q + 32);
    w[0] = __builtin_amdgcn_alignbyte(d0.y, d0.x, s);
    w[1] = __builtin_amdgcn_alignbyte(d0.z, d0.y, s);
    w[2] = __builtin_amdgcn_alignbyte(d0.w, d0.z, s);
    w[3] = __builtin_amdgcn_alignbyte(d1.x, d0.w, s);
    w[4] = __builtin_amdgcn_alignbyte(d1.y, d1.x, s);
    w[5] = __builtin_amdgcn_alignbyte(d1.z, d1.y, s);
    w[6] = __builtin_amdgcn_alignbyte(d1.w, d1.z, s);
    w[7] = __builtin_amdgcn_alignbyte(d2.x, d1.w, s);
    w[8] = __builtin_amdgcn_alignbyte(d2.y, d2.x, s);
}
// Packed horizontal 1-4-6-4-1 sums of channel C of the 11 BGR pixels in w[] (a u8 image): outputs o = 0..3 use pixels 2o .. 2o + 4; as
// pairs (out0, out1) = (p0, p2) + 4 (p1, p3) + 6 (p2, p4) + 4 (p3, p5) + (p4, p6), (out2, out3) likewise from p4 .. p10.
// MIRROR: the run was loaded in the opposite order (pixel k of the task is pixel 10 - k of w[]): the same sums from mirrored byte picks.
template <bool MIRROR, int C>
STX_DEV void dn_pack5_channel(const uint32_t* w, short* hs)
{
#define STX_PXB(k) (MIRROR ? 30 - 3 * (k) + C : 3 * (k) + C)
    const pk16 A = pk(pair_u8<STX_PXB(0), STX_PXB(2)>(w)), Bp = pk(pair_u8<STX_PXB(1), STX_PXB(3)>(w));
    const pk16 Cp = pk(pair_u8<STX_PXB(2), STX_PXB(4)>(w)), D = pk(pair_u8<STX_PXB(3), STX_PXB(5)>(w));
    const pk16 E = pk(pair_u8<STX_PXB(4), STX_PXB(6)>(w)), F = pk(pair_u8<STX_PXB(5), STX_PXB(7)>(w));
    const pk16 G = pk(pair_u8<STX_PXB(6), STX_PXB(8)>(w)), H = pk(pair_u8<STX_PXB(7), STX_PXB(9)>(w));
    const pk16 I = pk(pair_u8<STX_PXB(8), STX_PXB(10)>(w));
#undef STX_PXB
    const pk16 o01 = A + E + Cp * pk_splat(6) + (Bp + D) * pk_splat(4);
    const pk16 o23 = E + I + G * pk_splat(6) + (F + H) * pk_splat(4);
    *reinterpret_cast<uint2*>(hs) = make_uint2(unpk(o01), unpk(o23));
}

// The four 1-4-6-4-1 row sums (stride 2) of 11 mask bits held as bytes 0 / 1 in mb[0..2] (byte 11 may hold anything); sums <= 16: exact
// as floats.  STX_DN0_DOT4 = 1 (round 6, visit ab): bytes are what v_dot4_u32_u8 multiplies — 8 dot products against constant weight
// dwords where the packed 16-bit form takes 9 v_perm + 10 v_pk_*; measured SLOWER (mb_down0 108.1 / 109.7 / 110.1 -> 110.5 / 111.3 /
// 113.7 us, three interleaved runs of config 2): the dot product does not issue at the rate of the packed instructions.  Default 0.
#ifndef STX_DN0_DOT4
#define STX_DN0_DOT4 0
#endif
STX_DEV float4 dn_mask_sums4(const uint32_t (&mb)[3])
{
#if STX_DN0_DOT4
    const uint32_t o0 = __builtin_amdgcn_udot4(mb[0], 0x04060401u, mb[1] & 1u, false);                                      // bytes 0 .. 4
    const uint32_t o1 = __builtin_amdgcn_udot4(mb[1], 0x00010406u, __builtin_amdgcn_udot4(mb[0], 0x04010000u, 0u, false), false);  // 2 .. 6
    const uint32_t o2 = __builtin_amdgcn_udot4(mb[1], 0x04060401u, mb[2] & 1u, false);                                      // 4 .. 8
    const uint32_t o3 = __builtin_amdgcn_udot4(mb[2], 0x00010406u, __builtin_amdgcn_udot4(mb[1], 0x04010000u, 0u, false), false);  // 6 .. 10
    return make_float4((float)o0, (float)o1, (float)o2, (float)o3);
#else
    const pk16 o01 = pk(pair_u8<0, 2>(mb)) + pk(pair_u8<4, 6>(mb)) + pk(pair_u8<2, 4>(mb)) * pk_splat(6) +
                     (pk(pair_u8<1, 3>(mb)) + pk(pair_u8<3, 5>(mb))) * pk_splat(4);
    const pk16 o23 = pk(pair_u8<4, 6>(mb)) + pk(pair_u8<8, 10>(mb)) + pk(pair_u8<6, 8>(mb)) * pk_splat(6) +
                     (pk(pair_u8<5, 7>(mb)) + pk(pair_u8<7, 9>(mb))) * pk_splat(4);
    return make_float4((float)(unpk(o01) & 0xffffu), (float)(unpk(o01) >> 16), (float)(unpk(o23) & 0xffffu), (float)(unpk(o23) >> 16));
#endif
}

// level 0 (u8 BGR + u8 mask): 4 outputs from 11 input pixels
// PK: every image of the launch is u8 with a 0 / 255 mask (decided on the host: no per-task branch in the kernel)
// NEAR: the border of the image inside its feed rectangle is narrower than the image (left, right <= iw, top, bottom <= ih: every
// position is at most one mirror image away: branch-free index maps, no division); !NEAR (an exchange strip a few columns wide inside
// a 96-column border): cv::borderInterpolate's general form
// What the level-0 pyrDown reads of an image's descriptor, fetched ONCE per workgroup as one batch of scalar loads and pinned in scalar
// registers (round 6): read through the descriptor inside the task loop, the fields came back as dependent scalar-cache round trips
// in every iteration (the compiler sinks a kernel-argument load to its first use: three s_load + wait pairs per task).
struct DnImage0 {
    int fw, fh, iw, ih, left, top;
    uint32_t img0_stride, mask0_stride;
    const uint8_t* img0;
    const uint8_t* mask0;
};
template <bool PK, bool NEAR, class IM>
STX_DEV void dn_task_level0(const IM& im, int row, int xo, short* hs0, short* hs1, short* hs2, float* hw)
{
    const int by = reflect101_near(row, im.fh) - im.top;  // bordered row -> image row
    const bool yin = (unsigned)by < (unsigned)im.ih;
    const int sy = NEAR ? reflect_near(by, im.ih) : reflect(by, im.ih);
    // global address space, 32-bit offsets from the image's (uniform) base: scalar base + lane offset loads
    const STX_GAS uint8_t* img = gp(im.img0);
    const uint32_t rowoff = (uint32_t)sy * (uint32_t)im.img0_stride;
    const int c0 = 2 * xo - 2;     // first bordered column of the 11 this task reads
    const int a0 = c0 - im.left;   // ... as an image column
    int px[11][3];
    float f[11];
    if (c0 >= 0 && c0 + 10 < im.fw && a0 >= 0 && a0 + 10 < im.iw) {
        uint32_t w[9];
        dn_load36(img, rowoff + (uint32_t)a0 * 3u, w);  // 33 bytes
        dn_pack5_channel<false, 0>(w, hs0);  // packed image sums: the image is u8 in this kernel whatever its mask is
        dn_pack5_channel<false, 1>(w, hs1);
        dn_pack5_channel<false, 2>(w, hs2);
        // the mask's 11 bytes.  PK (host: every mask of the launch is 0 / 255): packed counts.  Otherwise (round 6) the SAME packed counts
        // whenever every lane of the wavefront reads nothing but 0 and 255 — a resized seam mask (SeamFinder.resize) is grey only along
        // its seams, a strip a dozen pixels wide — and the fp32 form (m / 255, row sums in pyrDown's order) for the wavefronts on a seam:
        // for 0 / 255 bytes the two agree exactly (W_0 is 0.f or 1.f, the sums are small integers whatever the order).
        uint32_t mw[3] = {0u, 0u, 0u};
        if (yin) {
            const uint32_t moff = (uint32_t)by * (uint32_t)im.mask0_stride + (uint32_t)a0;  // 11 bytes
            const STX_GAS uint8_t* mq = gp(im.mask0) + (moff & ~3u);
            const uint32_t ms = moff & 3u;
            const v4u m = *reinterpret_cast<const STX_GAS v4u_a4*>(mq);
            mw[0] = __builtin_amdgcn_alignbyte(m.y, m.x, ms);
            mw[1] = __builtin_amdgcn_alignbyte(m.z, m.y, ms);
            mw[2] = __builtin_amdgcn_alignbyte(m.w, m.z, ms) & 0x00ffffffu;
        }
        bool packed = PK;
        if (!PK) {
            // byte b is 0 or 255  <=>  b == 255 * (b >> 7)
            const uint32_t odd = (mw[0] ^ (((mw[0] >> 7) & 0x01010101u) * 255u)) | (mw[1] ^ (((mw[1] >> 7) & 0x01010101u) * 255u)) |
                                 (mw[2] ^ (((mw[2] >> 7) & 0x01010101u) * 255u));
            packed = __builtin_amdgcn_ballot_w64(odd != 0u) == 0ull;
        }
        if (packed) {
            uint32_t mb[3];  // 0 / 255 -> 0 / 1: W_0 is exactly 0.f or 1.f, the fp32 row sums are small integers
            mb[0] = mw[0] & 0x01010101u;
            mb[1] = mw[1] & 0x01010101u;
            mb[2] = mw[2] & 0x01010101u;
            *reinterpret_cast<float4*>(hw) = dn_mask_sums4(mb);
            return;
        }
#pragma unroll
        for (int j = 0; j < 11; j++) f[j] = fmul((float)byte_of(mw, j), INV255);  // (a row outside the image: mw = 0)
#pragma unroll
        for (int o = 0; o < 4; o++) hw[o] = h5f(f[2 * o], f[2 * o + 1], f[2 * o + 2], f[2 * o + 3], f[2 * o + 4]);
        return;
    }
    // Round 6.  An 11-pixel run that lies WHOLLY in the left or right border of the feed rectangle (copyMakeBorder's frame: 3 * 2^B
    // columns either side, a fifth to a quarter of all runs at 7 bands or for seam-cell crops): BORDER_REFLECT maps it onto 11 ADJACENT
    // pixels inside the image, in the opposite order — the same four vector loads and packed sums as an interior run with the byte
    // picks mirrored, instead of 44 byte loads; the weight's border is CONSTANT 0.
    if (c0 >= 0 && c0 + 10 < im.fw && (a0 + 10 < 0 || a0 >= im.iw)) {
        const int s_low = a0 + 10 < 0 ? -(a0 + 10) - 1 : 2 * im.iw - 1 - (a0 + 10);  // source column of the run's LAST pixel = the lowest
        if (s_low >= 0 && s_low + 10 < im.iw) {  // one mirror image away, all of it
            uint32_t w[9];
            dn_load36(img, rowoff + (uint32_t)s_low * 3u, w);
            dn_pack5_channel<true, 0>(w, hs0);
            dn_pack5_channel<true, 1>(w, hs1);
            dn_pack5_channel<true, 2>(w, hs2);
            *reinterpret_cast<float4*>(hw) = make_float4(0.f, 0.f, 0.f, 0.f);
            return;
        }
    }
    {
        // an 11-pixel run that meets a border: every load unconditional and from a position inside the image (so that all 44 of them
        // are in flight together), the CONSTANT-0 border of the weight as a select afterwards
        const STX_GAS uint8_t* mrow = gp(im.mask0) + (uint32_t)min(max(by, 0), im.ih - 1) * (uint32_t)im.mask0_stride;
#pragma unroll
        for (int j = 0; j < 11; j++) {
            const int bx = reflect101_near(c0 + j, im.fw) - im.left;
            const int sx = NEAR ? reflect_near(bx, im.iw) : reflect(bx, im.iw);
            const STX_GAS uint8_t* p = img + (rowoff + (uint32_t)sx * 3u);
            px[j][0] = p[0]; px[j][1] = p[1]; px[j][2] = p[2];
            const float mv = fmul((float)mrow[sx], INV255);
            f[j] = (yin && (unsigned)bx < (unsigned)im.iw) ? mv : 0.f;
        }
    }
#pragma unroll
    for (int o = 0; o < 4; o++) {
        hs0[o] = (short)h5i(px[2 * o][0], px[2 * o + 1][0], px[2 * o + 2][0], px[2 * o + 3][0], px[2 * o + 4][0]);
        hs1[o] = (short)h5i(px[2 * o][1], px[2 * o + 1][1], px[2 * o + 2][1], px[2 * o + 3][1], px[2 * o + 4][1]);
        hs2[o] = (short)h5i(px[2 * o][2], px[2 * o + 1][2], px[2 * o + 2][2], px[2 * o + 3][2], px[2 * o + 4][2]);
        hw[o] = h5f(f[2 * o], f[2 * o + 1], f[2 * o + 2], f[2 * o + 3], f[2 * o + 4]);
    }
}

// Occupancy map of the level the pyrDown kernels write (StxMbImage::occ): called by every lane that reached the end of phase 2
// with the OR of the bit patterns of its weights (lane = column pair p = tid & 31 of row pair rg = tid >> 5 of the 64 x 14 tile
// at (X0, Y0)); lane p = 0 — never beyond the image — stores the half-wavefront's verdict.
// bytes per row of an occupancy map over a level `lw` columns wide: one per 64 columns, rows on dword boundaries
STX_DEV int occ_pitch(int lw) { return (((lw + 63) >> 6) + 3) & ~3; }
STX_DEV void dn_note_occ(uint8_t* __restrict__ occ, uint32_t nzbits, int tid, int X0, int Y0, int ow, int oh)
{
    const unsigned long long bal = __ballot(nzbits != 0u);
    const int rg = tid >> 5;
    if (occ == nullptr || (tid & 31) != 0 || 2 * rg >= DN_TOH || Y0 + 2 * rg >= oh) return;
    const uint32_t half = (tid & 32) ? (uint32_t)(bal >> 32) : (uint32_t)bal;
    occ[(long long)((Y0 >> 1) + rg) * occ_pitch(ow) + (X0 >> 6)] = half != 0u ? 1 : 0;
}

#ifndef STX_DN0_REV
#define STX_DN0_REV 1
#endif
// bit 0: the binary-mask instantiation, bit 1: the grey-mask one
#ifndef STX_DN0_BATCH
#define STX_DN0_BATCH 3
#endif
// blockIdx.z = image: all fed images are processed by one launch (deferred pyramid build)
template <bool PK>
__global__ __launch_bounds__(256) void mb_down0_lds_kernel(const StxMbImage* __restrict__ images, StxTileMap M)
{
    __shared__ __attribute__((aligned(16))) short s_h[3][DN_ROWS][DN_TOW];  // horizontal sums, <= 255*16
    __shared__ __attribute__((aligned(16))) float s_w[DN_ROWS][DN_TOW];
    // (round 6, STX_DN0_REV) The launch walks the images and their tiles in the REVERSE of the order the batched warp wrote them: the
    // warped images of a panorama (317 MB on config 2) are a little more than the 256 MB Infinity Cache holds, so a second pass in the
    // same order finds every line evicted just before it asks for it, while the reverse pass starts on what was written last.
    // (gridDim.x - 1 - b keeps b mod 8: the tiles of a band still meet on one XCD.)
    const uint32_t zz = STX_DN0_REV ? gridDim.z - 1u - blockIdx.z : blockIdx.z, bb = STX_DN0_REV ? gridDim.x - 1u - blockIdx.x : blockIdx.x;
    const StxMbImage& im = images[zz];
    const int tid = threadIdx.x;
    const int ow = im.fw >> 1, oh = im.fh >> 1;
    int tile_tx, tile_ty;
    if (!xcd_tile(M, bb, tile_tx, tile_ty)) return;
    const int X0 = tile_tx * DN_TOW, Y0 = tile_ty * DN_TOH;
    if (im.img0_is_s16 || X0 >= ow || Y0 >= oh) return;  // int16 sources take the generic kernel
    uint8_t* const occ = im.occ[1];
    const bool w1h = im.w1_f16 != 0;
    asm volatile("" ::"s"(occ));  // fetched with the other descriptor fields, not at the tail where nothing hides the round trip
    DnImage0 D;
    {
        int fw = im.fw, fh = im.fh, iw_ = im.iw, ih_ = im.ih, left = im.left, top = im.top;
        uint32_t is = (uint32_t)im.img0_stride, ms = (uint32_t)im.mask0_stride;
        unsigned long long ia = (unsigned long long)im.img0, ma = (unsigned long long)im.mask0;
        // (the binary-mask instantiation only: the grey-mask one measured 148 -> 155 us with its fields pinned, three interleaved runs of the
        // reference-default leg, tools/gpu_r6r.sh; the binary one 121.7 -> 120.1)
        if (PK) asm volatile("" : "+s"(fw), "+s"(fh), "+s"(iw_), "+s"(ih_), "+s"(left), "+s"(top), "+s"(is), "+s"(ms), "+s"(ia), "+s"(ma));
        D.fw = fw; D.fh = fh; D.iw = iw_; D.ih = ih_; D.left = left; D.top = top; D.img0_stride = is; D.mask0_stride = ms;
        D.img0 = (const uint8_t*)ia; D.mask0 = (const uint8_t*)ma;
    }
    // ... and what the second phase stores through (they came back as five more scalar round trips behind the barrier, once per row)
    uint32_t g1_plane = 0, g1_stride = 0, w1_stride = 0;
    unsigned long long g1_a = 0, w1_a = 0;
    if (PK) {
        g1_plane = (uint32_t)im.g_plane[1]; g1_stride = (uint32_t)im.g_stride[1]; w1_stride = (uint32_t)im.wt_stride[1];
        g1_a = (unsigned long long)im.g[1]; w1_a = (unsigned long long)im.wt[1];
        asm volatile("" : "+s"(g1_plane), "+s"(g1_stride), "+s"(w1_stride), "+s"(g1_a), "+s"(w1_a));
    }
    // rows / columns of the tile past the image's last output feed nothing (narrow exchange strips and the right / bottom
    // edge tiles would otherwise run the reflecting slow path for them)
    const int r_end = 2 * min(DN_TOH, oh - Y0) + 3;
    // uniform for the workgroup: which form of the border index maps this image needs (dn_task_level0)
    const bool near = PK ? (D.left <= D.iw && D.fw - D.left - D.iw <= D.iw && D.top <= D.ih && D.fh - D.top - D.ih <= D.ih)
                         : (im.left <= im.iw && im.fw - im.left - im.iw <= im.iw && im.top <= im.ih && im.fh - im.top - im.ih <= im.ih);
    // (round 6) Every 11-pixel run of the wavefront either inside the image or wholly inside the left / right frame one mirror image away
    // (the two forms dn_task_level0 serves with four vector loads): a lane's TWO tasks — the same column group q in the rows r and r + 16 —
    // take their six image loads and two mask loads in ONE batch.  The task loop below makes four dependent memory round trips of them
    // (image, then the mask behind its row test; twice) in front of the workgroup's barrier.
    bool batched = false;
    if ((STX_DN0_BATCH & (PK ? 1 : 2)) && near) {
        const int q = tid & 15, rA = tid >> 4;
        const int xo = X0 + 4 * q, c0 = 2 * xo - 2, a0 = c0 - D.left;
        const bool col_ok = xo < ow;
        const bool in_frame = c0 >= 0 && c0 + 10 < D.fw;
        const bool interior = in_frame && a0 >= 0 && a0 + 10 < D.iw;
        const int s_low = a0 + 10 < 0 ? -(a0 + 10) - 1 : 2 * D.iw - 1 - (a0 + 10);  // a run in the frame: the source column of its LAST pixel = the lowest
        const bool mirrored = in_frame && (a0 + 10 < 0 || a0 >= D.iw) && s_low >= 0 && s_low + 10 < D.iw;
        batched = __builtin_amdgcn_ballot_w64(col_ok && !(interior || mirrored)) == 0ull;  // wave-uniform; the four wavefronts decide for themselves
        if (batched && col_ok) {
            const int acol = interior ? a0 : s_low;  // first image column of the 11 that are loaded
            v4u d0[2], d1[2], mq4[2];
            v2u d2[2];
            uint32_t sh[2], msh[2];
            bool yin[2], live[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int r = rA + 16 * t;
                live[t] = r < r_end && r < DN_ROWS;
                const int row = 2 * Y0 - 2 + min(r, r_end - 1);  // (a row past the tile's last is loaded from the last and not stored)
                const int by = reflect101_near(row, D.fh) - D.top;
                yin[t] = (unsigned)by < (unsigned)D.ih;
                const int sy = reflect_near(by, D.ih);
                const uint32_t off = (uint32_t)sy * (uint32_t)D.img0_stride + (uint32_t)acol * 3u;
                const STX_GAS uint8_t* p = gp(D.img0) + (off & ~3u);
                sh[t] = off & 3u;
                d0[t] = *reinterpret_cast<const STX_GAS v4u_a4*>(p);
                d1[t] = *reinterpret_cast<const STX_GAS v4u_a4*>(p + 16);
                d2[t] = *reinterpret_cast<const STX_GAS v2u_a4*>(p + 32);
                // the mask's 11 bytes (cleared below for a row outside the image and for a run in the frame: the weight's border is CONSTANT 0)
                const uint32_t moff = (uint32_t)min(max(by, 0), D.ih - 1) * (uint32_t)D.mask0_stride + (uint32_t)acol;
                msh[t] = moff & 3u;
#if STX_ABLATE_MASK
                mq4[t] = v4u{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};  // timing experiment only
#else
                mq4[t] = *reinterpret_cast<const STX_GAS v4u_a4*>(gp(D.mask0) + (moff & ~3u));
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2; t++) {
                if (!live[t]) continue;
                const int r = rA + 16 * t;
                uint32_t w[9];
                w[0] = __builtin_amdgcn_alignbyte(d0[t].y, d0[t].x, sh[t]);
                w[1] = __builtin_amdgcn_alignbyte(d0[t].z, d0[t].y, sh[t]);
                w[2] = __builtin_amdgcn_alignbyte(d0[t].w, d0[t].z, sh[t]);
                w[3] = __builtin_amdgcn_alignbyte(d1[t].x, d0[t].w, sh[t]);
                w[4] = __builtin_amdgcn_alignbyte(d1[t].y, d1[t].x, sh[t]);
                w[5] = __builtin_amdgcn_alignbyte(d1[t].z, d1[t].y, sh[t]);
                w[6] = __builtin_amdgcn_alignbyte(d1[t].w, d1[t].z, sh[t]);
                w[7] = __builtin_amdgcn_alignbyte(d2[t].x, d1[t].w, sh[t]);
                w[8] = __builtin_amdgcn_alignbyte(d2[t].y, d2[t].x, sh[t]);
                if (interior) {
                    dn_pack5_channel<false, 0>(w, &s_h[0][r][4 * q]);
                    dn_pack5_channel<false, 1>(w, &s_h[1][r][4 * q]);
                    dn_pack5_channel<false, 2>(w, &s_h[2][r][4 * q]);
                } else {
                    dn_pack5_channel<true, 0>(w, &s_h[0][r][4 * q]);
                    dn_pack5_channel<true, 1>(w, &s_h[1][r][4 * q]);
                    dn_pack5_channel<true, 2>(w, &s_h[2][r][4 * q]);
                }
                const uint32_t keep = (yin[t] && interior) ? 0xffffffffu : 0u;
                uint32_t mw[3];
                mw[0] = __builtin_amdgcn_alignbyte(mq4[t].y, mq4[t].x, msh[t]) & keep;
                mw[1] = __builtin_amdgcn_alignbyte(mq4[t].z, mq4[t].y, msh[t]) & keep;
                mw[2] = __builtin_amdgcn_alignbyte(mq4[t].w, mq4[t].z, msh[t]) & keep & 0x00ffffffu;
                bool packed = PK;
                if (!PK) {  // byte b is 0 or 255  <=>  b == 255 * (b >> 7); over the lanes of this task (see dn_task_level0)
                    const uint32_t odd = (mw[0] ^ (((mw[0] >> 7) & 0x01010101u) * 255u)) | (mw[1] ^ (((mw[1] >> 7) & 0x01010101u) * 255u)) |
                                         (mw[2] ^ (((mw[2] >> 7) & 0x01010101u) * 255u));
                    packed = __builtin_amdgcn_ballot_w64(odd != 0u) == 0ull;
                }
                if (packed) {
                    uint32_t mb[3];  // 0 / 255 -> 0 / 1: W_0 is exactly 0.f or 1.f, the fp32 row sums are small integers
                    mb[0] = mw[0] & 0x01010101u;
                    mb[1] = mw[1] & 0x01010101u;
                    mb[2] = mw[2] & 0x01010101u;
                    *reinterpret_cast<float4*>(&s_w[r][4 * q]) = dn_mask_sums4(mb);
                } else {
                    float f[11];
#pragma unroll
                    for (int j = 0; j < 11; j++) f[j] = fmul((float)byte_of(mw, j), INV255);
#pragma unroll
                    for (int o = 0; o < 4; o++) s_w[r][4 * q + o] = h5f(f[2 * o], f[2 * o + 1], f[2 * o + 2], f[2 * o + 3], f[2 * o + 4]);
                }
            }
        }
    }
    for (int task = tid; task < DN_ROWS * (DN_TOW / 4) && !batched; task += 256) {
        const int r = task / (DN_TOW / 4), q = task % (DN_TOW / 4);
        if (X0 + 4 * q >= ow || r >= r_end) continue;
        if (PK) {  // the pinned copy of the descriptor
            if (near) dn_task_level0<PK, true>(D, 2 * Y0 - 2 + r, X0 + 4 * q, &s_h[0][r][4 * q], &s_h[1][r][4 * q], &s_h[2][r][4 * q], &s_w[r][4 * q]);
            else dn_task_level0<PK, false>(D, 2 * Y0 - 2 + r, X0 + 4 * q, &s_h[0][r][4 * q], &s_h[1][r][4 * q], &s_h[2][r][4 * q], &s_w[r][4 * q]);
        } else {   // the descriptor itself, as rounds 1 - 5 read it
            if (near) dn_task_level0<PK, true>(im, 2 * Y0 - 2 + r, X0 + 4 * q, &s_h[0][r][4 * q], &s_h[1][r][4 * q], &s_h[2][r][4 * q], &s_w[r][4 * q]);
            else dn_task_level0<PK, false>(im, 2 * Y0 - 2 + r, X0 + 4 * q, &s_h[0][r][4 * q], &s_h[1][r][4 * q], &s_h[2][r][4 * q], &s_w[r][4 * q]);
        }
    }
    __syncthreads();
    const int p = tid & 31, rg = tid >> 5;  // output pair, row group (2 rows)
    const int xo = X0 + 2 * p;
    if (xo >= ow) return;
    const bool two = xo + 1 < ow;
    uint32_t nz = 0u;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int yl = 2 * rg + rr, y = Y0 + yl;
        if (yl >= DN_TOH || y >= oh) break;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            // two outputs per register: the row sums are <= 255 * 16, the column sum + 128 <= 65408 fits 16 bits
            pk16 a[5];
#pragma unroll
            for (int k = 0; k < 5; k++) a[k] = pk(*reinterpret_cast<const uint32_t*>(&s_h[c][2 * yl + k][2 * p]));
            const pk16 v = (a[0] + a[4] + a[2] * pk_splat(6) + (a[1] + a[3]) * pk_splat(4) + pk_splat(128)) >> pk_splat(8);
            // G_1 of a u8 image is <= 255: one byte per sample (StxMbImage::g_u8)
            STX_GAS uint8_t* o = PK ? gp(reinterpret_cast<uint8_t*>(g1_a)) + ((uint32_t)c * g1_plane + (uint32_t)y * g1_stride + (uint32_t)xo)
                                    : gp(reinterpret_cast<uint8_t*>(im.g[1])) + ((uint32_t)c * (uint32_t)im.g_plane[1] + (uint32_t)y * (uint32_t)im.g_stride[1] + (uint32_t)xo);
            const uint32_t b2 = __builtin_amdgcn_perm(0u, unpk(v), 0x0c0c0200u);
            if (two) *reinterpret_cast<STX_GAS uint16_t*>(o) = (uint16_t)b2;
            else o[0] = (uint8_t)b2;
        }
        float fa[5], fb[5];
        {
            // Five 8-byte LDS reads (rows 2 yl .. 2 yl + 4, row pitch 256 bytes).  Written as ds_read_b64 by hand: the compiler
            // turns each float2 into two dword accesses (ds_read2_b32), whose 32 lanes then hit every other bank twice
            // (1.1 conflict cycles per LDS instruction measured); the 8-byte form is conflict free.
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 r0, r1, r2, r3, r4;
            const uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)&s_w[2 * yl][2 * p];
            asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:256\n\tds_read_b64 %2, %5 offset:512\n\t"
                         "ds_read_b64 %3, %5 offset:768\n\tds_read_b64 %4, %5 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4) : "v"(a) : "memory");
            fa[0] = r0.x; fb[0] = r0.y; fa[1] = r1.x; fb[1] = r1.y; fa[2] = r2.x; fb[2] = r2.y;
            fa[3] = r3.x; fb[3] = r3.y; fa[4] = r4.x; fb[4] = r4.y;
        }
        const float wa = fmul(h5f(fa[0], fa[1], fa[2], fa[3], fa[4]), INV256);
        const float wb = fmul(h5f(fb[0], fb[1], fb[2], fb[3], fb[4]), INV256);
        if (w1h) {  // StxMbImage::w1_f16: k / 256 as halves (exact: a 9-bit significand)
            STX_GAS _Float16* o = PK ? gp(reinterpret_cast<_Float16*>(w1_a)) + ((uint32_t)y * w1_stride + (uint32_t)xo)
                                     : reinterpret_cast<STX_GAS _Float16*>(gp(im.wt[1])) + ((uint32_t)y * (uint32_t)im.wt_stride[1] + (uint32_t)xo);
            typedef _Float16 v2h16 __attribute__((ext_vector_type(2)));
            if (two) { const v2h16 wab = {(_Float16)wa, (_Float16)wb}; *reinterpret_cast<STX_GAS v2h16*>(o) = wab; }
            else o[0] = (_Float16)wa;
        } else {
            STX_GAS float* o = PK ? gp(reinterpret_cast<float*>(w1_a)) + ((uint32_t)y * w1_stride + (uint32_t)xo)
                                  : gp(im.wt[1]) + ((uint32_t)y * (uint32_t)im.wt_stride[1] + (uint32_t)xo);
            typedef float v2fl __attribute__((ext_vector_type(2)));
            if (two) { const v2fl wab = {wa, wb}; *reinterpret_cast<STX_GAS v2fl*>(o) = wab; }
            else o[0] = wa;
        }
        nz |= __float_as_uint(wa) | (two ? __float_as_uint(wb) : 0u);
    }
    dn_note_occ(occ, nz, tid, X0, Y0, ow, oh);
}

// level >= 1 of a byte pyramid: the 19 samples p[c0 .. c0 + 18], c0 = 16 t - 2: dword, 16 bytes, dword around them.
// p = base + off (global address space, 32-bit offset from the uniform plane base: scalar base + lane offset loads)
STX_DEV void dn_load19_u8(const STX_GAS uint8_t* __restrict__ base, uint32_t off, int c0, int iw, bool fast, int s[19])
{
    if (fast) {
        const STX_GAS uint8_t* p = base + (off + (uint32_t)c0);
        const uint32_t a = *reinterpret_cast<const STX_GAS uint32_t*>(p - 2);
        const v4u b = *reinterpret_cast<const STX_GAS v4u*>(p + 2);
        const uint32_t c = *reinterpret_cast<const STX_GAS uint32_t*>(p + 18);
        const uint32_t w[6] = {a, b.x, b.y, b.z, b.w, c};
#pragma unroll
        for (int j = 0; j < 19; j++) s[j] = (int)byte_of(w, j + 2);
    } else {
#pragma unroll
        for (int j = 0; j < 19; j++) s[j] = base[off + (uint32_t)reflect101_near(c0 + j, iw)];
    }
}

// level >= 1 (planar int16 x3 + fp32): 8 outputs from 19 input elements per plane
STX_DEV void dn_load19_s16(const STX_GAS short* __restrict__ base, uint32_t off, int c0, int iw, bool fast, int s[19])
{
    if (fast) {  // c0 = 16t - 2: dword, 2 x 16-byte, short
        const STX_GAS short* p = base + (off + (uint32_t)c0);
        const uint32_t a = *reinterpret_cast<const STX_GAS uint32_t*>(p);
        const v4u b = *reinterpret_cast<const STX_GAS v4u*>(p + 2);
        const v4u c = *reinterpret_cast<const STX_GAS v4u*>(p + 10);
        s[0] = s16lo(a); s[1] = s16hi(a);
        s[2] = s16lo(b.x); s[3] = s16hi(b.x); s[4] = s16lo(b.y); s[5] = s16hi(b.y);
        s[6] = s16lo(b.z); s[7] = s16hi(b.z); s[8] = s16lo(b.w); s[9] = s16hi(b.w);
        s[10] = s16lo(c.x); s[11] = s16hi(c.x); s[12] = s16lo(c.y); s[13] = s16hi(c.y);
        s[14] = s16lo(c.z); s[15] = s16hi(c.z); s[16] = s16lo(c.w); s[17] = s16hi(c.w);
        s[18] = p[18];
    } else {
#pragma unroll
        for (int j = 0; j < 19; j++) s[j] = base[off + (uint32_t)reflect101_near(c0 + j, iw)];
    }
}

// horizontal 1-4-6-4-1 sums of 19 samples -> 8 outputs, parked in the LDS column order of mb_down_lds_kernel (even output columns
// in the first half of the row, odd ones in the second: see the kernel)
STX_DEV void dn_hsum_store_i(const int (&s)[19], int* row_q)
{
    int4 lo, hi;
    lo.x = h5i(s[0], s[1], s[2], s[3], s[4]);
    lo.y = h5i(s[2], s[3], s[4], s[5], s[6]);
    lo.z = h5i(s[4], s[5], s[6], s[7], s[8]);
    lo.w = h5i(s[6], s[7], s[8], s[9], s[10]);
    hi.x = h5i(s[8], s[9], s[10], s[11], s[12]);
    hi.y = h5i(s[10], s[11], s[12], s[13], s[14]);
    hi.z = h5i(s[12], s[13], s[14], s[15], s[16]);
    hi.w = h5i(s[14], s[15], s[16], s[17], s[18]);
    *reinterpret_cast<int4*>(row_q) = make_int4(lo.x, lo.z, hi.x, hi.z);
    *reinterpret_cast<int4*>(row_q + 32) = make_int4(lo.y, lo.w, hi.y, hi.w);
}
STX_DEV void dn_hsum_store_f(const float (&f)[19], float* row_q)
{
    float4 lo, hi;
    lo.x = h5f(f[0], f[1], f[2], f[3], f[4]);
    lo.y = h5f(f[2], f[3], f[4], f[5], f[6]);
    lo.z = h5f(f[4], f[5], f[6], f[7], f[8]);
    lo.w = h5f(f[6], f[7], f[8], f[9], f[10]);
    hi.x = h5f(f[8], f[9], f[10], f[11], f[12]);
    hi.y = h5f(f[10], f[11], f[12], f[13], f[14]);
    hi.z = h5f(f[12], f[13], f[14], f[15], f[16]);
    hi.w = h5f(f[14], f[15], f[16], f[17], f[18]);
    *reinterpret_cast<float4*>(row_q) = make_float4(lo.x, lo.z, hi.x, hi.z);
    *reinterpret_cast<float4*>(row_q + 32) = make_float4(lo.y, lo.w, hi.y, hi.w);
}

__global__ __launch_bounds__(256) void mb_down_lds_kernel(const StxMbImage* __restrict__ images, int lv, StxTileMap M)
{
    __shared__ __attribute__((aligned(16))) int s_h[3][DN_ROWS][DN_TOW];
    __shared__ __attribute__((aligned(16))) float s_w[DN_ROWS][DN_TOW];
    const StxMbImage& im = images[blockIdx.z];
    const int tid = threadIdx.x;
    const int iw = im.fw >> lv, ih = im.fh >> lv;
    const int ow = iw >> 1, oh = ih >> 1;
    int tile_tx, tile_ty;
    if (!xcd_tile(M, blockIdx.x, tile_tx, tile_ty)) return;
    const int X0 = tile_tx * DN_TOW, Y0 = tile_ty * DN_TOH;
    if (X0 >= ow || Y0 >= oh) return;
    const bool g8b = im.g_u8 != 0;  // byte planes (u8 image) or int16 planes: uniform for the workgroup
    // every descriptor field of both phases in ONE batch of scalar loads, pinned (round 6: as in the level-0 kernel — the weight pointer
    // came back inside the task loop, the output pointers behind the barrier, once per row: seven dependent scalar round trips)
    unsigned long long G_a = (unsigned long long)im.g[lv], W_a = (unsigned long long)im.wt[lv];
    unsigned long long Go_a = (unsigned long long)im.g[lv + 1], Wo_a = (unsigned long long)im.wt[lv + 1];
    uint32_t gs = (uint32_t)im.g_stride[lv], gpl = (uint32_t)im.g_plane[lv], wst = (uint32_t)im.wt_stride[lv];
    uint32_t gos = (uint32_t)im.g_stride[lv + 1], gopl = (uint32_t)im.g_plane[lv + 1], wost = (uint32_t)im.wt_stride[lv + 1];
    asm volatile("" : "+s"(G_a), "+s"(W_a), "+s"(Go_a), "+s"(Wo_a), "+s"(gs), "+s"(gpl), "+s"(wst), "+s"(gos), "+s"(gopl), "+s"(wost));
    const short* G = (const short*)G_a;
    uint8_t* const occ = im.occ[lv + 1];
    const bool w_half = lv == 1 && im.w1_f16 != 0;
    asm volatile("" ::"s"(occ));  // as in the level-0 kernel
    const int r_end = 2 * min(DN_TOH, oh - Y0) + 3;  // as in the level-0 kernel
    for (int task = tid; task < DN_ROWS * (DN_TOW / 8); task += 256) {
        const int r = task / (DN_TOW / 8), q = task % (DN_TOW / 8);
        if (X0 + 8 * q >= ow || r >= r_end) continue;
        const int sy = reflect101_near(2 * Y0 - 2 + r, ih);
        const int c0 = 2 * (X0 + 8 * q) - 2;
        const bool fast = c0 >= 0 && c0 + 18 < iw;
        typedef _Float16 v2h16 __attribute__((ext_vector_type(2)));
        typedef _Float16 v8h16 __attribute__((ext_vector_type(8)));
        // (round 6) a wavefront of a byte pyramid whose lanes are all clear of the border: the loads of the three planes and of the weights
        // leave in ONE batch.  Plane by plane — a branch on `fast` sits in front of every plane's loads — a task was four dependent memory
        // round trips, and a workgroup of this kernel is its tasks' round trips + a barrier.
        if (g8b && __builtin_amdgcn_ballot_w64(!fast) == 0ull) {
            uint32_t ra[3], rc[3];
            v4u rb[3];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const STX_GAS uint8_t* pp = gp(reinterpret_cast<const uint8_t*>(G)) + ((uint32_t)c * gpl + (uint32_t)sy * gs + (uint32_t)c0);
                ra[c] = *reinterpret_cast<const STX_GAS uint32_t*>(pp - 2);
                rb[c] = *reinterpret_cast<const STX_GAS v4u*>(pp + 2);
                rc[c] = *reinterpret_cast<const STX_GAS uint32_t*>(pp + 18);
            }
            float f[19];
            if (w_half) {  // wave-uniform
                const STX_GAS _Float16* hq = gp(reinterpret_cast<const _Float16*>(W_a)) + (uint32_t)sy * wst;
                const v2h16 ha = *reinterpret_cast<const STX_GAS v2h16*>(hq + c0);
                const v8h16 hb = *reinterpret_cast<const STX_GAS v8h16*>(hq + c0 + 2), hc = *reinterpret_cast<const STX_GAS v8h16*>(hq + c0 + 10);
                const _Float16 hl = hq[c0 + 18];
                __builtin_amdgcn_sched_barrier(0);
                f[0] = (float)ha.x; f[1] = (float)ha.y;
#pragma unroll
                for (int k = 0; k < 8; k++) { f[2 + k] = (float)hb[k]; f[10 + k] = (float)hc[k]; }
                f[18] = (float)hl;
            } else {
                const STX_GAS float* wq0 = gp(reinterpret_cast<const float*>(W_a)) + (uint32_t)sy * wst;
                const v2u wa = *reinterpret_cast<const STX_GAS v2u*>(wq0 + c0);
                v4u wb4[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) wb4[k4] = *reinterpret_cast<const STX_GAS v4u*>(wq0 + c0 + 2 + 4 * k4);
                const float wl = wq0[c0 + 18];
                __builtin_amdgcn_sched_barrier(0);
                f[0] = __uint_as_float(wa.x); f[1] = __uint_as_float(wa.y);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    f[2 + 4 * k] = __uint_as_float(wb4[k].x); f[3 + 4 * k] = __uint_as_float(wb4[k].y); f[4 + 4 * k] = __uint_as_float(wb4[k].z);
                    f[5 + 4 * k] = __uint_as_float(wb4[k].w);
                }
                f[18] = wl;
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const uint32_t w6[6] = {ra[c], rb[c].x, rb[c].y, rb[c].z, rb[c].w, rc[c]};
                int s[19];
#pragma unroll
                for (int j = 0; j < 19; j++) s[j] = (int)byte_of(w6, j + 2);
                dn_hsum_store_i(s, &s_h[c][r][4 * q]);
            }
            dn_hsum_store_f(f, &s_w[r][4 * q]);
            continue;
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int s[19];
            if (g8b) dn_load19_u8(gp(reinterpret_cast<const uint8_t*>(G)), (uint32_t)c * gpl + (uint32_t)sy * gs, c0, iw, fast, s);
            else dn_load19_s16(gp(G), (uint32_t)c * gpl + (uint32_t)sy * gs, c0, iw, fast, s);
            // LDS column order: even output columns in the first half of the row, odd ones in the second (column c at
            // (c & 1) * 32 + (c >> 1)).  Stores: the eight lanes of a 16-byte store group write 128 contiguous bytes (natural
            // order: 32-byte lane pitch, lanes q and q + 4 on the same banks); loads: the pair (2p, 2p+1) is two dword
            // reads of 32 consecutive banks each (natural order: every other bank, twice).  3.5 conflict cycles per LDS
            // instruction measured with the natural order.
            dn_hsum_store_i(s, &s_h[c][r][4 * q]);
        }
        const STX_GAS float* wq = gp(reinterpret_cast<const float*>(W_a)) + (uint32_t)sy * wst;
        float f[19];
        if (w_half) {  // level 1 as halves (StxMbImage::w1_f16): the same 19 samples from half the bytes
            const STX_GAS _Float16* hq = gp(reinterpret_cast<const _Float16*>(W_a)) + (uint32_t)sy * wst;
            if (fast) {
                const v2h16 a = *reinterpret_cast<const STX_GAS v2h16*>(hq + c0);
                const v8h16 b = *reinterpret_cast<const STX_GAS v8h16*>(hq + c0 + 2), c = *reinterpret_cast<const STX_GAS v8h16*>(hq + c0 + 10);
                f[0] = (float)a.x; f[1] = (float)a.y;
#pragma unroll
                for (int k = 0; k < 8; k++) { f[2 + k] = (float)b[k]; f[10 + k] = (float)c[k]; }
                f[18] = (float)hq[c0 + 18];
            } else {
#pragma unroll
                for (int j = 0; j < 19; j++) f[j] = (float)hq[reflect101_near(c0 + j, iw)];
            }
        } else if (fast) {
            const v2u a = *reinterpret_cast<const STX_GAS v2u*>(wq + c0);
            f[0] = __uint_as_float(a.x); f[1] = __uint_as_float(a.y);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const v4u b = *reinterpret_cast<const STX_GAS v4u*>(wq + c0 + 2 + 4 * k);
                f[2 + 4 * k] = __uint_as_float(b.x); f[3 + 4 * k] = __uint_as_float(b.y); f[4 + 4 * k] = __uint_as_float(b.z);
                f[5 + 4 * k] = __uint_as_float(b.w);
            }
            f[18] = wq[c0 + 18];
        } else {
#pragma unroll
            for (int j = 0; j < 19; j++) f[j] = wq[reflect101_near(c0 + j, iw)];
        }
        dn_hsum_store_f(f, &s_w[r][4 * q]);
    }
    __syncthreads();
    const int p = tid & 31, rg = tid >> 5;
    const int xo = X0 + 2 * p;
    if (xo >= ow) return;
    const bool two = xo + 1 < ow;
    uint32_t nz = 0u;
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
        const int yl = 2 * rg + rr, y = Y0 + yl;
        if (yl >= DN_TOH || y >= oh) break;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int a[5], b[5];
#pragma unroll
            for (int k = 0; k < 5; k++) {
                a[k] = s_h[c][2 * yl + k][p];
                b[k] = s_h[c][2 * yl + k][32 + p];
            }
            const int va = (h5i(a[0], a[1], a[2], a[3], a[4]) + 128) >> 8;
            const int vb = (h5i(b[0], b[1], b[2], b[3], b[4]) + 128) >> 8;
            const uint32_t oo = (uint32_t)c * gopl + (uint32_t)y * gos + (uint32_t)xo;
            if (g8b) {
                STX_GAS uint8_t* o = gp(reinterpret_cast<uint8_t*>(Go_a)) + oo;
                if (two) *reinterpret_cast<STX_GAS uint16_t*>(o) = (uint16_t)((uint32_t)va | ((uint32_t)vb << 8));
                else o[0] = (uint8_t)va;
            } else {
                STX_GAS short* o = gp(reinterpret_cast<short*>(Go_a)) + oo;
                if (two) *reinterpret_cast<STX_GAS uint32_t*>(o) = pack16(va, vb);
                else o[0] = (short)va;
            }
        }
        float fa[5], fb[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            fa[k] = s_w[2 * yl + k][p];
            fb[k] = s_w[2 * yl + k][32 + p];
        }
        const float wa = fmul(h5f(fa[0], fa[1], fa[2], fa[3], fa[4]), INV256);
        const float wb = fmul(h5f(fb[0], fb[1], fb[2], fb[3], fb[4]), INV256);
        STX_GAS float* o = gp(reinterpret_cast<float*>(Wo_a)) + ((uint32_t)y * wost + (uint32_t)xo);
        typedef float v2fl __attribute__((ext_vector_type(2)));
        if (two) { const v2fl wab = {wa, wb}; *reinterpret_cast<STX_GAS v2fl*>(o) = wab; }
        else o[0] = wa;
        nz |= __float_as_uint(wa) | (two ? __float_as_uint(wb) : 0u);
    }
    dn_note_occ(occ, nz, tid, X0, Y0, ow, oh);
}

// ---------------------------------------------------------------------------------------------
// gather + normalise + collapse.  One lane = 8 adjacent pixels x 2 rows; tile 512 x 8.
// ---------------------------------------------------------------------------------------------
// pyrUp's FixPtCast<short,6>: (v + 32) >> 6.  The (short) cast is the identity here: the taps sum to 64,
// so the result is a convex combination of int16 inputs (+ rounding) and stays inside int16.
STX_DEV int s6(int v) { return (v + 32) >> 6; }

// static_cast<short>(float) for |v| <= 32768 (products L*w with w <= 1, quotients acc/(w+eps) — see
// DESIGN.md §4.4): v_cvt_i32_f32 truncates toward zero like cvttss2si and no wrap can occur.
STX_DEV int trunc_small(float v) { return (int)v; }

// pyrUp_<FixPtCast<short,6>> of one plane for the 8x2 patch whose coarse origin is (cx, cy);
// cx is a multiple of 4 and cx+3 < cw.  T = short: an int16 plane (finished levels, the pyramids of int16 images);
// T = uint8_t: a byte plane (the pyramids of u8 images)
template <class T>
STX_DEV void up_patch(const T* __restrict__ plane, long long stride, int cw, int ch, int cx, int cy, int up[2][8])
{
    const int rr[3] = {up_idx(cy - 1, ch), cy, up_idx(cy + 1, ch)};
    const bool le = cx == 0, re = cx + 4 >= cw;  // pyrUp's border rule: column -1 -> 1, column cw -> cw - 1
    int he[3][4], ho[3][4];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        const T* p = plane + (long long)rr[r] * stride;
        int c[6];
        if (sizeof(T) == 2) {
            // one dword-aligned 16-byte window plane[cx - 2 .. cx + 5] instead of an 8-byte load and two single samples
            const v4u w = *reinterpret_cast<const v4u_a4*>(p + cx - 2);
            c[1] = s16lo(w.y); c[2] = s16hi(w.y); c[3] = s16lo(w.z); c[4] = s16hi(w.z);
            c[0] = s16hi(w.x);
            c[5] = s16lo(w.w);
        } else {
            // bytes: one dword-aligned 12-byte window plane[cx - 4 .. cx + 7]
            const v3u w = *reinterpret_cast<const v3u_a4*>(p + cx - 4);
            c[0] = (int)(w.x >> 24);
            c[1] = (int)(w.y & 255u); c[2] = (int)((w.y >> 8) & 255u); c[3] = (int)((w.y >> 16) & 255u); c[4] = (int)(w.y >> 24);
            c[5] = (int)(w.z & 255u);
        }
        if (le) c[0] = c[2];
        if (re) c[5] = c[4];
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

// normalizeUsingWeightMap, + pyrUp(finished coarser level) saturating, store (level >= 1: planar int16;
// level 0: u8 panorama via convertScaleAbs, mask = weight > eps, optional int16 result)
template <bool L0>
STX_DEV void level_epilogue(const MbLevelK& P, int X0, int Y0, int (&acc)[2][8][3], float (&ws)[2][8])
{
    // normalizeUsingWeightMap, then + pyrUp(finished coarser level), saturating
    int v[2][8][3];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float den = fadd(ws[r][j], WEIGHT_EPS);
            float q0, q1, q2;
            div3_shared(den, (float)(short)acc[r][j][0], (float)(short)acc[r][j][1], (float)(short)acc[r][j][2], q0, q1, q2);
            v[r][j][0] = trunc_small(q0);
            v[r][j][1] = trunc_small(q1);
            v[r][j][2] = trunc_small(q2);
        }
    if (P.up) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int up[2][8];
            up_patch(P.up + c * P.up_plane - ((long long)P.up_y0 * P.up_stride + P.up_x0), P.up_stride, P.pw >> 1, P.ph >> 1,
                     X0 >> 1, Y0 >> 1, up);
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
                if (Y0 + r < P.y1)
                    *reinterpret_cast<uint4*>(P.out + c * P.out_plane + (long long)(Y0 + r - P.out_y0) * P.out_stride + (X0 - P.out_x0)) = o;
            }
    } else {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int y = Y0 + r;
            if (y >= P.y1) break;
            const int oy = y - P.pano_y0, ox = X0 - P.pano_x0;
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
            uint32_t* po = reinterpret_cast<uint32_t*>(P.pano + (long long)oy * P.pano_stride + (long long)ox * 3);
            *reinterpret_cast<uint2*>(po) = make_uint2(ob[0], ob[1]);
            *reinterpret_cast<uint2*>(po + 2) = make_uint2(ob[2], ob[3]);
            *reinterpret_cast<uint2*>(po + 4) = make_uint2(ob[4], ob[5]);
            *reinterpret_cast<uint2*>(P.pmask + (long long)oy * P.pmask_stride + ox) = make_uint2(om[0], om[1]);
            if (P.pano16) {
                uint32_t* p16 = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(P.pano16) + (long long)oy * P.pano16_stride +
                                                            (long long)ox * 6);
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

// Does image `im` have a non-zero weight in the rows Y0, Y0 + 1 of level lv under the 512 columns from tile_x on?  (Occupancy
// map written by the pyramid kernels; true when there is none.)  Level 0 has no map of its own: a non-zero mask value at the
// frame position (x, y) makes W_1(x >> 1, y >> 1) non-zero (the taps 2 x', 2 x' + 1 of the 5-tap kernel are direct, the weights
// are non-negative and far above underflow), so the level-1 entries over (x >> 1, y >> 1) bound it.  Passing over an image whose
// weights are all exactly 0 changes nothing: (short)(L * 0.f) = 0 and w + 0.f = w.
// (the fields as values: a caller that has them in registers — loaded in one batch with the rest of its image search — spares the
// dependent loads of the pointer and of the rectangle)
template <bool L0>
STX_DEV bool occ_hit_f(const uint8_t* occ, int fx, int fy, int fw, int fh, int lv, int tile_x, int Y0)
{
    const int sl = L0 ? 1 : lv;
    if (occ == nullptr) return true;
    const int lw = fw >> sl, lh = fh >> sl;
    const int fr = L0 ? (Y0 - fy) >> 1 : Y0 - (fy >> lv);  // row of level sl, frame coordinates
    int x0 = L0 ? (tile_x - fx) >> 1 : tile_x - (fx >> lv);
    int x1 = L0 ? (tile_x + 511 - fx) >> 1 : x0 + 511;
    x0 = max(x0, 0); x1 = min(x1, lw - 1);
    if (fr < 0 || fr >= lh || x0 > x1) return true;
    // entries t0 .. t1 (at most 5 at level 0: 256 columns of level 1; 9 otherwise) from one dword-aligned 8 / 12-byte load (rows
    // start on dword boundaries and the arena ends with slack); entries outside the range are masked off
    const int t0 = x0 >> 6, n = (x1 >> 6) - t0 + 1;
    const uint32_t* q = reinterpret_cast<const uint32_t*>(occ + (long long)(fr >> 1) * occ_pitch(lw) + (t0 & ~3));
    const uint32_t sh = (uint32_t)t0 & 3u;
    if (L0) {
        const uint2 d = *reinterpret_cast<const uint2*>(q);
        const uint32_t w0 = __builtin_amdgcn_alignbyte(d.y, d.x, sh), w1 = d.y >> (8u * sh);
        return ((w0 & (n >= 4 ? 0xffffffffu : (1u << (8 * n)) - 1u)) | (n > 4 ? w1 & 0xffu : 0u)) != 0u;
    }
    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
    const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh), w1 = __builtin_amdgcn_alignbyte(d2, d1, sh), w2 = d2 >> (8u * sh);
    const uint32_t m0 = n >= 4 ? 0xffffffffu : (1u << (8 * n)) - 1u;
    const uint32_t m1 = n >= 8 ? 0xffffffffu : (n > 4 ? (1u << (8 * (n - 4))) - 1u : 0u);
    return ((w0 & m0) | (w1 & m1) | (n > 8 ? w2 & 0xffu : 0u)) != 0u;
}
template <bool L0>
STX_DEV bool occ_hit(const StxMbImage& im, int lv, int tile_x, int Y0)
{
    return occ_hit_f<L0>(im.occ[L0 ? 1 : lv], im.fx, im.fy, im.fw, im.fh, lv, tile_x, Y0);
}

template <bool WF, bool AHEAD = true>
STX_DEV void level0_epilogue_pk(const MbLevelK& P, int X0, int Y0, uint32_t (&acc)[2][3][4], uint32_t (&cnt)[2][4], const float (*ws)[8]);

// CONTRIB: the image table may hold received contribution strips (kind 1); EMIT: write un-normalised sums.
// Both are compile-time so that the common single-GPU instantiation carries neither path.
// U8SRC (levels >= 1): every image was fed as u8, so G_i is 0..255 and pyrUp / the Laplacian run in packed 16-bit lanes
template <bool L0, bool CONTRIB, bool EMIT, bool U8SRC>
STX_DEV void mb_level_fast_body(const MbLevelK& P)
{
    const int tid = threadIdx.x;
    const int lv = P.level;
    int tile_tx, tile_ty;
    if (!xcd_tile(P.tiles, blockIdx.x, tile_tx, tile_ty)) return;
    const int tile_x = P.x0 + tile_tx * 512, tile_y = P.y0 + tile_ty * LV_TH;
    // a wavefront owns two rows: its row index is wave-uniform, and saying so (readfirstlane) moves every row test, row
    // offset and row pointer below to the scalar unit
    const int X0 = tile_x + (tid & 63) * 8, Y0 = tile_y + __builtin_amdgcn_readfirstlane(tid >> 6) * 2;
    const bool active = X0 < P.x1 && Y0 < P.y1;

    int acc[2][8][3];
    float ws[2][8];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc[r][j][0] = acc[r][j][1] = acc[r][j][2] = 0;
            ws[r][j] = 0.f;
        }
    // Level 0 of u8 images with fp32 weights, single blender (no strips of products, no export): the sums live as int16 PAIRS
    // (order (0,2)(4,6)(1,3)(5,7) of the packed pyrUp), 24 registers instead of 48.  OpenCV's accumulator is a short that wraps
    // (dst += (short)(L * w)); a wrapping 16-bit add is that very operation, and the epilogue's (short) cast of the int sums of
    // the other instantiations gives the same residue.
    // The weight sums are COUNTS (16-bit pairs) until the first image with a grey mask byte under this wavefront arrives; then they
    // become the fp32 sums `ws` (a count converts exactly) and stay so.  A wavefront that never meets a grey byte — most of them: a
    // resized seam mask is grey only along the seam — runs the arithmetic of mb_level0_pk_kernel, epilogue included.
    constexpr bool PKACC = L0 && U8SRC && !CONTRIB && !EMIT;
    uint32_t accp[2][3][4], cntp[2][4];
    bool ws_live = false;  // per lane: a lane outside the grey image keeps counting until a grey image reaches it; a wavefront whose lanes
                           // disagree at the end runs both epilogues under their masks (one common fp32 epilogue measured slower)
#pragma unroll
    for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int c = 0; c < 3; c++) accp[r][c][0] = accp[r][c][1] = accp[r][c][2] = accp[r][c][3] = 0u;
        cntp[r][0] = cntp[r][1] = cntp[r][2] = cntp[r][3] = 0u;
    }

    // every wavefront finds the images under ITS two rows of the tile with one ballot (no LDS list, no barrier:
    // the other three wavefronts of the block no longer wait for the first one's descriptor loads)
    for (int base = 0; base < P.n_images; base += 64) {
        bool hit = false;
        {
            const int kk = base + (tid & 63);
            if (kk < P.n_images) {
                const StxMbImage& im = P.images[kk];
                int rx, ry, rw, rh;
                if (L0 && !(CONTRIB && im.kind == 1)) { rx = im.ix; ry = im.iy; rw = im.iw; rh = im.ih; }
                else { rx = im.fx >> lv; ry = im.fy >> lv; rw = im.fw >> lv; rh = im.fh >> lv; }
                hit = rx < tile_x + 512 && rx + rw > tile_x && ry < Y0 + 2 && ry + rh > Y0;
                if (hit) hit = occ_hit<L0>(im, lv, tile_x, Y0);
            }
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int k = base + (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const StxMbImage& im = P.images[k];
            if (!active) continue;
            if (!L0 || (CONTRIB && im.kind == 1)) {
                const bool contrib = CONTRIB && im.kind == 1;  // rows already hold (short)(L * W)
                const int lx0 = X0 - (im.fx >> lv), ly0 = Y0 - (im.fy >> lv);
                const int lw = im.fw >> lv, lh = im.fh >> lv;
                if ((unsigned)lx0 >= (unsigned)lw || (unsigned)ly0 >= (unsigned)lh) continue;
                float w[2][8];
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    if (lv == 1 && im.w1_f16) {  // level 1 as halves (StxMbImage::w1_f16)
                        typedef _Float16 v8h16 __attribute__((ext_vector_type(8)));
                        const v8h16 a = *reinterpret_cast<const v8h16*>(reinterpret_cast<const _Float16*>(im.wt[1]) + (long long)(ly0 + r) * im.wt_stride[1] + lx0);
#pragma unroll
                        for (int j = 0; j < 8; j++) w[r][j] = (float)a[j];
                        continue;
                    }
                    const float* q = im.wt[lv] + (long long)(ly0 + r) * im.wt_stride[lv] + lx0;
                    float4 a = *reinterpret_cast<const float4*>(q), b = *reinterpret_cast<const float4*>(q + 4);
                    w[r][0] = a.x; w[r][1] = a.y; w[r][2] = a.z; w[r][3] = a.w;
                    w[r][4] = b.x; w[r][5] = b.y; w[r][6] = b.z; w[r][7] = b.w;
                }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (U8SRC && !contrib) {
                        pk16 upk[2][4];
                        up_patch_pk(g8(im, lv + 1, c), (uint32_t)im.g_stride[lv + 1], lh >> 1, (uint32_t)(lx0 >> 1), ly0 >> 1,
                                    up_sel_u8(lx0 == 0, (lx0 >> 1) + 4 >= (lw >> 1)), upk);
#pragma unroll
                        for (int r = 0; r < 2; r++) {
                            uint32_t gq[4];  // the pyrUp pair order (0,2)(4,6)(1,3)(5,7)
                            g8_row_pairs(g8(im, lv, c), ly0 + r, (uint32_t)im.g_stride[lv], (uint32_t)lx0, gq);
                            const uint32_t Lq[4] = {unpk(pk(gq[0]) - upk[r][0]), unpk(pk(gq[1]) - upk[r][1]), unpk(pk(gq[2]) - upk[r][2]),
                                                    unpk(pk(gq[3]) - upk[r][3])};
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                const int q = (j & 1) ? 2 + (j >> 2) : (j >> 2);
                                const int L = ((j >> 1) & 1) ? s16hi(Lq[q]) : s16lo(Lq[q]);  // in [-255, 255]: subtract never saturates
                                acc[r][j][c] += trunc_small(fmul((float)L, w[r][j]));
                            }
                        }
                        continue;
                    }
                    // images of either kind: the pyramid of a u8 image is bytes, that of an int16 image (and a received contribution
                    // strip) int16 — a wave-uniform branch per image
                    const bool g8b = !contrib && im.g_u8 != 0;
                    int up[2][8];
                    if (!contrib) {
                        if (g8b) up_patch(reinterpret_cast<const uint8_t*>(im.g[lv + 1]) + c * im.g_plane[lv + 1], im.g_stride[lv + 1], lw >> 1,
                                          lh >> 1, lx0 >> 1, ly0 >> 1, up);
                        else up_patch(im.g[lv + 1] + c * im.g_plane[lv + 1], im.g_stride[lv + 1], lw >> 1, lh >> 1, lx0 >> 1, ly0 >> 1, up);
                    }
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        int gs8[8];
                        if (g8b) {
                            const uint2 gb = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint8_t*>(im.g[lv]) + c * im.g_plane[lv] +
                                                                             (long long)(ly0 + r) * im.g_stride[lv] + lx0);
                            const uint32_t gw2[2] = {gb.x, gb.y};
#pragma unroll
                            for (int j = 0; j < 8; j++) gs8[j] = (int)byte_of(gw2, j);
                        } else {
                            const short* grow = im.g[lv] + c * im.g_plane[lv] + (long long)(ly0 + r) * im.g_stride[lv] + lx0;
                            uint4 gv = *reinterpret_cast<const uint4*>(grow);
                            const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                            for (int j = 0; j < 8; j++) gs8[j] = (j & 1) ? s16hi(gw[j >> 1]) : s16lo(gw[j >> 1]);
                        }
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            int g = gs8[j];
                            if (contrib) {
                                acc[r][j][c] += g;
                            } else {
                                int L = sat_s16(g - up[r][j]);
                                acc[r][j][c] += trunc_small(fmul((float)L, w[r][j]));
                            }
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int j = 0; j < 8; j++) ws[r][j] = fadd(ws[r][j], w[r][j]);
            } else if (U8SRC) {
                // level 0 of u8 images with fp32 weights (masks that may be grey: resized seam masks): G_1 is 0..255, so pyrUp
                // and the Laplacian run in packed 16-bit lanes exactly as in mb_level0_pk_kernel (pair order (0,2)(4,6)(1,3)(5,7));
                // only the product with the weight and the weight sum are per pixel in fp32.  Pixels outside the image have
                // weight 0 and add nothing.  (num_bands > 0: the launcher's condition.)
                const int lx0 = X0 - im.ix, ly0 = Y0 - im.iy;
                if (lx0 + 8 <= 0 || lx0 >= im.iw || ly0 + 2 <= 0 || ly0 >= im.ih) continue;
                // lanes partly left / right of the image: the same aligned loads, mask bytes outside the image cleared (weight 0:
                // whatever the pixel bytes are, (short)(L * 0.f) = 0) — see mb_level0_pk_kernel
                uint32_t vm0, vm1;
                lane_valid_bytes(lx0, im.iw, vm0, vm1);
                uint32_t pw_[2][6], mw[2][2];
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int ly = ly0 + r;
#pragma unroll
                    for (int q = 0; q < 6; q++) pw_[r][q] = 0;
                    mw[r][0] = mw[r][1] = 0;
                    if ((unsigned)ly >= (unsigned)im.ih) continue;
                    load_px8_u8(im, lx0, ly, vm0, vm1, pw_[r], mw[r]);
                }
                float w[2][8];
                if (!PKACC) {
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int j = 0; j < 8; j++) w[r][j] = fmul((float)byte_of(mw[r], j), INV255);
                }
                // Resized seam masks are grey only along the seams: where every mask byte under this wavefront is 0 or 255 the product
                // (short)(L * w) is L or 0 — a mask AND on the packed Laplacian pairs instead of convert / multiply / convert per
                // pixel and channel (a byte is 0 / 255 iff it equals its sign bit replicated: t = the sign bits, (t << 8) - t = 255 t)
                bool img_binary = false;
                uint32_t mk[2][4];
                if (PKACC) {
                    uint32_t grey = 0u;
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int hlf = 0; hlf < 2; hlf++) {
                            const uint32_t t = (mw[r][hlf] >> 7) & 0x01010101u;
                            grey |= mw[r][hlf] ^ ((t << 8) - t);
                        }
                    img_binary = __ballot(grey != 0u) == 0ull;
#pragma unroll
                    for (int r = 0; r < 2; r++) {  // 0xffff / 0 per pixel in the pair order (0,2)(4,6)(1,3)(5,7)
                        mk[r][0] = __builtin_amdgcn_perm(0u, mw[r][0], 0x02020000u);
                        mk[r][1] = __builtin_amdgcn_perm(0u, mw[r][1], 0x02020000u);
                        mk[r][2] = __builtin_amdgcn_perm(0u, mw[r][0], 0x03030101u);
                        mk[r][3] = __builtin_amdgcn_perm(0u, mw[r][1], 0x03030101u);
                    }
                    if (!img_binary && !ws_live) {  // the first grey image of this wavefront: counts -> fp32 sums
                        ws_live = true;
#pragma unroll
                        for (int r = 0; r < 2; r++)
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                ws[r][(q & 1) * 4 + (q >> 1)] = (float)(cntp[r][q] & 0xffffu);
                                ws[r][(q & 1) * 4 + (q >> 1) + 2] = (float)(cntp[r][q] >> 16);
                            }
                    }
                    if (ws_live) {
#pragma unroll
                        for (int r = 0; r < 2; r++)
#pragma unroll
                            for (int j = 0; j < 8; j++) w[r][j] = fmul((float)byte_of(mw[r], j), INV255);
                    } else {
#pragma unroll
                        for (int r = 0; r < 2; r++)
#pragma unroll
                            for (int q = 0; q < 4; q++) cntp[r][q] = unpk(pk(cntp[r][q]) + pk(mk[r][q] & 0x00010001u));
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    pk16 upk[2][4];
                    up_patch_pk(g8(im, 1, c), (uint32_t)im.g_stride[1], im.fh >> 1, (uint32_t)((X0 - im.fx) >> 1),
                                (Y0 - im.fy) >> 1, up_sel_u8(X0 == im.fx, ((X0 - im.fx) >> 1) + 4 >= (im.fw >> 1)), upk);
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        uint32_t px[4];
                        if (c == 0) {
                            px[0] = pair_u8<0, 6>(pw_[r]); px[1] = pair_u8<12, 18>(pw_[r]);
                            px[2] = pair_u8<3, 9>(pw_[r]); px[3] = pair_u8<15, 21>(pw_[r]);
                        } else if (c == 1) {
                            px[0] = pair_u8<1, 7>(pw_[r]); px[1] = pair_u8<13, 19>(pw_[r]);
                            px[2] = pair_u8<4, 10>(pw_[r]); px[3] = pair_u8<16, 22>(pw_[r]);
                        } else {
                            px[0] = pair_u8<2, 8>(pw_[r]); px[1] = pair_u8<14, 20>(pw_[r]);
                            px[2] = pair_u8<5, 11>(pw_[r]); px[3] = pair_u8<17, 23>(pw_[r]);
                        }
                        uint32_t Lq[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) Lq[q] = unpk(pk(px[q]) - upk[r][q]);  // in [-255, 255]
                        if (PKACC) {
                            if (img_binary) {
#pragma unroll
                                for (int q = 0; q < 4; q++) accp[r][c][q] = unpk(pk(accp[r][c][q]) + pk(Lq[q] & mk[r][q]));
                            } else {
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    // pair q = pixels (j0, j0 + 2): j0 = 0, 4, 1, 5
                                    const int j0 = (q & 1) * 4 + (q >> 1);
                                    const int lo = trunc_small(fmul((float)s16lo(Lq[q]), w[r][j0]));
                                    const int hi = trunc_small(fmul((float)s16hi(Lq[q]), w[r][j0 + 2]));
                                    accp[r][c][q] = unpk(pk(accp[r][c][q]) + pk(pack16(lo, hi)));
                                }
                            }
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; j++) {
                                const int q = (j & 1) ? 2 + (j >> 2) : (j >> 2);
                                const int L = ((j >> 1) & 1) ? s16hi(Lq[q]) : s16lo(Lq[q]);
                                acc[r][j][c] += trunc_small(fmul((float)L, w[r][j]));
                            }
                        }
                    }
                }
                if (!PKACC || ws_live) {
#pragma unroll
                    for (int r = 0; r < 2; r++)
#pragma unroll
                        for (int j = 0; j < 8; j++) ws[r][j] = fadd(ws[r][j], w[r][j]);
                }
            } else {
                const int lx0 = X0 - im.ix, ly0 = Y0 - im.iy;
                if (lx0 + 8 <= 0 || lx0 >= im.iw || ly0 + 2 <= 0 || ly0 >= im.ih) continue;
                int up[3][2][8];
                if (P.num_bands > 0) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        if (im.g_u8) up_patch(reinterpret_cast<const uint8_t*>(im.g[1]) + c * im.g_plane[1], im.g_stride[1], im.fw >> 1, im.fh >> 1,
                                              (X0 - im.fx) >> 1, (Y0 - im.fy) >> 1, up[c]);
                        else up_patch(im.g[1] + c * im.g_plane[1], im.g_stride[1], im.fw >> 1, im.fh >> 1, (X0 - im.fx) >> 1, (Y0 - im.fy) >> 1, up[c]);
                    }
                }
                uint32_t vm0, vm1;
                lane_valid_bytes(lx0, im.iw, vm0, vm1);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int ly = ly0 + r;
                    if ((unsigned)ly >= (unsigned)im.ih) continue;
                    uint32_t pw_[6], mw[2];
                    load_px8_u8(im, lx0, ly, vm0, vm1, pw_, mw);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        if ((unsigned)(lx0 + j) >= (unsigned)im.iw) continue;  // outside the image: nothing is added (not even 0.f to the weight sum)
                        const float w = fmul((float)byte_of(mw, j), INV255);
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            int L = (int)byte_of(pw_, 3 * j + c);
                            if (P.num_bands > 0) L = sat_s16(L - up[c][r][j]);
                            acc[r][j][c] += trunc_small(fmul((float)L, w));
                        }
                        ws[r][j] = fadd(ws[r][j], w);
                    }
                }
            }
        }
    }
    if (!active) return;

    if (EMIT) {  // contribution strip for another rank: (short)acc and the weight sum, un-normalised
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (Y0 + r >= P.y1) break;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                uint4 o;
                o.x = pack16(acc[r][0][c], acc[r][1][c]);
                o.y = pack16(acc[r][2][c], acc[r][3][c]);
                o.z = pack16(acc[r][4][c], acc[r][5][c]);
                o.w = pack16(acc[r][6][c], acc[r][7][c]);
                *reinterpret_cast<uint4*>(P.out + c * P.out_plane + (long long)(Y0 + r - P.out_y0) * P.out_stride + (X0 - P.out_x0)) = o;
            }
            float* ow = P.out_w + (long long)(Y0 + r - P.out_y0) * P.out_w_stride + (X0 - P.out_x0);
            *reinterpret_cast<float4*>(ow) = make_float4(ws[r][0], ws[r][1], ws[r][2], ws[r][3]);
            *reinterpret_cast<float4*>(ow + 4) = make_float4(ws[r][4], ws[r][5], ws[r][6], ws[r][7]);
        }
        return;
    }

    if (PKACC) {
        if (!ws_live) {
            level0_epilogue_pk<false>(P, X0, Y0, accp, cntp, nullptr);
        } else {
            // compare(dst_band_weights_0, WEIGHT_EPS, CMP_GT) as 1 / 0 per pixel
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    cntp[r][q] = (ws[r][(q & 1) * 4 + (q >> 1)] > WEIGHT_EPS ? 1u : 0u) | (ws[r][(q & 1) * 4 + (q >> 1) + 2] > WEIGHT_EPS ? 0x10000u : 0u);
            level0_epilogue_pk<true>(P, X0, Y0, accp, cntp, ws);
        }
        return;
    }
    level_epilogue<L0>(P, X0, Y0, acc, ws);
}

template <bool L0, bool CONTRIB, bool EMIT, bool U8SRC>
__global__ __launch_bounds__(LV_THREADS) void mb_level_fast_kernel(MbLevelK P)
{
    mb_level_fast_body<L0, CONTRIB, EMIT, U8SRC>(P);
}


// Strip export for sharded blending: ONE launch for all (strip, level) pairs that take the same instantiation; blockIdx.z
// picks the argument block from a device array, blocks beyond a member's own tile grid leave at once.
template <bool L0, bool U8SRC>
__global__ __launch_bounds__(LV_THREADS) void mb_emit_multi_kernel(const MbLevelK* __restrict__ Ps)
{
    mb_level_fast_body<L0, false, true, U8SRC>(Ps[blockIdx.z]);
}


// ---------------------------------------------------------------------------------------------
// Packed 16-bit level-0 gather for the reference's actual operating point: every fed image is u8
// (stitching/blender.py:41 widens u8 warps to int16, so all values are 0..255) and every mask is
// binary 0/255 (warped masks, seam masks).  Then
//   * G_1 is 0..255, so pyrUp's sums (<= 64*255) fit unsigned 16-bit lanes: two pixels per VALU op;
//   * W_0 is exactly 0.f or 1.f (255 * (float)(1/255.) rounds to 1.f), so (short)(L * W) is L or 0:
//     the product / truncation become a bitwise AND, and the fp32 weight sum is an exact count.
// Pixel pairs inside a lane's 8-pixel strip are kept in the order pyrUp produces them:
//   pair 0 = (px0, px2), 1 = (px4, px6), 2 = (px1, px3), 3 = (px5, px7)   [lo half, hi half]
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Packed epilogue of the level-0 kernel (pair layout, see above).  Everything stays two pixels per register:
//   normalise : where no pixel of the lane is covered by more than one image, (short)(a / (1 + 1e-5f)) is
//               a - sign(a) (a / 1.00001 lies strictly between a - 1 and a for 0 < a < 1e5), i.e. three packed ops;
//               otherwise the exact shared-reciprocal division per element;
//   collapse  : pyrUp of the finished level 1 in signed 16-bit lanes when all 18 taps of the patch are within
//               +-500 (then no intermediate leaves int16), else the 32-bit patch; saturating add;
//   store     : |v| clamped to 255 (convertScaleAbs), zero where no image, bytes gathered with v_perm.
// ---------------------------------------------------------------------------------------------
typedef short pk16s __attribute__((ext_vector_type(2)));
STX_DEV pk16s pks(uint32_t v) { return __builtin_bit_cast(pk16s, v); }
STX_DEV uint32_t unpks(pk16s v) { return __builtin_bit_cast(uint32_t, v); }
STX_DEV pk16s pks_splat(short v) { pk16s r = {v, v}; return r; }

// returns false (and leaves `up` untouched) when a tap is outside [-500, 500].  In two halves like up_patch_pk: the three 16-byte
// window loads, then the arithmetic.
STX_DEV void up_patch_pks_load(const STX_GAS short* __restrict__ plane, uint32_t stride, int ch, uint32_t boff, int cy, v4u (&w)[3])
{
    const int rr[3] = {up_idx_s(cy - 1, ch), cy, up_idx_s(cy + 1, ch)};
    asm("" : "+v"(boff));  // as in up_patch_pk
#pragma unroll
    for (int r = 0; r < 3; r++)
        w[r] = *reinterpret_cast<const STX_GAS v4u_a4*>(reinterpret_cast<const STX_GAS char*>(row_ptr_s(plane, rr[r], stride)) + (size_t)boff - 4);
}
STX_DEV bool up_patch_pks_math(const v4u (&w)[3], UpSel sel, pk16s up[2][4])
{
    pk16s HE[3][2], HO[3][2];
    pk16 worst = pk_splat(0);
#pragma unroll
    for (int r = 0; r < 3; r++) {
        UpRow t;  // up_row_window's five registers from the window (x,c0) (c1,c2) (c3,c4) (c5,x)
        t.B0 = w[r].y;
        t.B1 = w[r].z;
        t.A1 = __builtin_amdgcn_alignbit(w[r].z, w[r].y, 16);
        t.A0 = __builtin_amdgcn_perm(w[r].y, w[r].x, sel.a0);
        t.A2 = __builtin_amdgcn_perm(w[r].w, w[r].z, sel.a2);
        worst = __builtin_elementwise_max(worst, pk(t.A0) + pk_splat(500));
        worst = __builtin_elementwise_max(worst, pk(t.A1) + pk_splat(500));
        worst = __builtin_elementwise_max(worst, pk(t.A2) + pk_splat(500));
        HE[r][0] = pks(t.A0) + pks(t.B0) * pks_splat(6) + pks(t.A1);
        HE[r][1] = pks(t.A1) + pks(t.B1) * pks_splat(6) + pks(t.A2);
        HO[r][0] = pks(t.B0) + pks(t.A1);
        HO[r][1] = pks(t.B1) + pks(t.A2);
    }
    if (unpk(__builtin_elementwise_min(worst, pk_splat(1000))) != unpk(worst)) return false;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        up[0][k] = (HE[0][k] + HE[1][k] * pks_splat(6) + HE[2][k] + pks_splat(32)) >> pks_splat(6);
        up[0][2 + k] = (HO[0][k] + HO[1][k] * pks_splat(6) + HO[2][k] + pks_splat(8)) >> pks_splat(4);
        up[1][k] = (HE[1][k] + HE[2][k] + pks_splat(8)) >> pks_splat(4);
        up[1][2 + k] = (HO[1][k] + HO[2][k] + pks_splat(2)) >> pks_splat(2);
    }
    return true;
}
STX_DEV bool up_patch_pks(const STX_GAS short* __restrict__ plane, uint32_t stride, int ch, uint32_t boff, int cy, UpSel sel, pk16s up[2][4])
{
    v4u w[3];
    up_patch_pks_load(plane, stride, ch, boff, cy, w);
    return up_patch_pks_math(w, sel, up);
}

// pair register / half that hold pixel j of a lane's 8-pixel strip
constexpr int pair_of(int j) { return (j & 1) ? 2 + (j >> 2) : (j >> 2); }
constexpr int half_of(int j) { return (j >> 1) & 1; }

// output dword K of a row of 8 BGR pixels from the per-channel pair registers U[c][q] (values 0..255 in each half)
template <int K>
STX_DEV uint32_t bgr_dword(const uint32_t (&U)[3][4])
{
    constexpr int b0 = 4 * K, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3;
    constexpr uint32_t s01 = (uint32_t)(2 * half_of(b0 / 3)) | ((uint32_t)(4 + 2 * half_of(b1 / 3)) << 8) | 0x0c0c0000u;
    constexpr uint32_t s23 = 0x00000c0cu | ((uint32_t)(2 * half_of(b2 / 3)) << 16) | ((uint32_t)(4 + 2 * half_of(b3 / 3)) << 24);
    const uint32_t lo = __builtin_amdgcn_perm(U[b1 % 3][pair_of(b1 / 3)], U[b0 % 3][pair_of(b0 / 3)], s01);
    const uint32_t hi = __builtin_amdgcn_perm(U[b3 % 3][pair_of(b3 / 3)], U[b2 % 3][pair_of(b2 / 3)], s23);
    return lo | hi;
}

// WF: the weight sums are fp32 values `ws` (grey masks somewhere under the wavefront): every pixel takes the division, and `cnt` holds
// 1 / 0 per pixel for "weight sum > WEIGHT_EPS" (only the final mask and the zeroing look at it)
// AHEAD: the pyrUp windows of the finished level 1 are loaded ahead of their use (see below); costs registers
template <bool WF, bool AHEAD>
STX_DEV void level0_epilogue_pk(const MbLevelK& P, int X0, int Y0, uint32_t (&acc)[2][3][4], uint32_t (&cnt)[2][4], const float (*ws)[8])
{
    // The constants (1, 1) and (-1, -1) as values the compiler cannot see through: against literal constants LLVM rewrites
    // min(max(a, -1), 1) and min(count, 1) into per-half compares and selects — 8 and 5 VALU per register instead of 3 and 2
    // (measured in the ISA: 193 -> 72 VALU for the normalisation of a lane).
    uint32_t k_p1 = 0x00010001u, k_m1 = 0xffffffffu;
    asm("" : "+s"(k_p1), "+s"(k_m1));
    // ---- normalizeUsingWeightMap
    uint32_t v[2][3][4];
    const uint32_t anyc = (cnt[0][0] | cnt[0][1] | cnt[0][2] | cnt[0][3]) | (cnt[1][0] | cnt[1][1] | cnt[1][2] | cnt[1][3]);
    // every count + 1 ORed: no bit above 2 set <=> every count <= 2 (the counts are below 256: no carry between the halves)
    const uint32_t k11 = 0x00010001u;
    const uint32_t cnt3 = ((cnt[0][0] + k11) | (cnt[0][1] + k11) | (cnt[0][2] + k11) | (cnt[0][3] + k11)) |
                          ((cnt[1][0] + k11) | (cnt[1][1] + k11) | (cnt[1][2] + k11) | (cnt[1][3] + k11));
    if (!WF && (anyc & 0xfffefffeu) == 0u) {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const pk16s a = pks(acc[r][c][q]);
                    v[r][c][q] = unpks(a - __builtin_elementwise_min(__builtin_elementwise_max(a, pks(k_m1)), pks(k_p1)));  // a - sign(a)
                }
    } else if (!WF && (STX_L0_NORM & 1) && (cnt3 & 0xfffcfffcu) == 0u) {
        // (round 6) at most TWO images over every pixel of the lane — all an un-pitched ring ever has.  For an integer count n <= 16 and
        // |a| <= 32768, (int)(a / (n + 1e-5f)) = trunc((a - sign(a)) / n): the quotient falls short of a / n by a * 1e-5 / n^2 < 1 / n,
        // 40+ ulps when a / n is an integer (tests/test_host_logic.py checks every a and n against IEEE division).  For n = 1, 2 that
        // is a shift in the packed lanes: 7 VALU per pixel pair and channel where the shared-reciprocal division takes ~18.
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const pk16 h = pk(cnt[r][q]) >> pk_splat(1);  // 1 where two images cover the pixel (a is 0 where none does)
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const pk16s a0 = pks(acc[r][c][q]);
                    const pk16s a = a0 - __builtin_elementwise_min(__builtin_elementwise_max(a0, pks(k_m1)), pks(k_p1));
                    const pk16 neg = pk(unpks(a)) >> pk_splat(15);
                    v[r][c][q] = unpks(pks(unpk(pk(unpks(a)) + (neg & h))) >> pks(unpk(h)));  // towards zero
                }
            }
    } else {
        // (round 6) integer counts below 16: the same quotients from ONE multiplication by v_rcp_f32's reciprocal (1 ulp) — the distance
        // of a / (n + 1e-5f) to the next integer is 1000 x the error of the product (checked for every a, n <= 30 and reciprocals 2 ulps off)
        const bool small = !WF && (STX_L0_NORM & 2) && (anyc & 0xfff0fff0u) == 0u;
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int o[2][3];
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    // pair q = pixels (j0, j0 + 2), j0 = 0, 4, 1, 5
                    const float den = WF ? fadd(ws[r][(q & 1) * 4 + (q >> 1) + 2 * hf], WEIGHT_EPS)
                                         : fadd((float)(hf ? (cnt[r][q] >> 16) : (cnt[r][q] & 0xffffu)), WEIGHT_EPS);
                    const float n0 = (float)(hf ? s16hi(acc[r][0][q]) : s16lo(acc[r][0][q])), n1 = (float)(hf ? s16hi(acc[r][1][q]) : s16lo(acc[r][1][q])),
                                n2 = (float)(hf ? s16hi(acc[r][2][q]) : s16lo(acc[r][2][q]));
                    float q0, q1, q2;
                    if (small) {
                        const float rr = __builtin_amdgcn_rcpf(den);
                        q0 = fmul(n0, rr); q1 = fmul(n1, rr); q2 = fmul(n2, rr);
                    } else {
                        div3_shared(den, n0, n1, n2, q0, q1, q2);
                    }
                    o[hf][0] = trunc_small(q0); o[hf][1] = trunc_small(q1); o[hf][2] = trunc_small(q2);
                }
#pragma unroll
                for (int c = 0; c < 3; c++) v[r][c][q] = pack16(o[0][c], o[1][c]);
            }
    }
    // ---- + pyrUp(finished level 1), saturating
    if (P.up) {
        // (round 6) AHEAD: the windows of the planes of the finished level 1 are loaded ahead of their use — planes 0 and 1 together,
        // plane 2 while plane 0 is worked on: two memory round trips at the end of every wavefront where a branch between the planes
        // made three.  It costs 12 registers: free for the instantiations that hold 88 anyway (grey-mask deferral, contribution strips:
        // the reference-default leg's level 0 193 -> 179 us together with the batched image search), one wavefront per SIMD less for the
        // binary-mask instantiation at 80 (measured: +1.4 us) — which therefore keeps the plane-by-plane form.
        v4u uw[3][3];
        const short* const plane0 = P.up - ((long long)P.up_y0 * P.up_stride + P.up_x0);
        if (AHEAD) {
            up_patch_pks_load(gp(plane0), (uint32_t)P.up_stride, P.ph >> 1, (uint32_t)(X0 >> 1) * 2u, Y0 >> 1, uw[0]);
            up_patch_pks_load(gp(plane0 + P.up_plane), (uint32_t)P.up_stride, P.ph >> 1, (uint32_t)(X0 >> 1) * 2u, Y0 >> 1, uw[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const short* plane = P.up + c * P.up_plane - ((long long)P.up_y0 * P.up_stride + P.up_x0);
            if (AHEAD) {
                if (c == 0) {
                    up_patch_pks_load(gp(plane0 + 2 * P.up_plane), (uint32_t)P.up_stride, P.ph >> 1, (uint32_t)(X0 >> 1) * 2u, Y0 >> 1, uw[2]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                up_patch_pks_load(gp(plane), (uint32_t)P.up_stride, P.ph >> 1, (uint32_t)(X0 >> 1) * 2u, Y0 >> 1, uw[c]);
            }
            pk16s up[2][4];
            if (!up_patch_pks_math(uw[c], up_sel(X0 == 0, (X0 >> 1) + 4 >= (P.pw >> 1)), up)) {
                int u32[2][8];
                up_patch(plane, P.up_stride, P.pw >> 1, P.ph >> 1, X0 >> 1, Y0 >> 1, u32);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    up[r][0] = pks(pack16(u32[r][0], u32[r][2]));
                    up[r][1] = pks(pack16(u32[r][4], u32[r][6]));
                    up[r][2] = pks(pack16(u32[r][1], u32[r][3]));
                    up[r][3] = pks(pack16(u32[r][5], u32[r][7]));
                }
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int q = 0; q < 4; q++) v[r][c][q] = unpks(__builtin_elementwise_add_sat(up[r][q], pks(v[r][c][q])));
        }
    }
    // ---- mask = weight > eps, zero outside, convertScaleAbs, store
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int y = Y0 + r;
        if (y >= P.y1) break;
        const int oy = y - P.pano_y0, ox = X0 - P.pano_x0;
        uint32_t keep[4], U[3][4];
#pragma unroll
        for (int q = 0; q < 4; q++) keep[q] = unpk(pk_splat(0) - __builtin_elementwise_min(pk(cnt[r][q]), pk(k_p1)));  // 0xffff where covered
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                v[r][c][q] &= keep[q];
                const pk16s x = pks(v[r][c][q]);
                const pk16s ax = __builtin_elementwise_max(x, __builtin_elementwise_sub_sat(pks_splat(0), x));
                U[c][q] = unpks(__builtin_elementwise_min(ax, pks_splat(255)));
            }
        uint32_t* po = reinterpret_cast<uint32_t*>(P.pano + (long long)oy * P.pano_stride + (long long)ox * 3);
        *reinterpret_cast<uint2*>(po) = make_uint2(bgr_dword<0>(U), bgr_dword<1>(U));
        *reinterpret_cast<uint2*>(po + 2) = make_uint2(bgr_dword<2>(U), bgr_dword<3>(U));
        *reinterpret_cast<uint2*>(po + 4) = make_uint2(bgr_dword<4>(U), bgr_dword<5>(U));
        // mask bytes of pixels 0..3 = keep[0].lo, keep[2].lo, keep[0].hi, keep[2].hi; 4..7 likewise from keep[1], keep[3]
        *reinterpret_cast<uint2*>(P.pmask + (long long)oy * P.pmask_stride + ox) =
            make_uint2(__builtin_amdgcn_perm(keep[2], keep[0], 0x06020400u), __builtin_amdgcn_perm(keep[3], keep[1], 0x06020400u));
        if (P.pano16) {
            uint32_t* p16 = reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(P.pano16) + (long long)oy * P.pano16_stride + (long long)ox * 6);
            int w[8][3];
#pragma unroll
            for (int j = 0; j < 8; j++)
#pragma unroll
                for (int c = 0; c < 3; c++) w[j][c] = half_of(j) ? s16hi(v[r][c][pair_of(j)]) : s16lo(v[r][c][pair_of(j)]);
            uint32_t t[12];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                t[(j >> 1) * 3 + 0] = pack16(w[j][0], w[j][1]);
                t[(j >> 1) * 3 + 1] = pack16(w[j][2], w[j + 1][0]);
                t[(j >> 1) * 3 + 2] = pack16(w[j + 1][1], w[j + 1][2]);
            }
            *reinterpret_cast<uint4*>(p16) = make_uint4(t[0], t[1], t[2], t[3]);
            *reinterpret_cast<uint4*>(p16 + 4) = make_uint4(t[4], t[5], t[6], t[7]);
            *reinterpret_cast<uint4*>(p16 + 8) = make_uint4(t[8], t[9], t[10], t[11]);
        }
    }
}

// DEFER (masks that may be grey, MbLevelK::defer_list): the packed arithmetic is exact for a lane as long as every mask byte under its
// 8 x 2 patch is 0 or 255, whatever the other lanes of the wavefront see — so the kernel runs as it is, a lane notes "grey byte seen" (a
// byte is 0 / 255 iff it equals its sign bit replicated), and at the end such a lane queues its patch for the fp32-weight pass instead of
// storing.  A resized seam mask is grey along the seams only: 1 - 2 % of the lanes (round 3 switched the WHOLE wavefront to fp32 sums at
// its first grey byte: 40 % of the wavefronts of the default pipeline).
template <bool CONTRIB, bool DEFER = false>
__global__ __launch_bounds__(LV_THREADS) __attribute__((amdgpu_waves_per_eu(STX_L0_WAVES, 8))) void mb_level0_pk_kernel(MbLevelK P)
{
    const int tid = threadIdx.x;
    int tile_tx, tile_ty;
    if (!xcd_tile(P.tiles, blockIdx.x, tile_tx, tile_ty)) return;
    const int tile_x = P.x0 + tile_tx * 512, tile_y = P.y0 + tile_ty * LV_TH;
    const int X0 = tile_x + (tid & 63) * 8, Y0 = tile_y + __builtin_amdgcn_readfirstlane(tid >> 6) * 2;  // Y0: wave-uniform (scalar)
    const bool active = X0 < P.x1 && Y0 < P.y1;

    uint32_t acc[2][3][4];  // [row][channel][pair]: int16 sums, wrap-around like OpenCV's short +=
    uint32_t cntb[2][2];    // [row][px / 4]: number of images whose mask covers the pixel, one byte per pixel (<= 255 images)
#pragma unroll
    for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int k = 0; k < 4; k++) acc[r][0][k] = acc[r][1][k] = acc[r][2][k] = 0;
        cntb[r][0] = cntb[r][1] = 0;
    }
    uint32_t grey_seen = 0u;  // DEFER: non-zero once a mask byte under this lane was neither 0 nor 255

    // every wavefront finds the images under ITS two rows with one ballot (no LDS list, no barrier)
    for (int base = 0; base < P.n_images; base += 64) {
        bool hit = false;
        {
            const int kk = base + (tid & 63);
            // (round 6) size and corner of the lane's image as ONE 16-byte load and tests without short circuits: `&&` had become three
            // dependent loads, each behind its own branch (see mb_level_pk_kernel)
            // (round 6) size and corner of the lane's image as ONE 16-byte load and tests without short circuits (`&&` had become three
            // dependent loads, each behind its own branch: see mb_level_pk_kernel) — where its 8 registers are free (see level0_epilogue_pk)
            if (DEFER || CONTRIB || STX_L0_FULL) {
                const StxMbImage& im = P.images[min(kk, P.n_images - 1)];
                const v4u f = *reinterpret_cast<const v4u_a4*>(&im.iw);  // iw, ih, ix, iy
                const v4u ff = *reinterpret_cast<const v4u_a4*>(&im.fx);  // fx, fy, fw, fh
                const uint8_t* const occp = im.occ[1];
                const int kind = CONTRIB ? im.kind : 0;
                int rx = (int)f.z, ry = (int)f.w, rw = (int)f.x, rh = (int)f.y;
                if (CONTRIB && kind == 1) { rx = (int)ff.x; ry = (int)ff.y; rw = (int)ff.z; rh = (int)ff.w; }
                hit = (kk < P.n_images) & (rx < tile_x + 512) & (rx + rw > tile_x) & (ry < Y0 + 2) & (ry + rh > Y0);
                if (hit) hit = occ_hit_f<true>(occp, (int)ff.x, (int)ff.y, (int)ff.z, (int)ff.w, 0, tile_x, Y0);
            } else if (kk < P.n_images) {
                const StxMbImage& im = P.images[kk];
                const int rx = im.ix, ry = im.iy, rw = im.iw, rh = im.ih;
                hit = rx < tile_x + 512 && rx + rw > tile_x && ry < Y0 + 2 && ry + rh > Y0;
                if (hit) hit = occ_hit<true>(im, 0, tile_x, Y0);
            }
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int k = base + (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const StxMbImage& im = P.images[k];
            if (!active) continue;
            if (CONTRIB && im.kind == 1) {
                // strip received from another rank: rows hold (short)(L * W) = L or 0 and W = 0.f / 1.f
                const int cx0 = X0 - im.fx, cy0 = Y0 - im.fy;
                if ((unsigned)cx0 >= (unsigned)im.fw || (unsigned)cy0 >= (unsigned)im.fh) continue;
#pragma unroll
                for (int r = 0; r < 2; r++) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const v4u gv = *reinterpret_cast<const STX_GAS v4u*>(
                            gp(im.g[0]) + c * im.g_plane[0] + ((uint32_t)(cy0 + r) * (uint32_t)im.g_stride[0] + (uint32_t)cx0));
                        acc[r][c][0] = unpk(pk(acc[r][c][0]) + pk(__builtin_amdgcn_perm(gv.y, gv.x, 0x05040100u)));
                        acc[r][c][1] = unpk(pk(acc[r][c][1]) + pk(__builtin_amdgcn_perm(gv.w, gv.z, 0x05040100u)));
                        acc[r][c][2] = unpk(pk(acc[r][c][2]) + pk(__builtin_amdgcn_perm(gv.y, gv.x, 0x07060302u)));
                        acc[r][c][3] = unpk(pk(acc[r][c][3]) + pk(__builtin_amdgcn_perm(gv.w, gv.z, 0x07060302u)));
                    }
                    const STX_GAS float* wq = gp(im.wt[0]) + ((uint32_t)(cy0 + r) * (uint32_t)im.wt_stride[0] + (uint32_t)cx0);
                    const v4u w0 = *reinterpret_cast<const STX_GAS v4u*>(wq), w1 = *reinterpret_cast<const STX_GAS v4u*>(wq + 4);
                    // 1.f = 0x3f800000, 0.f = 0: bit 23 is the count
                    const uint32_t i0 = (w0.x >> 23) & 1u, i1 = (w0.y >> 23) & 1u, i2 = (w0.z >> 23) & 1u, i3 = (w0.w >> 23) & 1u;
                    const uint32_t i4 = (w1.x >> 23) & 1u, i5 = (w1.y >> 23) & 1u, i6 = (w1.z >> 23) & 1u, i7 = (w1.w >> 23) & 1u;
                    cntb[r][0] += i0 | (i1 << 8) | (i2 << 16) | (i3 << 24);
                    cntb[r][1] += i4 | (i5 << 8) | (i6 << 16) | (i7 << 24);
                }
                continue;
            }
            // every field of the image's descriptor this iteration reads, as ONE batch of scalar loads (round 6, as in mb_level_pk_kernel)
            int i_ix = im.ix, i_iy = im.iy, i_iw = im.iw, i_ih = im.ih, i_fx = im.fx, i_fy = im.fy, i_fw = im.fw, i_fh = im.fh;
            unsigned long long I_a = (unsigned long long)im.img0, M_a = (unsigned long long)im.mask0, G1_a = (unsigned long long)im.g[1];
            uint32_t ist = (uint32_t)im.img0_stride, mst = (uint32_t)im.mask0_stride, g1s = (uint32_t)im.g_stride[1], g1p = (uint32_t)im.g_plane[1];
            asm("" : "+s"(i_ix), "+s"(i_iy), "+s"(i_iw), "+s"(i_ih), "+s"(i_fx), "+s"(i_fy), "+s"(i_fw), "+s"(i_fh), "+s"(I_a), "+s"(M_a), "+s"(G1_a),
                "+s"(ist), "+s"(mst), "+s"(g1s), "+s"(g1p));
            const int lx0 = X0 - i_ix, ly0 = Y0 - i_iy;
            if (lx0 + 8 <= 0 || lx0 >= i_iw || ly0 + 2 <= 0 || ly0 >= i_ih) continue;
            // lanes partly left / right of the image take the same aligned loads (load_px8_u8): no per-pixel path
            uint32_t pw_[2][6], mw[2][2];
#if STX_L0_FULL
            // every load of the image in one batch: both pixel rows (clamped row, mask cleared when outside) and the pyrUp windows of the planes
            const uint32_t g1_boff_f = (uint32_t)((X0 - i_fx) >> 1);
            v3u win[3][3];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int ly = ly0 + r;
                const uint32_t rm = (unsigned)ly < (unsigned)i_ih ? 0xffffffffu : 0u;
                load_px8_u8((const uint8_t*)I_a, ist, (const uint8_t*)M_a, mst, lx0, min(max(ly, 0), i_ih - 1), rm, rm, pw_[r], mw[r]);
            }
#pragma unroll
            for (int c = 0; c < 3; c++)
                up_patch_pk_load(gp(reinterpret_cast<const uint8_t*>(G1_a)) + (uint32_t)c * g1p, g1s, i_fh >> 1, g1_boff_f, (Y0 - i_fy) >> 1, win[c]);
            __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int ly = ly0 + r;
#pragma unroll
                for (int q = 0; q < 6; q++) pw_[r][q] = 0;
                mw[r][0] = mw[r][1] = 0;
                if ((unsigned)ly >= (unsigned)i_ih) continue;
                load_px8_u8((const uint8_t*)I_a, ist, (const uint8_t*)M_a, mst, lx0, ly, 0xffffffffu, 0xffffffffu, pw_[r], mw[r]);
            }
#endif
            {   // mask bytes of the pixels outside the image: cleared (after the loads: two registers less while they are in flight)
                uint32_t vm0, vm1;
                lane_valid_bytes(lx0, i_iw, vm0, vm1);
                mw[0][0] &= vm0; mw[0][1] &= vm1; mw[1][0] &= vm0; mw[1][1] &= vm1;
            }
            if (DEFER) {
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int hlf = 0; hlf < 2; hlf++) {
                        const uint32_t t = (mw[r][hlf] >> 7) & 0x01010101u;
                        grey_seen |= mw[r][hlf] ^ ((t << 8) - t);
                    }
            }
            // (no early-out on an all-zero mask: it would put the G_1 loads behind the mask loads' round trip)
            const uint32_t g1_boff = (uint32_t)((X0 - i_fx) >> 1);  // this lane's samples of G_1 (bytes): offset in a row
            const UpSel g1_sel = up_sel_u8(X0 == i_fx, ((X0 - i_fx) >> 1) + 4 >= (i_fw >> 1));
            uint32_t M[2][4];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                // the mask as 0 / 1: the coverage counters add its bytes, and as 16-bit pairs (short)(L * W) = L * m folds into
                // the accumulation as one v_pk_mad_u16 (the low 16 bits of L * 1 are L for negative L too)
                const uint32_t m01[2] = {mw[r][0] & 0x01010101u, mw[r][1] & 0x01010101u};
                M[r][0] = pair_u8<0, 2>(m01);
                M[r][1] = pair_u8<4, 6>(m01);
                M[r][2] = pair_u8<1, 3>(m01);
                M[r][3] = pair_u8<5, 7>(m01);
                cntb[r][0] += m01[0];  // four byte counters per register: no carry below 256 images
                cntb[r][1] += m01[1];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                pk16 up[2][4];
#if STX_L0_FULL
                up_patch_pk_math(win[c], g1_sel, up);
#else
                up_patch_pk(gp(reinterpret_cast<const uint8_t*>(G1_a)) + (uint32_t)c * g1p, g1s, i_fh >> 1, g1_boff, (Y0 - i_fy) >> 1, g1_sel, up);
#endif
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    uint32_t px[4];
                    if (c == 0) {
                        px[0] = pair_u8<0, 6>(pw_[r]); px[1] = pair_u8<12, 18>(pw_[r]);
                        px[2] = pair_u8<3, 9>(pw_[r]); px[3] = pair_u8<15, 21>(pw_[r]);
                    } else if (c == 1) {
                        px[0] = pair_u8<1, 7>(pw_[r]); px[1] = pair_u8<13, 19>(pw_[r]);
                        px[2] = pair_u8<4, 10>(pw_[r]); px[3] = pair_u8<16, 22>(pw_[r]);
                    } else {
                        px[0] = pair_u8<2, 8>(pw_[r]); px[1] = pair_u8<14, 20>(pw_[r]);
                        px[2] = pair_u8<5, 11>(pw_[r]); px[3] = pair_u8<17, 23>(pw_[r]);
                    }
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const pk16 L = pk(px[q]) - up[r][q];  // in [-255, 255]: the saturating subtract never clips
                        acc[r][c][q] = unpk(L * pk(M[r][q]) + pk(acc[r][c][q]));
                    }
                }
            }
        }
    }
    if (!active) return;
    if (DEFER) {
        // one atomic per wavefront that holds deferred lanes, on the counter of this workgroup's segment of the queue
        const bool mine = grey_seen != 0u;
        const unsigned long long dm = __ballot(mine);
        if (dm != 0ull) {
            const unsigned seg = (unsigned)tile_tx % (unsigned)STX_DEFER_SEGS;
            unsigned base_i = 0u;
            if ((unsigned)(tid & 63) == (unsigned)__builtin_ctzll(dm)) base_i = atomicAdd(P.defer_count + 32u * seg, (unsigned)__builtin_popcountll(dm));
            base_i = (unsigned)__builtin_amdgcn_readlane((int)base_i, (int)__builtin_ctzll(dm));
            if (mine) {
                const unsigned slot = base_i + (unsigned)__builtin_popcountll(dm & ((1ull << (tid & 63)) - 1ull));
                if (slot < P.defer_cap)
                    P.defer_list[(size_t)seg * P.defer_cap + slot] = (unsigned long long)(unsigned)X0 | ((unsigned long long)(unsigned)Y0 << 32);
                return;
            }
        }
    }

    uint32_t cnt[2][4];  // the epilogue wants the counts in the pair layout of acc
#pragma unroll
    for (int r = 0; r < 2; r++) {
        cnt[r][0] = pair_u8<0, 2>(cntb[r]);
        cnt[r][1] = pair_u8<4, 6>(cntb[r]);
        cnt[r][2] = pair_u8<1, 3>(cntb[r]);
        cnt[r][3] = pair_u8<5, 7>(cntb[r]);
    }
    level0_epilogue_pk<false, DEFER || CONTRIB || STX_L0_FULL>(P, X0, Y0, acc, cnt, nullptr);
}


// Second pass of a level-0 gather with deferral: one lane per queued 8 x 2 patch, at any position — nothing here is wave-uniform but the
// image loop.  Per pixel, channel and image: L = sat(img - pyrUp(G_1)), acc += (short)(L * w) with w = mask / 255 in fp32, weight sums
// in fp32, in feed order; then the common epilogue (division, collapse, convertScaleAbs): the arithmetic of mb_level_fast_body's general
// level-0 branch, for 1 - 2 % of the patches of a panorama.  A wavefront takes 64 consecutive entries of one segment (one 512-pixel
// column of the panorama: two or three images) and blocks of 64 entries are dealt round-robin over ALL wavefronts of the launch,
// whichever segment they come from: the seams of a panorama sit in a few columns, and with a fixed number of wavefronts per segment
// those columns' queues were worked off in 4 - 7 serial steps (80 us); a wavefront's own step is a dozen microseconds of dependent loads.
__global__ __launch_bounds__(256) void mb_level0_deferred_kernel(MbLevelK P)
{
    const unsigned wave = blockIdx.x * 4u + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), waves = gridDim.x * 4u;
    const unsigned lane = threadIdx.x & 63u;
    unsigned g0 = 0u;  // global index of the current segment's first block of 64 entries
    // every counter in one batch of loads (lane l holds those of the segments l, l + 64, ...): read one by one inside the loop they were
    // a chain of defer_segs dependent memory round trips in front of every wavefront's work — 40 of the 45 us this kernel took
    unsigned cnt_l[STX_DEFER_SEGS / 64];
#pragma unroll
    for (int j = 0; j < STX_DEFER_SEGS / 64; j++) {
        const unsigned sg = lane + 64u * (unsigned)j;
        cnt_l[j] = sg < (unsigned)P.defer_segs ? min(P.defer_count[32u * sg], P.defer_cap) : 0u;
    }
#pragma unroll
    for (int j = 0; j < STX_DEFER_SEGS / 64; j++)
    for (unsigned sl = 0; sl < 64u && 64u * (unsigned)j + sl < (unsigned)P.defer_segs; sl++) {
        const unsigned seg = 64u * (unsigned)j + sl;
        const unsigned n = (unsigned)__builtin_amdgcn_readlane((int)cnt_l[j], (int)sl), nb = (n + 63u) >> 6;
        // the blocks g of [g0, g0 + nb) with g = wave (mod waves)
        for (unsigned g = g0 + (wave + waves - g0 % waves) % waves; g < g0 + nb; g += waves) {
            const unsigned i = (g - g0) * 64u + lane;
            const bool live = i < n;
            const unsigned long long e = live ? P.defer_list[(size_t)seg * P.defer_cap + i] : 0ull;
            const int X0 = (int)(unsigned)(e & 0xffffffffull), Y0 = (int)(unsigned)(e >> 32);
            int acc[2][8][3];
            float ws[2][8];
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    acc[r][j][0] = acc[r][j][1] = acc[r][j][2] = 0;
                    ws[r][j] = 0.f;
                }
            for (int k = 0; k < P.n_images; k++) {
                const StxMbImage& im = P.images[k];
                const int lx0 = X0 - im.ix, ly0 = Y0 - im.iy;
                const bool inside = live && !(lx0 + 8 <= 0 || lx0 >= im.iw || ly0 + 2 <= 0 || ly0 >= im.ih);
                if (__ballot(inside) == 0ull) continue;
                if (!inside) continue;
                // Straight-line code from here on (no branch per channel, row or pixel), so that every load of this image — nine pyrUp
                // windows, two rows of pixels and mask bytes — is in flight before the first use: the 64 patches of this wavefront lie in
                // 64 different rows, every load instruction touches 64 cache lines, and with the loads issued behind one another's uses a
                // wavefront spent six dependent round trips per image (42 us for the pass).  A pixel outside the image gets weight 0
                // through its mask byte: (short)(L * 0.f) = 0 and w + 0.f = w, exactly what skipping it does.
                uint32_t vm0, vm1;
                {   // lane_valid_bytes without its wave-wide shortcut (the lanes of this kernel sit in different images' columns)
                    const int lo = max(-lx0, 0), hi = min(im.iw - lx0, 8);
                    const uint32_t bits = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                    vm0 = (((bits & 15u) * 0x00204081u) & 0x01010101u) * 0xffu;
                    vm1 = (((bits >> 4) * 0x00204081u) & 0x01010101u) * 0xffu;
                }
                uint32_t pw_[2][6], mw[2][2];
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int ly = ly0 + r;
                    const bool rin = (unsigned)ly < (unsigned)im.ih;
                    load_px8_u8(im, lx0, min(max(ly, 0), im.ih - 1), rin ? vm0 : 0u, rin ? vm1 : 0u, pw_[r], mw[r]);
                }
                int up[3][2][8];
#pragma unroll
                for (int c = 0; c < 3; c++)  // every image of this launch was fed as u8 (the launcher's condition): byte planes
                    up_patch(reinterpret_cast<const uint8_t*>(im.g[1]) + c * im.g_plane[1], im.g_stride[1], im.fw >> 1, im.fh >> 1,
                             (X0 - im.fx) >> 1, (Y0 - im.fy) >> 1, up[c]);
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const float w = fmul((float)byte_of(mw[r], j), INV255);
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const int L = sat_s16((int)byte_of(pw_[r], 3 * j + c) - up[c][r][j]);
                            acc[r][j][c] += trunc_small(fmul((float)L, w));
                        }
                        ws[r][j] = fadd(ws[r][j], w);
                    }
            }
            if (live) level_epilogue<true>(P, X0, Y0, acc, ws);
        }
        g0 += nb;
    }
}

// ---------------------------------------------------------------------------------------------
// Levels 1 .. B-3 of u8-sourced images (every fed image is u8x3, no received contribution strips): the reference's own
// operating point (stitching/blender.py:41 widens u8 warps).  G_lv is 0..255, so pyrUp and the Laplacian run in packed 16-bit
// lanes (as mb_level_fast_body<.., U8SRC>), and on top of that:
//   * the sums are kept as wrapping int16 pairs (OpenCV's `short +=`; only the low 16 bits of a sum are ever used);
//   * a weight pyramid is exactly 1.f in the interior of its mask (the taps sum to 256 / 256) and exactly 0.f outside its
//     reach.  Where all 16 weights of an image under a lane's 8 x 2 patch are 1.f for every lane of the wavefront
//     (min over the bit patterns: weights are in [0, 1]), (short)(L * 1.f) is L and the products / truncations / conversions
//     (5 VALU per sample and channel) become one packed add per pixel pair;
//   * a lane whose weight sums are all 0.f or 1.f — at most one all-ones image, every other image all zeros there —
//     normalises with a - sign(a) (three packed ops per pair, see level0_epilogue_pk) instead of 48 divisions;
//   * the collapse runs in signed 16-bit lanes when all taps are within +-500 (up_patch_pks), else the 32-bit patch.
// Same integers and the same fp32 operations in the same order as mb_level_fast_body; 104 -> ? us for levels 1 + 2 of config 2.
// ---------------------------------------------------------------------------------------------
STX_DEV uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { return min(min(a, b), c); }
STX_DEV uint32_t max3u(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }

// at least 3 wavefronts per SIMD (168 registers): the batched loads of round 6 would otherwise take 174 and leave two.
// Measured forms (one box, interleaved, tools/gpu_r6o.sh / gpu_r6p.sh; mb_level of config 2 = levels 1 + 2 / the four launches of the
// reference-default leg / of config 4's share):  HEAD (loads plane by plane, 103 registers, 4 per SIMD): 84.8 / 128.0 / 332.4 us;
// STX_LVPK_LOOP = 1 (all loads of an image in one batch, 3 per SIMD: shipped): 79.1 / 116.3 / 308.7;  STX_LVPK_LOOP = 0 with the batched
// image search and the windows of the epilogue ahead, held to 4 per SIMD (128 registers, 19 spill instructions): 89.5 / 146.5 / 376.6.
#ifndef STX_LVPK_WAVES
#define STX_LVPK_WAVES 3
#endif
#ifndef STX_LVPK_EPI
#define STX_LVPK_EPI 3
#endif
#ifndef STX_LVPK_LOOP
#define STX_LVPK_LOOP 1
#endif
__global__ __launch_bounds__(LV_THREADS) __attribute__((amdgpu_waves_per_eu(STX_LVPK_WAVES, 8))) void mb_level_pk_kernel(MbLevelK P)
{
    const int tid = threadIdx.x;
    const int lv = P.level;
    int tile_tx, tile_ty;
    if (!xcd_tile(P.tiles, blockIdx.x, tile_tx, tile_ty)) return;
    const int tile_x = P.x0 + tile_tx * 512, tile_y = P.y0 + tile_ty * LV_TH;
    const int X0 = tile_x + (tid & 63) * 8, Y0 = tile_y + __builtin_amdgcn_readfirstlane(tid >> 6) * 2;  // Y0: wave-uniform
    const bool active = X0 < P.x1 && Y0 < P.y1;

    uint32_t acc[2][3][4];  // [row][channel][pair]: wrapping int16 sums, pair order (0,2)(4,6)(1,3)(5,7)
    float ws[2][8];
#pragma unroll
    for (int r = 0; r < 2; r++) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc[r][0][q] = acc[r][1][q] = acc[r][2][q] = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) ws[r][j] = 0.f;
    }
    uint32_t ones = 0;     // images whose 16 weights here are all 1.f
    uint32_t mixed = 0;    // != 0: some image had a weight here that is neither covered by `ones` nor 0.f

    for (int base = 0; base < P.n_images; base += 64) {
        // (round 6) the rectangle of the lane's image as ONE 16-byte load, the tests without short circuits: left to `&&` the compiler
        // made three dependent loads of it, each behind its own branch — three memory round trips before a wavefront knew its images
        bool hit = false;
        {
            const int kk = base + (tid & 63);
            const StxMbImage& im = P.images[min(kk, P.n_images - 1)];
            const v4u f = *reinterpret_cast<const v4u_a4*>(&im.fx);  // fx, fy, fw, fh
            const uint8_t* const occp = im.occ[lv];                  // (in the same batch: the occupancy test then costs one round trip, not two)
            const int rx = (int)f.x >> lv, ry = (int)f.y >> lv, rw = (int)f.z >> lv, rh = (int)f.w >> lv;
            hit = (kk < P.n_images) & (rx < tile_x + 512) & (rx + rw > tile_x) & (ry < Y0 + 2) & (ry + rh > Y0);
            if (hit) hit = occ_hit_f<false>(occp, (int)f.x, (int)f.y, (int)f.z, (int)f.w, lv, tile_x, Y0);
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int k = base + (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const StxMbImage& im = P.images[k];
            if (!active) continue;
            // every field of the image's descriptor this iteration reads, as ONE batch of scalar loads (round 6: left to the compiler
            // they came back one dependent scalar-cache round trip after the other — rectangle, weight pointer, value pointers — in
            // front of the vector loads they address)
            int i_fx = im.fx, i_fy = im.fy, i_fw = im.fw, i_fh = im.fh, i_w1h = im.w1_f16;
            unsigned long long W_a = (unsigned long long)im.wt[lv], G0_a = (unsigned long long)im.g[lv], G1_a = (unsigned long long)im.g[lv + 1];
            uint32_t wst = (uint32_t)im.wt_stride[lv], g0s = (uint32_t)im.g_stride[lv], g1s = (uint32_t)im.g_stride[lv + 1];
            uint32_t g0p = (uint32_t)im.g_plane[lv], g1p = (uint32_t)im.g_plane[lv + 1];
            // (not volatile: a side-effecting asm counts as a memory clobber and would turn the NEXT iteration's descriptor loads into vector loads)
            asm("" : "+s"(i_fx), "+s"(i_fy), "+s"(i_fw), "+s"(i_fh), "+s"(i_w1h), "+s"(W_a), "+s"(G0_a), "+s"(G1_a), "+s"(wst), "+s"(g0s),
                "+s"(g1s), "+s"(g0p), "+s"(g1p));
            const int lx0 = X0 - (i_fx >> lv), ly0 = Y0 - (i_fy >> lv);
            const int lw = i_fw >> lv, lh = i_fh >> lv;
            if ((unsigned)lx0 >= (unsigned)lw || (unsigned)ly0 >= (unsigned)lh) continue;
            // the 16 weights (bit patterns; all in [0, 1], so unsigned order = float order)
            const bool w_half = lv == 1 && i_w1h != 0;
            uint32_t wb[2][8];
            typedef _Float16 v8h16 __attribute__((ext_vector_type(8)));
            v8h16 hw[2];
            // (round 6) every load of the image — its weights, the three pyrUp windows and the two value rows of each plane — leaves before
            // the first is used: one memory round trip per image where the plane-by-plane form made four (a wavefront of this kernel is a
            // chain of round trips and little else: 3 wavefronts per SIMD, 13 k wavefronts at level 1 of config 2)
            if (w_half) {  // level 1 as halves (StxMbImage::w1_f16): one 16-byte load per row
#pragma unroll
                for (int r = 0; r < 2; r++)
                    hw[r] = *reinterpret_cast<const STX_GAS v8h16*>(gp(reinterpret_cast<const _Float16*>(W_a)) + ((uint32_t)(ly0 + r) * wst + (uint32_t)lx0));
            } else {
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const STX_GAS float* q = gp(reinterpret_cast<const float*>(W_a)) + ((uint32_t)(ly0 + r) * wst + (uint32_t)lx0);
                    const v4u a = *reinterpret_cast<const STX_GAS v4u*>(q), b = *reinterpret_cast<const STX_GAS v4u*>(q + 4);
                    wb[r][0] = a.x; wb[r][1] = a.y; wb[r][2] = a.z; wb[r][3] = a.w;
                    wb[r][4] = b.x; wb[r][5] = b.y; wb[r][6] = b.z; wb[r][7] = b.w;
                }
            }
            const uint32_t g1_boff = (uint32_t)(lx0 >> 1);
            const UpSel g1_sel = up_sel_u8(lx0 == 0, (lx0 >> 1) + 4 >= (lw >> 1));
            v3u win[3][3];
            v2u grow[3][2];
#define STX_LVPK_LOAD(c)                                                                                                            \
    {                                                                                                                              \
        up_patch_pk_load(gp(reinterpret_cast<const uint8_t*>(G1_a)) + (uint32_t)(c) * g1p, g1s, lh >> 1, g1_boff, ly0 >> 1, win[c]);    \
        _Pragma("unroll") for (int r = 0; r < 2; r++)                                                                              \
            grow[c][r] = g8_row_load(gp(reinterpret_cast<const uint8_t*>(G0_a)) + (uint32_t)(c) * g0p, ly0 + r, g0s, (uint32_t)lx0);   \
    }
#if STX_LVPK_LOOP
#pragma unroll
            for (int c = 0; c < 3; c++) STX_LVPK_LOAD(c)
            __builtin_amdgcn_sched_barrier(0);
#endif
            bool all1;
            if (w_half) {
                // "all sixteen are 1" is decided on the half patterns themselves (weights are in [0, 1]: unsigned order = float order;
                // 1.0 = 0x3c00): seven packed minima instead of sixteen conversions in front of the path almost every wavefront takes
                const v4u h0 = __builtin_bit_cast(v4u, hw[0]), h1 = __builtin_bit_cast(v4u, hw[1]);
                const pk16 m = __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_min(pk(h0.x), pk(h0.y)), __builtin_elementwise_min(pk(h0.z), pk(h0.w))),
                                                         __builtin_elementwise_min(__builtin_elementwise_min(pk(h1.x), pk(h1.y)), __builtin_elementwise_min(pk(h1.z), pk(h1.w))));
                all1 = unpk(m) == 0x3c003c00u;
            } else {
                uint32_t lo = min3u(wb[0][0], wb[0][1], wb[0][2]);
                lo = min3u(lo, wb[0][3], wb[0][4]); lo = min3u(lo, wb[0][5], wb[0][6]); lo = min3u(lo, wb[0][7], wb[1][0]);
                lo = min3u(lo, wb[1][1], wb[1][2]); lo = min3u(lo, wb[1][3], wb[1][4]); lo = min3u(lo, wb[1][5], wb[1][6]);
                lo = min(lo, wb[1][7]);
                all1 = lo == 0x3f800000u;
            }
            const bool wave_all1 = __ballot(!all1) == 0ull;  // over the lanes that reached this point
            if (wave_all1) {
                ones += 1u;
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    pk16 upk[2][4];
#if !STX_LVPK_LOOP
                    STX_LVPK_LOAD(c)
#endif
                    up_patch_pk_math(win[c], g1_sel, upk);
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        uint32_t gq[4];  // the pyrUp pair order (0,2)(4,6)(1,3)(5,7); L in [-255, 255]
                        g8_pairs_of(grow[c][r], gq);
#pragma unroll
                        for (int q = 0; q < 4; q++) acc[r][c][q] = unpk(pk(acc[r][c][q]) + (pk(gq[q]) - upk[r][q]));
                    }
                }
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int j = 0; j < 8; j++) ws[r][j] = fadd(ws[r][j], 1.0f);
                continue;
            }
            if (w_half) {  // the fp32 patterns of the halves (exact) for the weighted path
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int j = 0; j < 8; j++) wb[r][j] = __float_as_uint((float)hw[r][j]);
            }
            {   // book-keeping for the epilogue's short cut
                uint32_t hi = max3u(wb[0][0], wb[0][1], wb[0][2]);
                hi = max3u(hi, wb[0][3], wb[0][4]); hi = max3u(hi, wb[0][5], wb[0][6]); hi = max3u(hi, wb[0][7], wb[1][0]);
                hi = max3u(hi, wb[1][1], wb[1][2]); hi = max3u(hi, wb[1][3], wb[1][4]); hi = max3u(hi, wb[1][5], wb[1][6]);
                hi = max(hi, wb[1][7]);
                if (all1) ones += 1u;
                else mixed |= hi;
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                pk16 upk[2][4];
#if !STX_LVPK_LOOP
                STX_LVPK_LOAD(c)
#endif
                up_patch_pk_math(win[c], g1_sel, upk);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    uint32_t gq[4];
                    g8_pairs_of(grow[c][r], gq);
                    const uint32_t Lq[4] = {unpk(pk(gq[0]) - upk[r][0]), unpk(pk(gq[1]) - upk[r][1]), unpk(pk(gq[2]) - upk[r][2]),
                                            unpk(pk(gq[3]) - upk[r][3])};
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        // pair q holds the pixels (jl, jh): q = 0: (0, 2), 1: (4, 6), 2: (1, 3), 3: (5, 7)
                        const int jl = (q & 1) * 4 + (q >> 1), jh = jl + 2;
                        const int tl = trunc_small(fmul((float)s16lo(Lq[q]), __uint_as_float(wb[r][jl])));
                        const int th = trunc_small(fmul((float)s16hi(Lq[q]), __uint_as_float(wb[r][jh])));
                        acc[r][c][q] = unpk(pk(acc[r][c][q]) + pk(pack16(tl, th)));
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int j = 0; j < 8; j++) ws[r][j] = fadd(ws[r][j], __uint_as_float(wb[r][j]));
        }
    }
    if (!active) return;

    // ---- normalizeUsingWeightMap
    uint32_t k_p1 = 0x00010001u, k_m1 = 0xffffffffu;  // opaque constants: see level0_epilogue_pk
    asm("" : "+s"(k_p1), "+s"(k_m1));
    uint32_t v[2][3][4];
    if (mixed == 0u && ones <= 1u) {  // every weight sum of the lane is 0.f or 1.f: (short)(a / (w + 1e-5f)) = a - sign(a) (0 stays 0)
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 3; c++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const pk16s a = pks(acc[r][c][q]);
                    v[r][c][q] = unpks(a - __builtin_elementwise_min(__builtin_elementwise_max(a, pks(k_m1)), pks(k_p1)));
                }
    } else {
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int jl = (q & 1) * 4 + (q >> 1), jh = jl + 2;
                int o[2][3];
#pragma unroll
                for (int hf = 0; hf < 2; hf++) {
                    const float den = fadd(ws[r][hf ? jh : jl], WEIGHT_EPS);
                    float q0, q1, q2;
                    div3_shared(den, (float)(hf ? s16hi(acc[r][0][q]) : s16lo(acc[r][0][q])),
                                (float)(hf ? s16hi(acc[r][1][q]) : s16lo(acc[r][1][q])),
                                (float)(hf ? s16hi(acc[r][2][q]) : s16lo(acc[r][2][q])), q0, q1, q2);
                    o[hf][0] = trunc_small(q0); o[hf][1] = trunc_small(q1); o[hf][2] = trunc_small(q2);
                }
#pragma unroll
                for (int c = 0; c < 3; c++) v[r][c][q] = pack16(o[0][c], o[1][c]);
            }
    }
    // ---- + pyrUp(finished coarser level), saturating
    if (P.up) {
        const UpSel usel = up_sel(X0 == 0, (X0 >> 1) + 4 >= (P.pw >> 1));
        // (round 6) the windows of the planes are loaded ahead of their use (they were three round trips: a branch sat between the planes)
        v4u uw[3][3];
        const short* const plane0 = P.up - ((long long)P.up_y0 * P.up_stride + P.up_x0);
#if STX_LVPK_EPI == 3
#pragma unroll
        for (int c = 0; c < 3; c++) up_patch_pks_load(gp(plane0 + c * P.up_plane), (uint32_t)P.up_stride, P.ph >> 1, (uint32_t)(X0 >> 1) * 2u, Y0 >> 1, uw[c]);
        __builtin_amdgcn_sched_barrier(0);
#elif STX_LVPK_EPI == 2
        up_patch_pks_load(gp(plane0), (uint32_t)P.up_stride, P.ph >> 1, (uint32_t)(X0 >> 1) * 2u, Y0 >> 1, uw[0]);
        up_patch_pks_load(gp(plane0 + P.up_plane), (uint32_t)P.up_stride, P.ph >> 1, (uint32_t)(X0 >> 1) * 2u, Y0 >> 1, uw[1]);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const short* plane = P.up + c * P.up_plane - ((long long)P.up_y0 * P.up_stride + P.up_x0);
#if STX_LVPK_EPI == 2
            if (c == 0) {
                up_patch_pks_load(gp(plane0 + 2 * P.up_plane), (uint32_t)P.up_stride, P.ph >> 1, (uint32_t)(X0 >> 1) * 2u, Y0 >> 1, uw[2]);
                __builtin_amdgcn_sched_barrier(0);
            }
#elif STX_LVPK_EPI < 2
            up_patch_pks_load(gp(plane), (uint32_t)P.up_stride, P.ph >> 1, (uint32_t)(X0 >> 1) * 2u, Y0 >> 1, uw[c]);
#endif
            pk16s up[2][4];
            if (!up_patch_pks_math(uw[c], usel, up)) {
                int u32[2][8];
                up_patch(plane, P.up_stride, P.pw >> 1, P.ph >> 1, X0 >> 1, Y0 >> 1, u32);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    up[r][0] = pks(pack16(u32[r][0], u32[r][2]));
                    up[r][1] = pks(pack16(u32[r][4], u32[r][6]));
                    up[r][2] = pks(pack16(u32[r][1], u32[r][3]));
                    up[r][3] = pks(pack16(u32[r][5], u32[r][7]));
                }
            }
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int q = 0; q < 4; q++) v[r][c][q] = unpks(__builtin_elementwise_add_sat(up[r][q], pks(v[r][c][q])));
        }
    }
    // ---- store: pair order -> natural order, 16 bytes per row and plane
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if (Y0 + r >= P.y1) break;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            v4u o;
            o.x = __builtin_amdgcn_perm(v[r][c][2], v[r][c][0], 0x05040100u);  // (px0, px1)
            o.y = __builtin_amdgcn_perm(v[r][c][2], v[r][c][0], 0x07060302u);  // (px2, px3)
            o.z = __builtin_amdgcn_perm(v[r][c][3], v[r][c][1], 0x05040100u);  // (px4, px5)
            o.w = __builtin_amdgcn_perm(v[r][c][3], v[r][c][1], 0x07060302u);  // (px6, px7)
            *reinterpret_cast<STX_GAS v4u*>(gp(P.out) + c * P.out_plane + (long long)(Y0 + r - P.out_y0) * P.out_stride + (X0 - P.out_x0)) = o;
        }
    }
}

bool launched_ok() { return hipGetLastError() == hipSuccess; }

}  // namespace

// One launch per pyramid level for ALL fed images (grid.z = image).  h_images mirrors d_images.
bool stx_fast_mb_down_batch(stx_ctx* ctx, const StxMbImage* d_images, const StxMbImage* h_images, int n, int level)
{
    int mw = 0, mh = 0;
    for (int i = 0; i < n; i++) {
        mw = std::max(mw, (h_images[i].fw >> level) >> 1);
        mh = std::max(mh, (h_images[i].fh >> level) >> 1);
    }
    if (n <= 0 || mw < 1 || mh < 1 || n > 65535) return false;
    // Tile order: XCD bands (2 tile rows per band) cut the level-0 fetch from 584 to 340 MB for 314 MB of input at equal
    // kernel time and equal end-to-end throughput (A/B on one box, two panoramas in flight: 116.1 / 116.9 against
    // 116.5 / 116.9 Gpix/s) — kept for the HBM traffic they leave to the co-running kernels.
    StxTileMap M = stx_tile_map((mw + DN_TOW - 1) / DN_TOW, (mh + DN_TOH - 1) / DN_TOH, DN_BAND);
    // Levels 1 and 2 in XCD bands too (round 6, visits af / ag).  Round 4 had measured the bands 15 % SLOWER on level 1 -> 2 (103 against
    // 88 us) and kept the plain row-major order from level 1 on, at 1.8 x the algorithmic bytes fetched; on the batched-load kernel of round 6
    // the bands win: mb_down x 4 88.7 -> 80 us, value +1.0 %, latency 0.683 -> 0.66 ms (three interleaved runs; bands on the levels >= 3 as well
    // change nothing more).  STX_DN_PLAIN_FROM = first level in plain order.
#ifndef STX_DN_PLAIN_FROM
#define STX_DN_PLAIN_FROM 3
#endif
    M.plain = level >= STX_DN_PLAIN_FROM;
    bool pk_ok = true;  // packed 16-bit row sums need u8 images with 0 / 255 masks
    for (int i = 0; i < n; i++) pk_ok = pk_ok && !h_images[i].img0_is_s16 && h_images[i].mask_binary;
    dim3 grid(stx_tile_grid(M), 1, n);
    static const unsigned pad_lds = getenv("STITCHING_AMD_D0_LDS") ? (unsigned)atoi(getenv("STITCHING_AMD_D0_LDS")) : 0u;  // diagnostic, see stx_warp.hip
    if (level == 0 && pk_ok) hipLaunchKernelGGL(mb_down0_lds_kernel<true>, grid, dim3(256), pad_lds, ctx->stream, d_images, M);
    else if (level == 0) hipLaunchKernelGGL(mb_down0_lds_kernel<false>, grid, dim3(256), 0, ctx->stream, d_images, M);
    else hipLaunchKernelGGL(mb_down_lds_kernel, grid, dim3(256), 0, ctx->stream, d_images, level, M);
    return launched_ok();
}

// preconditions of the register-blocked gather kernels; fills the tile map
static bool fast_level_ok(const MbLevelK& K, MbLevelK* KT)
{
    // 8-pixel strips must never straddle a feed-rectangle edge: 2^(B - level) >= 8; all origins 8-aligned
    if (K.num_bands - K.level < 3) return false;
    if ((K.x0 | K.y0 | K.out_x0 | K.out_y0 | K.pano_x0 | K.pano_y0) & 7) return false;
    if (K.up && ((K.up_x0 | K.up_y0) & 3)) return false;
    if (K.n_images > 255) return false;
    // vertically adjacent 512 x 8 tiles share the G_{i+1} / finished-level rows of their pyrUp halos: keep them on one XCD
    *KT = K;
    KT->tiles = stx_tile_map((K.x1 - K.x0 + 511) / 512, (K.y1 - K.y0 + LV_TH - 1) / LV_TH, LV_BAND);
    return true;
}

// Batched strip export (emit): class of the instantiation a (strip, level) argument block needs — 0: level 0, 1: level >= 1 of
// u8 pyramids, 2: level >= 1 general, -1: not eligible for the register-blocked kernels (generic kernel).  Fills the tile map.
int stx_fast_mb_emit_class(const MbLevelK& K, MbLevelK* KT)
{
    if (!(K.level > 0 || K.all_u8) || !fast_level_ok(K, KT)) { *KT = K; return -1; }
    return K.level == 0 ? 0 : (K.all_u8 ? 1 : 2);
}

// h_Ks: host copies of the `count` argument blocks at d_Ks, all of class `cls`
bool stx_fast_mb_emit_launch(stx_ctx* ctx, int cls, const MbLevelK* d_Ks, const MbLevelK* h_Ks, int count)
{
    unsigned gx = 1;
    for (int i = 0; i < count; i++) gx = std::max(gx, stx_tile_grid(h_Ks[i].tiles));
    const dim3 grid(gx, 1, (unsigned)count);
    if (cls == 0) hipLaunchKernelGGL((mb_emit_multi_kernel<true, false>), grid, dim3(LV_THREADS), 0, ctx->stream, d_Ks);
    else if (cls == 1) hipLaunchKernelGGL((mb_emit_multi_kernel<false, true>), grid, dim3(LV_THREADS), 0, ctx->stream, d_Ks);
    else hipLaunchKernelGGL((mb_emit_multi_kernel<false, false>), grid, dim3(LV_THREADS), 0, ctx->stream, d_Ks);
    return launched_ok();
}

// STITCHING_AMD_NO_DEFER (diagnostic): grey masks through round 3's wave-level switch (mb_level_fast_kernel<true, false, false, true>)
static bool no_defer()
{
    static const bool off = getenv("STITCHING_AMD_NO_DEFER") != nullptr;
    return off;
}

bool stx_fast_mb_level(stx_ctx* ctx, const MbLevelK& K)
{
    MbLevelK KT;
    if (!fast_level_ok(K, &KT)) return false;
    dim3 grid(stx_tile_grid(KT.tiles), 1);
    hipStream_t st = ctx->stream;
    if (K.level == 0 && K.pk_ok && !K.emit && K.num_bands > 0) {
        // STITCHING_AMD_L0_LDS (diagnostic): dynamic LDS as an occupancy limit, see stx_warp.hip
        static const unsigned pad_lds = getenv("STITCHING_AMD_L0_LDS") ? (unsigned)atoi(getenv("STITCHING_AMD_L0_LDS")) : 0u;
        if (K.has_contrib) hipLaunchKernelGGL(mb_level0_pk_kernel<true>, grid, dim3(LV_THREADS), pad_lds, st, KT);
        else hipLaunchKernelGGL(mb_level0_pk_kernel<false>, grid, dim3(LV_THREADS), pad_lds, st, KT);
    } else if (K.level == 0 && K.all_u8 && !K.has_contrib && !K.emit && K.num_bands > 0 && !no_defer()) {
        // u8 images whose masks are not known to be binary (resized seam masks: the reference's default pipeline): the packed kernel
        // with per-lane deferral + the fp32-weight pass over the queued patches.  The queue has room for every patch of the region; it
        // and its counter come from the stream-ordered allocator and go back to it right behind the second launch.
        // a tile (one wavefront: LV_THREADS lanes) holds at most LV_THREADS patches; segment s takes the tile columns tx = s (mod SEGS), so
        // only the first min(tiles_x, SEGS) segments exist and only they get room (a 1024-column panorama used to pay for all 256)
        const int segs = std::min(KT.tiles.tiles_x, STX_DEFER_SEGS);
        const size_t seg_cap = (size_t)((KT.tiles.tiles_x + STX_DEFER_SEGS - 1) / STX_DEFER_SEGS) * (size_t)KT.tiles.tiles_y * (unsigned)LV_THREADS;
        const size_t counters = (size_t)STX_DEFER_SEGS * 128;
        KT.defer_segs = segs;
        void* q = nullptr;
        if (stx_dev_alloc(ctx, counters + seg_cap * (size_t)segs * sizeof(unsigned long long), &q) != STX_OK) return false;
        KT.defer_count = reinterpret_cast<unsigned*>(q);
        KT.defer_list = reinterpret_cast<unsigned long long*>(reinterpret_cast<uint8_t*>(q) + counters);
        KT.defer_cap = (unsigned)seg_cap;
        hipMemsetAsync(q, 0, counters, st);
        hipLaunchKernelGGL((mb_level0_pk_kernel<false, true>), grid, dim3(LV_THREADS), 0, st, KT);
        {
            StxProfScope prof2(ctx, "mb_level0_deferred", 0.0);  // inside the caller's "mb_level0" bracket: that one times both launches
            hipLaunchKernelGGL(mb_level0_deferred_kernel, dim3(1024), dim3(256), 0, st, KT);
        }
        static const bool stats = getenv("STITCHING_AMD_DEFER_STATS") != nullptr;  // diagnostic: how many patches took the second pass
        if (stats) {
            std::vector<unsigned> hc(counters / 4);
            hipStreamSynchronize(st);
            hipMemcpy(hc.data(), q, counters, hipMemcpyDeviceToHost);
            unsigned long long tot = 0; unsigned mx = 0;
            for (int sgi = 0; sgi < segs; sgi++) { tot += hc[32 * sgi]; mx = std::max(mx, hc[32 * sgi]); }
            fprintf(stderr, "[stitching_amd] level-0 deferral: %llu of %llu patches queued (%d segments, fullest %u of %zu)\n", tot,
                    (unsigned long long)KT.tiles.tiles_x * KT.tiles.tiles_y * (unsigned long long)LV_THREADS, segs, mx, seg_cap);
        }
        stx_dev_free(ctx, q);
    } else if (K.emit) {
        if (K.level == 0) hipLaunchKernelGGL((mb_level_fast_kernel<true, false, true, false>), grid, dim3(LV_THREADS), 0, st, KT);
        else if (K.all_u8) hipLaunchKernelGGL((mb_level_fast_kernel<false, false, true, true>), grid, dim3(LV_THREADS), 0, st, KT);
        else hipLaunchKernelGGL((mb_level_fast_kernel<false, false, true, false>), grid, dim3(LV_THREADS), 0, st, KT);
    } else if (K.has_contrib) {
        if (K.level == 0) hipLaunchKernelGGL((mb_level_fast_kernel<true, true, false, false>), grid, dim3(LV_THREADS), 0, st, KT);
        else if (K.all_u8) hipLaunchKernelGGL((mb_level_fast_kernel<false, true, false, true>), grid, dim3(LV_THREADS), 0, st, KT);
        else hipLaunchKernelGGL((mb_level_fast_kernel<false, true, false, false>), grid, dim3(LV_THREADS), 0, st, KT);
    } else {
        static const bool no_pk_levels = getenv("STITCHING_AMD_NO_PK_LEVELS") != nullptr;  // diagnostic: A/B against mb_level_fast_kernel
        // (4 waves per SIMD at 121 registers; forced to 5 it spills 19 of them: 309 against 203 us on the resized-seam-mask leg)
        if (K.level == 0 && K.all_u8 && K.num_bands > 0) hipLaunchKernelGGL((mb_level_fast_kernel<true, false, false, true>), grid, dim3(LV_THREADS), 0, st, KT);
        else if (K.level == 0) hipLaunchKernelGGL((mb_level_fast_kernel<true, false, false, false>), grid, dim3(LV_THREADS), 0, st, KT);
        else if (K.all_u8 && K.level < K.num_bands && !no_pk_levels) hipLaunchKernelGGL(mb_level_pk_kernel, grid, dim3(LV_THREADS), 0, st, KT);
        else if (K.all_u8) hipLaunchKernelGGL((mb_level_fast_kernel<false, false, false, true>), grid, dim3(LV_THREADS), 0, st, KT);
        else hipLaunchKernelGGL((mb_level_fast_kernel<false, false, false, false>), grid, dim3(LV_THREADS), 0, st, KT);
    }
    return launched_ok();
}
