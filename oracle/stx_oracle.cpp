// stx_oracle.cpp — CPU ORACLE for the warp + blend hot path.  TEST INFRASTRUCTURE ONLY.
//
// This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load it.  The product (stitching_amd/) never links,
// imports or calls anything in oracle/.
//
// WHAT IT RESTATES
//   The reference (OpenStitching/stitching) has no arithmetic of its own on this path; it
//   calls OpenCV (opencv-python, requirements.txt:1 pins 5.0.0.93, setup.cfg:21 allows
//   >=4.0.1,<6) at
//     stitching/warper.py:44-51   cv.PyRotationWarper(type, scale).warp(img,K,R,INTER_LINEAR,BORDER_REFLECT)
//     stitching/warper.py:59-67   ... .warp(mask255,K,R,INTER_NEAREST,BORDER_CONSTANT)
//     stitching/warper.py:80-82   ... .warpRoi(size,K,R)
//     stitching/blender.py:24     cv.detail.resultRoi
//     stitching/blender.py:28     cv.detail.Blender_createDefault(Blender_NO)
//     stitching/blender.py:31-32  cv.detail_MultiBandBlender().setNumBands
//     stitching/blender.py:35-36  cv.detail_FeatherBlender().setSharpness
//     stitching/blender.py:38,41  blender.prepare / blender.feed(int16 image, mask, corner)
//     stitching/blender.py:46-47  blender.blend ; cv.convertScaleAbs
//   OpenCV is a third-party dependency that is ABSENT from /root/reference and from this
//   image (no cv2, no headers, no network).  This file therefore restates OpenCV's
//   published algorithm (modules/stitching: warpers_inl.hpp, warpers.cpp, blenders.cpp,
//   util.cpp; modules/imgproc: imgwarp.cpp remap, pyramids.cpp, distransform.cpp;
//   modules/core: copy.cpp borderInterpolate, matrix inv/gemm for 3x3) from memory of the
//   4.x sources.  Upstream function names are cited at each function; line numbers cannot be.
//
// PARITY UNPINNED.  The reference's own tests hold no golden vector, known-answer test or
// fixture on this path (tests/test_stitcher.py:229-231 checks panorama *shapes* only), and no
// OpenCV binary exists here to generate any.  What pins this oracle instead: closed-form
// known-answer tests and seeded SHA-256 goldens of its own output (tests/, tests/golden/).
//
// Warpers.  All sixteen names cv.PyRotationWarper accepts (stitching/warper.py:10-27): plane, affine,
// cylindrical, spherical with their detectResultRoi overrides, and the twelve RotationWarperBase<P>
// warpers (fisheye, stereographic, compressedPlane*, panini*, mercator, transverseMercator) with the
// generic every-pixel detectResultRoi; (a, b) as PyRotationWarper's constructor fixes them.
//
// Trig modes.  OpenCV calls libm sinf/cosf/atan2f/acosf (and tanf/asinf/atanf/logf/sinhf/coshf in the
// twelve other projectors).  trig=0 ("libm") does the same (literal restatement).  trig=1 ("exact")
// uses the double-precision polynomial routines below and rounds once to fp32; it is what the HIP
// kernels reproduce bit-for-bit, and tests/ measure its drift from trig=0 (<= 1 ULP fp32 per call).
//
// Build: see oracle/Makefile (g++ -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp).
// -ffp-contract=off matters: OpenCV's baseline x86-64 build has no FMA contraction.

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API extern "C" __attribute__((visibility("default")))

namespace {

int g_threads = 1;

// Alternative arithmetic models of the two places where OpenCV builds differ from one another (sensitivity studies,
// tests/test_oracle_models.py, tools/oracle_sensitivity.py).  Defaults = the literal scalar / classic restatement.
//   g_pyr32f: evaluation order of pyrDown on CV_32F (the blend weights).  bit 0: vertical pass in the universal-intrinsic
//             order  (r1 + r3 + r2) * 4 + (r0 + r4 + (r2 + r2))  (PyrDownVecV<float,float>; the SSE2 code of 3.x / 4.0 has the
//             same order); bit 1: horizontal pass  r2 * 6 + ((r1 + r3) * 4 + (r0 + r4))  (PyrDownVecH<float,float,1>, 4.2+);
//             bit 2: v_muladd is fused (FMA3 / NEON builds).  Vector groups of g_pyr32f_lanes outputs start at x = 1
//             (horizontal; x = 0 is the left-border column) / x = 0 (vertical); the leftover columns and the border
//             columns use the scalar order.  [OCV-MEM]
//   g_remap : 0 = classic fixed-point remap (1/32-px positions, Q15 table); 1 = fp32 bilinear on the unquantised
//             position, the shape of the rewritten 5.x kernels (floor, alpha/beta in fp32, p + a * (q - p), round to
//             nearest even): an UNVERIFIED model, for sensitivity only; bit 1: the three lerps are fused (fma).
int g_pyr32f = 0, g_pyr32f_lanes = 4, g_remap = 0;
inline float muladd_model(float a, float b, float c, bool fused) { return fused ? __builtin_fmaf(a, b, c) : a * b + c; }

// ------------------------------------------------------------------------------------------
// "exact" trig: double precision, explicit fma, then one rounding to fp32.
// Coefficients are the classic fdlibm minimax sets (k_sin.c, k_cos.c, s_atan.c).
// ------------------------------------------------------------------------------------------
inline double dfma(double a, double b, double c) { return __builtin_fma(a, b, c); }

void sincos_d(double x, double* s, double* c)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double PIO2_1T = 6.07710050650619224932e-11;  // pi/2 - PIO2_1
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    if (!(std::fabs(x) < 1.0e6)) {  // far outside any panorama (NaN/Inf land here too): defined as NaN
        *s = *c = std::numeric_limits<double>::quiet_NaN();
        return;
    }
    double kd = std::nearbyint(x * INV_PIO2);
    double r = dfma(-kd, PIO2_1, x);
    r = dfma(-kd, PIO2_1T, r);
    double z = r * r;
    double ps = dfma(z, S6, S5);
    ps = dfma(z, ps, S4);
    ps = dfma(z, ps, S3);
    ps = dfma(z, ps, S2);
    ps = dfma(z, ps, S1);
    double sr = dfma(r * z, ps, r);
    double pc = dfma(z, C6, C5);
    pc = dfma(z, pc, C4);
    pc = dfma(z, pc, C3);
    pc = dfma(z, pc, C2);
    pc = dfma(z, pc, C1);
    double cr = dfma(z * z, pc, dfma(-0.5, z, 1.0));
    long long k = (long long)kd;
    switch (k & 3) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
    }
}

double atan_d(double x)
{
    static const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                                     9.82793723247329054082e-01, 1.57079632679489655800e+00};
    static const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                                     1.39033110312309984516e-17, 6.12323399573676603587e-17};
    static const double aT[11] = {3.33333333333329318027e-01,  -1.99999999998764832476e-01,
                                  1.42857142725034663711e-01,  -1.11111104054623557880e-01,
                                  9.09088713343650656196e-02,  -7.69187620504482999495e-02,
                                  6.66107313738753120669e-02,  -5.83357013379057348645e-02,
                                  4.97687799461593236017e-02,  -3.65315727442169155270e-02,
                                  1.62858201153657823623e-02};
    if (x != x) return x;
    bool neg = std::signbit(x);
    double a = std::fabs(x);
    int id;
    if (a < 0.4375) {
        id = -1;
    } else if (a < 1.1875) {
        if (a < 0.6875) { id = 0; a = (2.0 * a - 1.0) / (2.0 + a); }
        else { id = 1; a = (a - 1.0) / (a + 1.0); }
    } else if (a < 2.4375) {
        id = 2; a = (a - 1.5) / (1.0 + 1.5 * a);
    } else {
        id = 3; a = -1.0 / a;
    }
    double z = a * a;
    double w = z * z;
    double s1 = z * dfma(w, dfma(w, dfma(w, dfma(w, dfma(w, aT[10], aT[8]), aT[6]), aT[4]), aT[2]), aT[0]);
    double s2 = w * dfma(w, dfma(w, dfma(w, dfma(w, aT[9], aT[7]), aT[5]), aT[3]), aT[1]);
    double r;
    if (id < 0) r = a - a * (s1 + s2);
    else r = atanhi[id] - ((a * (s1 + s2) - atanlo[id]) - a);
    return neg ? -r : r;
}

double atan2_d(double y, double x)
{
    const double PI = 3.1415926535897931160E+00, PI_LO = 1.2246467991473531772E-16;
    if (x != x || y != y) return x + y;
    int m = (std::signbit(y) ? 1 : 0) | (std::signbit(x) ? 2 : 0);
    if (y == 0.0) {
        switch (m) {
        case 0: case 1: return y;
        case 2: return PI;
        default: return -PI;
        }
    }
    if (x == 0.0) return (m & 1) ? -PI / 2 : PI / 2;
    double z = atan_d(std::fabs(y / x));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return PI - (z - PI_LO);
    default: return (z - PI_LO) - PI;
    }
}

double acos_d(double w)
{
    // acos(w) = atan2(sqrt((1-w)(1+w)), w); NaN for |w| > 1 like libm
    double t = (1.0 - w) * (1.0 + w);
    return atan2_d(std::sqrt(t), w);
}

// The projectors beyond plane / cylindrical / spherical also call tanf, asinf, atanf, logf, sinhf, coshf.
double tan_d(double x)
{
    double s, c;
    sincos_d(x, &s, &c);
    return s / c;
}

double asin_d(double w)
{
    // asin(w) = atan2(w, sqrt((1-w)(1+w))); NaN for |w| > 1 like libm
    double t = (1.0 - w) * (1.0 + w);
    return atan2_d(w, std::sqrt(t));
}

inline uint64_t d2u(double v) { uint64_t u; std::memcpy(&u, &v, 8); return u; }
inline double u2d(uint64_t u) { double v; std::memcpy(&v, &u, 8); return v; }

// fdlibm e_log.c: x = 2^k (1 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)), s = f / (2 + f)
double log_d(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x < 0.0) return std::numeric_limits<double>::quiet_NaN();
    if (x == 0.0) return -std::numeric_limits<double>::infinity();
    if (x == std::numeric_limits<double>::infinity()) return x;
    int k = 0;
    if (x < 2.2250738585072014e-308) {
        x = x * 18014398509481984.0;  // 2^54
        k = -54;
    }
    uint64_t bits = d2u(x);
    int hx = (int)(bits >> 32);
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;  // mantissa >= sqrt(2): halve it, bump k
    bits = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (bits & 0xffffffffull);
    k += i >> 20;
    const double f = u2d(bits) - 1.0;
    const double dk = (double)k;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    return dk * LN2_HI - ((hfsq - (s * (hfsq + R) + dk * LN2_LO)) - f);
}

// fdlibm e_exp.c: x = k ln2 + r, exp(r) = 1 + r + r c / (2 - c)
double exp_d(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10,
                 INV_LN2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return std::numeric_limits<double>::infinity();
    if (x < -7.45133219101941108420e+02) return 0.0;
    double hi = x, lo = 0.0;
    int k = 0;
    if (std::fabs(x) > 0.34657359027997264) {  // 0.5 ln2
        k = (int)(INV_LN2 * x + (x < 0.0 ? -0.5 : 0.5));
        const double t = (double)k;
        hi = x - t * LN2_HI;
        lo = t * LN2_LO;
    }
    const double r = hi - lo;
    const double t = r * r;
    const double c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    const int k1 = k / 2, k2 = k - k1;  // 2^k in two normal factors
    return (y * u2d((uint64_t)(k1 + 1023) << 52)) * u2d((uint64_t)(k2 + 1023) << 52);
}

double sinh_d(double x)
{
    if (x != x) return x;
    const double a = std::fabs(x);
    double r;
    if (a < 0.03125) {  // odd Taylor series: the difference of exponentials would cancel
        const double z = a * a;
        const double p = z * (1.66666666666666657415e-01 + z * (8.33333333333333321769e-03 +
                         z * (1.98412698412698412526e-04 + z * 2.75573192239858925110e-06)));
        r = a + a * p;
    } else {
        const double e = exp_d(a);
        r = 0.5 * (e - 1.0 / e);
    }
    return x < 0.0 ? -r : r;
}

double cosh_d(double x)
{
    if (x != x) return x;
    const double e = exp_d(std::fabs(x));
    return 0.5 * (e + 1.0 / e);
}


// ------------------------------------------------------------------------------------------
// trig = 2 / 3 ("glibc" / "glibc-nofma"): glibc >= 2.28 sinf / cosf, restated.  [from memory of glibc's
// sysdeps/ieee754/flt-32/{s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c} — Szabolcs Nagy's routines from ARM's optimized-routines;
// pinned HERE against the host's own libm over >= 10^8 arguments: tests/test_glibc_trig.py, tools/check_glibc_trig.py]
//   double x = y;  |y| < pi/4: the polynomial directly (|y| < 2^-12: sin = y, cos = 1);
//   |y| < 120: reduce_fast — r = x * (2/pi * 2^24), n = ((int32)r + 0x800000) >> 24 (the x86-64 build has no TOINT_INTRINSICS),
//              x -= n * pi/2 (one double: exact enough up to 120), sign table, 2nd coefficient set (negated cosine) for n & 2;
//   else     : reduce_large — 32 x 96 -> 128-bit fixed-point product with 4/pi to 192 bits.
// The polynomial is sinf_poly (sine: x + x^3 s1 + x^7 (s2 + x^2 s3); cosine: (c0 + x^2 c1) + x^4 c2 + x^6 (c3 + x^2 c4)), evaluated in
// double and rounded once to float.  x86-64 glibc ships two builds behind an ifunc: __sinf_sse2 (every operation rounded:
// mode 3) and __sinf_fma (compiled -mfma with GCC's default -ffp-contract=fast: every a * b whose only uses are additions /
// subtractions fuses — mode 2, what any x86-64-v3 host runs).  The two differ on about one argument in 10^8.
// ------------------------------------------------------------------------------------------
struct SincosTab { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
const SincosTab g_sincosf_table[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5,
     -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5,
     0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
const uint32_t g_inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
inline uint32_t f2u(float v) { uint32_t u; std::memcpy(&u, &v, 4); return u; }
inline uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ff; }
// a * b + c as the build in question evaluates it
inline double gl_mad(double a, double b, double c, bool fma) { return fma ? __builtin_fma(a, b, c) : a * b + c; }

inline float gl_sinf_poly(double x, double x2, const SincosTab& p, int n, bool fma)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = gl_mad(x2, p.s3, p.s2, fma);
        const double x7 = x3 * x2;
        const double s = gl_mad(x3, p.s1, x, fma);
        return (float)gl_mad(x7, s1, s, fma);
    }
    const double x4 = x2 * x2;
    const double c2 = gl_mad(x2, p.c4, p.c3, fma);
    const double c1 = gl_mad(x2, p.c1, p.c0, fma);
    const double x6 = x4 * x2;
    const double c = gl_mad(x4, p.c2, c1, fma);
    return (float)gl_mad(x6, c2, c, fma);
}
inline double gl_reduce_fast(double x, const SincosTab& p, int* np, bool fma)
{
    const double r = x * p.hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return gl_mad(-(double)n, p.hpi, x, fma);  // x - n * hpi
}
inline double gl_reduce_large(uint32_t xi, int* np)
{
    const uint32_t* arr = &g_inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = xi * arr[0];  // 32-bit product, as in the source
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    const double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;  // pi63
}
float glibc_sincosf1(float y, bool want_cos, bool fma)
{
    double x = y, s;
    int n;
    const SincosTab* p = &g_sincosf_table[0];
    const int flip = want_cos ? 1 : 0;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return want_cos ? 1.0f : y;
        return gl_sinf_poly(x, x * x, *p, flip, fma);
    }
    if (abstop12(y) < abstop12(120.0f)) {
        x = gl_reduce_fast(x, *p, &n, fma);
        s = p->sign[n & 3];
        if (n & 2) p = &g_sincosf_table[1];
        return gl_sinf_poly(x * s, x * x, *p, n ^ flip, fma);
    }
    if (abstop12(y) < abstop12(std::numeric_limits<float>::infinity())) {
        const uint32_t xi = f2u(y);
        const int sign = (int)(xi >> 31);
        x = gl_reduce_large(xi, &n);
        s = p->sign[(n + sign) & 3];
        if ((n + sign) & 2) p = &g_sincosf_table[1];
        return gl_sinf_poly(x * s, x * x, *p, n ^ flip, fma);
    }
    return std::numeric_limits<float>::quiet_NaN();  // __math_invalidf
}

// mode 0: the host's libm for everything; 1: the correctly rounded fp64 routines above; 2 / 3: sinf / cosf as glibc >= 2.28
// computes them (FMA build / SSE2 build), everything else as mode 1
struct Trig {
    int mode;
    float sin_(float x) const
    {
        if (!mode) return sinf(x);
        if (mode >= 2) return glibc_sincosf1(x, false, mode == 2);
        double s, c; sincos_d((double)x, &s, &c); return (float)s;
    }
    float cos_(float x) const
    {
        if (!mode) return cosf(x);
        if (mode >= 2) return glibc_sincosf1(x, true, mode == 2);
        double s, c; sincos_d((double)x, &s, &c); return (float)c;
    }
    float atan2_(float y, float x) const { return mode ? (float)atan2_d((double)y, (double)x) : atan2f(y, x); }
    float acos_(float w) const { return mode ? (float)acos_d((double)w) : acosf(w); }
    float tan_(float x) const { return mode ? (float)tan_d((double)x) : tanf(x); }
    float asin_(float w) const { return mode ? (float)asin_d((double)w) : asinf(w); }
    float atan_(float x) const { return mode ? (float)atan_d((double)x) : atanf(x); }
    float log_(float x) const { return mode ? (float)log_d((double)x) : logf(x); }
    float sinh_(float x) const { return mode ? (float)sinh_d((double)x) : sinhf(x); }
    float cosh_(float x) const { return mode ? (float)cosh_d((double)x) : coshf(x); }
};

// ------------------------------------------------------------------------------------------
// Small helpers restating OpenCV core semantics
// ------------------------------------------------------------------------------------------
// cvRound(float) on x86-64 = _mm_cvtss_si32: round-half-even; out of range / NaN -> INT_MIN
inline int cv_round(float v)
{
    if (!(v >= -2147483648.f && v < 2147483648.f)) return INT_MIN;
    return (int)std::nearbyintf(v);
}
inline short sat_s16(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
// static_cast<short>(float) as x86 compiles it: cvttss2si (trunc; INT_MIN when out of
// int32 range) then low 16 bits.
inline short trunc_s16(float v)
{
    int i;
    if (!(v >= -2147483648.f && v < 2147483648.f)) i = INT_MIN;
    else i = (int)v;
    return (short)(unsigned short)(unsigned)i;
}

enum { B_CONSTANT = 0, B_REPLICATE = 1, B_REFLECT = 2, B_REFLECT_101 = 4 };

// cv::borderInterpolate (modules/core/src/copy.cpp)
int border_interpolate(int p, int len, int type)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (type == B_REPLICATE) return p < 0 ? 0 : len - 1;
    if (type == B_REFLECT || type == B_REFLECT_101) {
        int delta = type == B_REFLECT_101;
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    return -1;  // B_CONSTANT
}

// ------------------------------------------------------------------------------------------
// Projectors (modules/stitching/include/opencv2/stitching/detail/warpers_inl.hpp,
// modules/stitching/src/warpers.cpp)
// ------------------------------------------------------------------------------------------
// warper ids = the 16 names cv.PyRotationWarper accepts (stitching/warper.py:11-28), in that order of appearance in
// OpenCV's PyRotationWarper constructor; `family` + (a, b) is what the projector templates are instantiated with
enum { W_PLANE = 0, W_AFFINE = 1, W_CYLINDRICAL = 2, W_SPHERICAL = 3, W_FISHEYE = 4, W_STEREOGRAPHIC = 5,
       W_CPLANE_A2B1 = 6, W_CPLANE_A15B1 = 7, W_CPLANE_PORTRAIT_A2B1 = 8, W_CPLANE_PORTRAIT_A15B1 = 9,
       W_PANINI_A2B1 = 10, W_PANINI_A15B1 = 11, W_PANINI_PORTRAIT_A2B1 = 12, W_PANINI_PORTRAIT_A15B1 = 13,
       W_MERCATOR = 14, W_TRANSVERSE_MERCATOR = 15, W_COUNT = 16 };
enum { F_PLANE = 0, F_CYLINDRICAL, F_SPHERICAL, F_FISHEYE, F_STEREOGRAPHIC, F_CRECT, F_CRECT_PORTRAIT, F_PANINI,
       F_PANINI_PORTRAIT, F_MERCATOR, F_TRANSVERSE_MERCATOR };

struct Projector {
    int type;    // W_*
    int family;  // F_*
    float scale;
    float a, b;  // compressed-rectilinear / panini parameters
    float k[9], rinv[9], r_kinv[9], k_rinv[9], t[3];
    Trig tr;
};

// Mat::inv() for 3x3 CV_32F: closed form in double, cast to float (cv::invert, n == 3)
bool inv3x3(const float* m, float* o)
{
#define M(i, j) m[(i) * 3 + (j)]
    double d = M(0, 0) * ((double)M(1, 1) * M(2, 2) - (double)M(1, 2) * M(2, 1)) -
               M(0, 1) * ((double)M(1, 0) * M(2, 2) - (double)M(1, 2) * M(2, 0)) +
               M(0, 2) * ((double)M(1, 0) * M(2, 1) - (double)M(1, 1) * M(2, 0));
    if (d == 0.) { for (int i = 0; i < 9; i++) o[i] = 0.f; return false; }
    d = 1. / d;
    double t[9];
    t[0] = (((double)M(1, 1) * M(2, 2) - (double)M(1, 2) * M(2, 1)) * d);
    t[1] = (((double)M(0, 2) * M(2, 1) - (double)M(0, 1) * M(2, 2)) * d);
    t[2] = (((double)M(0, 1) * M(1, 2) - (double)M(0, 2) * M(1, 1)) * d);
    t[3] = (((double)M(1, 2) * M(2, 0) - (double)M(1, 0) * M(2, 2)) * d);
    t[4] = (((double)M(0, 0) * M(2, 2) - (double)M(0, 2) * M(2, 0)) * d);
    t[5] = (((double)M(0, 2) * M(1, 0) - (double)M(0, 0) * M(1, 2)) * d);
    t[6] = (((double)M(1, 0) * M(2, 1) - (double)M(1, 1) * M(2, 0)) * d);
    t[7] = (((double)M(0, 1) * M(2, 0) - (double)M(0, 0) * M(2, 1)) * d);
    t[8] = (((double)M(0, 0) * M(1, 1) - (double)M(0, 1) * M(1, 0)) * d);
#undef M
    for (int i = 0; i < 9; i++) o[i] = (float)t[i];
    return true;
}

// 3x3 * 3x3 CV_32F product as cv::gemm's small-matrix path: float dot products, left to right
void mul3x3(const float* a, const float* b, float* d)
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            float t = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
            d[i * 3 + j] = t;
        }
}

// ProjectorBase::setCameraParams(K, R, T)
void set_camera_params(Projector& p, const float* K, const float* R, const float* T)
{
    for (int i = 0; i < 9; i++) p.k[i] = K[i];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) p.rinv[i * 3 + j] = R[j * 3 + i];
    float kinv[9];
    inv3x3(K, kinv);
    mul3x3(R, kinv, p.r_kinv);
    mul3x3(K, p.rinv, p.k_rinv);
    p.t[0] = T[0]; p.t[1] = T[1]; p.t[2] = T[2];
}

// AffineWarper::getRTfromHomogeneous; cv::AffineWarper::create(scale) = makePtr<detail::AffineWarper>(scale) and
// detail::AffineWarper(float scale = 1.f) : PlaneWarper(scale): the caller's scale is kept
void make_projector(Projector& p, int type, float scale, const float* K, const float* Rin, int trig)
{
    p.type = type;
    p.scale = scale;
    p.a = p.b = 1.f;
    p.tr.mode = trig;
    switch (type) {
    case W_PLANE: case W_AFFINE: p.family = F_PLANE; break;
    case W_CYLINDRICAL: p.family = F_CYLINDRICAL; break;
    case W_SPHERICAL: p.family = F_SPHERICAL; break;
    case W_FISHEYE: p.family = F_FISHEYE; break;
    case W_STEREOGRAPHIC: p.family = F_STEREOGRAPHIC; break;
    // PyRotationWarper: "compressedPlaneA2B1" -> CompressedRectilinearWarper(2.0f, 1.0f), "...A1.5B1" -> (1.5f, 1.0f), etc.
    case W_CPLANE_A2B1: p.family = F_CRECT; p.a = 2.0f; break;
    case W_CPLANE_A15B1: p.family = F_CRECT; p.a = 1.5f; break;
    case W_CPLANE_PORTRAIT_A2B1: p.family = F_CRECT_PORTRAIT; p.a = 2.0f; break;
    case W_CPLANE_PORTRAIT_A15B1: p.family = F_CRECT_PORTRAIT; p.a = 1.5f; break;
    case W_PANINI_A2B1: p.family = F_PANINI; p.a = 2.0f; break;
    case W_PANINI_A15B1: p.family = F_PANINI; p.a = 1.5f; break;
    case W_PANINI_PORTRAIT_A2B1: p.family = F_PANINI_PORTRAIT; p.a = 2.0f; break;
    case W_PANINI_PORTRAIT_A15B1: p.family = F_PANINI_PORTRAIT; p.a = 1.5f; break;
    case W_MERCATOR: p.family = F_MERCATOR; break;
    default: p.family = F_TRANSVERSE_MERCATOR; break;
    }
    float T[3] = {0.f, 0.f, 0.f};
    if (type == W_AFFINE) {
        // R = H with H[0,2]=H[1,2]=0, transposed; T = -(R * (H[0,2], H[1,2], 0))
        float H[9];
        for (int i = 0; i < 9; i++) H[i] = Rin[i];
        float t0 = H[2], t1 = H[5];
        H[2] = 0.f; H[5] = 0.f;
        float Rt[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rt[i * 3 + j] = H[j * 3 + i];
        for (int i = 0; i < 3; i++) {
            float v = Rt[i * 3 + 0] * t0 + Rt[i * 3 + 1] * t1 + Rt[i * 3 + 2] * 0.f;
            T[i] = v * -1.f;
        }
        set_camera_params(p, K, Rt, T);
        return;
    }
    set_camera_params(p, K, Rin, T);
}

const float PI_F = static_cast<float>(3.14159265358979323846);

void map_forward(const Projector& p, float x, float y, float& u, float& v)
{
    const float* rk = p.r_kinv;
    const Trig& T = p.tr;
    float x_ = rk[0] * x + rk[1] * y + rk[2];
    float y_ = rk[3] * x + rk[4] * y + rk[5];
    float z_ = rk[6] * x + rk[7] * y + rk[8];
    if (p.family == F_CRECT_PORTRAIT || p.family == F_PANINI_PORTRAIT) {  // the portrait projectors swap the roles of rows 0 and 1
        float t = x_; x_ = y_; y_ = t;
    }
    switch (p.family) {
    case F_PLANE:
        x_ = p.t[0] + x_ / z_ * (1 - p.t[2]);
        y_ = p.t[1] + y_ / z_ * (1 - p.t[2]);
        u = p.scale * x_;
        v = p.scale * y_;
        break;
    case F_CYLINDRICAL:
        u = p.scale * T.atan2_(x_, z_);
        v = p.scale * y_ / sqrtf(x_ * x_ + z_ * z_);
        break;
    case F_SPHERICAL: {
        u = p.scale * T.atan2_(x_, z_);
        float w = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
        v = p.scale * (PI_F - T.acos_(w == w ? w : 0));
        break;
    }
    case F_FISHEYE: {  // FisheyeProjector::mapForward
        float u_ = T.atan2_(x_, z_);
        float v_ = PI_F - T.acos_(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        u = p.scale * v_ * T.cos_(u_);
        v = p.scale * v_ * T.sin_(u_);
        break;
    }
    case F_STEREOGRAPHIC: {  // StereographicProjector::mapForward
        float u_ = T.atan2_(x_, z_);
        float v_ = PI_F - T.acos_(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float r = T.sin_(v_) / (1 - T.cos_(v_));
        u = p.scale * r * T.cos_(u_);
        v = p.scale * r * T.sin_(u_);
        break;
    }
    case F_CRECT:            // CompressedRectilinearProjector::mapForward
    case F_CRECT_PORTRAIT: {  // CompressedRectilinearPortraitProjector::mapForward (u negated)
        float u_ = T.atan2_(x_, z_);
        float v_ = T.asin_(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float s = p.family == F_CRECT ? p.scale : -p.scale;
        u = s * p.a * T.tan_(u_ / p.a);
        v = p.scale * p.b * T.tan_(v_) / T.cos_(u_);
        break;
    }
    case F_PANINI:            // PaniniProjector::mapForward
    case F_PANINI_PORTRAIT: {  // PaniniPortraitProjector::mapForward (u negated)
        float u_ = T.atan2_(x_, z_);
        float v_ = T.asin_(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float tg = p.a * T.tan_(u_ / p.a);
        float s = p.family == F_PANINI ? p.scale : -p.scale;
        u = s * tg;
        float sinu = T.sin_(u_);
        if (std::fabs((double)sinu) < 1E-7) v = p.scale * p.b * T.tan_(v_);
        else v = p.scale * p.b * tg * T.tan_(v_) / sinu;
        break;
    }
    case F_MERCATOR: {  // MercatorProjector::mapForward
        float u_ = T.atan2_(x_, z_);
        float v_ = T.asin_(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        u = p.scale * u_;
        v = p.scale * T.log_(T.tan_((float)(3.14159265358979323846 / 4) + v_ / 2));
        break;
    }
    case F_TRANSVERSE_MERCATOR: {  // TransverseMercatorProjector::mapForward
        float u_ = T.atan2_(x_, z_);
        float v_ = T.asin_(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float B = T.cos_(v_) * T.sin_(u_);
        u = p.scale / 2 * T.log_((1 + B) / (1 - B));
        v = p.scale * T.atan2_(T.tan_(v_), T.cos_(u_));
        break;
    }
    default:
        u = v = std::numeric_limits<float>::quiet_NaN();
    }
}

// Row/column-separable parts of mapBackward are kept unseparated here on purpose (literal).
void map_backward(const Projector& p, float u, float v, float& x, float& y)
{
    const float* kr = p.k_rinv;
    const Trig& T = p.tr;
    float x_, y_, z_, z;
    switch (p.family) {
    case F_PLANE:
        u = u / p.scale - p.t[0];
        v = v / p.scale - p.t[1];
        x = kr[0] * u + kr[1] * v + kr[2] * (1 - p.t[2]);
        y = kr[3] * u + kr[4] * v + kr[5] * (1 - p.t[2]);
        z = kr[6] * u + kr[7] * v + kr[8] * (1 - p.t[2]);
        x /= z;
        y /= z;
        return;
    case F_CYLINDRICAL:
        u /= p.scale;
        v /= p.scale;
        x_ = T.sin_(u);
        y_ = v;
        z_ = T.cos_(u);
        break;
    case F_SPHERICAL: {
        u /= p.scale;
        v /= p.scale;
        float sinv = T.sin_(PI_F - v);
        x_ = sinv * T.sin_(u);
        y_ = T.cos_(PI_F - v);
        z_ = sinv * T.cos_(u);
        break;
    }
    case F_FISHEYE: {  // FisheyeProjector::mapBackward
        u /= p.scale;
        v /= p.scale;
        float u_ = T.atan2_(v, u);
        float v_ = sqrtf(u * u + v * v);
        float sinv = T.sin_(PI_F - v_);
        x_ = sinv * T.sin_(u_);
        y_ = T.cos_(PI_F - v_);
        z_ = sinv * T.cos_(u_);
        break;
    }
    case F_STEREOGRAPHIC: {  // StereographicProjector::mapBackward
        u /= p.scale;
        v /= p.scale;
        float u_ = T.atan2_(v, u);
        float r = sqrtf(u * u + v * v);
        float v_ = 2 * T.atan_(1.f / r);
        float sinv = T.sin_(PI_F - v_);
        x_ = sinv * T.sin_(u_);
        y_ = T.cos_(PI_F - v_);
        z_ = sinv * T.cos_(u_);
        break;
    }
    case F_CRECT:
    case F_CRECT_PORTRAIT: {  // CompressedRectilinear[Portrait]Projector::mapBackward
        u /= (p.family == F_CRECT ? p.scale : -p.scale);
        v /= p.scale;
        float aatg = p.a * T.atan_(u / p.a);
        float u_ = aatg;
        float v_ = T.atan_(v * T.cos_(aatg) / p.b);
        float cosv = T.cos_(v_);
        x_ = cosv * T.sin_(u_);
        y_ = T.sin_(v_);
        z_ = cosv * T.cos_(u_);
        break;
    }
    case F_PANINI:
    case F_PANINI_PORTRAIT: {  // Panini[Portrait]Projector::mapBackward
        u /= (p.family == F_PANINI ? p.scale : -p.scale);
        v /= p.scale;
        float lamda = p.a * T.atan_(u / p.a);
        float u_ = lamda;
        float v_;
        if (lamda == lamda) v_ = T.atan_(v * T.sin_(lamda) / (p.b * p.a * T.tan_(lamda / p.a)));
        else v_ = 0.f;
        float cosv = T.cos_(v_);
        x_ = cosv * T.sin_(u_);
        y_ = T.sin_(v_);
        z_ = cosv * T.cos_(u_);
        break;
    }
    case F_MERCATOR: {  // MercatorProjector::mapBackward
        u /= p.scale;
        v /= p.scale;
        float v_ = T.atan_(T.sinh_(v));
        float u_ = u;
        float cosv = T.cos_(v_);
        x_ = cosv * T.sin_(u_);
        y_ = T.sin_(v_);
        z_ = cosv * T.cos_(u_);
        break;
    }
    case F_TRANSVERSE_MERCATOR: {  // TransverseMercatorProjector::mapBackward
        u /= p.scale;
        v /= p.scale;
        float v_ = T.asin_(T.sin_(v) / T.cosh_(u));
        float u_ = T.atan2_(T.sinh_(u), T.cos_(v));
        float cosv = T.cos_(v_);
        x_ = cosv * T.sin_(u_);
        y_ = T.sin_(v_);
        z_ = cosv * T.cos_(u_);
        break;
    }
    default:
        x = y = -1;
        return;
    }
    if (p.family == F_CRECT_PORTRAIT || p.family == F_PANINI_PORTRAIT) {  // portrait: y_ = cosv sin u_, x_ = sin v_
        float t = x_; x_ = y_; y_ = t;
    }
    x = kr[0] * x_ + kr[1] * y_ + kr[2] * z_;
    y = kr[3] * x_ + kr[4] * y_ + kr[5] * z_;
    z = kr[6] * x_ + kr[7] * y_ + kr[8] * z_;
    if (z > 0) { x /= z; y /= z; }
    else x = y = -1;
}

struct MinMax {
    float tl_u = (std::numeric_limits<float>::max)(), tl_v = (std::numeric_limits<float>::max)();
    float br_u = -(std::numeric_limits<float>::max)(), br_v = -(std::numeric_limits<float>::max)();
    void add(float u, float v)
    {
        tl_u = (std::min)(tl_u, u); tl_v = (std::min)(tl_v, v);
        br_u = (std::max)(br_u, u); br_v = (std::max)(br_v, v);
    }
};

// RotationWarperBase::detectResultRoiByBorder / SphericalWarper::detectResultRoi /
// PlaneWarper::detectResultRoi.  tl/br inclusive, (int) truncation.
void detect_result_roi(const Projector& p, int w, int h, int* tl, int* br)
{
    MinMax mm;
    float u, v;
    if (p.family > F_SPHERICAL) {
        // RotationWarperBase::detectResultRoi: every source pixel (the warpers without an override)
        // (rows folded per thread: min / max over the non-NaN values does not depend on the order)
        std::vector<MinMax> rows((size_t)h);
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (int y = 0; y < h; ++y) {
            float uu, vv;
            for (int x = 0; x < w; ++x) {
                map_forward(p, (float)x, (float)y, uu, vv);
                rows[y].add(uu, vv);
            }
        }
        for (int y = 0; y < h; ++y) {
            mm.tl_u = (std::min)(mm.tl_u, rows[y].tl_u); mm.tl_v = (std::min)(mm.tl_v, rows[y].tl_v);
            mm.br_u = (std::max)(mm.br_u, rows[y].br_u); mm.br_v = (std::max)(mm.br_v, rows[y].br_v);
        }
    } else if (p.type == W_PLANE || p.type == W_AFFINE) {
        map_forward(p, 0, 0, u, v); mm.add(u, v);
        map_forward(p, 0, (float)(h - 1), u, v); mm.add(u, v);
        map_forward(p, (float)(w - 1), 0, u, v); mm.add(u, v);
        map_forward(p, (float)(w - 1), (float)(h - 1), u, v); mm.add(u, v);
    } else {
        for (int x = 0; x < w; ++x) {
            map_forward(p, (float)x, 0, u, v); mm.add(u, v);
            map_forward(p, (float)x, (float)(h - 1), u, v); mm.add(u, v);
        }
        for (int y = 0; y < h; ++y) {
            map_forward(p, 0, (float)y, u, v); mm.add(u, v);
            map_forward(p, (float)(w - 1), (float)y, u, v); mm.add(u, v);
        }
    }
    tl[0] = (int)mm.tl_u; tl[1] = (int)mm.tl_v;
    br[0] = (int)mm.br_u; br[1] = (int)mm.br_v;
    if (p.type == W_SPHERICAL) {
        float tl_uf = (float)tl[0], tl_vf = (float)tl[1], br_uf = (float)br[0], br_vf = (float)br[1];
        float x = p.rinv[1], y = p.rinv[4], z = p.rinv[7];
        if (y > 0.f) {
            float x_ = (p.k[0] * x + p.k[1] * y) / z + p.k[2];
            float y_ = p.k[4] * y / z + p.k[5];
            if (x_ > 0.f && x_ < w && y_ > 0.f && y_ < h) {
                float pv = static_cast<float>(3.14159265358979323846 * p.scale);
                tl_uf = (std::min)(tl_uf, 0.f); tl_vf = (std::min)(tl_vf, pv);
                br_uf = (std::max)(br_uf, 0.f); br_vf = (std::max)(br_vf, pv);
            }
        }
        x = p.rinv[1]; y = -p.rinv[4]; z = p.rinv[7];
        if (y > 0.f) {
            float x_ = (p.k[0] * x + p.k[1] * y) / z + p.k[2];
            float y_ = p.k[4] * y / z + p.k[5];
            if (x_ > 0.f && x_ < w && y_ > 0.f && y_ < h) {
                tl_uf = (std::min)(tl_uf, 0.f); tl_vf = (std::min)(tl_vf, 0.f);
                br_uf = (std::max)(br_uf, 0.f); br_vf = (std::max)(br_vf, 0.f);
            }
        }
        tl[0] = (int)tl_uf; tl[1] = (int)tl_vf;
        br[0] = (int)br_uf; br[1] = (int)br_vf;
    }
}

// ------------------------------------------------------------------------------------------
// cv::remap (modules/imgproc/src/imgwarp.cpp), classic fixed-point path
// ------------------------------------------------------------------------------------------
enum { INTER_BITS = 5, INTER_TAB_SIZE = 32, REMAP_COEF_BITS = 15, REMAP_COEF_SCALE = 1 << 15 };

// initInterTab2D(INTER_LINEAR, fixpt=true): Q15 weights incl. the saturate_cast<short> of
// 32768 at (fx,fy)=(0,0) and the "fix the sum" step that follows it.
short g_bilinear_tab[INTER_TAB_SIZE * INTER_TAB_SIZE * 4 + 8];
bool g_tab_ready = false;
void init_bilinear_tab()
{
    if (g_tab_ready) return;
    float t1[INTER_TAB_SIZE * 2];
    for (int i = 0; i < INTER_TAB_SIZE; i++) {
        float x = (float)i * (1.f / INTER_TAB_SIZE);
        t1[i * 2 + 0] = 1.f - x;
        t1[i * 2 + 1] = x;
    }
    std::memset(g_bilinear_tab, 0, sizeof(g_bilinear_tab));
    const int ksize = 2;
    for (int i = 0; i < INTER_TAB_SIZE; i++)
        for (int j = 0; j < INTER_TAB_SIZE; j++) {
            short* itab = g_bilinear_tab + (i * INTER_TAB_SIZE + j) * ksize * ksize;
            int isum = 0;
            for (int k1 = 0; k1 < ksize; k1++) {
                float vy = t1[i * ksize + k1];
                for (int k2 = 0; k2 < ksize; k2++) {
                    float v = vy * t1[j * ksize + k2];
                    itab[k1 * ksize + k2] = sat_s16(cv_round(v * REMAP_COEF_SCALE));
                    isum += itab[k1 * ksize + k2];
                }
            }
            if (isum != REMAP_COEF_SCALE) {
                int diff = isum - REMAP_COEF_SCALE;
                int ksize2 = ksize / 2, Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (int k1 = ksize2; k1 < ksize2 + 2; k1++)
                    for (int k2 = ksize2; k2 < ksize2 + 2; k2++) {
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2]) mk1 = k1, mk2 = k2;
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2]) Mk1 = k1, Mk2 = k2;
                    }
                if (diff < 0) itab[Mk1 * ksize + Mk2] = (short)(itab[Mk1 * ksize + Mk2] - diff);
                else itab[mk1 * ksize + mk2] = (short)(itab[mk1 * ksize + mk2] - diff);
            }
        }
    g_tab_ready = true;
}

// map quantisation of remap()'s planar-fp32-map branch
inline void quantise_linear(float x, float y, int& ix, int& iy, int& fxy)
{
    int sx = cv_round(x * (float)INTER_TAB_SIZE);
    int sy = cv_round(y * (float)INTER_TAB_SIZE);
    fxy = (sy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (sx & (INTER_TAB_SIZE - 1));
    ix = sat_s16(sx >> INTER_BITS);
    iy = sat_s16(sy >> INTER_BITS);
}

// remapBilinear<FixedPtCast<int,uchar,15>, ..., short>, one pixel, cn channels
inline void bilinear_px(const uint8_t* S0, int sw, int sh, size_t sstep, int cn, int sx, int sy, int fxy,
                        int border, uint8_t* D)
{
    const short* w = g_bilinear_tab + fxy * 4;
    if ((unsigned)sx < (unsigned)(sw - 1) && (unsigned)sy < (unsigned)(sh - 1)) {
        const uint8_t* S = S0 + sy * sstep + sx * cn;
        for (int k = 0; k < cn; k++) {
            int t = S[k] * w[0] + S[k + cn] * w[1] + S[sstep + k] * w[2] + S[sstep + k + cn] * w[3];
            D[k] = sat_u8((t + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS);
        }
        return;
    }
    if (border == B_CONSTANT && (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0)) {
        for (int k = 0; k < cn; k++) D[k] = 0;
        return;
    }
    int sx0 = border_interpolate(sx, sw, border), sx1 = border_interpolate(sx + 1, sw, border);
    int sy0 = border_interpolate(sy, sh, border), sy1 = border_interpolate(sy + 1, sh, border);
    static const uint8_t cval[4] = {0, 0, 0, 0};
    const uint8_t* v0 = sx0 >= 0 && sy0 >= 0 ? S0 + sy0 * sstep + sx0 * cn : cval;
    const uint8_t* v1 = sx1 >= 0 && sy0 >= 0 ? S0 + sy0 * sstep + sx1 * cn : cval;
    const uint8_t* v2 = sx0 >= 0 && sy1 >= 0 ? S0 + sy1 * sstep + sx0 * cn : cval;
    const uint8_t* v3 = sx1 >= 0 && sy1 >= 0 ? S0 + sy1 * sstep + sx1 * cn : cval;
    for (int k = 0; k < cn; k++) {
        int t = v0[k] * w[0] + v1[k] * w[1] + v2[k] * w[2] + v3[k] * w[3];
        D[k] = sat_u8((t + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS);
    }
}

// fp32 bilinear on the unquantised position (g_remap & 1): sensitivity model of the rewritten remap kernels, see g_remap
inline void bilinear_px_float(const uint8_t* S0, int sw, int sh, size_t sstep, int cn, float x, float y, int border, uint8_t* D)
{
    const bool fused = g_remap & 2;
    if (!(x > -1.0e9f && x < 1.0e9f && y > -1.0e9f && y < 1.0e9f)) { x = -1.f; y = -1.f; }
    const float fxf = std::floor(x), fyf = std::floor(y);
    const int ix = (int)fxf, iy = (int)fyf;
    const float a = x - fxf, b = y - fyf;
    int sx0 = border_interpolate(ix, sw, border), sx1 = border_interpolate(ix + 1, sw, border);
    int sy0 = border_interpolate(iy, sh, border), sy1 = border_interpolate(iy + 1, sh, border);
    static const uint8_t cval[4] = {0, 0, 0, 0};
    const uint8_t* v0 = sx0 >= 0 && sy0 >= 0 ? S0 + sy0 * sstep + sx0 * cn : cval;
    const uint8_t* v1 = sx1 >= 0 && sy0 >= 0 ? S0 + sy0 * sstep + sx1 * cn : cval;
    const uint8_t* v2 = sx0 >= 0 && sy1 >= 0 ? S0 + sy1 * sstep + sx0 * cn : cval;
    const uint8_t* v3 = sx1 >= 0 && sy1 >= 0 ? S0 + sy1 * sstep + sx1 * cn : cval;
    for (int k = 0; k < cn; k++) {
        const float p00 = v0[k], p01 = v1[k], p10 = v2[k], p11 = v3[k];
        const float t = muladd_model(a, p01 - p00, p00, fused), u = muladd_model(a, p11 - p10, p10, fused);
        D[k] = sat_u8(cv_round(muladd_model(b, u - t, t, fused)));
    }
}

// remapNearest<uchar>, BORDER_CONSTANT 0; map quantised by saturate_cast<short>(float)
inline void nearest_px(const uint8_t* S0, int sw, int sh, size_t sstep, int cn, float x, float y, uint8_t* D)
{
    int sx = sat_s16(cv_round(x)), sy = sat_s16(cv_round(y));
    if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) {
        const uint8_t* S = S0 + sy * sstep + sx * cn;
        for (int k = 0; k < cn; k++) D[k] = S[k];
    } else {
        for (int k = 0; k < cn; k++) D[k] = 0;
    }
}

// ------------------------------------------------------------------------------------------
// Pyramids (modules/imgproc/src/pyramids.cpp)
// ------------------------------------------------------------------------------------------
// pyrDown_<FixPtCast<short,8>>: 5x5 [1 4 6 4 1]^2, BORDER_REFLECT_101, int32 sums, (v+128)>>8
// (integer sums: the SIMD and scalar paths of OpenCV give the same bits)
void pyr_down_16s(const int16_t* src, int w, int h, int cn, int16_t* dst)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
#pragma omp parallel num_threads(g_threads)
    {
        std::vector<int> rows((size_t)5 * dw * cn);  // one ring of 5 filtered rows per thread
        std::vector<int> xi((size_t)5 * dw);
        for (int x = 0; x < dw; x++)
            for (int k = 0; k < 5; k++) xi[(size_t)5 * x + k] = border_interpolate(2 * x - 2 + k, w, B_REFLECT_101) * cn;
#pragma omp for schedule(static)
        for (int y = 0; y < dh; y++) {
            for (int k = 0; k < 5; k++) {
                int sy = border_interpolate(2 * y - 2 + k, h, B_REFLECT_101);
                const int16_t* s = src + (size_t)sy * w * cn;
                int* row = rows.data() + (size_t)k * dw * cn;
                for (int x = 0; x < dw; x++) {
                    const int* X = &xi[(size_t)5 * x];
                    for (int c = 0; c < cn; c++)
                        row[x * cn + c] = s[X[2] + c] * 6 + (s[X[1] + c] + s[X[3] + c]) * 4 + s[X[0] + c] + s[X[4] + c];
                }
            }
            int16_t* d = dst + (size_t)y * dw * cn;
            const int *r0 = rows.data(), *r1 = r0 + dw * cn, *r2 = r1 + dw * cn, *r3 = r2 + dw * cn, *r4 = r3 + dw * cn;
            for (int x = 0; x < dw * cn; x++) {
                int v = r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x];
                d[x] = (int16_t)((v + 128) >> 8);
            }
        }
    }
}

// pyrDown_<FltCast<float,8>>; g_pyr32f selects the evaluation order (default: the scalar C expression, no FMA)
void pyr_down_32f(const float* src, int w, int h, float* dst)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    const bool vsimd = g_pyr32f & 1, hsimd = g_pyr32f & 2, fused = g_pyr32f & 4;
    const int L = g_pyr32f_lanes > 0 ? g_pyr32f_lanes : 4;
    // outputs [1, hx1) of a row / [0, vx1) of the vertical pass are formed by the vector code
    const int width0 = std::min((w - 5 / 2 - 1) / 2 + 1, dw);
    const int hx1 = hsimd && width0 > 1 ? 1 + ((width0 - 1) / L) * L : 1;
    const int vx1 = vsimd ? (dw / L) * L : 0;
#pragma omp parallel num_threads(g_threads)
    {
        std::vector<float> rows((size_t)5 * dw);
#pragma omp for schedule(static)
        for (int y = 0; y < dh; y++) {
            for (int k = 0; k < 5; k++) {
                int sy = border_interpolate(2 * y - 2 + k, h, B_REFLECT_101);
                const float* s = src + (size_t)sy * w;
                float* row = rows.data() + (size_t)k * dw;
                for (int x = 0; x < dw; x++) {
                    int x0 = border_interpolate(2 * x - 2, w, B_REFLECT_101);
                    int x1 = border_interpolate(2 * x - 1, w, B_REFLECT_101);
                    int x2 = border_interpolate(2 * x, w, B_REFLECT_101);
                    int x3 = border_interpolate(2 * x + 1, w, B_REFLECT_101);
                    int x4 = border_interpolate(2 * x + 2, w, B_REFLECT_101);
                    if (x >= 1 && x < hx1) row[x] = muladd_model(s[x2], 6.f, muladd_model(s[x1] + s[x3], 4.f, s[x0] + s[x4], fused), fused);
                    else row[x] = s[x2] * 6 + (s[x1] + s[x3]) * 4 + s[x0] + s[x4];
                }
            }
            float* d = dst + (size_t)y * dw;
            const float *r0 = rows.data(), *r1 = r0 + dw, *r2 = r1 + dw, *r3 = r2 + dw, *r4 = r3 + dw;
            for (int x = 0; x < dw; x++) {
                float v;
                if (x < vx1) v = muladd_model(r1[x] + r3[x] + r2[x], 4.f, r0[x] + r4[x] + (r2[x] + r2[x]), fused);
                else v = r2[x] * 6 + (r1[x] + r3[x]) * 4 + r0[x] + r4[x];
                d[x] = v * (float)(1. / 256);
            }
        }
    }
}

// index rule of pyrUp_ on both axes: -1 -> 1 (reflect-101; 0 when n == 1), n -> n-1 (replicate)
inline int up_idx(int i, int n) { return i < 0 ? (n > 1 ? 1 : 0) : (i >= n ? n - 1 : i); }

// pyrUp_<FixPtCast<short,6>>; dst is exactly (2w, 2h) (always the case inside the blender)
void pyr_up_16s(const int16_t* src, int w, int h, int cn, int16_t* dst)
{
    int dw = 2 * w;
#pragma omp parallel num_threads(g_threads)
    {
        std::vector<int> rows((size_t)3 * dw * cn);
#pragma omp for schedule(static)
        for (int y = 0; y < h; y++) {
            for (int k = 0; k < 3; k++) {
                const int16_t* s = src + (size_t)up_idx(y - 1 + k, h) * w * cn;
                int* row = rows.data() + (size_t)k * dw * cn;
                for (int x = 0; x < w; x++) {
                    int xl = up_idx(x - 1, w) * cn, xc = x * cn, xr = up_idx(x + 1, w) * cn;
                    for (int c = 0; c < cn; c++) {
                        row[(2 * x) * cn + c] = s[xl + c] + s[xc + c] * 6 + s[xr + c];
                        row[(2 * x + 1) * cn + c] = (s[xc + c] + s[xr + c]) * 4;
                    }
                }
            }
            const int *r0 = rows.data(), *r1 = r0 + dw * cn, *r2 = r1 + dw * cn;
            int16_t* d0 = dst + (size_t)(2 * y) * dw * cn;
            int16_t* d1 = d0 + (size_t)dw * cn;
            for (int x = 0; x < dw * cn; x++) {
                d1[x] = (int16_t)(((r1[x] + r2[x]) * 4 + 32) >> 6);
                d0[x] = (int16_t)((r0[x] + r1[x] * 6 + r2[x] + 32) >> 6);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// distanceTransform(mask, DIST_L1, 3) -> CV_32F  (modules/imgproc/src/distransform.cpp,
// distanceTransform_3x3 with metrics {1, 2}; 16.16 fixed point, border = INT_MAX>>2)
// ------------------------------------------------------------------------------------------
void distance_transform_l1(const uint8_t* mask, int w, int h, float* dst)
{
    const int HV = 1 << 16, DIAG = 2 << 16, INIT = INT_MAX >> 2, DIST_MAX = INT_MAX >> 2;
    const float scale = 1.f / (1 << 16);
    int step = w + 2;
    std::vector<int> temp((size_t)(h + 2) * step, INIT);
    for (int i = 0; i < h; i++) {
        const uint8_t* s = mask + (size_t)i * w;
        int* tmp = temp.data() + (size_t)(i + 1) * step + 1;
        for (int j = 0; j < w; j++) {
            if (!s[j]) tmp[j] = 0;
            else {
                int t0 = tmp[j - step - 1] + DIAG;
                int t = tmp[j - step] + HV; if (t0 > t) t0 = t;
                t = tmp[j - step + 1] + DIAG; if (t0 > t) t0 = t;
                t = tmp[j - 1] + HV; if (t0 > t) t0 = t;
                tmp[j] = t0;
            }
        }
    }
    for (int i = h - 1; i >= 0; i--) {
        float* d = dst + (size_t)i * w;
        int* tmp = temp.data() + (size_t)(i + 1) * step + 1;
        for (int j = w - 1; j >= 0; j--) {
            int t0 = tmp[j];
            if (t0 > HV) {
                int t = tmp[j + step + 1] + DIAG; if (t0 > t) t0 = t;
                t = tmp[j + step] + HV; if (t0 > t) t0 = t;
                t = tmp[j + step - 1] + DIAG; if (t0 > t) t0 = t;
                t = tmp[j + 1] + HV; if (t0 > t) t0 = t;
                tmp[j] = t0;
            }
            t0 = (t0 > DIST_MAX) ? DIST_MAX : t0;
            d[j] = (float)(t0 * scale);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Blenders (modules/stitching/src/blenders.cpp)
// ------------------------------------------------------------------------------------------
const float WEIGHT_EPS = 1e-5f;

struct Img16 { int w = 0, h = 0; std::vector<int16_t> d; void create(int w_, int h_) { w = w_; h = h_; d.assign((size_t)w * h * 3, 0); } };
struct ImgF { int w = 0, h = 0; std::vector<float> d; void create(int w_, int h_) { w = w_; h = h_; d.assign((size_t)w * h, 0.f); } };
struct Img8 { int w = 0, h = 0; std::vector<uint8_t> d; void create(int w_, int h_) { w = w_; h = h_; d.assign((size_t)w * h, 0); } };

// normalizeUsingWeightMap, CV_32F weights
void normalize_using_weight_map(const ImgF& weight, Img16& src)
{
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int y = 0; y < src.h; ++y) {
        int16_t* row = src.d.data() + (size_t)y * src.w * 3;
        const float* wr = weight.d.data() + (size_t)y * src.w;
        for (int x = 0; x < src.w; ++x) {
            row[x * 3 + 0] = trunc_s16(row[x * 3 + 0] / (wr[x] + WEIGHT_EPS));
            row[x * 3 + 1] = trunc_s16(row[x * 3 + 1] / (wr[x] + WEIGHT_EPS));
            row[x * 3 + 2] = trunc_s16(row[x * 3 + 2] / (wr[x] + WEIGHT_EPS));
        }
    }
}

struct Blender {
    int kind;  // 0 = no (base Blender), 1 = feather, 2 = multiband
    int actual_num_bands = 5, num_bands = 0;
    float sharpness = 0.02f;
    int rx = 0, ry = 0, rw = 0, rh = 0;       // dst_roi_
    int fw = 0, fh = 0;                       // dst_roi_final_ size (multiband)
    Img16 dst;
    Img8 dst_mask;
    ImgF dst_weight;                          // feather
    std::vector<Img16> pyr_laplace;           // multiband (level 0 lives here, not in dst)
    std::vector<ImgF> band_weights;

    void prepare(int x, int y, int w, int h)
    {
        if (kind == 2) {
            fw = w; fh = h;
            double max_len = (double)std::max(w, h);
            num_bands = std::min(actual_num_bands, (int)std::ceil(std::log(max_len) / std::log(2.0)));
            w += ((1 << num_bands) - w % (1 << num_bands)) % (1 << num_bands);
            h += ((1 << num_bands) - h % (1 << num_bands)) % (1 << num_bands);
        }
        rx = x; ry = y; rw = w; rh = h;
        dst_mask.create(w, h);
        if (kind == 2) {
            pyr_laplace.resize(num_bands + 1);
            band_weights.resize(num_bands + 1);
            pyr_laplace[0].create(w, h);
            band_weights[0].create(w, h);
            for (int i = 1; i <= num_bands; ++i) {
                pyr_laplace[i].create((pyr_laplace[i - 1].w + 1) / 2, (pyr_laplace[i - 1].h + 1) / 2);
                band_weights[i].create((band_weights[i - 1].w + 1) / 2, (band_weights[i - 1].h + 1) / 2);
            }
        } else {
            dst.create(w, h);
            if (kind == 1) dst_weight.create(w, h);
        }
    }

    // Blender::feed
    void feed_no(const int16_t* img, const uint8_t* mask, int w, int h, int tlx, int tly)
    {
        int dx = tlx - rx, dy = tly - ry;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                size_t di = (size_t)(dy + y) * rw + dx + x;
                if (mask[(size_t)y * w + x])
                    for (int c = 0; c < 3; c++) dst.d[di * 3 + c] = img[((size_t)y * w + x) * 3 + c];
                dst_mask.d[di] |= mask[(size_t)y * w + x];
            }
    }

    // FeatherBlender::feed (+ createWeightMap)
    void feed_feather(const int16_t* img, const uint8_t* mask, int w, int h, int tlx, int tly)
    {
        std::vector<float> wm((size_t)w * h);
        distance_transform_l1(mask, w, h, wm.data());
        for (size_t i = 0; i < wm.size(); i++) {
            float t = wm[i] * sharpness;
            wm[i] = t > 1.f ? 1.f : t;  // threshold(..., 1, 1, THRESH_TRUNC)
        }
        int dx = tlx - rx, dy = tly - ry;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                size_t di = (size_t)(dy + y) * rw + dx + x, si = (size_t)y * w + x;
                for (int c = 0; c < 3; c++)
                    dst.d[di * 3 + c] = (int16_t)(dst.d[di * 3 + c] + trunc_s16(img[si * 3 + c] * wm[si]));
                dst_weight.d[di] += wm[si];
            }
    }

    // MultiBandBlender::feed, CV_16SC3 image, CV_32F weights (the branch the reference takes)
    void feed_multiband(const int16_t* img, const uint8_t* mask, int w, int h, int tlx, int tly)
    {
        const int nb = num_bands;
        int gap = 3 * (1 << nb);
        int tlnx = std::max(rx, tlx - gap), tlny = std::max(ry, tly - gap);
        int brnx = std::min(rx + rw, tlx + w + gap), brny = std::min(ry + rh, tly + h + gap);
        tlnx = rx + (((tlnx - rx) >> nb) << nb);
        tlny = ry + (((tlny - ry) >> nb) << nb);
        int width = brnx - tlnx, height = brny - tlny;
        width += ((1 << nb) - width % (1 << nb)) % (1 << nb);
        height += ((1 << nb) - height % (1 << nb)) % (1 << nb);
        brnx = tlnx + width; brny = tlny + height;
        int dy = std::max(brny - (ry + rh), 0), dx = std::max(brnx - (rx + rw), 0);
        tlnx -= dx; brnx -= dx; tlny -= dy; brny -= dy;
        int top = tly - tlny, left = tlx - tlnx, bottom = brny - tly - h, right = brnx - tlx - w;
        (void)bottom; (void)right;

        // copyMakeBorder(img, BORDER_REFLECT) ; weight = mask * (1/255.f), copyMakeBorder(CONSTANT 0)
        std::vector<Img16> lap(nb + 1);
        std::vector<ImgF> wp(nb + 1);
        lap[0].create(width, height);
        wp[0].create(width, height);
        const float inv255 = (float)(1. / 255.);
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (int y = 0; y < height; y++) {
            int sy = border_interpolate(y - top, h, B_REFLECT);
            int my = y - top;
            for (int x = 0; x < width; x++) {
                int sx = border_interpolate(x - left, w, B_REFLECT);
                int mx = x - left;
                for (int c = 0; c < 3; c++)
                    lap[0].d[((size_t)y * width + x) * 3 + c] = img[((size_t)sy * w + sx) * 3 + c];
                float wv = 0.f;
                if ((unsigned)mx < (unsigned)w && (unsigned)my < (unsigned)h) wv = mask[(size_t)my * w + mx] * inv255;
                wp[0].d[(size_t)y * width + x] = wv;
            }
        }
        // createLaplacePyr (16S branch): Gaussian chain, then L_i = G_i - pyrUp(G_{i+1}) saturating
        for (int i = 0; i < nb; ++i) {
            lap[i + 1].create((lap[i].w + 1) / 2, (lap[i].h + 1) / 2);
            pyr_down_16s(lap[i].d.data(), lap[i].w, lap[i].h, 3, lap[i + 1].d.data());
        }
        for (int i = 0; i < nb; ++i) {
            Img16 up;
            up.create(lap[i].w, lap[i].h);
            pyr_up_16s(lap[i + 1].d.data(), lap[i + 1].w, lap[i + 1].h, 3, up.d.data());
            const long long nk = (long long)lap[i].d.size();
#pragma omp parallel for num_threads(g_threads) schedule(static)
            for (long long k = 0; k < nk; k++) lap[i].d[k] = sat_s16((int)lap[i].d[k] - (int)up.d[k]);
        }
        for (int i = 0; i < nb; ++i) {
            wp[i + 1].create((wp[i].w + 1) / 2, (wp[i].h + 1) / 2);
            pyr_down_32f(wp[i].d.data(), wp[i].w, wp[i].h, wp[i + 1].d.data());
        }
        int y_tl = tlny - ry, y_br = brny - ry, x_tl = tlnx - rx, x_br = brnx - rx;
        for (int i = 0; i <= nb; ++i) {
            int rcw = x_br - x_tl, rch = y_br - y_tl;
            Img16& D = pyr_laplace[i];
            ImgF& DW = band_weights[i];
#pragma omp parallel for num_threads(g_threads) schedule(static)
            for (int y = 0; y < rch; ++y) {
                const int16_t* src_row = lap[i].d.data() + (size_t)y * lap[i].w * 3;
                const float* weight_row = wp[i].d.data() + (size_t)y * wp[i].w;
                int16_t* dst_row = D.d.data() + ((size_t)(y_tl + y) * D.w + x_tl) * 3;
                float* dst_weight_row = DW.d.data() + (size_t)(y_tl + y) * DW.w + x_tl;
                for (int x = 0; x < rcw; ++x) {
                    for (int c = 0; c < 3; c++)
                        dst_row[x * 3 + c] = (int16_t)(dst_row[x * 3 + c] + trunc_s16(src_row[x * 3 + c] * weight_row[x]));
                    dst_weight_row[x] += weight_row[x];
                }
            }
            x_tl /= 2; y_tl /= 2; x_br /= 2; y_br /= 2;
        }
    }

    // {Blender,FeatherBlender,MultiBandBlender}::blend.  Output size: (rw,rh) or (fw,fh).
    void blend(int16_t* out, uint8_t* out_mask)
    {
        int ow = rw, oh = rh;
        if (kind == 1) {
            normalize_using_weight_map(dst_weight, dst);
            for (size_t i = 0; i < dst_mask.d.size(); i++) dst_mask.d[i] = dst_weight.d[i] > WEIGHT_EPS ? 255 : 0;
        } else if (kind == 2) {
            for (int i = 0; i <= num_bands; ++i) normalize_using_weight_map(band_weights[i], pyr_laplace[i]);
            // restoreImageFromLaplacePyr
            for (int i = num_bands; i > 0; --i) {
                Img16 up;
                up.create(pyr_laplace[i - 1].w, pyr_laplace[i - 1].h);
                pyr_up_16s(pyr_laplace[i].d.data(), pyr_laplace[i].w, pyr_laplace[i].h, 3, up.d.data());
                Img16& L = pyr_laplace[i - 1];
                const long long nk = (long long)L.d.size();
#pragma omp parallel for num_threads(g_threads) schedule(static)
                for (long long k = 0; k < nk; k++) L.d[k] = sat_s16((int)up.d[k] + (int)L.d[k]);
            }
            ow = fw; oh = fh;
            dst.create(ow, oh);
            dst_mask.create(ow, oh);
#pragma omp parallel for num_threads(g_threads) schedule(static)
            for (int y = 0; y < oh; y++)
                for (int x = 0; x < ow; x++) {
                    for (int c = 0; c < 3; c++) dst.d[((size_t)y * ow + x) * 3 + c] = pyr_laplace[0].d[((size_t)y * rw + x) * 3 + c];
                    dst_mask.d[(size_t)y * ow + x] = band_weights[0].d[(size_t)y * rw + x] > WEIGHT_EPS ? 255 : 0;
                }
        }
        // Blender::blend: dst_.setTo(0, dst_mask_ == 0)
        const long long npx = (long long)ow * oh;
#pragma omp parallel for num_threads(g_threads) schedule(static)
        for (long long i = 0; i < npx; i++) {
            bool keep = dst_mask.d[i] != 0;
            for (int c = 0; c < 3; c++) out[i * 3 + c] = keep ? dst.d[i * 3 + c] : (int16_t)0;
            out_mask[i] = dst_mask.d[i];
        }
    }
};

}  // namespace

// ==========================================================================================
// C entry points (ctypes)
// ==========================================================================================
ORC_API void orc_set_num_threads(int n) { g_threads = n < 1 ? 1 : n; }
// arithmetic-model switches (see g_pyr32f / g_remap); returns the previous packed setting
ORC_API int orc_set_model(int pyr32f, int pyr32f_lanes, int remap)
{
    const int prev = g_pyr32f | (g_pyr32f_lanes << 8) | (g_remap << 16);
    g_pyr32f = pyr32f & 7;
    g_pyr32f_lanes = pyr32f_lanes > 0 ? pyr32f_lanes : 4;
    g_remap = remap & 3;
    return prev;
}
ORC_API int orc_get_max_threads()
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// RotationWarper::warpRoi -> (x, y, w, h)
ORC_API int orc_warp_roi(int type, float scale, const float* K, const float* R, int w, int h, int trig, int* out_xywh)
{
    if (type < 0 || type >= W_COUNT) return -1;
    Projector p;
    make_projector(p, type, scale, K, R, trig);
    int tl[2], br[2];
    detect_result_roi(p, w, h, tl, br);
    out_xywh[0] = tl[0]; out_xywh[1] = tl[1];
    out_xywh[2] = br[0] - tl[0] + 1; out_xywh[3] = br[1] - tl[1] + 1;
    return 0;
}

// RotationWarperBase::buildMaps (maps for u in [tlx, tlx+dw), v in [tly, tly+dh))
ORC_API int orc_build_maps(int type, float scale, const float* K, const float* R, int trig, int tlx, int tly, int dw,
                           int dh, float* xmap, float* ymap)
{
    if (type < 0 || type >= W_COUNT) return -1;
    Projector p;
    make_projector(p, type, scale, K, R, trig);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int v = 0; v < dh; ++v)
        for (int u = 0; u < dw; ++u) {
            float x, y;
            map_backward(p, (float)(u + tlx), (float)(v + tly), x, y);
            xmap[(size_t)v * dw + u] = x;
            ymap[(size_t)v * dw + u] = y;
        }
    return 0;
}

// cv::remap(src u8 x cn, fp32 maps, INTER_LINEAR, border)
ORC_API int orc_remap_linear_u8(const uint8_t* src, int sw, int sh, int cn, const float* xmap, const float* ymap,
                                int dw, int dh, int border, uint8_t* dst)
{
    init_bilinear_tab();
    size_t sstep = (size_t)sw * cn;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int v = 0; v < dh; ++v)
        for (int u = 0; u < dw; ++u) {
            if (g_remap & 1) {
                bilinear_px_float(src, sw, sh, sstep, cn, xmap[(size_t)v * dw + u], ymap[(size_t)v * dw + u], border,
                                  dst + ((size_t)v * dw + u) * cn);
                continue;
            }
            int ix, iy, fxy;
            quantise_linear(xmap[(size_t)v * dw + u], ymap[(size_t)v * dw + u], ix, iy, fxy);
            bilinear_px(src, sw, sh, sstep, cn, ix, iy, fxy, border, dst + ((size_t)v * dw + u) * cn);
        }
    return 0;
}

// cv::remap(src u8 x cn, fp32 maps, INTER_NEAREST, BORDER_CONSTANT 0)
ORC_API int orc_remap_nearest_u8(const uint8_t* src, int sw, int sh, int cn, const float* xmap, const float* ymap,
                                 int dw, int dh, uint8_t* dst)
{
    size_t sstep = (size_t)sw * cn;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int v = 0; v < dh; ++v)
        for (int u = 0; u < dw; ++u)
            nearest_px(src, sw, sh, sstep, cn, xmap[(size_t)v * dw + u], ymap[(size_t)v * dw + u],
                       dst + ((size_t)v * dw + u) * cn);
    return 0;
}

// RotationWarper::warp, fused (no map materialisation): the "optimised CPU" variant of BASELINE.md.
// Writes image (INTER_LINEAR/BORDER_REFLECT) and/or the 255-mask (INTER_NEAREST/BORDER_CONSTANT).
ORC_API int orc_warp_fused(int type, float scale, const float* K, const float* R, int trig, const uint8_t* src, int sw,
                           int sh, int cn, const int* xywh, uint8_t* dst_img, uint8_t* dst_mask)
{
    if (type < 0 || type >= W_COUNT) return -1;
    init_bilinear_tab();
    Projector p;
    make_projector(p, type, scale, K, R, trig);
    int tlx = xywh[0], tly = xywh[1], dw = xywh[2], dh = xywh[3];
    size_t sstep = (size_t)sw * cn;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 16)
    for (int v = 0; v < dh; ++v)
        for (int u = 0; u < dw; ++u) {
            float x, y;
            map_backward(p, (float)(u + tlx), (float)(v + tly), x, y);
            if (dst_img && (g_remap & 1)) {
                bilinear_px_float(src, sw, sh, sstep, cn, x, y, B_REFLECT, dst_img + ((size_t)v * dw + u) * cn);
            } else if (dst_img) {
                int ix, iy, fxy;
                quantise_linear(x, y, ix, iy, fxy);
                bilinear_px(src, sw, sh, sstep, cn, ix, iy, fxy, B_REFLECT, dst_img + ((size_t)v * dw + u) * cn);
            }
            if (dst_mask) {
                int sx = sat_s16(cv_round(x)), sy = sat_s16(cv_round(y));
                dst_mask[(size_t)v * dw + u] = ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) ? 255 : 0;
            }
        }
    return 0;
}

ORC_API int orc_pyr_down_16s(const int16_t* src, int w, int h, int cn, int16_t* dst) { pyr_down_16s(src, w, h, cn, dst); return 0; }
ORC_API int orc_pyr_down_32f(const float* src, int w, int h, float* dst) { pyr_down_32f(src, w, h, dst); return 0; }
ORC_API int orc_pyr_up_16s(const int16_t* src, int w, int h, int cn, int16_t* dst) { pyr_up_16s(src, w, h, cn, dst); return 0; }
ORC_API int orc_border_interpolate(int p, int len, int type) { return border_interpolate(p, len, type); }
ORC_API int orc_distance_transform_l1(const uint8_t* mask, int w, int h, float* dst) { distance_transform_l1(mask, w, h, dst); return 0; }
ORC_API int orc_bilinear_tab(short* out) { init_bilinear_tab(); std::memcpy(out, g_bilinear_tab, sizeof(short) * 32 * 32 * 4); return 0; }

// cv.convertScaleAbs on CV_16S: saturate_cast<uchar>(|x|)
ORC_API int orc_convert_scale_abs_16s(const int16_t* src, size_t n, uint8_t* dst)
{
    for (size_t i = 0; i < n; i++) {
        float v = std::fabs((float)src[i]);
        dst[i] = sat_u8(cv_round(v));
    }
    return 0;
}

// cv::detail::resultRoi(corners, sizes)
ORC_API int orc_result_roi(int n, const int* corners_xy, const int* sizes_wh, int* out_xywh)
{
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; i++) {
        tlx = std::min(tlx, corners_xy[2 * i]); tly = std::min(tly, corners_xy[2 * i + 1]);
        brx = std::max(brx, corners_xy[2 * i] + sizes_wh[2 * i]); bry = std::max(bry, corners_xy[2 * i + 1] + sizes_wh[2 * i + 1]);
    }
    out_xywh[0] = tlx; out_xywh[1] = tly; out_xywh[2] = brx - tlx; out_xywh[3] = bry - tly;
    return 0;
}

ORC_API void* orc_blender_create(int kind, int num_bands, float sharpness)
{
    Blender* b = new Blender();
    b->kind = kind;
    b->actual_num_bands = num_bands;
    b->sharpness = sharpness;
    return b;
}
ORC_API int orc_blender_prepare(void* h, int x, int y, int w, int hh) { ((Blender*)h)->prepare(x, y, w, hh); return 0; }
ORC_API int orc_blender_num_bands(void* h) { return ((Blender*)h)->num_bands; }
ORC_API int orc_blender_out_size(void* h, int* wh)
{
    Blender* b = (Blender*)h;
    wh[0] = b->kind == 2 ? b->fw : b->rw;
    wh[1] = b->kind == 2 ? b->fh : b->rh;
    return 0;
}
ORC_API int orc_blender_feed(void* h, const int16_t* img, const uint8_t* mask, int w, int hh, int tlx, int tly)
{
    Blender* b = (Blender*)h;
    if (tlx < b->rx || tly < b->ry || tlx + w > b->rx + b->rw || tly + hh > b->ry + b->rh) return -2;
    if (b->kind == 0) b->feed_no(img, mask, w, hh, tlx, tly);
    else if (b->kind == 1) b->feed_feather(img, mask, w, hh, tlx, tly);
    else b->feed_multiband(img, mask, w, hh, tlx, tly);
    return 0;
}
ORC_API int orc_blender_blend(void* h, int16_t* out, uint8_t* out_mask) { ((Blender*)h)->blend(out, out_mask); return 0; }
ORC_API void orc_blender_destroy(void* h) { delete (Blender*)h; }

// sinf / cosf of n arguments under a trig mode (0: host libm, 1: exact, 2: glibc FMA build, 3: glibc SSE2 build) — for the
// tests that pin the glibc restatement against the host's libm
ORC_API void orc_trig_eval(int mode, int want_cos, const float* x, float* out, long long n)
{
    Trig t{mode};
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long long i = 0; i < n; i++) out[i] = want_cos ? t.cos_(x[i]) : t.sin_(x[i]);
}
// number of arguments in the bit-pattern range [lo, hi) (as float patterns, sign bit included by the caller's choice) where
// mode_a and mode_b disagree for sinf (bit 0 of `which`) / cosf (bit 1); the first few are written to `examples`
ORC_API long long orc_trig_compare_range(int mode_a, int mode_b, uint32_t lo, uint32_t hi, int which, uint32_t* examples, int max_examples)
{
    Trig a{mode_a}, b{mode_b};
    long long bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(g_threads)
    for (long long u = (long long)lo; u < (long long)hi; u++) {
        float x;
        const uint32_t bits = (uint32_t)u;
        std::memcpy(&x, &bits, 4);
        bool d = false;
        if (which & 1) { const float p = a.sin_(x), q = b.sin_(x); d = d || (f2u(p) != f2u(q) && !(p != p && q != q)); }
        if (which & 2) { const float p = a.cos_(x), q = b.cos_(x); d = d || (f2u(p) != f2u(q) && !(p != p && q != q)); }
        if (d) bad++;
    }
    // examples: a serial second pass over the (rare) disagreements keeps the parallel loop simple
    if (examples && max_examples > 0 && bad > 0) {
        int k = 0;
        for (long long u = (long long)lo; u < (long long)hi && k < max_examples; u++) {
            float x;
            const uint32_t bits = (uint32_t)u;
            std::memcpy(&x, &bits, 4);
            bool d = false;
            if (which & 1) { const float p = a.sin_(x), q = b.sin_(x); d = d || (f2u(p) != f2u(q) && !(p != p && q != q)); }
            if (which & 2) { const float p = a.cos_(x), q = b.cos_(x); d = d || (f2u(p) != f2u(q) && !(p != p && q != q)); }
            if (d) examples[k++] = bits;
        }
    }
    return bad;
}

// exposed for tests of the "exact" trig routines
ORC_API void orc_sincos_d(double x, double* s, double* c) { sincos_d(x, s, c); }
ORC_API double orc_atan2_d(double y, double x) { return atan2_d(y, x); }
ORC_API double orc_acos_d(double w) { return acos_d(w); }
ORC_API double orc_tan_d(double x) { return tan_d(x); }
ORC_API double orc_asin_d(double w) { return asin_d(w); }
ORC_API double orc_atan_d(double x) { return atan_d(x); }
ORC_API double orc_log_d(double x) { return log_d(x); }
ORC_API double orc_exp_d(double x) { return exp_d(x); }
ORC_API double orc_sinh_d(double x) { return sinh_d(x); }
ORC_API double orc_cosh_d(double x) { return cosh_d(x); }
// mapForward / mapBackward of one point (tests)
ORC_API int orc_map_point(int type, float scale, const float* K, const float* R, int trig, int backward, float a, float b,
                          float* out2)
{
    if (type < 0 || type >= W_COUNT) return -1;
    Projector p;
    make_projector(p, type, scale, K, R, trig);
    if (backward) map_backward(p, a, b, out2[0], out2[1]);
    else map_forward(p, a, b, out2[0], out2[1]);
    return 0;
}
