// stx_device_math.h — device-side scalar helpers for the gfx950 kernels.
//
// Exact-trig contract: sin/cos/atan2/acos (and tan/asin/atan/log/sinh/cosh) are evaluated in fp64 with a fixed
// sequence of IEEE fma/mul/add/div/sqrt operations (fdlibm minimax coefficients), then rounded
// once to fp32.  The same sequence on any IEEE machine gives the same bits, so the oracle's
// trig=exact mode and these routines agree bit-for-bit; versus libm sinf/cosf/atan2f/acosf
// (what OpenCV calls) the result differs by at most 1 ULP fp32, and only when libm itself is
// not correctly rounded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define STX_DEV __device__ __forceinline__

namespace stxd {

STX_DEV void sincos_d(double x, double* s, double* c)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00;
    const double PIO2_1T = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    if (!(fabs(x) < 1.0e6)) {  // outside the contract's domain (no panorama gets here): NaN
        *s = *c = __longlong_as_double(0x7ff8000000000000ll);
        return;
    }
    double kd = rint(__dmul_rn(x, INV_PIO2));
    double r = fma(-kd, PIO2_1, x);
    r = fma(-kd, PIO2_1T, r);
    double z = __dmul_rn(r, r);
    double ps = fma(z, S6, S5);
    ps = fma(z, ps, S4);
    ps = fma(z, ps, S3);
    ps = fma(z, ps, S2);
    ps = fma(z, ps, S1);
    double sr = fma(__dmul_rn(r, z), ps, r);
    double pc = fma(z, C6, C5);
    pc = fma(z, pc, C4);
    pc = fma(z, pc, C3);
    pc = fma(z, pc, C2);
    pc = fma(z, pc, C1);
    double cr = fma(__dmul_rn(z, z), pc, fma(-0.5, z, 1.0));
    long long k = (long long)kd;
    switch (k & 3) {
    case 0: *s = sr; *c = cr; break;
    case 1: *s = cr; *c = -sr; break;
    case 2: *s = -sr; *c = -cr; break;
    default: *s = -cr; *c = sr; break;
    }
}

STX_DEV double atan_d(double x)
{
    const double atanhi[4] = {4.63647609000806093515e-01, 7.85398163397448278999e-01,
                              9.82793723247329054082e-01, 1.57079632679489655800e+00};
    const double atanlo[4] = {2.26987774529616870924e-17, 3.06161699786838301793e-17,
                              1.39033110312309984516e-17, 6.12323399573676603587e-17};
    const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
                 aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
                 aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
                 aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
                 aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
                 aT10 = 1.62858201153657823623e-02;
    if (x != x) return x;
    bool neg = signbit(x);
    double a = fabs(x);
    int id;
    if (a < 0.4375) {
        id = -1;
    } else if (a < 1.1875) {
        if (a < 0.6875) { id = 0; a = __ddiv_rn(__dsub_rn(__dmul_rn(2.0, a), 1.0), __dadd_rn(2.0, a)); }
        else { id = 1; a = __ddiv_rn(__dsub_rn(a, 1.0), __dadd_rn(a, 1.0)); }
    } else if (a < 2.4375) {
        id = 2; a = __ddiv_rn(__dsub_rn(a, 1.5), __dadd_rn(1.0, __dmul_rn(1.5, a)));
    } else {
        id = 3; a = __ddiv_rn(-1.0, a);
    }
    double z = __dmul_rn(a, a);
    double w = __dmul_rn(z, z);
    double s1 = __dmul_rn(z, fma(w, fma(w, fma(w, fma(w, fma(w, aT10, aT8), aT6), aT4), aT2), aT0));
    double s2 = __dmul_rn(w, fma(w, fma(w, fma(w, fma(w, aT9, aT7), aT5), aT3), aT1));
    double r;
    if (id < 0) r = __dsub_rn(a, __dmul_rn(a, __dadd_rn(s1, s2)));
    else r = __dsub_rn(atanhi[id], __dsub_rn(__dsub_rn(__dmul_rn(a, __dadd_rn(s1, s2)), atanlo[id]), a));
    return neg ? -r : r;
}

STX_DEV double atan2_d(double y, double x)
{
    const double PI = 3.1415926535897931160E+00, PI_LO = 1.2246467991473531772E-16;
    if (x != x || y != y) return x + y;
    int m = (signbit(y) ? 1 : 0) | (signbit(x) ? 2 : 0);
    if (y == 0.0) {
        switch (m) {
        case 0: case 1: return y;
        case 2: return PI;
        default: return -PI;
        }
    }
    if (x == 0.0) return (m & 1) ? -PI / 2 : PI / 2;
    double z = atan_d(fabs(__ddiv_rn(y, x)));
    switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return __dsub_rn(PI, __dsub_rn(z, PI_LO));
    default: return __dsub_rn(__dsub_rn(z, PI_LO), PI);
    }
}

STX_DEV double acos_d(double w)
{
    double t = __dmul_rn(__dsub_rn(1.0, w), __dadd_rn(1.0, w));
    return atan2_d(__dsqrt_rn(t), w);
}

// --- the projectors beyond plane / cylindrical / spherical also need tan, asin, atan, log, sinh, cosh (same
// contract: a fixed sequence of IEEE fp64 operations, one rounding to fp32)
STX_DEV double tan_d(double x)
{
    double s, c;
    sincos_d(x, &s, &c);
    return __ddiv_rn(s, c);
}

STX_DEV double asin_d(double w)
{
    double t = __dmul_rn(__dsub_rn(1.0, w), __dadd_rn(1.0, w));
    return atan2_d(w, __dsqrt_rn(t));
}

STX_DEV double pow2_d(int k) { return __longlong_as_double((long long)(k + 1023) << 52); }

// fdlibm e_log.c: x = 2^k (1 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)), s = f / (2 + f)
STX_DEV double log_d(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (x != x) return x;
    if (x < 0.0) return __longlong_as_double(0x7ff8000000000000ll);
    if (x == 0.0) return __longlong_as_double(0xfff0000000000000ll);
    if (x == __longlong_as_double(0x7ff0000000000000ll)) return x;
    int k = 0;
    if (x < 2.2250738585072014e-308) {
        x = __dmul_rn(x, 18014398509481984.0);
        k = -54;
    }
    unsigned long long bits = (unsigned long long)__double_as_longlong(x);
    int hx = (int)(bits >> 32);
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int i = (hx + 0x95f64) & 0x100000;
    bits = ((unsigned long long)(unsigned)(hx | (i ^ 0x3ff00000)) << 32) | (bits & 0xffffffffull);
    k += i >> 20;
    const double f = __dsub_rn(__longlong_as_double((long long)bits), 1.0);
    const double dk = (double)k;
    const double s = __ddiv_rn(f, __dadd_rn(2.0, f));
    const double z = __dmul_rn(s, s), w = __dmul_rn(z, z);
    const double t1 = __dmul_rn(w, __dadd_rn(Lg2, __dmul_rn(w, __dadd_rn(Lg4, __dmul_rn(w, Lg6)))));
    const double t2 = __dmul_rn(z, __dadd_rn(Lg1, __dmul_rn(w, __dadd_rn(Lg3, __dmul_rn(w, __dadd_rn(Lg5, __dmul_rn(w, Lg7)))))));
    const double R = __dadd_rn(t2, t1);
    const double hfsq = __dmul_rn(__dmul_rn(0.5, f), f);
    return __dsub_rn(__dmul_rn(dk, LN2_HI),
                     __dsub_rn(__dsub_rn(hfsq, __dadd_rn(__dmul_rn(s, __dadd_rn(hfsq, R)), __dmul_rn(dk, LN2_LO))), f));
}

// fdlibm e_exp.c: x = k ln2 + r, exp(r) = 1 + r + r c / (2 - c)
STX_DEV double exp_d(double x)
{
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10,
                 INV_LN2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return __longlong_as_double(0x7ff0000000000000ll);
    if (x < -7.45133219101941108420e+02) return 0.0;
    double hi = x, lo = 0.0;
    int k = 0;
    if (fabs(x) > 0.34657359027997264) {
        k = (int)__dadd_rn(__dmul_rn(INV_LN2, x), x < 0.0 ? -0.5 : 0.5);
        const double t = (double)k;
        hi = __dsub_rn(x, __dmul_rn(t, LN2_HI));
        lo = __dmul_rn(t, LN2_LO);
    }
    const double r = __dsub_rn(hi, lo);
    const double t = __dmul_rn(r, r);
    const double c = __dsub_rn(r, __dmul_rn(t, __dadd_rn(P1, __dmul_rn(t, __dadd_rn(P2, __dmul_rn(t, __dadd_rn(P3, __dmul_rn(t, __dadd_rn(P4, __dmul_rn(t, P5))))))))));
    const double y = __dsub_rn(1.0, __dsub_rn(__dsub_rn(lo, __ddiv_rn(__dmul_rn(r, c), __dsub_rn(2.0, c))), hi));
    const int k1 = k / 2, k2 = k - k1;
    return __dmul_rn(__dmul_rn(y, pow2_d(k1)), pow2_d(k2));
}

STX_DEV double sinh_d(double x)
{
    if (x != x) return x;
    const double a = fabs(x);
    double r;
    if (a < 0.03125) {
        const double z = __dmul_rn(a, a);
        const double p = __dmul_rn(z, __dadd_rn(1.66666666666666657415e-01, __dmul_rn(z, __dadd_rn(8.33333333333333321769e-03,
                         __dmul_rn(z, __dadd_rn(1.98412698412698412526e-04, __dmul_rn(z, 2.75573192239858925110e-06)))))));
        r = __dadd_rn(a, __dmul_rn(a, p));
    } else {
        const double e = exp_d(a);
        r = __dmul_rn(0.5, __dsub_rn(e, __ddiv_rn(1.0, e)));
    }
    return x < 0.0 ? -r : r;
}

STX_DEV double cosh_d(double x)
{
    if (x != x) return x;
    const double e = exp_d(fabs(x));
    return __dmul_rn(0.5, __dadd_rn(e, __ddiv_rn(1.0, e)));
}

STX_DEV float tanf_x(float x) { return (float)tan_d((double)x); }
STX_DEV float asinf_x(float w) { return (float)asin_d((double)w); }
STX_DEV float atanf_x(float x) { return (float)atan_d((double)x); }
STX_DEV float logf_x(float x) { return (float)log_d((double)x); }
STX_DEV float sinhf_x(float x) { return (float)sinh_d((double)x); }
STX_DEV float coshf_x(float x) { return (float)cosh_d((double)x); }
// ---- trig = glibc: sinf / cosf exactly as glibc >= 2.28 computes them (sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h:
// double-precision range reduction by one multiply-subtract below 120, a 192-bit 4/pi table above, an 8th-degree sine / cosine
// polynomial, one rounding to float).  What cv::detail's projectors get from the host's libm on a glibc machine
// (stitching/warper.py:44-51 -> PyRotationWarper -> sinf / cosf), bit for bit: the oracle's restatement of the same routine
// equals the host's sinf and cosf on every one of the 2^32 float arguments (tests/test_glibc_trig.py, tools/check_glibc_trig.py).
// fma: the __sinf_fma build every x86-64-v3 host selects (GCC contracts each product whose only uses are sums); !fma: __sinf_sse2.
// The two differ at 17 positive arguments of all floats.
STX_DEV double gl_mad(double a, double b, double c, bool use_fma) { return use_fma ? fma(a, b, c) : __dadd_rn(__dmul_rn(a, b), c); }
STX_DEV uint32_t gl_abstop12(float x) { return (__float_as_uint(x) >> 20) & 0x7ffu; }
STX_DEV float gl_sinf_poly(double x, double x2, bool neg_cos, int n, bool use_fma)
{
    // __sincosf_table[0] / [1] (the second entry holds the negated cosine polynomial)
    const double c0 = neg_cos ? -0x1p0 : 0x1p0, c1 = neg_cos ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2,
                 c2 = neg_cos ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5, c3 = neg_cos ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10,
                 c4 = neg_cos ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = __dmul_rn(x, x2);
        const double t1 = gl_mad(x2, s3, s2, use_fma);
        const double x7 = __dmul_rn(x3, x2);
        const double s = gl_mad(x3, s1, x, use_fma);
        return (float)gl_mad(x7, t1, s, use_fma);
    }
    const double x4 = __dmul_rn(x2, x2);
    const double t2 = gl_mad(x2, c4, c3, use_fma);
    const double t1 = gl_mad(x2, c1, c0, use_fma);
    const double x6 = __dmul_rn(x4, x2);
    const double c = gl_mad(x4, c2, t1, use_fma);
    return (float)gl_mad(x6, t2, c, use_fma);
}
STX_DEV double gl_reduce_large(uint32_t xi, int* np)
{
    const uint32_t inv_pio4[24] = {0xa2,       0xa2f9,     0xa2f983,   0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529,
                                   0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0,
                                   0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    const int i0 = (int)((xi >> 26) & 15u);
    const int shift = (int)((xi >> 23) & 7u);
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    unsigned long long res0 = (unsigned long long)(uint32_t)(xi * inv_pio4[i0]);  // 32-bit product, as in the source
    const unsigned long long res1 = (unsigned long long)xi * inv_pio4[i0 + 4];
    const unsigned long long res2 = (unsigned long long)xi * inv_pio4[i0 + 8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const unsigned long long n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    *np = (int)n;
    return __dmul_rn((double)(long long)res0, 0x1.921FB54442D18p-62);
}
STX_DEV float gl_sincosf1(float y, bool want_cos, bool use_fma)
{
    double x = (double)y;
    int n;
    const int flip = want_cos ? 1 : 0;
    if (gl_abstop12(y) < gl_abstop12(0x1.921FB6p-1f)) {
        if (gl_abstop12(y) < gl_abstop12(0x1p-12f)) return want_cos ? 1.0f : y;
        return gl_sinf_poly(x, __dmul_rn(x, x), false, flip, use_fma);
    }
    if (gl_abstop12(y) < gl_abstop12(120.0f)) {
        const double r = __dmul_rn(x, 0x1.45F306DC9C883p+23);
        n = ((int)r + 0x800000) >> 24;
        x = gl_mad(-(double)n, 0x1.921FB54442D18p0, x, use_fma);
        const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;  // sign[n & 3] = {1, -1, -1, 1}
        return gl_sinf_poly(__dmul_rn(x, sg), __dmul_rn(x, x), (n & 2) != 0, n ^ flip, use_fma);
    }
    if (gl_abstop12(y) < 0x7f8u) {
        const uint32_t xi = __float_as_uint(y);
        const int sign = (int)(xi >> 31);
        x = gl_reduce_large(xi, &n);
        const int m = (n + sign) & 3;
        const double sg = (m == 1 || m == 2) ? -1.0 : 1.0;
        return gl_sinf_poly(__dmul_rn(x, sg), __dmul_rn(x, x), (m & 2) != 0, n ^ flip, use_fma);
    }
    return __uint_as_float(0x7fc00000u);
}
// trig: STX_TRIG_EXACT (0, correctly rounded), STX_TRIG_GLIBC (1), STX_TRIG_GLIBC_NOFMA (2) — include/stitching_amd.h
STX_DEV void sincosf_m(float x, int trig, float* s, float* c);
STX_DEV float sinf_x(float x) { double s, c; sincos_d((double)x, &s, &c); return (float)s; }
STX_DEV float cosf_x(float x) { double s, c; sincos_d((double)x, &s, &c); return (float)c; }
STX_DEV void sincosf_x(float x, float* s, float* c) { double sd, cd; sincos_d((double)x, &sd, &cd); *s = (float)sd; *c = (float)cd; }
STX_DEV void sincosf_m(float x, int trig, float* s, float* c)
{
    if (trig == 0) { sincosf_x(x, s, c); return; }
    *s = gl_sincosf1(x, false, trig == 1);
    *c = gl_sincosf1(x, true, trig == 1);
}
STX_DEV float sinf_m(float x, int trig) { return trig == 0 ? sinf_x(x) : gl_sincosf1(x, false, trig == 1); }
STX_DEV float cosf_m(float x, int trig) { return trig == 0 ? cosf_x(x) : gl_sincosf1(x, true, trig == 1); }
STX_DEV float atan2f_x(float y, float x) { return (float)atan2_d((double)y, (double)x); }
STX_DEV float acosf_x(float w) { return (float)acos_d((double)w); }

// fp32 arithmetic that must not be contracted or reassociated (OpenCV's baseline build has no FMA)
STX_DEV float fmul(float a, float b) { return __fmul_rn(a, b); }
STX_DEV float fadd(float a, float b) { return __fadd_rn(a, b); }
STX_DEV float fsub(float a, float b) { return __fsub_rn(a, b); }
STX_DEV float fdiv(float a, float b) { return __fdiv_rn(a, b); }
// correctly rounded fp32 square root (sqrtss): sqrtf gets the compiler's fix-up sequence, __fsqrt_rn is the bare
// v_sqrt_f32 (1 ulp) on this target
STX_DEV float fsqrt(float a) { return sqrtf(a); }
// a0*b0 + a1*b1 + a2*b2, left to right
STX_DEV float dot3(float a0, float b0, float a1, float b1, float a2, float b2)
{
    return fadd(fadd(fmul(a0, b0), fmul(a1, b1)), fmul(a2, b2));
}

// cvRound(float) as x86-64 _mm_cvtss_si32: round-half-even, INT_MIN when out of range / NaN
STX_DEV int cv_round(float v)
{
    if (!(v >= -2147483648.f && v < 2147483648.f)) return (int)0x80000000;
    return (int)rintf(v);
}
STX_DEV int sat_s16(int v) { return min(max(v, -32768), 32767); }
// static_cast<short>(float) as x86 compiles it (cvttss2si, low 16 bits)
STX_DEV int trunc_s16(float v)
{
    int i;
    if (!(v >= -2147483648.f && v < 2147483648.f)) i = (int)0x80000000;
    else i = (int)v;
    return (int)(short)(unsigned short)(unsigned)i;
}

// cv::borderInterpolate BORDER_REFLECT (fedcba|abcdefgh|hgfedcb), any distance
STX_DEV int reflect(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    // one mirror image away (the common case next to an image): no division
    const int q = p < 0 ? -p - 1 : 2 * len - 1 - p;
    if ((unsigned)q < (unsigned)len) return q;
    if (len == 1) return 0;
    int period = 2 * len;
    int m = p % period;
    if (m < 0) m += period;
    return m < len ? m : period - 1 - m;
}
// cv::borderInterpolate BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba), any distance
STX_DEV int reflect101(int p, int len)
{
    if ((unsigned)p < (unsigned)len) return p;
    // one mirror image away (every border tap of a 5-tap stencil on a level wider than 2 samples): no division.  The modulo form
    // below costs ~45 instructions per call; unrolled 19 x 4 times in the border path of the pyrDown kernels it made every
    // workgroup on an image edge take 15 us (round 3: the coarse pyramid levels are nothing but edge workgroups)
    const int q = p < 0 ? -p : 2 * len - 2 - p;
    if ((unsigned)q < (unsigned)len) return q;
    if (len == 1) return 0;
    int period = 2 * len - 2;
    int m = p % period;
    if (m < 0) m += period;
    return m < len ? m : period - m;
}
// cv::borderInterpolate BORDER_REFLECT for positions at most one mirror image away (-len <= p <= 2 len - 1), branch-free; any other
// p gives some index inside [0, len)
STX_DEV int reflect_near(int p, int len)
{
    const int a = p ^ (p >> 31);  // p < 0 ? -p - 1 : p
    const int b = a >= len ? 2 * len - 1 - a : a;
    return max(b, 0);
}
// reflect101 for taps at most one mirror image away (-(len - 1) <= p <= 2 len - 2), branch-free; any other p gives SOME index
// inside [0, len) (the LDS pyrDown kernels evaluate whole groups of 4 / 8 outputs and read — never use — the taps of the outputs
// past the image).  Exact for every tap of a 5-tap stencil centred on 2 x, x < len / 2, len >= 2.
STX_DEV int reflect101_near(int p, int len)
{
    const int a = abs(p);
    const int b = a >= len ? 2 * len - 2 - a : a;
    return max(b, 0);
}
// pyrUp_ index rule on both axes: -1 -> 1 (0 when n == 1), n -> n-1
STX_DEV int up_idx(int i, int n) { return i < 0 ? (n > 1 ? 1 : 0) : (i >= n ? n - 1 : i); }

// n[c] / d for three numerators sharing one denominator, bit-identical to IEEE-754 division
// (__fdiv_rn) whenever v_div_scale would not rescale — true here: d in [1e-5, #images], |n| <= 32768.
// It is LLVM's own f32 fdiv expansion (rcp, 2 Newton fmas; then mul + 4 fmas per quotient) with the
// reciprocal refinement shared.
STX_DEV void div3_shared(float d, float n0, float n1, float n2, float& q0, float& q1, float& q2)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __fmaf_rn(-d, r, 1.0f);
    r = __fmaf_rn(e, r, r);
    float t, u;
    t = __fmul_rn(n0, r); u = __fmaf_rn(-d, t, n0); t = __fmaf_rn(u, r, t); u = __fmaf_rn(-d, t, n0); q0 = __fmaf_rn(u, r, t);
    t = __fmul_rn(n1, r); u = __fmaf_rn(-d, t, n1); t = __fmaf_rn(u, r, t); u = __fmaf_rn(-d, t, n1); q1 = __fmaf_rn(u, r, t);
    t = __fmul_rn(n2, r); u = __fmaf_rn(-d, t, n2); t = __fmaf_rn(u, r, t); u = __fmaf_rn(-d, t, n2); q2 = __fmaf_rn(u, r, t);
}

}  // namespace stxd
