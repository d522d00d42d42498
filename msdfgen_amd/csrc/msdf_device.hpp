// msdf_device.hpp -- per-texel math of the MI355X MSDF hot path (device functions; also compilable for the host).
//
// Everything the reference evaluates per pixel is restructured around a pre-digested per-edge record (EdgeRec): all
// pixel-independent quantities (end tangents, corner bisectors, polynomial coefficients, divisions by constants) are computed
// once per edge by the prep kernel with the very same IEEE operations the reference uses per pixel, so results stay
// bit-identical while the per-pixel instruction count drops. fp64 throughout, no FMA contraction (-ffp-contract=off).
//
// Reference semantics restated: ShapeDistanceFinder::oneShotDistance (core/ShapeDistanceFinder.hpp:36-58) -- the EdgeCache
// pruning of distance() is a pure CPU optimisation.
#pragma once

#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <float.h>

#if defined(__HIPCC__)
#define MSDF_HD __host__ __device__ inline
#define MSDF_NOINLINE __attribute__((noinline))
#define MSDF_NOUNROLL _Pragma("clang loop unroll(disable)")
#define MSDF_UNROLL _Pragma("unroll")
#if defined(__HIP_DEVICE_COMPILE__)
#define MSDF_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)   // value is identical in every active lane: keep it in an SGPR
#else
#define MSDF_UNIFORM(x) (x)
#endif
#else
#define MSDF_HD inline
#define MSDF_NOINLINE
#define MSDF_NOUNROLL
#define MSDF_UNROLL
#define MSDF_UNIFORM(x) (x)
#endif

#if defined(MSDF_NO_UNLIKELY)
#define MSDF_UNLIKELY(x) (x)
#else
#define MSDF_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif   // a cold path: the register allocator places its spill code by block frequency

namespace msdfhip {

// Wave vote: does ANY lane of the wavefront need this? Used where skipping work is exact whenever no lane needs it and evaluating more is
// harmless (the walk of k_distance is wave-uniform: every lane looks at the same edge / contour).
#if defined(MSDF_NO_DYNAMIC_CULL)
#define MSDF_WAVE_ANY(pred) (true)
#elif defined(__HIP_DEVICE_COMPILE__)
#define MSDF_WAVE_ANY(pred) (__any((int) (pred)) != 0)
#else
#define MSDF_WAVE_ANY(pred) (pred)                   // host walk: one "lane" at a time, i.e. the most aggressive skipping
#endif

// ----------------------------------------------------------------------------------------------------- edge record

enum : int32_t {
    REC_A_ZERO = 1,   // direction(0) has zero length: normalize() yields (0,1), normalize(true) yields (0,0) (Vector2.hpp:42-46)
    REC_B_ZERO = 2,   // same for direction(1)
    REC_NORMED = 4,   // quadratic: a != 0 && |b/a| < 1e6 -> solveCubicNormed, else solveQuadratic (equation-solver.cpp:63-70)
    REC_CORNER = 8,   // colour changes between the previous edge and this one (MSDFErrorCorrection.cpp:127-129)
    REC_FASTDIV = 16  // every pixel-independent divisor of this edge has an exactly usable reciprocal in rcp[] (see divExact)
};

struct V2 { double x, y; };

// Laid out in 64-byte BLOCKS, in the order the per-texel walk needs them, so that a wavefront fetches what one edge evaluation reads
// with a few wide scalar loads issued together (msdf_kernels.hpp: EdgeRegs) instead of one dependent round trip per field:
//   R  (0x000, 128 B)  everything the per-texel relevance test and the selector's end-point logic read: control box, first / last
//                      control point, corner bisectors, normalised end tangents
//   E0 (0x080,  64 B)  ab + the per-type scalars k[]: a LINEAR edge is complete with R + E0
//   E1 (0x0c0,  64 B)  br, end tangents and their squared norms          (quadratic, cubic)
//   E2 (0x100,  64 B)  reciprocals, first interior control point, as     (quadratic, cubic)
//   T  (0x140,  64 B)  what only phase 1 / the scanline pass / the digest read: second interior control point, on-curve sample, ids
struct alignas(128) EdgeRec {
    // ---- R
    double lo[2];     // bounding box of the control points (tile culling, per-texel relevance)
    double hi[2];
    double p0[2];     // first control point
    double pe[2];     // LAST control point = point(1) (p1 / p2 / p3 for linear / quadratic / cubic)
    double na[2];     // (prevDirN+aDirN).normalize(true)   (edge-selectors.cpp:197)
    double nb[2];     // (bDirN+nextDirN).normalize(true)   (edge-selectors.cpp:198)
    double aDirN[2];  // direction(0).normalize(true)       (edge-selectors.cpp:193)
    double bDirN[2];  // direction(1).normalize(true)       (edge-selectors.cpp:194)
    // ---- E0
    double ab[2];     // p1-p0
    double k[6];      // linear:    abab, orthoN.x, orthoN.y, abN.x, abN.y, RN(1/abab)
                      // quadratic: a, b, 2*dot(ab,ab), b/a, (b/a)^2, (b/a)*(1/3.)
                      // cubic:     3*ab.x, 3*ab.y, 6*br.x, 6*br.y
    // ---- E1
    double br[2];     // (p2-p1)-ab                         (quadratic, cubic)
    double ep0[2];    // direction(0)                       (edge-segments.cpp:121-139)
    double ep1[2];    // direction(1)
    double e0dot;     // dot(ep0, ep0)
    double e1dot;     // dot(ep1, ep1)
    // ---- E2
    double rcp[4];    // RN(1/divisor) of the pixel-independent divisors: quadratic {a, e0dot, e1dot}; cubic {-, e0dot, e1dot}  (linear: k[5])
    double p1[2];     // second control point (quadratic, cubic)
    double as_[2];    // ((p3-p2)-(p2-p1))-br               (cubic)
    // ---- T
    double p2[2];     // third control point of a cubic (scanline pass, digest)
    double mid[2];    // point(0.5): an on-curve sample (tile culling only)
    int32_t type;     // 1, 2, 3
    int32_t color;    // EdgeColor bitmask
    int32_t flags;    // REC_*
    int32_t contour;  // contour index within the batch
    double pad_[2];

    // Accessors: the per-texel math below is written against these, so that it runs unchanged on a record in memory (this struct:
    // the loads happen where the compiler puts them) and on a record held in scalar registers (msdf_kernels.hpp: EdgeRegs).
    MSDF_HD V2 Lo() const; MSDF_HD V2 Hi() const; MSDF_HD V2 P0() const; MSDF_HD V2 PE() const; MSDF_HD V2 P1() const;
    MSDF_HD V2 NA() const; MSDF_HD V2 NB() const; MSDF_HD V2 ADirN() const; MSDF_HD V2 BDirN() const;
    MSDF_HD V2 AB() const; MSDF_HD V2 BR() const; MSDF_HD V2 AS() const; MSDF_HD V2 EP0() const; MSDF_HD V2 EP1() const;
    MSDF_HD double E0dot() const { return e0dot; }
    MSDF_HD double E1dot() const { return e1dot; }
    MSDF_HD double K(int i) const { return k[i]; }
    MSDF_HD double Rcp(int i) const { return rcp[i]; }
    MSDF_HD int Type() const { return type; }
    MSDF_HD int Color() const { return color; }
    MSDF_HD int Flags() const { return flags; }
};
static_assert(sizeof(EdgeRec) == 384, "EdgeRec layout");
static_assert(offsetof(EdgeRec, na) == 0x40 && offsetof(EdgeRec, ab) == 0x80 && offsetof(EdgeRec, br) == 0xc0 && offsetof(EdgeRec, rcp) == 0x100 &&
              offsetof(EdgeRec, p2) == 0x140 && offsetof(EdgeRec, type) == 0x160, "EdgeRec blocks");

MSDF_HD V2 mk(double x, double y) { V2 r; r.x = x; r.y = y; return r; }
MSDF_HD V2 operator+(V2 a, V2 b) { return mk(a.x+b.x, a.y+b.y); }
MSDF_HD V2 operator-(V2 a, V2 b) { return mk(a.x-b.x, a.y-b.y); }
MSDF_HD V2 operator-(V2 a) { return mk(-a.x, -a.y); }
MSDF_HD V2 operator*(double a, V2 b) { return mk(a*b.x, a*b.y); }
MSDF_HD double dot(V2 a, V2 b) { return a.x*b.x+a.y*b.y; }
MSDF_HD double cross(V2 a, V2 b) { return a.x*b.y-a.y*b.x; }

// ---- square root ------------------------------------------------------------------------------------------------------------------
// hipcc expands the correctly rounded fp64 sqrt(x) to: scale x by 2^256 if x < 2^-767; y = v_rsq_f64(x); g = x*y, h = y/2; one coupled
// Goldschmidt step on (g, h) and two Newton corrections of g (seven FMAs); scale back; patch x = 0 / inf. For 2^-767 <= x < inf the scaling
// is the identity and the patch never applies: the SAME seven FMAs without them yield the same bits with 10 instead of 18 instructions.
// The wavefront takes the compiler's sqrt whenever any lane is outside that range (a texel exactly on a control point: x = 0).
// msdfhip_debug_sqrt_mismatches compares the two on the device (tests/test_gpu_parity.py).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MSDF_NO_LEAN_SQRT)
__device__ inline double leanSqrtCore(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x*y;
    double h = y*.5;
    const double r = __builtin_fma(-h, g, .5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
__device__ inline double msdfSqrt(double x) {
    const unsigned hi = (unsigned) __double2hiint(x);
    if (__any((int) (hi-0x10000000u >= 0x7ff00000u-0x10000000u)))      // some lane: x < 2^-767 (incl. 0, negative) or inf / nan
        return sqrt(x);
    return leanSqrtCore(x);
}
#else
MSDF_HD double msdfSqrt(double x) { return sqrt(x); }
#endif
MSDF_HD double vlen(V2 a) { return msdfSqrt(a.x*a.x+a.y*a.y); }
MSDF_HD V2 ld(const double *p) { return mk(p[0], p[1]); }
MSDF_HD V2 EdgeRec::Lo() const { return ld(lo); }
MSDF_HD V2 EdgeRec::Hi() const { return ld(hi); }
MSDF_HD V2 EdgeRec::P0() const { return ld(p0); }
MSDF_HD V2 EdgeRec::PE() const { return ld(pe); }
MSDF_HD V2 EdgeRec::P1() const { return ld(p1); }
MSDF_HD V2 EdgeRec::NA() const { return ld(na); }
MSDF_HD V2 EdgeRec::NB() const { return ld(nb); }
MSDF_HD V2 EdgeRec::ADirN() const { return ld(aDirN); }
MSDF_HD V2 EdgeRec::BDirN() const { return ld(bDirN); }
MSDF_HD V2 EdgeRec::AB() const { return ld(ab); }
MSDF_HD V2 EdgeRec::BR() const { return ld(br); }
MSDF_HD V2 EdgeRec::AS() const { return ld(as_); }
MSDF_HD V2 EdgeRec::EP0() const { return ld(ep0); }
MSDF_HD V2 EdgeRec::EP1() const { return ld(ep1); }

MSDF_HD V2 normalize(V2 a, bool allowZero) {                                 // Vector2.hpp:42-46
    double len = vlen(a);
    if (len != 0)
        return mk(a.x/len, a.y/len);
    return mk(0, allowZero ? 0. : 1.);
}

// normalize(a, allowZero) given len = vlen(a) (the callers below have it at hand; the square root is not recomputed)
MSDF_HD V2 normalizeLen(V2 a, double len, bool allowZero) {
    if (len != 0)
        return mk(a.x/len, a.y/len);
    return mk(0, allowZero ? 0. : 1.);
}

MSDF_HD V2 orthonormalFalse(V2 a) {                                          // getOrthonormal(false, false), Vector2.hpp:54-58
    double len = vlen(a);
    if (len != 0)
        return mk(a.y/len, -a.x/len);
    return mk(0, -1.);
}

MSDF_HD V2 mixv(V2 a, V2 b, double w) { return (1.-w)*a+w*b; }               // arithmetics.hpp:27-31
MSDF_HD double nonZeroSign(double n) { return n > 0 ? 1. : -1.; }            // arithmetics.hpp:59-61 (int 2*(n>0)-1, exact in double)
MSDF_HD double dmin(double a, double b) { return b < a ? b : a; }
MSDF_HD double dmax(double a, double b) { return a < b ? b : a; }
// For the CONSERVATIVE tests only (tile cull, per-texel relevance: they decide what is evaluated, never a value): the hardware's
// v_max_f64 / v_min_f64 -- one instruction where the reference-exact dmax / dmin (b < a ? b : a, whose NaN and signed-zero behaviour
// differs from IEEE maxNum) costs a compare and two selects.
#if defined(MSDF_EXACT_MINMAX_EVERYWHERE)
MSDF_HD double cmax(double a, double b) { return dmax(a, b); }
MSDF_HD double cmin(double a, double b) { return dmin(a, b); }
#else
MSDF_HD double cmax(double a, double b) { return __builtin_fmax(a, b); }
MSDF_HD double cmin(double a, double b) { return __builtin_fmin(a, b); }
#endif
MSDF_HD double median(double a, double b, double c) { return dmax(dmin(a, b), dmin(dmax(a, b), c)); }
MSDF_HD float fmin_(float a, float b) { return b < a ? b : a; }
MSDF_HD float fmax_(float a, float b) { return a < b ? b : a; }
MSDF_HD float medianf(float a, float b, float c) { return fmax_(fmin_(a, b), fmin_(fmax_(a, b), c)); }
MSDF_HD float mixf(float a, float b, double w) { return (float) ((1.-w)*a+w*b); } // arithmetics.hpp:27-31 (T=float, S=double)

// Correctly rounded quotient a/b from y = RN(1/b) with two FMAs (Markstein, "Computation of elementary functions on the IBM RISC
// System/6000 processor", 1990, Thm 4.? / Cornea-Harrison-Tang: q = RN(a*y), r = a-b*q exactly, q' = RN(q+r*y) == RN(a/b) when y
// is the correctly rounded reciprocal, b's significand is not all ones, and nothing over/underflows). divSafe() is evaluated per
// divisor at digestion time; edges with an unsafe divisor keep the IEEE division. ~3 instructions instead of ~28 on gfx950.
// tests/test_device_logic_host.py fuzzes the identity against true division.
MSDF_HD double divExact(double a, double b, double y) {
    const double q = a*y;
    const double r = fma(-b, q, a);
    return fma(r, y, q);
}

MSDF_HD bool divSafe(double b) {
    const double m = fabs(b);
    if (!(m > 1e-100 && m < 1e100))                  // also rejects 0, inf, nan
        return false;
    int e;
    const double f = frexp(m, &e);                   // f in [0.5, 1)
    return f != 1-0x1p-53;                           // significand all ones: RN(1/b) is not close enough
}

// ------------------------------------------------------------------------------------------------- equation solver

MSDF_HD int solveQuadratic(double x[2], double a, double b, double c) {      // equation-solver.cpp:9-32
    if (a == 0 || fabs(b) > 1e12*fabs(a)) {
        if (b == 0) {
            if (c == 0)
                return -1;
            return 0;
        }
        x[0] = -c/b;
        return 1;
    }
    double dscr = b*b-4*a*c;
    if (dscr > 0) {
        dscr = msdfSqrt(dscr);
        x[0] = (-b+dscr)/(2*a);
        x[1] = (-b-dscr)/(2*a);
        return 2;
    } else if (dscr == 0) {
        x[0] = -b/(2*a);
        return 1;
    } else
        return 0;
}

// ---- lean transcendentals for solveCubicNormed ------------------------------------------------------------------------------
// The reference calls glibc's acos / cos / pow there (equation-solver.cpp:41-60); neither glibc nor OCML is correctly rounded, so
// the contract is "< 1 ulp", not bit equality. OCML's generic cos (full range reduction) and pow (log + exp in extended precision)
// cost ~100 and ~250 instructions and many VGPRs each; the arguments here live in tiny, known ranges, which allows equally
// accurate (< 1 ulp, checked in tests/test_device_logic_host.py against 80-bit references) but far cheaper evaluations:
//   cos(1/3.*t), cos(1/3.*(t+2pi)), cos(1/3.*(t-2pi)) with t = acos(.) in [0, pi]  -> arguments in [0, pi/3], [2pi/3, pi],
//   [-2pi/3, -pi/3]: exact reductions pi-x and pi/2-|x| (Sterbenz, with a tail) onto [-pi/4, pi/4], then fdlibm's kernels.
//   pow(x, 1/3.) with x >= 0 -> a division-free cbrt with an exactly evaluated residual, times x^((1/3.)-1/3) = 1-1.85e-17*ln x.
// The host walk of tests/hostemu keeps libm (to stay bit-comparable with the oracle) unless MSDF_LEAN_MATH is defined.
#if defined(__HIP_DEVICE_COMPILE__) || defined(MSDF_LEAN_MATH)
#define MSDF_USE_LEAN_MATH 1
#endif

// fdlibm's __kernel_cos / __kernel_sin (public domain, Sun Microsystems 1993): cos / sin of x+y for |x| <= pi/4, y the tail of x;
// error < 1 ulp. Written without FMA, exactly as designed.
MSDF_HD double cosKernel(double x, double y) {
    const double z = x*x;
    const double r = z*(4.16666666666666019037e-02+z*(-1.38888888888741095749e-03+z*(2.48015872894767294178e-05+
                     z*(-2.75573143513906633035e-07+z*(2.08757232129817482790e-09+z*-1.13596475577881948265e-11)))));
    if (fabs(x) < .3)
        return 1-(.5*z-(z*r-x*y));
    const double qx = fabs(x) > .78125 ? .28125 : (double) (float) (.25*fabs(x));
    const double hz = .5*z-qx, a = 1-qx;
    return a-(hz-(z*r-x*y));
}

MSDF_HD double sinKernel(double x, double y) {
    const double z = x*x, v = z*x;
    const double r = 8.33333333332248946124e-03+z*(-1.98412698298579493134e-04+z*(2.75573137070700676789e-06+
                     z*(-2.50507602534068634195e-08+z*1.58969099521155010221e-10)));
    return x-((z*(.5*y-v*r)-y)-v*-1.66666666666666324348e-01);
}

// hi+lo = (c_hi-x)+c_lo where c_hi-x is exact (Sterbenz: c_hi/2 <= x <= 2*c_hi) and c_hi+c_lo is pi or pi/2 to ~107 bits.
MSDF_HD void reduceFrom(double cHi, double cLo, double x, double &hi, double &lo) {
    const double d = cHi-x;
    hi = d+cLo;
    lo = (d-hi)+cLo;
    if (fabs(d) < fabs(cLo)) {                       // (never for the ranges used unless d == 0) keep the two-sum exact
        hi = cLo+d;
        lo = (cLo-hi)+d;
    }
}

#define MSDF_PI_HI 3.141592653589793116
#define MSDF_PI_LO 1.2246467991473532e-16
#define MSDF_PIO2_HI 1.5707963267948965580
#define MSDF_PIO2_LO 6.123233995736766e-17

// cos(x) for x in [0, pi], < 1 ulp: every case reduces exactly onto [-pi/4, pi/4].
MSDF_HD double cosZeroToPi(double x) {
    double hi, lo;
    if (x <= .78539816339744830962)
        return cosKernel(x, 0);
    if (x < 2.35619449019234492885) {                // (pi/4, 3pi/4): cos x = sin(pi/2-x)
        reduceFrom(MSDF_PIO2_HI, MSDF_PIO2_LO, x, hi, lo);
        return sinKernel(hi, lo);
    }
    reduceFrom(MSDF_PI_HI, MSDF_PI_LO, x, hi, lo);   // [3pi/4, pi]: cos x = -cos(pi-x)
    return -cosKernel(hi, lo);
}

// cos of the three arguments of equation-solver.cpp:50-52, given t in [0, pi].
MSDF_HD void cosThirds(double t, double &c0, double &c1, double &c2) {
    const double a0 = 1/3.*t, a1 = 1/3.*(t+2*M_PI), a2 = 1/3.*(t-2*M_PI);     // the reference's own argument arithmetic
#if defined(MSDF_USE_LEAN_MATH)
    c0 = cosZeroToPi(a0);                            // a0 in [0, pi/3]
    c1 = cosZeroToPi(a1);                            // a1 in [2pi/3, pi]
    c2 = cosZeroToPi(fabs(a2));                      // a2 in [-2pi/3, -pi/3], cos is even
#else
    c0 = cos(a0), c1 = cos(a1), c2 = cos(a2);
#endif
}

// pow(x, 1/3.) for x >= 0, error < 1 ulp. x = a*2^(3k), a in [0.5, 4): r ~ a^(-1/3) from an fp32 exp2/log2 seed refined by two
// division-free Newton steps; y = a*r^2 ~ cbrt(a); one correction with the residual a-y^3 evaluated exactly (FMA, double-double)
// and the factor x^((1/3.)-1/3) = 1-1.85e-17*ln x folded into the same, single final rounding.
MSDF_HD double powThirdLean(double x) {
    if (!(x > 0 && x < DBL_MAX))
        return x == 0 ? 0. : pow(x, 1/3.);           // 0, inf, nan (x < 0 cannot occur: fabs(r)+sqrt(.)); denormals take the slow road too
    int e;
    const double m = frexp(x, &e);                   // x = m*2^e, m in [0.5, 1)
    const int k = (e >= 0 ? e : e-2)/3, j = e-3*k;   // floor division: j in {0, 1, 2}
    const double a = ldexp(m, j);                    // [0.5, 4)
    double r = (double) exp2f(-.333333333f*log2f((float) a));
    for (int it = 0; it < 2; ++it) {                 // Newton on r^-3 = a: r <- r*(4-a*r^3)/3
        const double r3 = r*r*r;
        r = r*fma(-a, r3, 4.)*.33333333333333333;
    }
    const double r2 = r*r;
    const double y = a*r2;
    const double t = y*y, tl = fma(y, y, -t);        // y^2 = t+tl exactly
    const double u = t*y, ul = fma(t, y, -u)+tl*y;   // y^3 = u+ul to ~2^-104
    const double resid = (a-u)-ul;
    const double lnx = .6931471805599453*(e+2*(m-1));                  // ln x to within 0.06: only scales a 1.85e-17 correction
    const double corr = fma(y, -1.850371707708594e-17*lnx, resid*(r2*.33333333333333333));
    return ldexp(y+corr, k);
}

MSDF_HD double powThird(double x) {
#if defined(MSDF_USE_LEAN_MATH)
    return powThirdLean(x);
#else
    return pow(x, 1/3.);
#endif
}

// solveCubicNormed (equation-solver.cpp:34-61) with the pixel-independent a, a*a and a*(1/3.) taken from the edge record.
MSDF_HD int solveCubicNormedPre(double x[3], double a, double a2, double a3, double b, double c) {
    double q = 1/9.*(a2-3*b);
    double r = 1/54.*(a*(2*a2-9*b)+27*c);
    double r2 = r*r;
    double q3 = q*q*q;
    if (r2 < q3) {
        double t = r/sqrt(q3);
        if (t < -1) t = -1;
        if (t > 1) t = 1;
        t = acos(t);
        q = -2*sqrt(q);
        double c0, c1, c2;
        cosThirds(t, c0, c1, c2);
        x[0] = q*c0-a3;
        x[1] = q*c1-a3;
        x[2] = q*c2-a3;
        return 3;
    } else {
        double u = (r < 0 ? 1. : -1.)*powThird(fabs(r)+sqrt(r2-q3));
        double v = u == 0 ? 0 : q/u;
        x[0] = (u+v)-a3;
        if (u == v || fabs(u-v) < 1e-12*fabs(u+v)) {
            x[1] = -.5*(u+v)-a3;
            return 2;
        }
        return 1;
    }
}

// --------------------------------------------------------------------------------------------------- signed distance

struct SD { double d, dot; };                                                // SignedDistance.hpp:10-20

MSDF_HD bool sdLess(SD a, SD b) {                                            // SignedDistance.hpp:22-24
    return fabs(a.d) < fabs(b.d) || (fabs(a.d) == fabs(b.d) && a.dot < b.dot);
}

template <class Rec> MSDF_HD V2 dirN0(const Rec &e) { return (e.Flags()&REC_A_ZERO) ? mk(0, 1) : e.ADirN(); } // direction(0).normalize()
template <class Rec> MSDF_HD V2 dirN1(const Rec &e) { return (e.Flags()&REC_B_ZERO) ? mk(0, 1) : e.BDirN(); } // direction(1).normalize()

template <class Rec> MSDF_HD SD sdLinear(const Rec &e, V2 o, double &param) {                 // edge-segments.cpp:173-185
    V2 p0 = e.P0(), p1 = e.PE(), ab = e.AB();
    V2 aq = o-p0;
    param = (e.Flags()&REC_FASTDIV) ? divExact(dot(aq, ab), e.K(0), e.K(5)) : dot(aq, ab)/e.K(0);
    // eq = (param > .5 ? p1 : p0)-o in the reference. Here its NEGATIVE, picked from o-p1 and o-p0 -- which selAddEdge needs anyway (bp, ap):
    // four selects instead of eight moves, four selects and a subtraction. IEEE subtraction, multiplication and division are odd functions
    // (round-to-nearest is symmetric), so -eq yields the same length, and a normalised vector / dot product that differ in sign only --
    // taken away by the fabs below (an exact zero may change its sign; it is squared, or added to the other product, or fabs'd).
    V2 bq = o-p1;
    V2 eq = param > .5 ? bq : aq;
    double endpointDistance = vlen(eq);
    if (param > 0 && param < 1) {
        double orthoDistance = dot(mk(e.K(1), e.K(2)), aq);
        if (fabs(orthoDistance) < endpointDistance) {
            SD r = { orthoDistance, 0 };
            return r;
        }
    }
    SD r = { nonZeroSign(cross(aq, ab))*endpointDistance, fabs(dot(mk(e.K(3), e.K(4)), normalizeLen(eq, endpointDistance, false))) };
    return r;
}

// The three roots of solveCubicNormed's trigonometric branch (equation-solver.cpp:41-53), produced ONE AT A TIME: q*cos(arg_i)-a3.
// Evaluating them inside the (rolled) candidate loop of sdQuadratic keeps a single cosine's temporaries live instead of three
// interleaved evaluations -- the quadratic path was what set the kernel's register demand (134 VGPRs for the plain SDF selector).
MSDF_HD double cosThirdsOne(double t, int i) {
    const double arg = i == 0 ? 1/3.*t : i == 1 ? 1/3.*(t+2*M_PI) : 1/3.*(t-2*M_PI);     // the reference's own argument arithmetic
#if defined(MSDF_USE_LEAN_MATH)
    return cosZeroToPi(fabs(arg));                   // [0, pi/3], [2pi/3, pi], [-2pi/3, -pi/3] (cos is even)
#else
    return cos(arg);
#endif
}

template <class Rec> MSDF_HD SD sdQuadratic(const Rec &e, V2 o, double &param) {              // edge-segments.cpp:187-226
    V2 p0 = e.P0(), p1 = e.P1(), p2 = e.PE(), ab = e.AB(), br = e.BR();
    V2 qa = p0-o;
    const bool fast = (e.Flags()&REC_FASTDIV) != 0;

    // solveCubic (equation-solver.cpp:63-70) up to the point where the roots are formed; the roots themselves follow lazily below
    double x0 = 0, x1 = 0, trigT = 0, trigQ = 0;
    int solutions;
    bool trig = false;
    {
        const double c = e.K(2)+dot(qa, br);
        const double d = dot(qa, ab);
        if (e.Flags()&REC_NORMED) {                                             // solveCubicNormed, equation-solver.cpp:34-61 (a, a*a, a/3 from the record)
            const double b = fast ? divExact(c, e.K(0), e.Rcp(0)) : c/e.K(0);
            const double cc = fast ? divExact(d, e.K(0), e.Rcp(0)) : d/e.K(0);
            const double a = e.K(3), a2 = e.K(4);
            double q = 1/9.*(a2-3*b);
            const double r = 1/54.*(a*(2*a2-9*b)+27*cc);
            const double r2 = r*r;
            const double q3 = q*q*q;
            if (r2 < q3) {
                double t = r/msdfSqrt(q3);
                if (t < -1) t = -1;
                if (t > 1) t = 1;
                trigT = acos(t);
                trigQ = -2*msdfSqrt(q);
                trig = true;
                solutions = 3;
            } else {
                const double u = (r < 0 ? 1. : -1.)*powThird(fabs(r)+msdfSqrt(r2-q3));
                const double v = u == 0 ? 0 : q/u;
                x0 = (u+v)-e.K(5);
                solutions = 1;
                if (u == v || fabs(u-v) < 1e-12*fabs(u+v)) {
                    x1 = -.5*(u+v)-e.K(5);
                    solutions = 2;
                }
            }
        } else {
            double t[2] = { 0, 0 };
            solutions = solveQuadratic(t, e.K(1), c, d);
            x0 = t[0], x1 = t[1];
        }
    }

    V2 epDir = e.EP0();
    const double lenA = vlen(qa);
    double minDistance = nonZeroSign(cross(epDir, qa))*lenA;
    param = fast ? divExact(-dot(qa, epDir), e.E0dot(), e.Rcp(1)) : -dot(qa, epDir)/e.E0dot();
    const V2 qb = p2-o;
    const double lenB = vlen(qb);
    {
        double distance = lenB;
        if (distance < fabs(minDistance)) {
            epDir = e.EP1();
            minDistance = nonZeroSign(cross(epDir, qb))*distance;
            param = fast ? divExact(dot(o-p1, epDir), e.E1dot(), e.Rcp(2)) : dot(o-p1, epDir)/e.E1dot();
        }
    }
    MSDF_NOUNROLL
    for (int i = 0; i < 3; ++i) {
        if (i < solutions) {
            const double ti = trig ? trigQ*cosThirdsOne(trigT, i)-e.K(5) : i == 0 ? x0 : x1;
            if (ti > 0 && ti < 1) {
                V2 qe = qa+(2*ti)*ab+(ti*ti)*br;
                double distance = vlen(qe);
                if (distance <= fabs(minDistance)) {
                    minDistance = nonZeroSign(cross(ab+ti*br, qe))*distance;
                    param = ti;
                }
            }
        }
    }
    SD r;
    r.d = minDistance;
    if (param >= 0 && param <= 1)
        r.dot = 0;
    else if (param < .5)
        r.dot = fabs(dot(dirN0(e), normalizeLen(qa, lenA, false)));
    else
        r.dot = fabs(dot(dirN1(e), normalizeLen(qb, lenB, false)));
    return r;
}

template <class Rec> MSDF_HD SD sdCubic(const Rec &e, V2 o, double &param) {                  // edge-segments.cpp:228-277
    V2 p0 = e.P0(), p3 = e.PE(), ab = e.AB(), br = e.BR(), as = e.AS();
    V2 ab3 = mk(e.K(0), e.K(1)), br6 = mk(e.K(2), e.K(3));
    V2 qa = p0-o;
    const bool fast = (e.Flags()&REC_FASTDIV) != 0;
    V2 epDir = e.EP0();
    const double lenA = vlen(qa);
    double minDistance = nonZeroSign(cross(epDir, qa))*lenA;
    param = fast ? divExact(-dot(qa, epDir), e.E0dot(), e.Rcp(1)) : -dot(qa, epDir)/e.E0dot();
    const V2 qb = p3-o;
    const double lenB = vlen(qb);
    {
        double distance = lenB;
        if (distance < fabs(minDistance)) {
            epDir = e.EP1();
            minDistance = nonZeroSign(cross(epDir, qb))*distance;
            param = fast ? divExact(dot(epDir-qb, epDir), e.E1dot(), e.Rcp(2)) : dot(epDir-qb, epDir)/e.E1dot();
        }
    }
    for (int i = 0; i <= 4; ++i) {                                            // MSDFGEN_CUBIC_SEARCH_STARTS, edge-segments.h:11
        double t = 1./4*i;
        V2 qe = qa+(3*t)*ab+(3*t*t)*br+(t*t*t)*as;
        V2 d1 = ab3+(6*t)*br+(3*t*t)*as;
        V2 d2 = br6+(6*t)*as;
        double improvedT = t-dot(qe, d1)/(dot(d1, d1)+dot(qe, d2));
        if (improvedT > 0 && improvedT < 1) {
            int remainingSteps = 4;                                           // MSDFGEN_CUBIC_SEARCH_STEPS, edge-segments.h:12
            do {
                t = improvedT;
                qe = qa+(3*t)*ab+(3*t*t)*br+(t*t*t)*as;
                d1 = ab3+(6*t)*br+(3*t*t)*as;
                if (!--remainingSteps)
                    break;
                d2 = br6+(6*t)*as;
                improvedT = t-dot(qe, d1)/(dot(d1, d1)+dot(qe, d2));
            } while (improvedT > 0 && improvedT < 1);
            double distance = vlen(qe);
            if (distance < fabs(minDistance)) {
                minDistance = nonZeroSign(cross(d1, qe))*distance;
                param = t;
            }
        }
    }
    SD r;
    r.d = minDistance;
    if (param >= 0 && param <= 1)
        r.dot = 0;
    else if (param < .5)
        r.dot = fabs(dot(dirN0(e), normalizeLen(qa, lenA, false)));
    else
        r.dot = fabs(dot(dirN1(e), normalizeLen(qb, lenB, false)));
    return r;
}

template <class Rec> MSDF_HD SD signedDistance(const Rec &e, V2 o, double &param) {
#if defined(MSDF_ONLY_TYPE)
    if (MSDF_ONLY_TYPE == 1) return sdLinear(e, o, param);
    if (MSDF_ONLY_TYPE == 2) return sdQuadratic(e, o, param);
    if (MSDF_ONLY_TYPE == 3) return sdCubic(e, o, param);
#endif
    if (e.Type() == 1)
        return sdLinear(e, o, param);
    if (e.Type() == 2)
        return sdQuadratic(e, o, param);
    return sdCubic(e, o, param);
}


// EdgeSegment::distanceToPerpendicularDistance, edge-segments.cpp:28-52
template <class Rec> MSDF_HD void distanceToPerpendicular(const Rec &e, SD &distance, V2 o, double param) {
    if (param < 0) {
        V2 dir = dirN0(e);
        V2 aq = o-e.P0();
        double ts = dot(aq, dir);
        if (ts < 0) {
            double perp = cross(aq, dir);
            if (fabs(perp) <= fabs(distance.d)) {
                distance.d = perp;
                distance.dot = 0;
            }
        }
    } else if (param > 1) {
        V2 dir = dirN1(e);
        V2 bq = o-e.PE();
        double ts = dot(bq, dir);
        if (ts > 0) {
            double perp = cross(bq, dir);
            if (fabs(perp) <= fabs(distance.d)) {
                distance.d = perp;
                distance.dot = 0;
            }
        }
    }
}

// --------------------------------------------------------------------------------------------------------- selectors

// PerpendicularDistanceSelectorBase (edge-selectors.h:40-70) with the nearest edge replaced by what computeDistance() derives
// from it. The reference remembers (nearEdge, nearEdgeParam) and, at the end, calls
//   nearEdge->distanceToPerpendicularDistance(minTrueDistance, p, nearEdgeParam)                        (edge-selectors.cpp:110-113)
// -- a function of the winning edge, its own signed distance / param and the query point only. All of these are at hand (the
// record in scalar registers) at the moment an edge BECOMES the nearest one, so the converted distance is evaluated right there and
// travels with the minimum: through later updates, through merge() (edge-selectors.cpp:96-106, the winner's value is the merged
// selector's value) and into computeDistance(). One register pair instead of (index, param), and no per-lane gather of a record at
// the end of every contour. Same operands, same operations, same result.
struct PB {
    double td, tdot;                     // minTrueDistance
    double perp;                         // distanceToPerpendicularDistance of it w.r.t. the nearest edge; meaningful iff td != -DBL_MAX
    double neg, pos;                     // min negative / positive perpendicular distance
};

MSDF_HD void pbInit(PB &b) {             // edge-selectors.cpp:54 followed by reset(delta): -DBL_MAX-delta == -DBL_MAX for any realistic delta
    b.td = -DBL_MAX, b.tdot = 0;
    b.perp = 0;
    b.neg = -DBL_MAX, b.pos = DBL_MAX;
}

// nearEdge != NULL: an edge replaces the initial minimum only with |distance| < DBL_MAX (SignedDistance <, SignedDistance.hpp:22-24;
// the tie-break dot is never negative), so a stored minimum is never -DBL_MAX again.
MSDF_HD bool pbHasNear(const PB &b) { return b.td != -DBL_MAX; }

MSDF_HD void pbAddPerp(PB &b, double d) {                                    // edge-selectors.cpp:89-94
    if (d <= 0 && d > b.neg)
        b.neg = d;
    if (d >= 0 && d < b.pos)
        b.pos = d;
}
// ... for a channel the edge may or may not carry (on: wave-uniform in k_distance). Folded into the conditions, the channel test is a scalar
// AND on the compare masks; as "if (on) pbAddPerp()" the compiler selected twice (new value, then channel).
MSDF_HD void pbAddPerpIf(PB &b, double d, bool on) {
#if defined(MSDF_NO_DIET_PERPMASK)
    if (on)
        pbAddPerp(b, d);
#else
    if (on & (d <= 0) & (d > b.neg))
        b.neg = d;
    if (on & (d >= 0) & (d < b.pos))
        b.pos = d;
#endif
}

MSDF_HD void pbMerge(PB &b, const PB &o) {                                   // edge-selectors.cpp:96-106
    SD a = { o.td, o.tdot }, c = { b.td, b.tdot };
    if (sdLess(a, c))
        b.td = o.td, b.tdot = o.tdot, b.perp = o.perp;
    if (o.neg > b.neg)
        b.neg = o.neg;
    if (o.pos < b.pos)
        b.pos = o.pos;
}

// pbMerge for the wave-uniform walk of k_distance: the three selects of the minimum only when some lane's contour is nearer.
MSDF_HD void pbMergeWave(PB &b, const PB &o) {
#if defined(MSDF_NO_DIET_MERGE)
    pbMerge(b, o);
#else
    SD a = { o.td, o.tdot }, c = { b.td, b.tdot };
    const bool less = sdLess(a, c);
    if (MSDF_WAVE_ANY(less)) {
        if (less)
            b.td = o.td, b.tdot = o.tdot, b.perp = o.perp;
    }
    if (o.neg > b.neg)
        b.neg = o.neg;
    if (o.pos < b.pos)
        b.pos = o.pos;
#endif
}

MSDF_HD double pbCompute(const PB &b) {                                      // edge-selectors.cpp:108-117
    double m = b.td < 0 ? b.neg : b.pos;
    if (pbHasNear(b) && fabs(b.perp) < fabs(m))
        m = b.perp;
    return m;
}

MSDF_HD bool getPerpendicularDistance(double &distance, V2 ep, V2 edgeDir) { // edge-selectors.cpp:42-52
    double ts = dot(ep, edgeDir);
    if (ts > 0) {
        double perp = cross(ep, edgeDir);
        if (fabs(perp) < fabs(distance)) {
            distance = perp;
            return true;
        }
    }
    return false;
}

// SEL: 1 TrueDistanceSelector, 2 PerpendicularDistanceSelector, 3 MultiDistanceSelector, 4 MultiAndTrueDistanceSelector
template <int SEL> struct SelTraits { enum { NPB = SEL == 1 ? 0 : SEL == 2 ? 1 : 3, NCH = SEL <= 2 ? 1 : SEL }; };

template <int SEL>
struct Selector {
    SD m;                                        // TrueDistanceSelector::minDistance (SEL == 1)
    PB c[(int) SelTraits<SEL>::NPB > 0 ? (int) SelTraits<SEL>::NPB : 1];
    // Visit index (position in the reference's visit order, ShapeDistanceFinder.hpp:45-57) of the edge that holds each minimum.
    // The reference feeds a contour's edges in visit order and replaces the minimum only on a strict SignedDistance <, so among
    // exactly tied edges (shared corner points: common) the FIRST VISITED wins. Keeping the index makes that rule explicit and the
    // result independent of the order in which a contour's edges are fed -- the kernel walks them nearest-first, which lets the
    // per-texel relevance test drop most of the others. Only compared within one contour; merges never look at it.
    int idx[(int) SelTraits<SEL>::NPB > 0 ? (int) SelTraits<SEL>::NPB : 1];
};

template <int SEL>
MSDF_HD void selInit(Selector<SEL> &s) {
    s.m.d = -DBL_MAX, s.m.dot = 0;
    s.idx[0] = 0x7fffffff;
    for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i) {
        pbInit(s.c[i]);
        s.idx[i] = 0x7fffffff;
    }
}

// sd replaces the current minimum cur (held by the edge visited at curIdx): SignedDistance < (SignedDistance.hpp:22-24), or an exact
// tie with an edge that the reference would have visited earlier.
MSDF_HD bool sdReplaces(SD sd, int idx, SD cur, int curIdx) {
    return sdLess(sd, cur) || (fabs(sd.d) == fabs(cur.d) && sd.dot == cur.dot && idx < curIdx);
}
// The same predicate for the wave-uniform walk of k_distance (every lane looks at the same edge): |distance| decides nearly always; the
// tie-break (three more compares) is evaluated only when SOME lane has an exact tie (edges meeting in a corner point, seen from a texel
// beyond it).
MSDF_HD bool sdReplacesWave(SD sd, int idx, SD cur, int curIdx) {
#if defined(MSDF_NO_DIET_TIES)
    return sdReplaces(sd, idx, cur, curIdx);
#else
    const double a = fabs(sd.d), b = fabs(cur.d);
    bool r = a < b;
    if (MSDF_WAVE_ANY(a == b))
        r = r || (a == b && (sd.dot < cur.dot || (sd.dot == cur.dot && idx < curIdx)));
    return r;
#endif
}

// addEdge: edge-selectors.cpp:19-29 (true), :129-160 (perpendicular), :174-227 (multi)
template <int SEL, class Rec>
MSDF_HD void selAddEdge(Selector<SEL> &s, const Rec &e, int idx, V2 o) {
    if (SEL == 1) {
        double dummy;
        SD sd = signedDistance(e, o, dummy);
        if (sdReplacesWave(sd, idx, s.m, s.idx[0]))
            s.m = sd, s.idx[0] = idx;
        return;
    }
    const int mask = SEL == 2 ? 1 : (e.Color()&7);
    if (!mask)
        return;                                  // MultiDistanceSelector ignores BLACK edges (edge-selectors.cpp:175-179)
    double param;
    SD sd = signedDistance(e, o, param);
    bool nearer[(int) SelTraits<SEL>::NPB > 0 ? (int) SelTraits<SEL>::NPB : 1];
    bool any = false;
    for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i) {                     // addEdgeTrueDistance, edge-selectors.cpp:81-87
        SD cur = { s.c[i].td, s.c[i].tdot };
        nearer[i] = (mask&(1<<i)) && sdReplacesWave(sd, idx, cur, s.idx[i]);
        any = any || nearer[i];
    }
    if (any) {
        SD conv = sd;                            // what computeDistance() would make of this edge, were it still the nearest then
        distanceToPerpendicular(e, conv, o, param);
        for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i)
            if (nearer[i])
                s.c[i].td = sd.d, s.c[i].tdot = sd.dot, s.c[i].perp = conv.d, s.idx[i] = idx;
    }
    V2 ap = o-e.P0();
    V2 bp = o-e.PE();
    double add = dot(ap, e.NA());
    double bdd = -dot(bp, e.NB());
    if (add > 0) {
        double pd = sd.d;
        if (getPerpendicularDistance(pd, ap, -e.ADirN())) {
            pd = -pd;
            for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i)
                pbAddPerpIf(s.c[i], pd, (mask&(1<<i)) != 0);
        }
    }
    if (bdd > 0) {
        double pd = sd.d;
        if (getPerpendicularDistance(pd, bp, e.BDirN())) {
            for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i)
                pbAddPerpIf(s.c[i], pd, (mask&(1<<i)) != 0);
        }
    }
}

// Per-texel relevance of an edge given the selector's CURRENT state (second-level, dynamic cull; the first level is per tile,
// msdf_cull.hpp). Skipping the edge is exact when this returns false:
//  * its distance to the texel is at least LB (control-box distance) > |current minimum true distance| of every channel it
//    carries, and minima only decrease -> it can never become the nearest edge, ties included (1e-9 relative slack);
//  * for each end, either the reference's own domain tests fail (add <= 0 or ts <= 0: same expressions as selAddEdge, so the same
//    values) or the perpendicular distance exceeds that bound, in which case it cannot survive computeDistance()/merge() because
//    the nearest edge's own (pseudo-)distance is smaller (see msdf_cull.hpp).
// The kernel evaluates the edge if ANY lane of the wavefront needs it (wave-uniform control flow); evaluating more is harmless.
// Two stages, so that a wavefront can vote after the first: an edge that is relevant at all is nearly always relevant through the box
// test of SOME lane, and then nobody has to look at the wedges (34 of 60 tests per wavefront on the font set).
template <int SEL, class Rec>
MSDF_HD bool selEdgeRelevantBox(const Selector<SEL> &s, const Rec &e, V2 o, double &bound2) {
    if (SEL == 1)                                    // squared bound, inflated
        bound2 = s.m.d*s.m.d;
    else {
        const int mask = SEL == 2 ? 1 : (e.Color()&7);
        bound2 = -1;                                 // an edge without a channel: neither stage can say "relevant"
        if (!mask)
            return false;
        bound2 = 0;
#if defined(MSDF_NO_DIET_BOUND)
        for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i)
            if (mask&(1<<i))
                bound2 = cmax(bound2, s.c[i].td*s.c[i].td);
#elif defined(MSDF_NO_DIET_BOUNDMAX)
        // a channel the edge does not carry contributes (td*0)*td = +0 (also for td = -DBL_MAX): the wave-uniform channel bit becomes a scalar
        // factor of the product instead of two selects per channel
        for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i)
            bound2 = cmax(bound2, (s.c[i].td*((mask&(1<<i)) ? 1. : 0.))*s.c[i].td);
#else
        // max over the carried channels of td^2 == (max of |td|)^2: rounding is monotonic, so squaring the maximum once yields the very value the
        // maximum of the squares had (round 6, tools/isa_bbcount.py: this test is 10 % of the kernel's VALU instructions; 7 instead of 10 of them
        // build the bound). The channel bit stays a scalar factor (|td|*0 = +0, also for td = -DBL_MAX); the absolute value is an operand modifier.
        double t = 0;
        for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i)
            t = cmax(t, fabs(s.c[i].td)*((mask&(1<<i)) ? 1. : 0.));
        bound2 = t*t;
#endif
    }
    bound2 *= 1+1e-9;                                // (-DBL_MAX)^2 = inf: nothing is skipped until a channel has a candidate
    const double dx = cmax(cmax(e.Lo().x-o.x, o.x-e.Hi().x), 0.);
    const double dy = cmax(cmax(e.Lo().y-o.y, o.y-e.Hi().y), 0.);
    return !(dx*dx+dy*dy > bound2);
}
template <int SEL, class Rec>
MSDF_HD bool selEdgeRelevantWedges(const Rec &e, V2 o, double bound2) {
    if (SEL >= 2) {
        const V2 ap = o-e.P0(), aDir = e.ADirN();
        if (dot(ap, e.NA()) > 0 && dot(ap, -aDir) > 0) {             // add > 0 && ts > 0 (edge-selectors.cpp:199-202, :43-44)
            const double perp = cross(ap, aDir);
            if (!(perp*perp > bound2))
                return true;
        }
        const V2 bp = o-e.PE(), bDir = e.BDirN();
        if (-dot(bp, e.NB()) > 0 && dot(bp, bDir) > 0) {             // bdd > 0 && ts > 0 (:212-215)
            const double perp = cross(bp, bDir);
            if (!(perp*perp > bound2))
                return true;
        }
    }
    return false;
}
template <int SEL, class Rec>
MSDF_HD bool selEdgeRelevant(const Selector<SEL> &s, const Rec &e, V2 o) {
    double bound2;
    return selEdgeRelevantBox(s, e, o, bound2) || selEdgeRelevantWedges<SEL>(e, o, bound2);
}

template <int SEL>
MSDF_HD void selMerge(Selector<SEL> &s, const Selector<SEL> &o) {            // edge-selectors.cpp:31-34, 229-233
    if (SEL == 1) {
        if (sdLess(o.m, s.m))
            s.m = o.m;
        return;
    }
    for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i)
        pbMergeWave(s.c[i], o.c[i]);
}

// distance(): edge-selectors.cpp:36, :162, :235-260. out has NCH entries.
template <int SEL>
MSDF_HD void selDistance(const Selector<SEL> &s, double *out) {
    if (SEL == 1) {
        out[0] = s.m.d;
        return;
    }
    for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i)
        out[i] = pbCompute(s.c[i]);
    if (SEL == 4) {                                                           // trueDistance(), :243-250
        SD t = { s.c[0].td, s.c[0].tdot };
        SD g = { s.c[SEL == 4 ? 1 : 0].td, s.c[SEL == 4 ? 1 : 0].tdot };
        SD b = { s.c[SEL == 4 ? 2 : 0].td, s.c[SEL == 4 ? 2 : 0].tdot };
        if (sdLess(g, t))
            t = g;
        if (sdLess(b, t))
            t = b;
        out[SEL == 4 ? 3 : 0] = t.d;
    }
}

template <int SEL>
MSDF_HD double resolve(const double *d) {                                    // contour-combiners.cpp:26-32
    return SEL >= 3 ? median(d[0], d[SEL >= 3 ? 1 : 0], d[SEL >= 3 ? 2 : 0]) : d[0];
}

// ------------------------------------------------------------------------------------------------- shape distance
//
// rec:      pre-digested records of this glyph's edges, contour by contour, each contour already in the reference's visit order
//           (last edge first, ShapeDistanceFinder.hpp:45-57); indexed relative to the glyph's first edge.
// coff:     the glyph's slice of the global contour_offsets (C+1 entries); e0 = coff[0].
// windings: Contour::winding per contour (contour-combiners.cpp:57-63).
// res:      per-lane scratch for the per-contour distances of the overlapping combiner, element (c, ch) at res[(c*NCH+ch)*rstride].

// Edge enumeration of one glyph for the per-texel loops: contour c owns positions [begin(c), end(c)); at(k) is the record index.
struct EdgesAll {                       // every edge, straight from the CSR offsets
    const int32_t *coff;
    MSDF_HD int begin(int c) const { return coff[c]-coff[0]; }
    MSDF_HD int end(int c) const { return coff[c+1]-coff[0]; }
    MSDF_HD int at(int k) const { return k; }
};
struct EdgesCulled {                    // survivors of the per-tile cull (msdf_cull.hpp), still grouped by contour and in visit order
    const int *cstart;                  // C+1 compacted offsets
    const int *list;                    // record index per position, or NULL if the surviving records were copied in this order
    MSDF_HD int begin(int c) const { return MSDF_UNIFORM(cstart[c]); }
    MSDF_HD int end(int c) const { return MSDF_UNIFORM(cstart[c+1]); }
    MSDF_HD int at(int k) const { return list ? MSDF_UNIFORM(list[k])&0xfffff : k; }   // uniform index -> scalar loads of the record (entries are packed: msdf_kernels.hpp)
};

// Feeds contour c's edges to the selector, in visit order (ShapeDistanceFinder.hpp:45-57). The default walks them one after the
// other in every lane; an Edges policy may provide its own (msdf_kernels.hpp: lanes = edges for a wave-uniform query point).
template <int SEL, class Edges>
MSDF_HD void selAddContourSerial(Selector<SEL> &sel, const EdgeRec *rec, const Edges &edges, int c, V2 o) {
    const int e = edges.end(c);
    for (int k = edges.begin(c); k < e; ++k) {
        const int i = edges.at(k);
        if (MSDF_WAVE_ANY(selEdgeRelevant(sel, rec[i], o)))
            selAddEdge(sel, rec[i], i, o);
    }
}
template <int SEL> MSDF_HD void selAddContour(Selector<SEL> &sel, const EdgeRec *rec, const EdgesAll &edges, int c, V2 o) { selAddContourSerial(sel, rec, edges, c, o); }
template <int SEL> MSDF_HD void selAddContour(Selector<SEL> &sel, const EdgeRec *rec, const EdgesCulled &edges, int c, V2 o) { selAddContourSerial(sel, rec, edges, c, o); }

// Measurement hooks (no-ops unless an edge policy overloads them: msdf_kernels.hpp under MSDF_PROFILE_WAITS).
template <class Edges> MSDF_HD void profAdd(const Edges &, int, unsigned long long) { }
template <class Edges> MSDF_HD unsigned long long profNow(const Edges &) { return 0; }

// A TEAM of wavefronts may share one texel tile's walk (msdf_kernels.hpp: TeamExchange, k_single_call): every member walks ITS share of a contour's
// survivors into a partial selector, afterWalk() hands the helpers' partial selectors to the leader, which merges them (selMergePartial) and carries on
// alone; a helper gets `false` and has nothing else to do for that contour. The default is a team of one.
struct NoTeam {
    template <int SEL> MSDF_HD bool afterWalk(Selector<SEL> &) const { return true; }
    MSDF_HD bool isHelper() const { return false; }
};

// Two partial selectors of the SAME contour over DISJOINT sets of its edges -> the selector the reference builds from the union: the minimum true distance
// by SignedDistance < with the explicit visit-index tie-break (sdReplaces: a total order, so the union's winner is the better of the two partial winners;
// the initial state loses to any edge), the converted distance travelling with it; max / min of the perpendicular distances do not depend on order.
template <int SEL>
MSDF_HD void selMergePartial(Selector<SEL> &a, const Selector<SEL> &b) {
    if (SEL == 1) {
        if (sdReplaces(b.m, b.idx[0], a.m, a.idx[0]))
            a.m = b.m, a.idx[0] = b.idx[0];
        return;
    }
    for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i) {
        SD x = { b.c[i].td, b.c[i].tdot }, y = { a.c[i].td, a.c[i].tdot };
        if (sdReplaces(x, b.idx[i], y, a.idx[i]))
            a.c[i].td = b.c[i].td, a.c[i].tdot = b.c[i].tdot, a.c[i].perp = b.c[i].perp, a.idx[i] = b.idx[i];
        if (b.c[i].neg > a.c[i].neg)
            a.c[i].neg = b.c[i].neg;
        if (b.c[i].pos < a.c[i].pos)
            a.c[i].pos = b.c[i].pos;
    }
}

template <int SEL, class Edges, class Team = NoTeam>
MSDF_HD void shapeDistanceSimple(const EdgeRec *rec, const Edges &edges, int C, V2 o, double *out, const Team &team = Team()) { // contour-combiners.cpp:34-50
    Selector<SEL> sel;
    selInit(sel);
    for (int c = 0; c < C; ++c)
        selAddContour(sel, rec, edges, c, o);
    if (!team.afterWalk(sel))
        return;
    selDistance(sel, out);
}

// The selection part of OverlappingContourCombiner::distance (contour-combiners.cpp:104-133) over the stored per-contour distances, shared by the two
// forms of the walk below. shapeD: the shape selector's distance; parked: what the (rare, wave-uniform: `second`) second walks left for the lanes with two or
// more members in their inner / outer selector.
template <int SEL, class Edges, class Wind>
MSDF_HD void combinerEpilogue(const Edges &edges, const Wind windings, int C, const double *res, int rstride, int nInner, int nOuter, int firstInner, int firstOuter,
                              bool second, const double *shapeD, const volatile double *parked, double *out) {
    enum { NCH = SelTraits<SEL>::NCH };
    double innerD[NCH], outerD[NCH];
    const unsigned long long te0 = profNow(edges);
    // merged selector with exactly one member == that member's own selector; with none, the initial state (every channel -DBL_MAX)
    for (int ch = 0; ch < NCH; ++ch) {
        innerD[ch] = nInner == 1 ? res[(firstInner*NCH+ch)*rstride] : -DBL_MAX;
        outerD[ch] = nOuter == 1 ? res[(firstOuter*NCH+ch)*rstride] : -DBL_MAX;
        if (second) {                                            // (wave-uniform; a lane reads what ITS second walk parked)
            if (nInner >= 2)
                innerD[ch] = parked[NCH+ch];
            if (nOuter >= 2)
                outerD[ch] = parked[2*NCH+ch];
        }
    }
    const double innerScalar = resolve<SEL>(innerD);
    const double outerScalar = resolve<SEL>(outerD);
    double dist[NCH];
    for (int ch = 0; ch < NCH; ++ch)
        dist[ch] = -DBL_MAX;
    // dm = resolve(dist) travels with dist (the reference re-evaluates the median in every comparison, contour-combiners.cpp:113-130: the
    // same function of the same values -- when dist becomes a contour's distance, its median is that contour's cm)
    int winding = 0;
    double dm;
    if (innerScalar >= 0 && fabs(innerScalar) <= fabs(outerScalar)) {
        for (int ch = 0; ch < NCH; ++ch)
            dist[ch] = innerD[ch];
        dm = innerScalar;
        winding = 1;
        for (int c = 0; c < C; ++c)
            if (windings[c] > 0) {
                double cd[NCH];
                for (int ch = 0; ch < NCH; ++ch)
                    cd[ch] = res[(c*NCH+ch)*rstride];
                const double cm = resolve<SEL>(cd);
                if (fabs(cm) < fabs(outerScalar) && cm > dm) {
                    for (int ch = 0; ch < NCH; ++ch)
                        dist[ch] = cd[ch];
                    dm = cm;
                }
            }
    } else if (outerScalar <= 0 && fabs(outerScalar) < fabs(innerScalar)) {
        for (int ch = 0; ch < NCH; ++ch)
            dist[ch] = outerD[ch];
        dm = outerScalar;
        winding = -1;
        for (int c = 0; c < C; ++c)
            if (windings[c] < 0) {
                double cd[NCH];
                for (int ch = 0; ch < NCH; ++ch)
                    cd[ch] = res[(c*NCH+ch)*rstride];
                const double cm = resolve<SEL>(cd);
                if (fabs(cm) < fabs(innerScalar) && cm < dm) {
                    for (int ch = 0; ch < NCH; ++ch)
                        dist[ch] = cd[ch];
                    dm = cm;
                }
            }
    } else {
        for (int ch = 0; ch < NCH; ++ch)
            out[ch] = shapeD[ch];
        profAdd(edges, 11, profNow(edges)-te0);
        return;
    }
    for (int c = 0; c < C; ++c)
        if (windings[c] != winding) {
            double cd[NCH];
            for (int ch = 0; ch < NCH; ++ch)
                cd[ch] = res[(c*NCH+ch)*rstride];
            const double cm = resolve<SEL>(cd);
            if (cm*dm >= 0 && fabs(cm) < fabs(dm)) {
                for (int ch = 0; ch < NCH; ++ch)
                    dist[ch] = cd[ch];
                dm = cm;
            }
        }
    if (dm == resolve<SEL>(shapeD))
        for (int ch = 0; ch < NCH; ++ch)
            dist[ch] = shapeD[ch];
    for (int ch = 0; ch < NCH; ++ch)
        out[ch] = dist[ch];
    profAdd(edges, 11, profNow(edges)-te0);
}

// OverlappingContourCombiner::distance (contour-combiners.cpp:77-134), restructured around what a lane has to REMEMBER.
// The reference keeps three merged selectors (shape / inner / outer) next to the contour being walked. Here only the shape one is
// kept; for the other two a lane counts the member contours it has seen and remembers the first:
//   no member   -> the merged selector is still in its initial state: every channel -DBL_MAX;
//   one member  -> merge(initial, contour) is that contour's own selector, its distance is the res[] entry of the contour (bitwise);
//   2+ members  -> (nested / overlapping contours around this texel) the member contours are walked a second time and merged in
//                  contour order, exactly the sequence of merges the reference performs. Membership is re-derived from res[].
// The second walk is wave-uniform (any lane needing it) and rare for font outlines; it buys 60 fewer live registers per lane than
// three merged selectors, which is what lets the kernel run without scratch spills. All walks share ONE instance of the edge loop
// (pass 0: every contour; pass 1 / 2: the members of the inner / outer selector) to keep the kernel's code and register demand down.
// windings: anything indexable by contour that yields Contour::winding -- the int8 array in memory, or (k_distance) two bit masks in scalar
// registers: a wave-uniform byte load from global memory per contour and loop iteration was a third of this function's time.
template <int SEL, class Edges, class Wind>
MSDF_HD void shapeDistanceOverlap(const EdgeRec *rec, const Edges &edges, const Wind windings, int C, V2 o, double *res, int rstride, double *out) {
    enum { NCH = SelTraits<SEL>::NCH };
    Selector<SEL> acc;                               // pass 0: the shape selector; pass 1 / 2: the inner / outer selector of the lanes that need one
    int nInner = 0, nOuter = 0, firstInner = 0, firstOuter = 0;
    // What the epilogue needs from the passes -- shapeD (pass 0), innerD / outerD (passes 1 / 2, only in the lanes with nInner / nOuter >= 2) -- is NOT carried
    // around the pass loop in registers: as loop-carried values these nine doubles cross the walk, and the 128-VGPR build (four wavefronts per SIMD) kept them
    // in scratch, stored at the top of every tile and re-stored at the end of every pass iteration (2.4 GB of scratch stores per 8 192-glyph pass). Instead:
    // when no lane of the wavefront needs a second walk (the usual case) the shape selector `acc` is still intact after the loop and shapeD is taken from it
    // THEN; only when a second walk runs are the passes' results parked in `parked` (private memory on the device: volatile, so that it stays memory) --
    // `second` is wave-uniform.
    double shapeD[NCH];
    volatile double parked[3*NCH];
    bool second = false;
    MSDF_NOUNROLL
    for (int pass = 0; pass < 3; ++pass) {
        const bool mine = pass == 1 ? nInner >= 2 : nOuter >= 2;
        if (pass > 0 && !MSDF_WAVE_ANY(mine))
            continue;
        selInit(acc);
        MSDF_NOUNROLL
        for (int c = 0; c < C; ++c) {
            bool member = false;
            if (pass > 0) {
                const int w = windings[c];
                if (pass == 1 ? !(w > 0) : !(w < 0))
                    continue;
                double cd[NCH];
                for (int ch = 0; ch < NCH; ++ch)
                    cd[ch] = res[(c*NCH+ch)*rstride];
                const double m = resolve<SEL>(cd);
                member = mine && (pass == 1 ? m >= 0 : m <= 0);
                if (!MSDF_WAVE_ANY(member))
                    continue;
            }
            Selector<SEL> sel;
            selInit(sel);
            const unsigned long long tw0 = profNow(edges);
            selAddContour(sel, rec, edges, c, o);
            const unsigned long long tw1 = profNow(edges);
            profAdd(edges, pass == 0 ? 8 : 10, tw1-tw0);
            if (pass == 0 && C == 1) {
                // One contour: the shape/inner/outer selectors can only ever hold that contour's own state, and every branch of
                // contour-combiners.cpp:104-133 then returns that contour's distance -- identical to the simple combiner.
                selDistance(sel, out);
                return;
            }
            if (pass == 0) {
                double d[NCH];
                selDistance(sel, d);
                for (int ch = 0; ch < NCH; ++ch)
                    res[(c*NCH+ch)*rstride] = d[ch];
                const double m = resolve<SEL>(d);
                const int w = windings[c];
#if !defined(MSDF_DIET_FIRSTCONTOUR)
                selMerge(acc, sel);
#else                                                 // measured: +31 spilled VGPRs in <3,true,false> (the branch keeps both paths live); off
                // merge(initial state, sel) == sel, field by field (edge-selectors.cpp:96-106 against :54: any minimum beats -DBL_MAX, an
                // untouched channel IS the initial state): the first contour's selector is copied instead of merged
                if (c == 0) {
                    acc.m = sel.m;
                    for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i)
                        acc.c[i] = sel.c[i];
                } else
                    selMerge(acc, sel);
#endif
                if (w > 0 && m >= 0) {
                    if (!nInner)
                        firstInner = c;
                    ++nInner;
                }
                if (w < 0 && m <= 0) {
                    if (!nOuter)
                        firstOuter = c;
                    ++nOuter;
                }
            } else if (member)
                selMerge(acc, sel);
            profAdd(edges, 9, profNow(edges)-tw1);
        }
        if (pass == 0) {
            second = MSDF_WAVE_ANY(nInner >= 2 || nOuter >= 2);
            if (second) {
                double t[NCH];
                selDistance(acc, t);
                for (int ch = 0; ch < NCH; ++ch)
                    parked[ch] = t[ch];
            }
        } else if (mine) {
            double t[NCH];
            selDistance(acc, t);
            for (int ch = 0; ch < NCH; ++ch)
                parked[pass*NCH+ch] = t[ch];
        }
    }
    if (!second)
        selDistance(acc, shapeD);                                 // (no pass after the first ran: acc is still the shape selector)
    else
        for (int ch = 0; ch < NCH; ++ch)
            shapeD[ch] = parked[ch];
#if defined(MSDF_ABLATE_EPILOGUE)                                   // measurement only: what the combiner's selection loops over the stored distances cost
    for (int ch = 0; ch < NCH; ++ch)
        out[ch] = shapeD[ch]+(double) (nInner+nOuter+firstInner+firstOuter);
    return;
#endif
    combinerEpilogue<SEL>(edges, windings, C, res, rstride, nInner, nOuter, firstInner, firstOuter, second, shapeD, parked, out);
}

// The same combiner for the hot walk of k_distance, in TWO instances of the contour loop (round 6). tools/isa_bbcount.py (basic-block execution counts of
// the production ISA, profiles/r06_valu_attribution.json) showed what the single rolled pass loop above costs a kernel at its register caps: `pass`
// lives in a spill lane, and EVERY contour iteration re-derives the predicates pass == 0 / != 0 / == 1 / != 1 as 64-bit lane masks, parks them in spill
// lanes (8 v_writelane) and reads them back where the body branches on them -- about 40 of the ~180 VALU instructions a contour iteration executes
// outside its edge loop, 10 % of the kernel's VALU instructions with the per-contour selects, none of it arithmetic. Here pass 0 is a loop of its own (no
// pass variable, no predicates), walked through `edges` (the register-resident record batches); the rare second walks (0.1 repeated evaluations per tile
// next to 7.5 on the font set) go through `edges2`, a plain policy whose record fields the compiler loads where it needs them -- slow and small, and its
// registers are live only there. Same operations on the same values in the same order as the form above.
template <int SEL, class Edges, class Edges2, class Wind, class Team = NoTeam>
MSDF_HD void shapeDistanceOverlapSplit(const EdgeRec *rec, const Edges &edges, const Edges2 &edges2, const Wind windings, int C, V2 o, double *res, int rstride, double *out,
                                       const Team &team = Team()) {
    enum { NCH = SelTraits<SEL>::NCH };
    Selector<SEL> acc;
    int nInner = 0, nOuter = 0, firstInner = 0, firstOuter = 0;
    selInit(acc);
    MSDF_NOUNROLL
    for (int c = 0; c < C; ++c) {
        Selector<SEL> sel;
        selInit(sel);
        const unsigned long long tw0 = profNow(edges);
        selAddContour(sel, rec, edges, c, o);
        const unsigned long long tw1 = profNow(edges);
        profAdd(edges, 8, tw1-tw0);
        if (!team.afterWalk(sel))                                // (a helper wavefront of a team: its share of the contour is with the leader now)
            continue;
        if (C == 1) {                                            // (see above: identical to the simple combiner)
            selDistance(sel, out);
            return;
        }
        double d[NCH];
        selDistance(sel, d);
        for (int ch = 0; ch < NCH; ++ch)
            res[(c*NCH+ch)*rstride] = d[ch];
        const double m = resolve<SEL>(d);
        const int w = windings[c];
        selMerge(acc, sel);
        if (w > 0 && m >= 0) {
            if (!nInner)
                firstInner = c;
            ++nInner;
        }
        if (w < 0 && m <= 0) {
            if (!nOuter)
                firstOuter = c;
            ++nOuter;
        }
        profAdd(edges, 9, profNow(edges)-tw1);
    }
    if (team.isHelper())
        return;
    double shapeD[NCH];
    volatile double parked[3*NCH];
    const bool second = MSDF_WAVE_ANY(nInner >= 2 || nOuter >= 2);
    selDistance(acc, shapeD);
    if (MSDF_UNLIKELY(second)) {                                 // rare, wave-uniform: the results cross the second walks in private memory, not in registers
        for (int ch = 0; ch < NCH; ++ch)
            parked[ch] = shapeD[ch];
        MSDF_NOUNROLL
        for (int pass = 1; pass < 3; ++pass) {
            const bool mine = pass == 1 ? nInner >= 2 : nOuter >= 2;
            if (!MSDF_WAVE_ANY(mine))
                continue;
            selInit(acc);
            MSDF_NOUNROLL
            for (int c = 0; c < C; ++c) {
                const int w = windings[c];
                if (pass == 1 ? !(w > 0) : !(w < 0))
                    continue;
                double cd[NCH];
                for (int ch = 0; ch < NCH; ++ch)
                    cd[ch] = res[(c*NCH+ch)*rstride];
                const double m = resolve<SEL>(cd);
                const bool member = mine && (pass == 1 ? m >= 0 : m <= 0);
                if (!MSDF_WAVE_ANY(member))
                    continue;
                Selector<SEL> sel;
                selInit(sel);
                const unsigned long long tw0 = profNow(edges);
                selAddContour(sel, rec, edges2, c, o);
                profAdd(edges, 10, profNow(edges)-tw0);
                if (member)
                    selMerge(acc, sel);
            }
            if (mine) {
                double t[NCH];
                selDistance(acc, t);
                for (int ch = 0; ch < NCH; ++ch)
                    parked[pass*NCH+ch] = t[ch];
            }
        }
        for (int ch = 0; ch < NCH; ++ch)
            shapeD[ch] = parked[ch];
    }
#if defined(MSDF_ABLATE_EPILOGUE)
    for (int ch = 0; ch < NCH; ++ch)
        out[ch] = shapeD[ch]+(double) (nInner+nOuter+firstInner+firstOuter);
    return;
#endif
    combinerEpilogue<SEL>(edges, windings, C, res, rstride, nInner, nOuter, firstInner, firstOuter, second, shapeD, parked, out);
}

// -------------------------------------------------------------------------------------------------------- transform

struct Xform {
    double sx, sy, tx, ty;       // Projection (core/Projection.h:31-33)
    double mapScale, mapTranslate; // DistanceMapping (core/DistanceMapping.h:28-30)
};

MSDF_HD V2 unproject(const Xform &t, V2 c) { return mk(c.x/t.sx-t.tx, c.y/t.sy-t.ty); }         // Projection.cpp:14-16
MSDF_HD V2 project(const Xform &t, V2 c) { return mk(t.sx*(c.x+t.tx), t.sy*(c.y+t.ty)); }       // Projection.cpp:10-12
MSDF_HD float mapDistance(const Xform &t, double d) { return (float) (t.mapScale*(d+t.mapTranslate)); } // DistanceMapping.cpp:15-17, msdfgen.cpp:20-48

} // namespace msdfhip
