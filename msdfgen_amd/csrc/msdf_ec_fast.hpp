// msdf_ec_fast.hpp -- the lean, single-sweep form of the error-correction classifier used for ALL texels; the handful of texels whose
// verdict depends on an exact shape-distance query (ShapeDistanceChecker, MSDFErrorCorrection.cpp:58-82; ~0.1 % of texels) are
// deferred to the full per-texel pipeline of msdf_ec.hpp (k_ec_slow).
//
// What is restructured relative to the reference, and why each step is exact:
//  * findErrors(sdf) (:383-410) and findErrors(sdf, shape) (:412-457) enumerate the SAME interpolation candidates (t, xm) -- they
//    differ only in the classifier (protectedFlag, distance check).  One sweep computes each candidate once and derives both
//    verdicts.  The sweep runs in shape orientation: a row flip maps every neighbour test of the native-order sweep onto a test
//    with identical operands (hasDiagonalArtifact always receives (texel, horizontal, vertical, diagonal neighbour)), and the
//    per-texel result is an OR over the tests, so the order does not matter.
//  * The texel's ERROR flag is an OR of side-effect-free predicates, so all cheap predicates are evaluated first; only if none
//    fires and some candidate needs the distance check is the texel deferred (result: identical flag).
//  * Linear pairs: t = dA/(dA-dB) lies in (0,1) only if dA and dB have strictly opposite signs (float subtraction and division are
//    monotone), so the fp64 division is skipped otherwise.
//  * Diagonal pairs: solveQuadratic (sqrt + 2 divisions) is skipped when the quadratic provably has no real root in
//    [0.005, 0.995] (the reference keeps only roots in (0.01, 0.99); the computed roots are within ~1e-4 of the real ones because
//    |b| <= 1e12*|a| in that branch).  The extremum parameters tEx (3 divisions) are computed only when a root survives.
#pragma once

#include "msdf_ec.hpp"

namespace msdfhip {

enum { EC_DEFER = 0x80 };   // internal: texel must be re-evaluated by the full pipeline (needs a shape-distance query)

// Which of the reference's two findErrors passes apply (core/msdf-error-correction.cpp:36-46).
MSDF_HD bool ecHasBasePass(const EcParams &p) { return p.distanceCheck == EC_DO_NOT_CHECK || (p.distanceCheck == EC_CHECK_AT_EDGE && p.mode != EC_MODE_EDGE_ONLY); }
MSDF_HD bool ecHasShapePass(const EcParams &p) { return p.distanceCheck == EC_ALWAYS_CHECK || p.distanceCheck == EC_CHECK_AT_EDGE; }

struct FastCtx {
    double span;
    bool p1;          // protectedFlag of the base pass (and of the shape pass under ALWAYS_CHECK)
    bool pShape;      // protectedFlag of the shape pass
    bool basePass, shapePass;
};

// rangeTest (:30-40) for both classifiers at once. Returns flags of the base pass in bits 0-1 and of the shape pass in bits 2-3.
MSDF_HD int rangeTest2(const FastCtx &c, double at, double bt, double xt, float am, float bm, float xm) {
    const bool inversion = (am > .5f && bm > .5f && xm <= .5f) || (am < .5f && bm < .5f && xm >= .5f);
    const bool outside = medianf(am, bm, xm) != xm;
    const bool candBase = c.basePass && (inversion || (!c.p1 && outside));
    const bool candShape = c.shapePass && (inversion || (!c.pShape && outside));
    if (!(candBase || candShape))
        return 0;
    const double axSpan = (xt-at)*c.span, bxSpan = (bt-xt)*c.span;
    const int f = (xm >= am-axSpan && xm <= am+axSpan && xm >= bm-bxSpan && xm <= bm+bxSpan) ? 1 : 3;
    return (candBase ? f : 0)|(candShape ? f<<2 : 0);
}

// 1: ERROR decided; 2: needs the distance check; 0: nothing.
MSDF_HD int judge(int flags) {
    if (flags&2)
        return 1;                     // base classifier: evaluate() == (flags&ARTIFACT) (:42-44)
    const int fs = flags>>2;
    if (fs&1)
        return (fs&2) ? 1 : 2;        // shape classifier: artifact already, or candidate -> distance check (:60-64)
    return 0;
}

// Conservative: false only if a*t^2+b*t+c has no real root in [0.005, 0.995] (see header comment). a, b, c as passed to solveQuadratic.
MSDF_HD bool quadraticMayHaveRootInRange(double a, double b, double c) {
    if (a == 0 || fabs(b) > 1e12*fabs(a))
        return true;                                    // linear / degenerate branch of solveQuadratic: let it decide
    const double lo = .005, hi = .995;
    const double mag = fabs(a)+fabs(b)+fabs(c);
    const double eps = 1e-9*mag;
    const double flo = (a*lo+b)*lo+c, fhi = (a*hi+b)*hi+c;
    if (!(flo > eps && fhi > eps) && !(flo < -eps && fhi < -eps))
        return true;                                    // sign change (or too close to call) between the interval ends
    const double s = flo > 0 ? 1. : -1.;                // sign of f at both ends; a root needs an interior extremum of the opposite sign
    const double as = a*s, bs = b*s, cs = c*s;          // as*t^2+bs*t+cs is positive at both ends
    if (as <= 0)
        return false;                                   // concave: minimum over the interval is at an end
    // convex: vertex at -bs/(2*as); inside (lo, hi) iff 2*as*lo < -bs < 2*as*hi
    if (!(-bs > 2*as*lo*(1-1e-9) && -bs < 2*as*hi*(1+1e-9)))
        return false;                                   // monotone on the interval
    return !(4*as*cs-bs*bs > 1e-9*mag*mag);             // minimum value cs-bs^2/(4as) not clearly positive -> may have roots
}

// Cheapest necessary condition for a diagonal candidate, in fp32: along the diagonal the channel difference is the quadratic with
// Bernstein coefficients (dA, dBC/2, dD) on [0, 1] (f(0) = dA, f(1) = dD). If all three have the same sign with a margin that
// dwarfs the rounding of the reference's float coefficients (dD-dBC+dA, dBC-dA-dA: <= 5e-7*M), the quadratic has no real root in
// [0, 1], so solveQuadratic cannot return one in (0.01, 0.99). False = "cannot have a root".
MSDF_HD bool bernsteinMayHaveRoot(float dA, float dBC, float dD) {
    const float m = fmaxf(fmaxf(fabsf(dA), fabsf(dBC)), fabsf(dD));
    const float eps = 1e-5f*m;
    const float h = .5f*dBC;
    return !((dA > eps && h > eps && dD > eps) || (dA < -eps && h < -eps && dD < -eps));
}

// edgeBetweenTexelsChannel (MSDFErrorCorrection.cpp:154-168) with the fp64 division skipped when t = (a-.5)/(a-b) cannot lie in
// (0, 1): numerator n = a-.5 (exact in fp64) and denominator d = fl32(a-b) must have the same sign and |n| < |d|; both are
// fp32-derived, so n/d < 1 implies n/d <= 1-2^-26 and the rounded quotient is < 1 as well -- the predicate is exact.
MSDF_HD bool edgeBetweenTexelsChannelFast(const float *a, const float *b, int channel) {
    const double n = a[channel]-.5;
    const double d = a[channel]-b[channel];
    if (!((n > 0 && d > 0 && n < d) || (n < 0 && d < 0 && n > d)))
        return false;
    const double t = n/d;
    if (t > 0 && t < 1) {
        float c[3] = { mixf(a[0], b[0], t), mixf(a[1], b[1], t), mixf(a[2], b[2], t) };
        return medianf(c[0], c[1], c[2]) == c[channel];
    }
    return false;
}

MSDF_HD int edgeBetweenTexelsFast(const float *a, const float *b) {
    return 1*(int) edgeBetweenTexelsChannelFast(a, b, 0)+2*(int) edgeBetweenTexelsChannelFast(a, b, 1)+4*(int) edgeBetweenTexelsChannelFast(a, b, 2);
}

// One diagonal channel pair (:291-327) with lazy extremum parameters. Returns judge() of the accumulated flags, OR-ed over roots;
// candidates that need the distance check are handed to `sink(t)`.
template <class Sink>
MSDF_HD int diagonalPairFast(const FastCtx &cx, float am, float dm, const float *a, const float *l, const float *q,
                             float dA, float dBC, float dD, float l0, float q0, float l1, float q1, Sink &sink) {
    const double qa = dD-dBC+dA, qb = dBC-dA-dA, qc = dA;   // float expressions promoted, exactly as passed at :295
#if defined(MSDF_EC_ABLATE) && MSDF_EC_ABLATE == 7                       // measurement only: the diagonal pairs up to the root-range test (a result the compiler cannot fold)
    return quadraticMayHaveRootInRange(qa, qb, qc) ? (qa == 0.12345678 ? 1 : 0) : (qb == 0.12345678 ? 1 : 0);
#endif
    if (!quadraticMayHaveRootInRange(qa, qb, qc))
        return 0;
    double t[2];
    const int solutions = solveQuadratic(t, qa, qb, qc);
#if defined(MSDF_EC_ABLATE) && MSDF_EC_ABLATE == 8                       // measurement only: ... up to solveQuadratic
    return solutions >= 1 && t[0] == 1.2345e300 ? 1 : 0;
#endif
    int verdict = 0;
    for (int i = 0; i < solutions; ++i) {
        if (t[i] > MSDF_ARTIFACT_T_EPSILON && t[i] < 1-MSDF_ARTIFACT_T_EPSILON) {
            const float xm = interpolatedMedianQuad(a, l, q, t[i]);
            int rangeFlags = rangeTest2(cx, 0, 1, t[i], am, dm, xm);
            const double tEx0 = -.5*l0/q0, tEx1 = -.5*l1/q1;               // :361-365 (l, q of the two channels of the pair)
            if (tEx0 > 0 && tEx0 < 1) {
                double tEnd0 = 0, tEnd1 = 1;
                float em0 = am, em1 = dm;
                const float ex = interpolatedMedianQuad(a, l, q, tEx0);
                if (tEx0 > t[i])
                    tEnd1 = tEx0, em1 = ex;
                else
                    tEnd0 = tEx0, em0 = ex;
                rangeFlags |= rangeTest2(cx, tEnd0, tEnd1, t[i], em0, em1, xm);
            }
            if (tEx1 > 0 && tEx1 < 1) {
                double tEnd0 = 0, tEnd1 = 1;
                float em0 = am, em1 = dm;
                const float ex = interpolatedMedianQuad(a, l, q, tEx1);
                if (tEx1 > t[i])
                    tEnd1 = tEx1, em1 = ex;
                else
                    tEnd0 = tEx1, em0 = ex;
                rangeFlags |= rangeTest2(cx, tEnd0, tEnd1, t[i], em0, em1, xm);
            }
            const int v = judge(rangeFlags);
            if (v&2)
                sink(t[i]);
            verdict |= v;
            if (verdict&1)
                return verdict;
        }
    }
    return verdict;
}

// 3x3 neighbourhood of a texel in NATIVE row order, loaded once into registers: v[dy+1][dx+1][channel]; valid bit (dy+1)*3+(dx+1).
// dev[dy+1][dx+1] = fabsf(median(v) - .5f) of that texel: both first-level tests (protectEdges' radius test :201, findErrors' "the texel farther from
// the edge reports" :335, :349) compare only these. A texel's value is computed ONCE -- k_ec_fast does it while it stages the halo in LDS (round 6:
// every lane used to recompute the medians of all eight neighbours, twice: 18 medians of 8 instructions per texel, 15 % of the kernel's VALU stream by
// tools/isa_bbcount.py) -- with the reference's own expression, so the comparisons see the same floats.
struct Neighbourhood {
    float v[3][3][3];
    float dev[3][3];
    unsigned valid;
};

MSDF_HD float ecTexelDeviation(const float *t) { return fabsf(medianf(t[0], t[1], t[2])-.5f); }

MSDF_HD void loadNeighbourhood(Neighbourhood &nb, const SdfView &sdf, int x, int yn) {
    nb.valid = 0;
    MSDF_UNROLL
    for (int dy = -1; dy <= 1; ++dy) {
        MSDF_UNROLL
        for (int dx = -1; dx <= 1; ++dx) {
            const int nx = x+dx, ny = yn+dy;
            const bool in = nx >= 0 && ny >= 0 && nx < sdf.w && ny < sdf.h;
            const float *t = sdf.native(in ? nx : x, in ? ny : yn);
            nb.v[dy+1][dx+1][0] = t[0], nb.v[dy+1][dx+1][1] = t[1], nb.v[dy+1][dx+1][2] = t[2];
            nb.dev[dy+1][dx+1] = ecTexelDeviation(t);
            if (in)
                nb.valid |= 1u<<((dy+1)*3+(dx+1));
        }
    }
}

// protectEdges (MSDFErrorCorrection.cpp:189-250) as a gather, from the register neighbourhood; see protectedByEdges in msdf_ec.hpp. Two
// stages like findErrors below: stage 1 emits the neighbours m = (dy+1)*3+(dx+1) whose pair passes the radius test (:201, :217, :233),
// stage 2 (evaluateProtectPair) runs edgeBetweenTexels on such a pair -- up to three fp64 divisions -- and is queued by the kernel.
template <class Emit>
MSDF_HD void texelProtectPairs(const Neighbourhood &nb, const EcParams &p, Emit &emit) {
    const float sdev = nb.dev[1][1];
    MSDF_UNROLL
    for (int dy = -1; dy <= 1; ++dy) {
        MSDF_UNROLL
        for (int dx = -1; dx <= 1; ++dx) {
            if (!dx && !dy)
                continue;
            if (!(nb.valid&(1u<<((dy+1)*3+(dx+1)))))
                continue;
            const float radius = dy == 0 ? p.radiusH : dx == 0 ? p.radiusV : p.radiusD;
            const float odev = nb.dev[dy+1][dx+1];
            const bool selfIsA = dy > 0 || (dy == 0 && dx > 0);
            const float sum = selfIsA ? sdev+odev : odev+sdev;                     // fabsf(am-.5f)+fabsf(bm-.5f)
            if (sum < radius)
                emit((dy+1)*3+(dx+1));
        }
    }
}

// Is `self` protected through its pair with neighbour m (`other`)? Pair orientation as the reference's sweeps: `a` is the texel of the
// lower native row (same row: the left one).
MSDF_HD bool evaluateProtectPair(const float *self, const float *other, int m) {
    const int dy = m/3-1, dx = m%3-1;
    const bool selfIsA = dy > 0 || (dy == 0 && dx > 0);
    const int mask = selfIsA ? edgeBetweenTexelsFast(self, other) : edgeBetweenTexelsFast(other, self);
    return extremeChannelInMask(self, medianf(self[0], self[1], self[2]), mask);
}

MSDF_HD bool protectedByEdgesNb(const Neighbourhood &nb, const EcParams &p) {
    struct Items {
        unsigned char m[8];
        int n;
        MSDF_HD void operator()(int mm) { m[n++] = (unsigned char) mm; }
    } items;
    items.n = 0;
    texelProtectPairs(nb, p, items);
    for (int i = 0; i < items.n; ++i)
        if (evaluateProtectPair(nb.v[1][1], nb.v[items.m[i]/3][items.m[i]%3], items.m[i]))
            return true;
    return false;
}

// ---- fused findErrors, in two stages so that a wavefront can COMPACT the expensive part ----------------------------------------------
// A texel is tested against its 8 neighbours x 3 channel pairs. Nearly all of those 24 tests are settled by a sign comparison
// (stage 1); the few survivors need fp64 interpolation / a quadratic solve (stage 2). Run per texel in a wavefront, stage 2 executes
// for a test if ANY of the 64 texels needs it, with a handful of lanes active (measured: 7 of the 24 tests per tile, 37 surviving
// (texel, test) items per tile). The kernel therefore queues the items of all its texels and evaluates them densely, one item per lane.
//
// Neighbour k (native row order): 0..3 = (-1,0), (0,-1), (1,0), (0,1); 4..7 = diagonals (k&1 ? +1 : -1, k&2 ? +1 : -1).
// Channel pair j: (j, j+1 mod 3), i.e. (1,0), (2,1), (0,2) in the reference's (minuend, subtrahend) order.
MSDF_HD int ecNeighbourDx(int k) { return k < 4 ? (k == 0 ? -1 : k == 2 ? 1 : 0) : ((k&1) ? 1 : -1); }
MSDF_HD int ecNeighbourDy(int k) { return k < 4 ? (k == 1 ? -1 : k == 3 ? 1 : 0) : ((k&2) ? 1 : -1); }

// Stage 1: emit(k, j) for every test of the centre texel of `nb` that the cheap conditions cannot settle. Exact skips only:
//  * the neighbour must exist and the centre must be the texel farther from the edge (:335, :349);
//  * linear: t = dA/(dA-dB) lies in (0, 1) only if dA and dB have strictly opposite signs;
//  * diagonal: 0 == 0 has no usable root (equation-solver.cpp:13-17); Bernstein sign test (bernsteinMayHaveRoot).
template <class Emit>
MSDF_HD void texelCandidatePairs(const Neighbourhood &nb, Emit &emit) {
    const float *c = nb.v[1][1];
    const float cdev = nb.dev[1][1];
    MSDF_UNROLL
    for (int k = 0; k < 4; ++k) {
        const int dx = k == 0 ? -1 : k == 2 ? 1 : 0, dy = k == 1 ? -1 : k == 3 ? 1 : 0;
        if (!(nb.valid&(1u<<((dy+1)*3+(dx+1)))))
            continue;
        const float *b = nb.v[dy+1][dx+1];
        if (!(cdev >= nb.dev[dy+1][dx+1]))
            continue;
        MSDF_UNROLL
        for (int j = 0; j < 3; ++j) {
            const int i0 = j, i1 = j == 2 ? 0 : j+1;
            const float dA = c[i1]-c[i0], dB = b[i1]-b[i0];
            if ((dA > 0 && dB < 0) || (dA < 0 && dB > 0))
                emit(k, j);
        }
    }
    MSDF_UNROLL
    for (int k = 4; k < 8; ++k) {
        const int dx = (k&1) ? 1 : -1, dy = (k&2) ? 1 : -1;
        if (!(nb.valid&(1u<<((dy+1)*3+(dx+1)))))
            continue;
        const float *d = nb.v[dy+1][dx+1];
        if (!(cdev >= nb.dev[dy+1][dx+1]))
            continue;
        const float *b = nb.v[1][dx+1], *cc = nb.v[dy+1][1];
        MSDF_UNROLL
        for (int j = 0; j < 3; ++j) {
            const int i0 = j, i1 = j == 2 ? 0 : j+1;
            const float dA = c[i1]-c[i0], dBC = b[i1]-b[i0]+cc[i1]-cc[i0], dD = d[i1]-d[i0];
            if (dA == 0 && dBC == 0 && dD == 0)
                continue;
            if (bernsteinMayHaveRoot(dA, dBC, dD))
                emit(k, j);
        }
    }
}

MSDF_HD float pick3(const float *v, int i) { return i == 0 ? v[0] : i == 1 ? v[1] : v[2]; }

// Stage 2: one surviving test. c: centre texel, n: the neighbour k, hb / vc: the horizontal / vertical neighbours on the way to a
// diagonal one (unused for k < 4); any memory (registers, LDS). Returns judge() bits (1: ERROR decided, 2: needs the distance check, then
// also reported as sink(t, dx, dyShape)). Same operands and operations as hasLinearArtifact / hasDiagonalArtifact (:330-381).
template <class Sink>
MSDF_HD int evaluatePair(const float *c, const float *n, const float *hb, const float *vc, const EcParams &p, bool p1, int flip, int k, int j, Sink &sink) {
    FastCtx cx;
    cx.basePass = ecHasBasePass(p);
    cx.shapePass = ecHasShapePass(p);
    cx.p1 = p1;
    cx.pShape = (p.distanceCheck == EC_CHECK_AT_EDGE) ? true : p1;   // protectAll() precedes the shape pass only in that mode (:38-39, :33)
    const int dx = ecNeighbourDx(k), dy = ecNeighbourDy(k);
    const int i0 = j, i1 = j == 2 ? 0 : j+1;
    const float cm = medianf(c[0], c[1], c[2]), nm = medianf(n[0], n[1], n[2]);
#if defined(MSDF_EC_ABLATE) && MSDF_EC_ABLATE == 5                       // measurement only: stage 2 without the linear pairs
    if (k < 4)
        return 0;
#endif
#if defined(MSDF_EC_ABLATE) && MSDF_EC_ABLATE == 6                       // measurement only: stage 2 without the diagonal pairs
    if (k >= 4)
        return 0;
#endif
    if (k < 4) {
        cx.span = dy == 0 ? p.hSpan : p.vSpan;
        const float dA = pick3(c, i1)-pick3(c, i0), dB = pick3(n, i1)-pick3(n, i0);
        const double t = (double) dA/(dA-dB);            // :281
        if (t > MSDF_ARTIFACT_T_EPSILON && t < 1-MSDF_ARTIFACT_T_EPSILON) {
            const float xm = interpolatedMedianLin(c, n, t);
            const int v = judge(rangeTest2(cx, 0, 1, t, cm, nm, xm));
            if (v&2)
                sink(t, dx, flip ? -dy : dy);
            return v;
        }
        return 0;
    }
    cx.span = p.dSpan;
    const float *a = c, *b = hb, *cc = vc, *d = n;       // (texel, horizontal, vertical, diagonal neighbour), :404-407
    const float abc[3] = { a[0]-b[0]-cc[0], a[1]-b[1]-cc[1], a[2]-b[2]-cc[2] };
    const float l[3] = { -a[0]-abc[0], -a[1]-abc[1], -a[2]-abc[2] };
    const float q[3] = { d[0]+abc[0], d[1]+abc[1], d[2]+abc[2] };
    const float a3[3] = { a[0], a[1], a[2] };
    struct DirSink {
        Sink &sink;
        int dx, dy;
        MSDF_HD void operator()(double t) { sink(t, dx, dy); }
    } dirSink = { sink, dx, flip ? -dy : dy };
    const float dA = pick3(a, i1)-pick3(a, i0), dBC = pick3(b, i1)-pick3(b, i0)+pick3(cc, i1)-pick3(cc, i0), dD = pick3(d, i1)-pick3(d, i0);
    return diagonalPairFast(cx, cm, nm, a3, l, q, dA, dBC, dD, pick3(l, i0), pick3(q, i0), pick3(l, i1), pick3(q, i1), dirSink);
}

// Both stages for one texel, test after test (host walk; the kernel queues the items instead). Returns bit0: ERROR decided, bit1: some
// candidate needs the distance check; every such candidate is reported as sink(t, dx, dyShape) (MSDFErrorCorrection.cpp:446-453).
template <class Sink>
MSDF_HD int texelFindFast(const Neighbourhood &nb, const EcParams &p, bool p1, int flip, Sink &sink) {
    struct Items {
        unsigned char k[24], j[24];
        int n;
        MSDF_HD void operator()(int kk, int jj) { k[n] = (unsigned char) kk, j[n] = (unsigned char) jj; ++n; }
    } items;
    items.n = 0;
    texelCandidatePairs(nb, items);
    int verdict = 0;
    for (int i = 0; i < items.n; ++i) {
        const int k = items.k[i], dx = ecNeighbourDx(k), dy = ecNeighbourDy(k);
        verdict |= evaluatePair(nb.v[1][1], nb.v[dy+1][dx+1], nb.v[1][dx+1], nb.v[dy+1][1], p, p1, flip, k, items.j[i], sink);
    }
    return verdict;
}

// Stencil byte of texel (x, yn) [native order] by the fast path: PROTECTED/ERROR as in ecTexelStencil, plus EC_DEFER if the verdict
// hinges on shape-distance checks (each reported through `sink`). corners: (l, b) pairs of protectCorners (:131-134) in shape
// orientation, precomputed per tile.
template <class Sink>
MSDF_HD int ecTexelFast(const SdfView &sdf, const EcParams &p, const int *corners, int nCorners, int x, int yn, Sink &sink) {
    const int ys = sdf.flip ? sdf.h-1-yn : yn;
    Neighbourhood nb;
    loadNeighbourhood(nb, sdf, x, yn);
    int st = 0;
    if (p.mode == EC_MODE_EDGE_PRIORITY) {
        for (int i = 0; i < nCorners; ++i) {
            const int l = corners[2*i], b = corners[2*i+1];
            if ((x == l || x == l+1) && (ys == b || ys == b+1)) {
                st |= EC_PROTECTED;
                break;
            }
        }
        if (!(st&EC_PROTECTED) && protectedByEdgesNb(nb, p))
            st |= EC_PROTECTED;
    } else if (p.mode == EC_MODE_EDGE_ONLY)
        st |= EC_PROTECTED;
    const int verdict = texelFindFast(nb, p, (st&EC_PROTECTED) != 0, sdf.flip, sink);
    if (ecHasBasePass(p) && p.distanceCheck == EC_CHECK_AT_EDGE)
        st |= EC_PROTECTED;                              // protectAll (:38-39)
    if (verdict&1)
        st |= EC_ERROR;
    else if (verdict&2)
        st |= EC_DEFER;
    return st;
}

// The distance check of one deferred candidate: ShapeDistanceChecker::ArtifactClassifier::evaluate past its flag tests (:65-79).
// (x, ys): texel in shape orientation; t, (dx, dy): as reported by texelFindFast; query(V2) -> exact PSDF distance at a shape point.
template <class Query>
MSDF_HD bool ecEvaluateCandidate(const SdfView &sdf, const EcParams &p, int x, int ys, double t, int dx, int dy, const Query &query) {
    const float *msd = sdf.shape(x, ys);
    const V2 tVector = t*mk(dx, dy);
    float oldMSD[3], newMSD[3];
    interpolate3(oldMSD, sdf, mk(x+.5, ys+.5)+tVector);
    const double aWeight = (1-fabs(tVector.x))*(1-fabs(tVector.y));
    const float aPSD = medianf(msd[0], msd[1], msd[2]);
    newMSD[0] = (float) (oldMSD[0]+aWeight*(aPSD-msd[0]));
    newMSD[1] = (float) (oldMSD[1]+aWeight*(aPSD-msd[1]));
    newMSD[2] = (float) (oldMSD[2]+aWeight*(aPSD-msd[2]));
    const float oldPSD = medianf(oldMSD[0], oldMSD[1], oldMSD[2]);
    const float newPSD = medianf(newMSD[0], newMSD[1], newMSD[2]);
    const V2 q = unproject(p.t, mk(x+.5, ys+.5))+mk(tVector.x*p.texelX, tVector.y*p.texelY);
    const float refPSD = mapDistance(p.t, query(q));
    return p.minImproveRatio*fabsf(newPSD-refPSD) < (double) fabsf(oldPSD-refPSD);
}

} // namespace msdfhip
