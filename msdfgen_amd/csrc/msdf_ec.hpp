// msdf_ec.hpp -- MSDF error correction as a pure per-texel function of the pre-correction distance field.
//
// The reference runs a sequence of whole-bitmap passes over a stencil (core/msdf-error-correction.cpp:12-48):
//   protectCorners -> protectEdges -> findErrors(sdf) -> protectAll -> findErrors(sdf, shape) -> apply.
// protectEdges *scatters* into both texels of a pair; every other pass touches only a texel's own stencil byte, and no pass
// modifies the distance field before apply.  So the final stencil byte (and corrected colour) of a texel is a pure function of
// the 3x3 neighbourhood of the ORIGINAL field, the corner list and (for the distance check) the shape.  We evaluate exactly that
// per texel: protectEdges becomes a gather over the 8 pair partners (same pair orientation and operand order as the reference
// sweep, which matters for float rounding), no atomics, one kernel.
#pragma once

#include "msdf_device.hpp"

namespace msdfhip {

enum { EC_ERROR = 1, EC_PROTECTED = 2 };                                      // MSDFErrorCorrection.h:15-20
enum { EC_MODE_DISABLED = 0, EC_MODE_INDISCRIMINATE = 1, EC_MODE_EDGE_PRIORITY = 2, EC_MODE_EDGE_ONLY = 3 }; // generator-config.h:20-29
enum { EC_DO_NOT_CHECK = 0, EC_CHECK_AT_EDGE = 1, EC_ALWAYS_CHECK = 2 };      // generator-config.h:31-38

#define MSDF_ARTIFACT_T_EPSILON .01                                           // MSDFErrorCorrection.cpp:16
#define MSDF_PROTECTION_RADIUS_TOLERANCE 1.001                                // MSDFErrorCorrection.cpp:17

// Pre-correction field of one glyph tile, rows in MEMORY (native) order, tightly packed [h][w][N].
struct SdfView {
    const float *px;
    int w, h, N;
    int flip;     // shape orientation != bitmap orientation
    MSDF_HD const float *native(int x, int y) const { return px+((size_t) y*w+x)*N; }
    MSDF_HD const float *shape(int x, int y) const { return native(x, flip ? h-1-y : y); } // sdf.reorient(shape orientation), MSDFErrorCorrection.cpp:414
};

struct EcParams {
    Xform t;
    double minDeviationRatio, minImproveRatio;
    float radiusH, radiusV, radiusD;  // protectEdges radii, MSDFErrorCorrection.cpp:194, 210, 226
    double hSpan, vSpan, dSpan;       // findErrors spans,   MSDFErrorCorrection.cpp:387-389
    double texelX, texelY;            // unprojectVector(Vector2(1)), :90
    int mode, distanceCheck, overlap, stageLimit;
};

MSDF_HD void ecDerive(EcParams &p) {
    const double d = p.t.mapScale*1.;                                         // distanceMapping(Delta(1)), DistanceMapping.cpp:19-21
    const double lh = vlen(mk(d/p.t.sx, 0/p.t.sy));                           // unprojectVector(Vector2(d, 0)).length()
    const double lv = vlen(mk(0/p.t.sx, d/p.t.sy));
    const double ldg = vlen(mk(d/p.t.sx, d/p.t.sy));
    p.radiusH = (float) (MSDF_PROTECTION_RADIUS_TOLERANCE*lh);
    p.radiusV = (float) (MSDF_PROTECTION_RADIUS_TOLERANCE*lv);
    p.radiusD = (float) (MSDF_PROTECTION_RADIUS_TOLERANCE*ldg);
    p.hSpan = p.minDeviationRatio*lh;
    p.vSpan = p.minDeviationRatio*lv;
    p.dSpan = p.minDeviationRatio*ldg;
    p.texelX = 1/p.t.sx;
    p.texelY = 1/p.t.sy;
}

// ---- protectEdges pieces (MSDFErrorCorrection.cpp:154-187)

MSDF_HD bool edgeBetweenTexelsChannel(const float *a, const float *b, int channel) {
    double t = (a[channel]-.5)/(a[channel]-b[channel]);
    if (t > 0 && t < 1) {
        float c[3] = { mixf(a[0], b[0], t), mixf(a[1], b[1], t), mixf(a[2], b[2], t) };
        return medianf(c[0], c[1], c[2]) == c[channel];
    }
    return false;
}

MSDF_HD int edgeBetweenTexels(const float *a, const float *b) {
    return 1*(int) edgeBetweenTexelsChannel(a, b, 0)+2*(int) edgeBetweenTexelsChannel(a, b, 1)+4*(int) edgeBetweenTexelsChannel(a, b, 2);
}

MSDF_HD bool extremeChannelInMask(const float *msd, float m, int mask) {
    return (mask&1 && msd[0] != m) || (mask&2 && msd[1] != m) || (mask&4 && msd[2] != m);
}

// Is texel (x, y) [native order] marked by the protectEdges sweeps (:189-250)?  Pair (a, b) = (left,right), (bottom,top),
// (left-bottom,right-top), (right-bottom,left-top): `a` is always the texel of the lower native row (same row: the left one).
MSDF_HD bool protectedByEdges(const SdfView &sdf, const EcParams &p, int x, int y) {
    const float *self = sdf.native(x, y);
    const float sm = medianf(self[0], self[1], self[2]);
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            if (!dx && !dy)
                continue;
            const int nx = x+dx, ny = y+dy;
            if (nx < 0 || ny < 0 || nx >= sdf.w || ny >= sdf.h)
                continue;
            const float radius = dy == 0 ? p.radiusH : dx == 0 ? p.radiusV : p.radiusD;
            const float *other = sdf.native(nx, ny);
            const float om = medianf(other[0], other[1], other[2]);
            const bool selfIsA = dy > 0 || (dy == 0 && dx > 0);
            const float *a = selfIsA ? self : other, *b = selfIsA ? other : self;
            const float am = selfIsA ? sm : om, bm = selfIsA ? om : sm;
            if (fabsf(am-.5f)+fabsf(bm-.5f) < radius) {
                int mask = edgeBetweenTexels(a, b);
                if (extremeChannelInMask(self, sm, mask))
                    return true;
            }
        }
    return false;
}

// ---- artifact classifiers (MSDFErrorCorrection.cpp:26-102)

// Abstract access to "exact PSDF distance at a shape point" for the ShapeDistanceChecker; Q::operator()(V2) -> double.
struct NoDistanceQuery { MSDF_HD double operator()(V2) const { return 0; } };

template <class Query>
struct Classifier {
    double span;
    bool protectedFlag;
    bool shapeAware;                 // false: BaseArtifactClassifier, true: ShapeDistanceChecker::ArtifactClassifier
    V2 direction;
    // ShapeDistanceChecker state (:83-101)
    const SdfView *sdf;
    const EcParams *p;
    V2 shapeCoord, sdfCoord;
    const float *msd;
    const Query *query;
};

template <class Query>
MSDF_HD int rangeTest(const Classifier<Query> &c, double at, double bt, double xt, float am, float bm, float xm) { // :30-40
    if ((am > .5f && bm > .5f && xm <= .5f) || (am < .5f && bm < .5f && xm >= .5f) || (!c.protectedFlag && medianf(am, bm, xm) != xm)) {
        double axSpan = (xt-at)*c.span, bxSpan = (bt-xt)*c.span;
        if (!(xm >= am-axSpan && xm <= am+axSpan && xm >= bm-bxSpan && xm <= bm+bxSpan))
            return 3;                // CLASSIFIER_FLAG_CANDIDATE|CLASSIFIER_FLAG_ARTIFACT
        return 1;                    // CLASSIFIER_FLAG_CANDIDATE
    }
    return 0;
}

MSDF_HD double clampd(double n, double b) { return n >= 0 && n <= b ? n : (double) (n > 0)*b; } // arithmetics.hpp:41-43
MSDF_HD int clampi(int n, int b) { return n >= 0 && n <= b ? n : (int) (n > 0)*b; }

MSDF_HD void interpolate3(float *out, const SdfView &sdf, V2 pos) {           // bitmap-interpolation.hpp:10-25 (first 3 channels)
    pos.x = clampd(pos.x, (double) sdf.w);
    pos.y = clampd(pos.y, (double) sdf.h);
    pos.x -= .5, pos.y -= .5;
    int l = (int) floor(pos.x);
    int b = (int) floor(pos.y);
    int r = l+1;
    int t = b+1;
    double lr = pos.x-l;
    double bt = pos.y-b;
    l = clampi(l, sdf.w-1), r = clampi(r, sdf.w-1);
    b = clampi(b, sdf.h-1), t = clampi(t, sdf.h-1);
    const float *lb = sdf.shape(l, b), *rb = sdf.shape(r, b), *lt = sdf.shape(l, t), *rt = sdf.shape(r, t);
    for (int i = 0; i < 3; ++i)
        out[i] = mixf(mixf(lb[i], rb[i], lr), mixf(lt[i], rt[i], lr), bt);
}

template <class Query>
MSDF_HD bool evaluate(const Classifier<Query> &c, double t, float m, int flags) { // :42-44, :58-82
    (void) m;
    if (!c.shapeAware)
        return (flags&2) != 0;
    if (flags&1) {
        if (flags&2)
            return true;
        V2 tVector = t*c.direction;
        float oldMSD[3], newMSD[3];
        V2 sdfCoord = c.sdfCoord+tVector;
        interpolate3(oldMSD, *c.sdf, sdfCoord);
        double aWeight = (1-fabs(tVector.x))*(1-fabs(tVector.y));
        float aPSD = medianf(c.msd[0], c.msd[1], c.msd[2]);
        newMSD[0] = (float) (oldMSD[0]+aWeight*(aPSD-c.msd[0]));
        newMSD[1] = (float) (oldMSD[1]+aWeight*(aPSD-c.msd[1]));
        newMSD[2] = (float) (oldMSD[2]+aWeight*(aPSD-c.msd[2]));
        float oldPSD = medianf(oldMSD[0], oldMSD[1], oldMSD[2]);
        float newPSD = medianf(newMSD[0], newMSD[1], newMSD[2]);
        V2 q = c.shapeCoord+mk(tVector.x*c.p->texelX, tVector.y*c.p->texelY);
        float refPSD = mapDistance(c.p->t, (*c.query)(q));
        return c.p->minImproveRatio*fabsf(newPSD-refPSD) < (double) fabsf(oldPSD-refPSD);
    }
    return false;
}

MSDF_HD float interpolatedMedianLin(const float *a, const float *b, double t) { // :260-266
    return medianf(mixf(a[0], b[0], t), mixf(a[1], b[1], t), mixf(a[2], b[2], t));
}

MSDF_HD float interpolatedMedianQuad(const float *a, const float *l, const float *q, double t) { // :268-275
    return (float) median(t*(t*q[0]+l[0])+a[0], t*(t*q[1]+l[1])+a[1], t*(t*q[2]+l[2])+a[2]);
}

template <class Query>
MSDF_HD bool hasLinearArtifactInner(const Classifier<Query> &cl, float am, float bm, const float *a, const float *b, float dA, float dB) { // :278-288
    double t = (double) dA/(dA-dB);
    if (t > MSDF_ARTIFACT_T_EPSILON && t < 1-MSDF_ARTIFACT_T_EPSILON) {
        float xm = interpolatedMedianLin(a, b, t);
        return evaluate(cl, t, xm, rangeTest(cl, 0, 1, t, am, bm, xm));
    }
    return false;
}

template <class Query>
MSDF_HD bool hasDiagonalArtifactInner(const Classifier<Query> &cl, float am, float dm, const float *a, const float *l, const float *q,
                                      float dA, float dBC, float dD, double tEx0, double tEx1) { // :291-327
    double t[2];
    int solutions = solveQuadratic(t, dD-dBC+dA, dBC-dA-dA, dA);
    for (int i = 0; i < solutions; ++i) {
        if (t[i] > MSDF_ARTIFACT_T_EPSILON && t[i] < 1-MSDF_ARTIFACT_T_EPSILON) {
            float xm = interpolatedMedianQuad(a, l, q, t[i]);
            int rangeFlags = rangeTest(cl, 0, 1, t[i], am, dm, xm);
            if (tEx0 > 0 && tEx0 < 1) {
                double tEnd0 = 0, tEnd1 = 1;
                float em0 = am, em1 = dm;
                float ex = interpolatedMedianQuad(a, l, q, tEx0);
                if (tEx0 > t[i])
                    tEnd1 = tEx0, em1 = ex;
                else
                    tEnd0 = tEx0, em0 = ex;
                rangeFlags |= rangeTest(cl, tEnd0, tEnd1, t[i], em0, em1, xm);
            }
            if (tEx1 > 0 && tEx1 < 1) {
                double tEnd0 = 0, tEnd1 = 1;
                float em0 = am, em1 = dm;
                float ex = interpolatedMedianQuad(a, l, q, tEx1);
                if (tEx1 > t[i])
                    tEnd1 = tEx1, em1 = ex;
                else
                    tEnd0 = tEx1, em0 = ex;
                rangeFlags |= rangeTest(cl, tEnd0, tEnd1, t[i], em0, em1, xm);
            }
            if (evaluate(cl, t[i], xm, rangeFlags))
                return true;
        }
    }
    return false;
}

template <class Query>
MSDF_HD bool hasLinearArtifact(const Classifier<Query> &cl, float am, const float *a, const float *b) { // :330-342
    float bm = medianf(b[0], b[1], b[2]);
    if (!(fabsf(am-.5f) >= fabsf(bm-.5f)))
        return false;
    MSDF_NOUNROLL
    for (int k = 0; k < 3; ++k) {                 // channel pairs (1,0), (2,1), (0,2)
        const int i0 = k, i1 = k == 2 ? 0 : k+1;
        if (hasLinearArtifactInner(cl, am, bm, a, b, a[i1]-a[i0], b[i1]-b[i0]))
            return true;
    }
    return false;
}

template <class Query>
MSDF_HD bool hasDiagonalArtifact(const Classifier<Query> &cl, float am, const float *a, const float *b, const float *c, const float *d) { // :345-381
    float dm = medianf(d[0], d[1], d[2]);
    if (fabsf(am-.5f) >= fabsf(dm-.5f)) {
        float abc[3] = { a[0]-b[0]-c[0], a[1]-b[1]-c[1], a[2]-b[2]-c[2] };
        float l[3] = { -a[0]-abc[0], -a[1]-abc[1], -a[2]-abc[2] };
        float q[3] = { d[0]+abc[0], d[1]+abc[1], d[2]+abc[2] };
        double tEx[3] = { -.5*l[0]/q[0], -.5*l[1]/q[1], -.5*l[2]/q[2] };
        MSDF_NOUNROLL
        for (int k = 0; k < 3; ++k) {             // channel pairs (1,0), (2,1), (0,2)
            const int i0 = k, i1 = k == 2 ? 0 : k+1;
            if (hasDiagonalArtifactInner(cl, am, dm, a, l, q, a[i1]-a[i0], b[i1]-b[i0]+c[i1]-c[i0], d[i1]-d[i0], tEx[i0], tEx[i1]))
                return true;
        }
        return false;
    }
    return false;
}

// Body shared by findErrors<N>(sdf) (:383-410; native order, base classifier) and findErrors<CC,N>(sdf, shape) (:412-457; shape
// orientation, distance-checking classifier). `shapeOriented` selects the row order in which (x, y) and the neighbours are taken.
// The reference's short-circuit || chain over the 8 neighbours is an OR of side-effect-free tests, so it is evaluated here as two
// rolled loops (4 axis neighbours, 4 diagonal neighbours) to keep one copy of the classifier code per kernel.
template <class Query>
MSDF_HD bool texelHasError(const SdfView &sdf, const EcParams &p, int x, int y, bool shapeOriented, bool protectedFlag, const Query *query) {
    const int w = sdf.w, h = sdf.h;
    #define MSDF_AT(X, Y) (shapeOriented ? sdf.shape(X, Y) : sdf.native(X, Y))
    const float *c = MSDF_AT(x, y);
    const float cm = medianf(c[0], c[1], c[2]);
    Classifier<Query> cl;
    cl.protectedFlag = protectedFlag;
    cl.shapeAware = shapeOriented;
    cl.sdf = &sdf;
    cl.p = &p;
    cl.shapeCoord = unproject(p.t, mk(x+.5, y+.5));
    cl.sdfCoord = mk(x+.5, y+.5);
    cl.msd = c;
    cl.query = query;
    // l, b, r, t (MSDFErrorCorrection.cpp:400-403 / :446-449)
    MSDF_NOUNROLL
    for (int k = 0; k < 4; ++k) {
        const int dx = k == 0 ? -1 : k == 2 ? 1 : 0, dy = k == 1 ? -1 : k == 3 ? 1 : 0;
        const int nx = x+dx, ny = y+dy;
        if (nx < 0 || ny < 0 || nx >= w || ny >= h)
            continue;
        cl.span = dy == 0 ? p.hSpan : p.vSpan;
        cl.direction = mk(dx, dy);
        if (hasLinearArtifact(cl, cm, c, MSDF_AT(nx, ny)))
            return true;
    }
    // (l,b) (r,b) (l,t) (r,t) (:404-407 / :450-453): hasDiagonalArtifact(c, horizontal neighbour, vertical neighbour, diagonal neighbour)
    cl.span = p.dSpan;
    MSDF_NOUNROLL
    for (int k = 0; k < 4; ++k) {
        const int dx = (k&1) ? 1 : -1, dy = (k&2) ? 1 : -1;
        const int nx = x+dx, ny = y+dy;
        if (nx < 0 || ny < 0 || nx >= w || ny >= h)
            continue;
        cl.direction = mk(dx, dy);
        if (hasDiagonalArtifact(cl, cm, c, MSDF_AT(nx, y), MSDF_AT(x, ny), MSDF_AT(nx, ny)))
            return true;
    }
    #undef MSDF_AT
    return false;
}

// protectCorners as a gather (MSDFErrorCorrection.cpp:121-151): texel (x, ys) [shape orientation] is protected iff it is one of
// the 2x2 texels around floor(project(corner)-.5) of some colour-change corner. recs: the glyph's records (REC_CORNER flags).
MSDF_HD bool protectedByCorners(const EdgeRec *rec, int nE, const Xform &t, int x, int ys) {
    for (int i = 0; i < nE; ++i)
        if (rec[i].flags&REC_CORNER) {
            V2 pp = project(t, ld(rec[i].p0));
            int l = (int) floor(pp.x-.5);
            int b = (int) floor(pp.y-.5);
            if ((x == l || x == l+1) && (ys == b || ys == b+1))
                return true;
        }
    return false;
}

// Final stencil byte of texel (x, yn) [native order] after the configured pipeline (core/msdf-error-correction.cpp:12-48).
// Stage snapshots (p.stageLimit 1..4) return the byte as it stands after that stage.
template <class Query>
MSDF_HD int ecTexelStencil(const SdfView &sdf, const EcParams &p, const EdgeRec *rec, int nE, int x, int yn, const Query *query) {
    const int ys = sdf.flip ? sdf.h-1-yn : yn;
    int st = 0;
    if (p.mode == EC_MODE_EDGE_PRIORITY) {
        if (protectedByCorners(rec, nE, p.t, x, ys))
            st |= EC_PROTECTED;
        if (p.stageLimit == 1)
            return st;
        if (!(st&EC_PROTECTED) && protectedByEdges(sdf, p, x, yn))
            st |= EC_PROTECTED;
        if (p.stageLimit == 2)
            return st;
    } else if (p.mode == EC_MODE_EDGE_ONLY)
        st |= EC_PROTECTED;
    if (p.distanceCheck == EC_DO_NOT_CHECK || (p.distanceCheck == EC_CHECK_AT_EDGE && p.mode != EC_MODE_EDGE_ONLY)) {
        if (texelHasError(sdf, p, x, yn, false, (st&EC_PROTECTED) != 0, (const Query *) 0))
            st |= EC_ERROR;
        if (p.stageLimit == 3)
            return st;
        if (p.distanceCheck == EC_CHECK_AT_EDGE)
            st |= EC_PROTECTED;
    }
    if (p.distanceCheck == EC_ALWAYS_CHECK || p.distanceCheck == EC_CHECK_AT_EDGE) {
        if (!(st&EC_ERROR) && texelHasError(sdf, p, x, ys, true, (st&EC_PROTECTED) != 0, query))
            st |= EC_ERROR;
    }
    return st;
}

} // namespace msdfhip
