// msdf_shapeprep.hpp -- shape preparation on the flat edge buffer: what a caller runs on every glyph before the
// generators -- Shape::normalize (core/Shape.cpp:65-92) and edgeColoringSimple (core/edge-coloring.cpp:68-142) -- so that raw outlines
// can be uploaded once and never come back to the host.
//
// The reference walks a contour edge by edge (normalize modifies the previous and the current edge at every cusp; the colouring carries its
// colour / seed state from contour to contour of a shape). Here the LANES ARE THE EDGES (round 4): a wavefront takes a contour (normalize:
// k_prep_normalize_flat, one lane per output edge slot) or a glyph (colouring: k_prep_colour_wave -- corners by ballot, the colour sequence of a
// contour from prefix popcounts over the corner mask, the seed state carried across the glyph's contours in scalar registers), no per-thread
// scratch. Edge counts change (a single-edge contour is split in thirds; a one-corner contour with fewer than three edges is split in six),
// hence the passes   normalizedCount -> normalize -> colouredCount -> (device prefix over contours) -> colour.
// The helpers below (deconvergeEdge, curveOrderingAt, switchColor, symmetricalTrichotomy) are the per-edge leaves all forms share. Of the serial
// drivers of rounds 1-3, normalizeContour is still the device's route for the rare contour with a convergent junction (pass 2 of the flat
// normalize: one lane per FLAGGED contour); the serial colouring survives only in tests/hostemu, the twin the tests compare against the oracle.
// All arithmetic is the reference's, operation for operation (fp64, no contraction); sin(angleThreshold) is taken on the host.
#pragma once

#include "msdf_device.hpp"

namespace msdfhip {

struct PrepEdge {
    int type, color;
    V2 p[4];
};

// Flat edge arrays (the C ABI's shape model: 8 doubles per edge, unused control points 0).
struct EdgeArrays {
    double *points;
    uint8_t *types, *colors;
};

MSDF_HD PrepEdge loadEdge(const EdgeArrays &a, int e) {
    PrepEdge r;
    r.type = a.types[e], r.color = a.colors ? a.colors[e] : 7;
    for (int i = 0; i < 4; ++i)
        r.p[i] = mk(a.points[8*(size_t) e+2*i], a.points[8*(size_t) e+2*i+1]);
    return r;
}

MSDF_HD void storeEdge(const EdgeArrays &a, int e, const PrepEdge &v) {
    for (int i = 0; i < 4; ++i) {
        a.points[8*(size_t) e+2*i] = i <= v.type ? v.p[i].x : 0.;
        a.points[8*(size_t) e+2*i+1] = i <= v.type ? v.p[i].y : 0.;
    }
    a.types[e] = (uint8_t) v.type;
    a.colors[e] = (uint8_t) v.color;
}

MSDF_HD V2 vmix(V2 a, V2 b, double w) { return (1.-w)*a+w*b; }                       // arithmetics.hpp:27-31 on Vector2
MSDF_HD bool veq(V2 a, V2 b) { return a.x == b.x && a.y == b.y; }
MSDF_HD bool vnz(V2 a) { return a.x != 0 || a.y != 0; }
MSDF_HD V2 vorthogonal(V2 a, bool polarity) { return polarity ? mk(-a.y, a.x) : mk(a.y, -a.x); }   // Vector2.hpp:49-51

MSDF_HD V2 edgePoint(const PrepEdge &e, double t) {                                   // edge-segments.cpp:108-119
    if (e.type == 1)
        return vmix(e.p[0], e.p[1], t);
    if (e.type == 2)
        return vmix(vmix(e.p[0], e.p[1], t), vmix(e.p[1], e.p[2], t), t);
    const V2 p12 = vmix(e.p[1], e.p[2], t);
    return vmix(vmix(vmix(e.p[0], e.p[1], t), p12, t), vmix(p12, vmix(e.p[2], e.p[3], t), t), t);
}

MSDF_HD V2 edgeDirection(const PrepEdge &e, double t) {                               // edge-segments.cpp:121-139
    if (e.type == 1)
        return e.p[1]-e.p[0];
    if (e.type == 2) {
        const V2 tangent = vmix(e.p[1]-e.p[0], e.p[2]-e.p[1], t);
        return vnz(tangent) ? tangent : e.p[2]-e.p[0];
    }
    const V2 tangent = vmix(vmix(e.p[1]-e.p[0], e.p[2]-e.p[1], t), vmix(e.p[2]-e.p[1], e.p[3]-e.p[2], t), t);
    if (!vnz(tangent)) {
        if (t == 0) return e.p[2]-e.p[0];
        if (t == 1) return e.p[3]-e.p[1];
    }
    return tangent;
}

MSDF_HD PrepEdge mkEdge(int type, int color, V2 a, V2 b, V2 c, V2 d) {
    PrepEdge e;
    e.type = type, e.color = color;
    e.p[0] = a, e.p[1] = b, e.p[2] = c, e.p[3] = d;
    return e;
}

// The k-th part (0, 1, 2) of EdgeSegment::splitInThirds (edge-segments.cpp:508-527); no array, so that a lane can take one part.
MSDF_HD PrepEdge splitThird(const PrepEdge &e, int k) {
    const V2 *p = e.p;
    const V2 z = mk(0, 0);
    if (e.type == 1) {
        if (k == 0)
            return mkEdge(1, e.color, p[0], edgePoint(e, 1/3.), z, z);
        if (k == 1)
            return mkEdge(1, e.color, edgePoint(e, 1/3.), edgePoint(e, 2/3.), z, z);
        return mkEdge(1, e.color, edgePoint(e, 2/3.), p[1], z, z);
    }
    if (e.type == 2) {
        if (k == 0)
            return mkEdge(2, e.color, p[0], vmix(p[0], p[1], 1/3.), edgePoint(e, 1/3.), z);
        if (k == 1)
            return mkEdge(2, e.color, edgePoint(e, 1/3.), vmix(vmix(p[0], p[1], 5/9.), vmix(p[1], p[2], 4/9.), .5), edgePoint(e, 2/3.), z);
        return mkEdge(2, e.color, edgePoint(e, 2/3.), vmix(p[1], p[2], 2/3.), p[2], z);
    }
    if (k == 0)
        return mkEdge(3, e.color, p[0], veq(p[0], p[1]) ? p[0] : vmix(p[0], p[1], 1/3.),
                      vmix(vmix(p[0], p[1], 1/3.), vmix(p[1], p[2], 1/3.), 1/3.), edgePoint(e, 1/3.));
    if (k == 1)
        return mkEdge(3, e.color, edgePoint(e, 1/3.),
                      vmix(vmix(vmix(p[0], p[1], 1/3.), vmix(p[1], p[2], 1/3.), 1/3.), vmix(vmix(p[1], p[2], 1/3.), vmix(p[2], p[3], 1/3.), 1/3.), 2/3.),
                      vmix(vmix(vmix(p[0], p[1], 2/3.), vmix(p[1], p[2], 2/3.), 2/3.), vmix(vmix(p[1], p[2], 2/3.), vmix(p[2], p[3], 2/3.), 2/3.), 1/3.),
                      edgePoint(e, 2/3.));
    return mkEdge(3, e.color, edgePoint(e, 2/3.), vmix(vmix(p[1], p[2], 2/3.), vmix(p[2], p[3], 2/3.), 2/3.),
                  veq(p[2], p[3]) ? p[3] : vmix(p[2], p[3], 2/3.), p[3]);
}

MSDF_HD void splitInThirds(const PrepEdge &e, PrepEdge *part) {
    for (int k = 0; k < 3; ++k)
        part[k] = splitThird(e, k);
}

MSDF_HD void deconvergeEdge(PrepEdge &e, int param, V2 vector) {                      // Shape.cpp:44-62
    if (e.type == 2) {                                                                // QuadraticSegment::convertToCubic, edge-segments.cpp:529-531
        const V2 a = e.p[0], b = e.p[1], c = e.p[2];
        e = mkEdge(3, e.color, a, vmix(a, b, 2/3.), vmix(b, c, 1/3.), c);
    }
    if (e.type == 3) {
        if (param == 0)
            e.p[1] = e.p[1]+vlen(e.p[1]-e.p[0])*vector;
        else
            e.p[2] = e.p[2]+vlen(e.p[2]-e.p[3])*vector;
    }
}

MSDF_HD void simplifyDegenerateCurve(V2 *cp, int &order) {                            // convergent-curve-ordering.cpp:34-46
    if (order == 3 && (veq(cp[1], cp[0]) || veq(cp[1], cp[3])) && (veq(cp[2], cp[0]) || veq(cp[2], cp[3]))) {
        cp[1] = cp[3];
        order = 1;
    }
    if (order == 2 && (veq(cp[1], cp[0]) || veq(cp[1], cp[2]))) {
        cp[1] = cp[2];
        order = 1;
    }
    if (order == 1 && veq(cp[0], cp[1]))
        order = 0;
}

MSDF_HD int signOf(double n) { return (0 < n)-(n < 0); }                              // arithmetics.hpp:53-55

MSDF_HD int curveOrderingAt(const V2 *corner, int before, int after) {                // convergent-curve-ordering.cpp:48-119
    if (!(before > 0 && after > 0))
        return 0;
    V2 a1, a2 = mk(0, 0), a3 = mk(0, 0), b1, b2 = mk(0, 0), b3 = mk(0, 0);
    a1 = corner[-1]-corner[0];
    b1 = corner[1]-corner[0];
    if (before >= 2)
        a2 = corner[-2]-corner[-1]-a1;
    if (after >= 2)
        b2 = corner[2]-corner[1]-b1;
    if (before >= 3) {
        a3 = corner[-3]-corner[-2]-(corner[-2]-corner[-1])-a2;
        a2 = 3.*a2;
    }
    if (after >= 3) {
        b3 = corner[3]-corner[2]-(corner[2]-corner[1])-b2;
        b2 = 3.*b2;
    }
    a1 = (double) before*a1;
    b1 = (double) after*b1;
    double d;
    if (vnz(a1) && vnz(b1)) {
        const double as = vlen(a1), bs = vlen(b1);
        if ((d = as*cross(a1, b2)+bs*cross(a2, b1)) != 0)
            return signOf(d);
        if ((d = as*as*cross(a1, b3)+as*bs*cross(a2, b2)+bs*bs*cross(a3, b1)) != 0)
            return signOf(d);
        if ((d = as*cross(a2, b3)+bs*cross(a3, b2)) != 0)
            return signOf(d);
        return signOf(cross(a3, b3));
    }
    int s = 1;
    if (vnz(a1)) {
        b1 = a1;
        a1 = b2, b2 = a2, a2 = a1;
        a1 = b3, b3 = a3, a3 = a1;
        s = -1;
    }
    if (vnz(b1)) {
        if ((d = cross(a3, b1)) != 0)
            return s*signOf(d);
        if ((d = cross(a2, b2)) != 0)
            return s*signOf(d);
        if ((d = cross(a3, b2)) != 0)
            return s*signOf(d);
        if ((d = cross(a2, b3)) != 0)
            return s*signOf(d);
        return s*signOf(cross(a3, b3));
    }
    if ((d = sqrt(vlen(a2))*cross(a2, b3)+sqrt(vlen(b2))*cross(a3, b2)) != 0)
        return signOf(d);
    return signOf(cross(a3, b3));
}

MSDF_HD int convergentCurveOrdering(const PrepEdge &a, const PrepEdge &b) {           // convergent-curve-ordering.cpp:121-138
    V2 cps[12];
    for (int i = 0; i < 12; ++i)
        cps[i] = mk(0, 0);
    V2 *corner = cps+4, *tmp = cps+8;
    int ao = a.type, bo = b.type;
    for (int i = 0; i <= ao; ++i)
        tmp[i] = a.p[i];
    for (int i = 0; i <= bo; ++i)
        corner[i] = b.p[i];
    if (!veq(tmp[ao], corner[0]))
        return 0;
    simplifyDegenerateCurve(tmp, ao);
    simplifyDegenerateCurve(corner, bo);
    for (int i = 0; i < ao; ++i)
        corner[i-ao] = tmp[i];
    return curveOrderingAt(corner, ao, bo);
}

#define MSDF_CORNER_DOT_EPSILON .000001                 // core/Shape.h:12
#define MSDF_DECONVERGE_OVERSHOOT 1.11111111111111111   // core/Shape.cpp:7

MSDF_HD int normalizedCount(int n) { return n == 1 ? 3 : n; }

// Shape::normalize on one contour: n input edges starting at `ib` -> normalizedCount(n) output edges starting at `ob`.
MSDF_HD void normalizeContour(const EdgeArrays &in, int ib, int n, const EdgeArrays &out, int ob) {
    if (n == 1) {
        PrepEdge parts[3];
        splitInThirds(loadEdge(in, ib), parts);
        for (int i = 0; i < 3; ++i)
            storeEdge(out, ob+i, parts[i]);
        return;
    }
    for (int i = 0; i < n; ++i)
        storeEdge(out, ob+i, loadEdge(in, ib+i));
    int prev = n-1;
    for (int i = 0; i < n; ++i) {                                                     // push apart convergent edge segments
        PrepEdge pe = loadEdge(out, ob+prev), ce = loadEdge(out, ob+i);
        const V2 prevDir = normalize(edgeDirection(pe, 1), false);
        const V2 curDir = normalize(edgeDirection(ce, 0), false);
        if (dot(prevDir, curDir) < MSDF_CORNER_DOT_EPSILON-1) {
            const double factor = MSDF_DECONVERGE_OVERSHOOT*sqrt(1-(MSDF_CORNER_DOT_EPSILON-1)*(MSDF_CORNER_DOT_EPSILON-1))/(MSDF_CORNER_DOT_EPSILON-1);
            V2 axis = factor*normalize(curDir-prevDir, false);
            if (convergentCurveOrdering(pe, ce) < 0)
                axis = -axis;
            deconvergeEdge(pe, 1, vorthogonal(axis, true));
            storeEdge(out, ob+prev, pe);
            if (prev == i)                                                            // cannot happen for n >= 2; keeps the two views coherent anyway
                ce = pe;
            deconvergeEdge(ce, 0, vorthogonal(axis, false));
            storeEdge(out, ob+i, ce);
        }
        prev = i;
    }
}

// Shape::normalize with LANES = EDGES over the whole batch (round 4). What is sequential in Shape.cpp:74-90 is only the cusp repair: junction i
// reads edge i-1 as junction i-1 left it. A junction that is NOT convergent modifies nothing, so as long as no junction of a contour is convergent
// ON THE RAW EDGES the sequential pass modifies nothing either (induction from junction 0, which reads raw edges) -- and cusps are rare (none in
// 8 192 DejaVu glyphs). Pass 1, one lane per OUTPUT edge: copy (or the lane's third of a single-edge contour, Shape.cpp:67-73) and the convergence
// test of the junction before it; a contour with a convergent junction is flagged and redone by normalizeContour (pass 2, one lane per flagged contour).
MSDF_HD bool junctionConvergent(const PrepEdge &pe, const PrepEdge &ce) {
    const V2 prevDir = normalize(edgeDirection(pe, 1), false);
    const V2 curDir = normalize(edgeDirection(ce, 0), false);
    return dot(prevDir, curDir) < MSDF_CORNER_DOT_EPSILON-1;
}

// Output edge `i` of a contour of n raw edges at `ib` -> slot ob+i. Returns true if the junction before edge i is convergent.
MSDF_HD bool normalizeEdgeFlat(const EdgeArrays &in, int ib, int n, const EdgeArrays &out, int ob, int i, bool doNormalize) {
    if (doNormalize && n == 1) {
        storeEdge(out, ob+i, splitThird(loadEdge(in, ib), i));
        return false;
    }
    const PrepEdge ce = loadEdge(in, ib+i);
    storeEdge(out, ob+i, ce);
    return doNormalize && junctionConvergent(loadEdge(in, ib+(i+n-1)%n), ce);
}

// ---- edgeColoringSimple

MSDF_HD int seedExtract2(unsigned long long &seed) { const int v = (int) seed&1; seed >>= 1; return v; }   // edge-coloring.cpp:36-40
MSDF_HD int seedExtract3(unsigned long long &seed) { const int v = (int) (seed%3); seed /= 3; return v; }  // :42-46
MSDF_HD int initColor(unsigned long long &seed) {                                     // :48-51: CYAN, MAGENTA, YELLOW
    const int v = seedExtract3(seed);
    return v == 0 ? 6 : v == 1 ? 5 : 3;
}
MSDF_HD void switchColor(int &color, unsigned long long &seed) {                      // :53-56
    const int shifted = color<<(1+seedExtract2(seed));
    color = (shifted|shifted>>3)&7;
}
MSDF_HD void switchColorBanned(int &color, unsigned long long &seed, int banned) {    // :58-64
    const int combined = color&banned;
    if (combined == 1 || combined == 2 || combined == 4)
        color = combined^7;
    else
        switchColor(color, seed);
}
MSDF_HD int symmetricalTrichotomy(int position, int n) { return (int) (3+2.875*position/(n-1)-1.4375+.5)-3; }   // :18-20
MSDF_HD bool isCorner(V2 aDir, V2 bDir, double crossThreshold) { return dot(aDir, bDir) <= 0 || fabs(cross(aDir, bDir)) > crossThreshold; }   // :22-24

// Corner i of a contour: the junction before edge i (edge-coloring.cpp:76-85). Calls visit(i) for every corner in order; returns the count.
template <class Visit>
MSDF_HD int contourCorners(const EdgeArrays &a, int b, int n, double crossThreshold, Visit &visit) {
    if (n == 0)
        return 0;
    int count = 0;
    V2 prevDirection = edgeDirection(loadEdge(a, b+n-1), 1);
    for (int i = 0; i < n; ++i) {
        const PrepEdge e = loadEdge(a, b+i);
        if (isCorner(normalize(prevDirection, false), normalize(edgeDirection(e, 0), false), crossThreshold)) {
            visit(i);
            ++count;
        }
        prevDirection = edgeDirection(e, 1);
    }
    return count;
}

struct NoVisit { MSDF_HD void operator()(int) const { } };
struct FirstVisit { int first; MSDF_HD void operator()(int i) { if (first < 0) first = i; } };

// Edges the contour has after colouring: a one-corner contour with fewer than three edges is split (edge-coloring.cpp:106-123).
MSDF_HD int colouredCount(const EdgeArrays &a, int b, int n, double crossThreshold) {
    if (n == 0 || n >= 3)
        return n;
    NoVisit none;
    return contourCorners(a, b, n, crossThreshold, none) == 1 ? 3*n : n;
}

// A one-corner contour of fewer than three edges: every edge split in thirds, colours by thirds (edge-coloring.cpp:106-123).
MSDF_HD void teardropSplit(const EdgeArrays &in, int ib, int n, const EdgeArrays &out, int ob, int corner, const int colors[3]) {
    PrepEdge parts[6];
    splitInThirds(loadEdge(in, ib), parts+3*corner);
    if (n >= 2) {
        splitInThirds(loadEdge(in, ib+1), parts+3-3*corner);
        parts[0].color = parts[1].color = colors[0];
        parts[2].color = parts[3].color = colors[1];
        parts[4].color = parts[5].color = colors[2];
    } else {
        parts[0].color = colors[0];
        parts[1].color = colors[1];
        parts[2].color = colors[2];
    }
    for (int i = 0; i < 3*n; ++i)
        storeEdge(out, ob+i, parts[i]);
}

// edgeColoringSimple of one contour (edge-coloring.cpp:73-141): input edges [ib, ib+n) of `in`, output edges from `ob` of `out`
// (colouredCount of them). color / seed: the shape-wide running state.
MSDF_HD void colourContour(const EdgeArrays &in, int ib, int n, const EdgeArrays &out, int ob, double crossThreshold, int &color, unsigned long long &seed) {
    if (n == 0)
        return;
    FirstVisit fv;
    fv.first = -1;
    const int nCorners = contourCorners(in, ib, n, crossThreshold, fv);
    if (nCorners == 0) {                                                              // smooth contour
        switchColor(color, seed);
        for (int i = 0; i < n; ++i) {
            PrepEdge e = loadEdge(in, ib+i);
            e.color = color;
            storeEdge(out, ob+i, e);
        }
    } else if (nCorners == 1) {                                                       // "teardrop"
        int colors[3];
        switchColor(color, seed);
        colors[0] = color;
        colors[1] = 7;
        switchColor(color, seed);
        colors[2] = color;
        const int corner = fv.first;
        if (n >= 3) {
            for (int i = 0; i < n; ++i) {
                const int index = (corner+i)%n;
                PrepEdge e = loadEdge(in, ib+index);
                e.color = colors[1+symmetricalTrichotomy(i, n)];
                storeEdge(out, ob+index, e);
            }
        } else
            teardropSplit(in, ib, n, out, ob, corner, colors);
    } else {                                                                          // multiple corners
        // The reference walks from the first corner and switches colour when it reaches the next one (corners[spline+1] == index):
        // with the corners in increasing order that is "edge `index` starts at a corner other than the first", decided on the fly here.
        int spline = 0;
        const int start = fv.first;
        switchColor(color, seed);
        const int initialColor = color;
        V2 prevDirection = edgeDirection(loadEdge(in, ib+(start+n-1)%n), 1);
        for (int i = 0; i < n; ++i) {
            const int index = (start+i)%n;
            PrepEdge e = loadEdge(in, ib+index);
            const bool corner = isCorner(normalize(prevDirection, false), normalize(edgeDirection(e, 0), false), crossThreshold);
            if (i > 0 && corner && spline+1 < nCorners) {
                ++spline;
                switchColorBanned(color, seed, (spline == nCorners-1)*initialColor);
            }
            e.color = color;
            storeEdge(out, ob+index, e);
            prevDirection = edgeDirection(e, 1);
        }
    }
}

// ---- edgeColoringInkTrap (edge-coloring.cpp:151-258)

MSDF_HD double estimateEdgeLength(const PrepEdge &e) {                                // :26-34, MSDFGEN_EDGE_LENGTH_PRECISION 4
    double len = 0;
    V2 prev = edgePoint(e, 0);
    for (int i = 1; i <= 4; ++i) {
        const V2 cur = edgePoint(e, 1./4*i);
        len += vlen(cur-prev);
        prev = cur;
    }
    return len;
}

// Per-corner work arrays (EdgeColoringInkTrapCorner, :144-149); a contour with n edges has at most n corners: slots [base, base+n).
struct CornerWork {
    int *index;
    double *length;          // prevEdgeLengthEstimate
    uint8_t *minor, *color;
};

// edgeColoringInkTrap of one contour: same contract as colourContour; cw: scratch for this contour's corners at `cb`.
MSDF_HD void colourContourInkTrap(const EdgeArrays &in, int ib, int n, const EdgeArrays &out, int ob, double crossThreshold, int &color, unsigned long long &seed,
                                  const CornerWork &cw, int cb) {
    if (n == 0)
        return;
    double splineLength = 0;
    int nCorners = 0;
    {
        V2 prevDirection = edgeDirection(loadEdge(in, ib+n-1), 1);
        for (int i = 0; i < n; ++i) {
            const PrepEdge e = loadEdge(in, ib+i);
            if (isCorner(normalize(prevDirection, false), normalize(edgeDirection(e, 0), false), crossThreshold)) {
                cw.index[cb+nCorners] = i, cw.length[cb+nCorners] = splineLength, cw.minor[cb+nCorners] = 0, cw.color[cb+nCorners] = 0;
                ++nCorners;
                splineLength = 0;
            }
            splineLength += estimateEdgeLength(e);
            prevDirection = edgeDirection(e, 1);
        }
    }
    if (nCorners <= 1) {                                                              // smooth / teardrop: exactly edgeColoringSimple's branches (:174-213)
        colourContour(in, ib, n, out, ob, crossThreshold, color, seed);
        return;
    }
    const int cornerCount = nCorners;
    int majorCornerCount = cornerCount;
    if (cornerCount > 3) {
        cw.length[cb] += splineLength;
        for (int i = 0; i < cornerCount; ++i)
            if (cw.length[cb+i] > cw.length[cb+(i+1)%cornerCount] && cw.length[cb+(i+1)%cornerCount] < cw.length[cb+(i+2)%cornerCount]) {
                cw.minor[cb+i] = 1;
                --majorCornerCount;
            }
    }
    int initialColor = 0;
    for (int i = 0; i < cornerCount; ++i)
        if (!cw.minor[cb+i]) {
            --majorCornerCount;
            switchColorBanned(color, seed, !majorCornerCount*initialColor);
            cw.color[cb+i] = (uint8_t) color;
            if (!initialColor)
                initialColor = color;
        }
    for (int i = 0; i < cornerCount; ++i) {
        if (cw.minor[cb+i]) {
            const int nextColor = cw.color[cb+(i+1)%cornerCount];
            cw.color[cb+i] = (uint8_t) ((color&nextColor)^7);
        } else
            color = cw.color[cb+i];
    }
    int spline = 0;
    const int start = cw.index[cb];
    color = cw.color[cb];
    for (int i = 0; i < n; ++i) {
        const int index = (start+i)%n;
        if (spline+1 < cornerCount && cw.index[cb+spline+1] == index)
            color = cw.color[cb+(++spline)];
        PrepEdge e = loadEdge(in, ib+index);
        e.color = color;
        storeEdge(out, ob+index, e);
    }
}

// ---- edgeColoringSimple / edgeColoringInkTrap with LANES = EDGES (and lanes = corners), one wavefront per glyph (round 4).
// What is sequential in edge-coloring.cpp is only the colour STATE: it advances once per smooth contour, twice per teardrop, once per (major)
// corner otherwise (switchColor, :53-64) -- a handful of wave-uniform scalar steps per contour. Everything per edge is parallel:
//   * the corner test of the junction before edge i (:76-85, :159-170) reads edges i-1 and i only -> one ballot per 64 edges;
//   * an edge's colour is its spline's, and its spline is the number of corners between the contour's first corner and the edge, cyclically --
//     a prefix population count of the ballots (:128-139, :248-256); the teardrop's thirds are a closed form of the edge's position (:101-104);
//   * ink trap: prevEdgeLengthEstimate of corner k is the reference's running sum over the edges of the spline before it -- one LANE PER CORNER
//     adds its spline's edge lengths (computed with lanes = edges) in edge order, so every sum rounds as the reference's; the minor test (:222-228)
//     reads three lengths; a minor corner's colour (:241-247) is (previous corner's & next corner's) ^ 7 -- its neighbours are always major
//     (minor[i] needs length[i+1] < length[i+2], minor[i+1] the opposite), so that step is parallel over corners as well.
// The driver is written against a wave context (lanes / ballot / leader / sync): the kernel's runs one lane each, tests/hostemu's loops over 64 --
// the same source is checked against the oracle in the GPU-less container.
enum { PREP_WAVE = 64 };

struct ColourTables {
    unsigned long long *cornerMask;    // [ceil(n / 64)] ballots of the corner test
    unsigned char *splineColor;        // [n] colour per spline (ink trap: EdgeColoringInkTrapCorner::color)
    double *edgeLength;                // ink trap: [n] estimateEdgeLength per edge
    double *cornerLength;              //           [n] prevEdgeLengthEstimate per corner
    int *cornerIndex;                  //           [n] edge index of the k-th corner
    unsigned char *minor;              //           [n]
};

// corners among edges [0, k): full ballot words below k's, plus the low bits of k's word
MSDF_HD int cornersBelow(const unsigned long long *mask, int k) {
    int sum = 0;
    for (int w = 0; w < k/PREP_WAVE; ++w)
        sum += __builtin_popcountll(mask[w]);
    if (k%PREP_WAVE)
        sum += __builtin_popcountll(mask[k/PREP_WAVE]&((1ull<<(k%PREP_WAVE))-1ull));
    return sum;
}

MSDF_HD int pick3(int a, int b, int c, int k) { return k == 0 ? a : k == 1 ? b : c; }

// One contour: input edges [ib, ib+n) of `in`, output edges from `ob` of `out`; color / seed: the shape-wide running state (identical in every lane).
template <bool INKTRAP, class Ctx>
MSDF_HD void colourContourWave(const Ctx &ctx, const ColourTables &t, const EdgeArrays &in, int ib, int n, const EdgeArrays &out, int ob, double crossThreshold,
                               int &color, unsigned long long &seed) {
    if (n == 0)
        return;
    // ---- corners: the junction before edge i
    int nCorners = 0, first = -1;
    for (int base = 0; base < n; base += PREP_WAVE) {
        const unsigned long long mask = ctx.ballot([&](int lane) {
            const int i = base+lane;
            if (i >= n)
                return false;
            const PrepEdge e = loadEdge(in, ib+i), pe = loadEdge(in, ib+(i+n-1)%n);
            return isCorner(normalize(edgeDirection(pe, 1), false), normalize(edgeDirection(e, 0), false), crossThreshold);
        });
        ctx.leader([&]() { t.cornerMask[base/PREP_WAVE] = mask; });
        if (first < 0 && mask)
            first = base+__builtin_ctzll(mask);
        nCorners += __builtin_popcountll(mask);
    }
    ctx.sync();
    if (nCorners == 0) {                                                              // smooth contour (:87-92, :174-179)
        switchColor(color, seed);
        const int c = color;
        ctx.lanes([&](int lane) {
            for (int i = lane; i < n; i += PREP_WAVE) {
                PrepEdge e = loadEdge(in, ib+i);
                e.color = c;
                storeEdge(out, ob+i, e);
            }
        });
    } else if (nCorners == 1) {                                                       // "teardrop" (:93-123, :180-213)
        switchColor(color, seed);
        const int c0 = color, c1 = 7;
        switchColor(color, seed);
        const int c2 = color;
        if (n >= 3)
            ctx.lanes([&](int lane) {
                for (int index = lane; index < n; index += PREP_WAVE) {
                    const int i = (index-first+n)%n;                                  // the edge's position counted from the corner
                    PrepEdge e = loadEdge(in, ib+index);
                    e.color = pick3(c0, c1, c2, 1+symmetricalTrichotomy(i, n));
                    storeEdge(out, ob+index, e);
                }
            });
        else                                                                          // fewer than three edges: thirds of each, a lane per part (teardropSplit)
            ctx.lanes([&](int lane) {
                if (lane < 3*n) {
                    const int src = n == 2 ? (lane/3 == first ? 0 : 1) : 0;           // the corner's edge comes first
                    PrepEdge part = splitThird(loadEdge(in, ib+src), lane%3);
                    part.color = pick3(c0, c1, c2, n >= 2 ? lane/2 : lane);
                    storeEdge(out, ob+lane, part);
                }
            });
    } else {                                                                          // multiple corners: one colour per spline, then lanes = edges
        if (!INKTRAP) {                                                               // (:124-140)
            switchColor(color, seed);
            const int initialColor = color;
            ctx.leader([&]() { t.splineColor[0] = (unsigned char) initialColor; });
            for (int spline = 1; spline < nCorners; ++spline) {
                switchColorBanned(color, seed, (spline == nCorners-1)*initialColor);
                const int c = color;
                ctx.leader([&]() { t.splineColor[spline] = (unsigned char) c; });
            }
            ctx.sync();
        } else {                                                                      // (:214-247)
            const int cornerCount = nCorners;
            ctx.lanes([&](int lane) {                                                 // lanes = edges: length estimates; the position of the k-th corner
                for (int i = lane; i < n; i += PREP_WAVE) {
                    t.edgeLength[i] = estimateEdgeLength(loadEdge(in, ib+i));
                    if (t.cornerMask[i/PREP_WAVE]>>(i%PREP_WAVE)&1ull)
                        t.cornerIndex[cornersBelow(t.cornerMask, i)] = i;
                }
            });
            ctx.sync();
            ctx.lanes([&](int lane) {                                                 // lanes = corners: the running sum of the spline before the corner, in edge order
                for (int k = lane; k < cornerCount; k += PREP_WAVE) {
                    double length = 0;
                    for (int i = k ? t.cornerIndex[k-1] : 0; i < t.cornerIndex[k]; ++i)
                        length += t.edgeLength[i];
                    if (k == 0 && cornerCount > 3) {                                  // corners[0].prevEdgeLengthEstimate += splineLength (:220)
                        double tail = 0;
                        for (int i = t.cornerIndex[cornerCount-1]; i < n; ++i)
                            tail += t.edgeLength[i];
                        length += tail;
                    }
                    t.cornerLength[k] = length;
                }
            });
            ctx.sync();
            int majorCornerCount = cornerCount;
            for (int base = 0; base < cornerCount; base += PREP_WAVE) {               // lanes = corners (:221-228)
                const unsigned long long mask = ctx.ballot([&](int lane) {
                    const int i = base+lane;
                    if (i >= cornerCount)
                        return false;
                    const double a = t.cornerLength[i], b = t.cornerLength[(i+1)%cornerCount], c = t.cornerLength[(i+2)%cornerCount];
                    const bool minor = cornerCount > 3 && a > b && b < c;
                    t.minor[i] = (unsigned char) minor;
                    return minor;
                });
                majorCornerCount -= __builtin_popcountll(mask);
            }
            ctx.sync();
            int initialColor = 0;
            for (int i = 0; i < cornerCount; ++i)                                     // the colour state: one step per major corner (:229-240)
                if (!t.minor[i]) {
                    --majorCornerCount;
                    switchColorBanned(color, seed, !majorCornerCount*initialColor);
                    const int c = color;
                    ctx.leader([&]() { t.splineColor[i] = (unsigned char) c; });
                    if (!initialColor)
                        initialColor = color;
                }
            ctx.sync();
            ctx.lanes([&](int lane) {                                                 // lanes = minor corners: between two major ones (:241-247)
                for (int i = lane; i < cornerCount; i += PREP_WAVE)
                    if (t.minor[i])
                        t.splineColor[i] = (unsigned char) ((t.splineColor[(i+cornerCount-1)%cornerCount]&t.splineColor[(i+1)%cornerCount])^7);
            });
            ctx.sync();
            color = t.splineColor[cornerCount-1];                                     // the walk (:248-256) ends in the last corner's spline
        }
        const int uptoStart = cornersBelow(t.cornerMask, first+1);
        ctx.lanes([&](int lane) {
            for (int index = lane; index < n; index += PREP_WAVE) {
                const int upto = cornersBelow(t.cornerMask, index+1);
                const int spline = index >= first ? upto-uptoStart : nCorners-uptoStart+upto;   // corners in (first, index], cyclically
                PrepEdge e = loadEdge(in, ib+index);
                e.color = t.splineColor[spline];
                storeEdge(out, ob+index, e);
            }
        });
    }
    ctx.sync();                                                                       // the tables are the next contour's
}

} // namespace msdfhip
