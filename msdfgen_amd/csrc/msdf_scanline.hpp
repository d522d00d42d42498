// msdf_scanline.hpp -- scanline fill test and distance sign correction: the step that sits between
// generate* and msdfErrorCorrection in every no-Skia flow (main.cpp:1281-1298), so that the field never has to leave the GPU
// mid-pipeline.
//
// Reference: distanceSignCorrection (core/rasterization.cpp:19-88) over Shape::scanline (core/Shape.cpp:117-135),
// EdgeSegment::scanlineIntersections (core/edge-segments.cpp:279-403) and Scanline::filled (core/Scanline.cpp:66-122).
// The reference sorts a row's intersections and prefix-sums their directions; filled(x) then reads the sum over all intersections
// with x_i <= x -- an order-independent sum, evaluated here directly. The correction of a texel is a pure function of its own value
// and fill bit, plus (only for texels whose median is EXACTLY the zero value) the match state of its four neighbours, which in turn
// is a pure function of the ORIGINAL field and the neighbours' fill bits -- hence one gather pass, input and output buffers distinct.
#pragma once

#include "msdf_device.hpp"

namespace msdfhip {

MSDF_HD int isign(double n) { return (0 < n)-(n < 0); }                      // arithmetics.hpp:53-55

// EdgeSegment::scanlineIntersections from the digested record (control points, ab, br, as). Returns the count; x[3], dy[3].
MSDF_HD int scanlineIntersections(const EdgeRec &e, double x[3], int dy[3], double y) {
    const V2 p0 = ld(e.p0), p1 = e.type == 1 ? ld(e.pe) : ld(e.p1), p2 = e.type == 2 ? ld(e.pe) : ld(e.p2), p3 = ld(e.pe);   // control points as stored (EdgeRec blocks)
    if (e.type == 1) {                                                        // edge-segments.cpp:279-287
        if ((y >= p0.y && y < p1.y) || (y >= p1.y && y < p0.y)) {
            const double param = (y-p0.y)/(p1.y-p0.y);
            x[0] = (1.-param)*p0.x+param*p1.x;
            dy[0] = isign(p1.y-p0.y);
            return 1;
        }
        return 0;
    }
    const V2 ab = ld(e.ab), br = ld(e.br), as = ld(e.as_);
    const V2 pe = e.type == 2 ? p2 : p3;                                      // end point
    int total = 0;
    int nextDY = y > p0.y ? 1 : -1;
    x[total] = p0.x;
    if (p0.y == y) {
        const bool rising = e.type == 2 ? (p0.y < p1.y || (p0.y == p1.y && p0.y < p2.y))                                   // :293-298
                                        : (p0.y < p1.y || (p0.y == p1.y && (p0.y < p2.y || (p0.y == p2.y && p0.y < p3.y)))); // :347-352
        if (rising)
            dy[total++] = 1;
        else
            nextDY = 1;
    }
    {
        double t[3] = { 0, 0, 0 };
        int solutions;
        if (e.type == 2) {                                                    // :300-318
            solutions = solveQuadratic(t, br.y, 2*ab.y, p0.y-y);
            if (solutions >= 2 && t[0] > t[1]) {
                const double tmp = t[0];
                t[0] = t[1], t[1] = tmp;
            }
        } else {                                                              // :354-380
            const double a = as.y, b = 3*br.y, c = 3*ab.y, d = p0.y-y;        // solveCubic, equation-solver.cpp:63-70
            solutions = -2;
            if (a != 0) {
                const double bn = b/a;
                if (fabs(bn) < 1e6)
                    solutions = solveCubicNormedPre(t, bn, bn*bn, bn*(1/3.), c/a, d/a);
            }
            if (solutions == -2)
                solutions = solveQuadratic(t, b, c, d);
            if (solutions >= 2) {
                if (t[0] > t[1]) {
                    const double tmp = t[0];
                    t[0] = t[1], t[1] = tmp;
                }
                if (solutions >= 3 && t[1] > t[2]) {
                    double tmp = t[1];
                    t[1] = t[2], t[2] = tmp;
                    if (t[0] > t[1])
                        tmp = t[0], t[0] = t[1], t[1] = tmp;
                }
            }
        }
        const int cap = e.type == 2 ? 2 : 3;
        MSDF_UNROLL
        for (int i = 0; i < 3; ++i) {
            if (i < solutions && total < cap && t[i] >= 0 && t[i] <= 1) {
                double slope;
                if (e.type == 2) {
                    x[total] = p0.x+2*t[i]*ab.x+t[i]*t[i]*br.x;
                    slope = ab.y+t[i]*br.y;
                } else {
                    x[total] = p0.x+3*t[i]*ab.x+3*t[i]*t[i]*br.x+t[i]*t[i]*t[i]*as.x;
                    slope = ab.y+2*t[i]*br.y+t[i]*t[i]*as.y;
                }
                if (nextDY*slope >= 0) {
                    dy[total++] = nextDY;
                    nextDY = -nextDY;
                }
            }
        }
    }
    const int cap = e.type == 2 ? 2 : 3;
    if (pe.y == y) {                                                          // :320-332, :382-394
        if (nextDY > 0 && total > 0) {
            --total;
            nextDY = -1;
        }
        const bool falling = e.type == 2 ? (p2.y < p1.y || (p2.y == p1.y && p2.y < p0.y))
                                         : (p3.y < p2.y || (p3.y == p2.y && (p3.y < p1.y || (p3.y == p1.y && p3.y < p0.y))));
        if (falling && total < cap) {
            x[total] = pe.x;
            if (nextDY < 0) {
                dy[total++] = -1;
                nextDY = 1;
            }
        }
    }
    if (nextDY != (y >= pe.y ? 1 : -1)) {                                     // :333-340, :395-402
        if (total > 0)
            --total;
        else {
            if (fabs(pe.y-y) < fabs(p0.y-y))
                x[total] = pe.x;
            dy[total++] = nextDY;
        }
    }
    return total;
}

// Can edge e intersect row y at all? A Bezier lies in the hull of its control points; rows more than a rounding margin outside the
// control box yield no intersection in any branch of scanlineIntersections (the end-point fix-ups need p.y == y or a sign change).
MSDF_HD bool rowMayIntersect(const EdgeRec &e, double y) {
    const double m = 1e-9*(fabs(e.lo[1])+fabs(e.hi[1])+fabs(y));
    return y >= e.lo[1]-m && y <= e.hi[1]+m;
}

MSDF_HD bool interpretFillRule(int intersections, int rule) {                 // Scanline.cpp:13-25: 0 nonzero, 1 odd, 2 positive, 3 negative
    return rule == 0 ? intersections != 0 : rule == 1 ? (intersections&1) != 0 : rule == 2 ? intersections > 0 : intersections < 0;
}

// match value of a texel (rasterization.cpp:55-66): 0 ambiguous, -1 sign flipped, +1 sign kept.
MSDF_HD int signMatch(const float *msd, bool fill, float zero) {
    const float sd = medianf(msd[0], msd[1], msd[2]);
    if (sd == zero)
        return 0;
    return ((sd > zero) != fill) ? -1 : 1;
}

// ---- estimateSDFError (core/sdf-error-estimation.cpp:134-154), SURVEY.md 8 row f4 ------------------------------------------------
// One lane evaluates one scanline: the shape's intersections and the intersections reconstructed from the distance field, both
// sorted with prefix-summed directions (Scanline::preprocess, Scanline.cpp:66-77), then Scanline::overlap (:27-62). The lists live in
// a global workspace, element i of a lane at [i*stride] (lanes adjacent: coalesced).

struct StridedList {
    double *x;
    int *dir;
    size_t stride;
    int n;
    MSDF_HD double &X(int i) const { return x[(size_t) i*stride]; }
    MSDF_HD int &D(int i) const { return dir[(size_t) i*stride]; }
    MSDF_HD void push(double xv, int d) { X(n) = xv, D(n) = d; ++n; }
};

MSDF_HD void listPreprocess(StridedList &l) {                                 // qsort by x (ties: zero-length spans, any order) + prefix sums
    for (int i = 1; i < l.n; ++i) {
        const double xv = l.X(i);
        const int d = l.D(i);
        int j = i-1;
        while (j >= 0 && l.X(j) > xv) {
            l.X(j+1) = l.X(j), l.D(j+1) = l.D(j);
            --j;
        }
        l.X(j+1) = xv, l.D(j+1) = d;
    }
    int total = 0;
    for (int i = 0; i < l.n; ++i) {
        total += l.D(i);
        l.D(i) = total;
    }
}

MSDF_HD void shapeScanline(StridedList &l, const EdgeRec *rec, int nE, double y) { // Shape::scanline, Shape.cpp:117-135
    l.n = 0;
    for (int e = 0; e < nE; ++e) {
        if (!rowMayIntersect(rec[e], y))
            continue;
        double x[3];
        int dy[3];
        const int n = scanlineIntersections(rec[e], x, dy, y);
        for (int k = 0; k < 3; ++k)
            if (k < n)
                l.push(x[k], dy[k]);
    }
    listPreprocess(l);
}

MSDF_HD double scanlineOverlap(const StridedList &a, const StridedList &b, double xFrom, double xTo, int fillRule) {   // Scanline.cpp:27-62
    double total = 0;
    bool aInside = false, bInside = false;
    int ai = 0, bi = 0;
    double ax = a.n ? a.X(ai) : xTo;
    double bx = b.n ? b.X(bi) : xTo;
    while (ax < xFrom || bx < xFrom) {
        const double xNext = dmin(ax, bx);
        if (ax == xNext && ai < a.n) {
            aInside = interpretFillRule(a.D(ai), fillRule);
            ax = ++ai < a.n ? a.X(ai) : xTo;
        }
        if (bx == xNext && bi < b.n) {
            bInside = interpretFillRule(b.D(bi), fillRule);
            bx = ++bi < b.n ? b.X(bi) : xTo;
        }
    }
    double x = xFrom;
    while (ax < xTo || bx < xTo) {
        const double xNext = dmin(ax, bx);
        if (aInside == bInside)
            total += xNext-x;
        if (ax == xNext && ai < a.n) {
            aInside = interpretFillRule(a.D(ai), fillRule);
            ax = ++ai < a.n ? a.X(ai) : xTo;
        }
        if (bx == xNext && bi < b.n) {
            bInside = interpretFillRule(b.D(bi), fillRule);
            bx = ++bi < b.n ? b.X(bi) : xTo;
        }
        x = xNext;
    }
    if (aInside == bInside)
        total += xTo-x;
    return total;
}

MSDF_HD double clampTo(double n, double b) { return n >= 0 && n <= b ? n : (double) (n > 0)*b; }   // arithmetics.hpp:41-43

// scanlineSDF (N == 1, sdf-error-estimation.cpp:9-48) / scanlineMSDF (N >= 3, :50-125). px: row-major [h][w][N], memory rows.
template <int N>
MSDF_HD void scanlineFromSdf(StridedList &line, const float *px, int w, int h, double sx, double sy, double tx, double ty, double y, bool yDown) {
    line.n = 0;
    if (!(w > 0 && h > 0))
        return;
    double pixelY = clampTo(sy*(y+ty)-.5, (double) (h-1));                    // projection.projectY(y)
    if (yDown)
        pixelY = h-1-pixelY;
    int b = (int) floor(pixelY);
    int t = b+1;
    double bt = pixelY-b;
    if (t >= h) {
        b = h-1;
        t = h-1;
        bt = 1;
    }
    const float *rowB = px+(size_t) b*w*N, *rowT = px+(size_t) t*w*N;
    bool inside = false;
    if (N == 1) {
        float lv, rv = mixf(rowB[0], rowT[0], bt);
        if ((inside = rv > .5f))
            line.push(-1e240, 1);
        for (int l = 0, r = 1; r < w; ++l, ++r) {
            lv = rv;
            rv = mixf(rowB[r], rowT[r], bt);
            if (lv != rv) {
                const double lr = (double) (.5f-lv)/(double) (rv-lv);
                if (lr >= 0 && lr <= 1)
                    line.push((l+lr+.5)/sx-tx, (0.f < rv-lv)-(rv-lv < 0.f));  // projection.unprojectX, sign(rv-lv)
            }
        }
    } else {
        float lv[3], rv[3];
        for (int i = 0; i < 3; ++i)
            rv[i] = mixf(rowB[i], rowT[i], bt);
        if ((inside = medianf(rv[0], rv[1], rv[2]) > .5f))
            line.push(-1e240, 1);
        for (int l = 0, r = 1; r < w; ++l, ++r) {
            for (int i = 0; i < 3; ++i) {
                lv[i] = rv[i];
                rv[i] = mixf(rowB[(size_t) r*N+i], rowT[(size_t) r*N+i], bt);
            }
            double nx[3];
            int nd[3], count = 0;
            for (int i = 0; i < 3; ++i) {
                if (lv[i] != rv[i]) {
                    const double lr = (double) (.5f-lv[i])/(double) (rv[i]-lv[i]);
                    if (lr >= 0 && lr <= 1) {
                        const float v0 = mixf(lv[0], rv[0], lr), v1 = mixf(lv[1], rv[1], lr), v2 = mixf(lv[2], rv[2], lr);
                        const float vi = i == 0 ? v0 : i == 1 ? v1 : v2;
                        if (medianf(v0, v1, v2) == vi) {
                            const double xv = (l+lr+.5)/sx-tx;
                            const int dv = (0.f < rv[i]-lv[i])-(rv[i]-lv[i] < 0.f);
                            if (count == 0) nx[0] = xv, nd[0] = dv;             // constant indices keep the tiny arrays in registers
                            else if (count == 1) nx[1] = xv, nd[1] = dv;
                            else nx[2] = xv, nd[2] = dv;
                            ++count;
                        }
                    }
                }
            }
            if (count >= 2) {                                                 // sort new intersections (:100-108)
                if (nx[0] > nx[1]) {
                    const double tx_ = nx[0]; const int td = nd[0];
                    nx[0] = nx[1], nd[0] = nd[1], nx[1] = tx_, nd[1] = td;
                }
                if (count >= 3 && nx[1] > nx[2]) {
                    double tx_ = nx[1]; int td = nd[1];
                    nx[1] = nx[2], nd[1] = nd[2], nx[2] = tx_, nd[2] = td;
                    if (nx[0] > nx[1]) {
                        tx_ = nx[0], td = nd[0];
                        nx[0] = nx[1], nd[0] = nd[1], nx[1] = tx_, nd[1] = td;
                    }
                }
            }
            for (int i = 0; i < 3; ++i)
                if (i < count && (nd[i] > 0) == !inside) {
                    line.push(nx[i], nd[i]);
                    inside = !inside;
                }
            const float rvScalar = medianf(rv[0], rv[1], rv[2]);              // consistency check (:116-121)
            if ((rvScalar > .5f) != inside && rvScalar != .5f && line.n > 0) {
                --line.n;
                inside = !inside;
            }
        }
    }
    listPreprocess(line);
}

// One summand of estimateSDFErrorInner (:143-149): 1 - overlapFactor*overlap for scanline (row, subRow) of a glyph.
template <int N>
MSDF_HD double sdfErrorOfLine(const EdgeRec *rec, int nE, const float *px, int w, int h, double sx, double sy, double tx, double ty, bool yDown,
                              int row, int subRow, int scanlinesPerRow, int fillRule, StridedList &refList, StridedList &sdfList) {
    const double subRowSize = 1./scanlinesPerRow;
    const double xFrom = .5/sx-tx, xTo = (w-.5)/sx-tx;
    const double overlapFactor = 1/(xTo-xFrom);
    const double bt = (subRow+.5)*subRowSize;
    const double y = (row+bt+.5)/sy-ty;
    shapeScanline(refList, rec, nE, y);
    scanlineFromSdf<N>(sdfList, px, w, h, sx, sy, tx, ty, y, yDown);
    return 1-overlapFactor*scanlineOverlap(refList, sdfList, xFrom, xTo, fillRule);
}

} // namespace msdfhip
