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
    const V2 p0 = ld(e.p), p1 = ld(e.p+2), p2 = ld(e.p+4), p3 = ld(e.p+6);
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

} // namespace msdfhip
