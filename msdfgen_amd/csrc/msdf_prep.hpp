// msdf_prep.hpp -- once-per-upload digestion of the raw CSR edge buffer into EdgeRec records and contour windings.
// Every quantity is produced with the same IEEE operations, in the same order, that the reference applies per pixel
// (edge-selectors.cpp:189-198, edge-segments.cpp:121-139, 173-277, equation-solver.cpp:63-70), so hoisting them is exact.
#pragma once

#include "msdf_device.hpp"

namespace msdfhip {

struct RawEdge {
    V2 p[4];
    int type, color;
};

MSDF_HD RawEdge loadRaw(const double *points, const uint8_t *types, const uint8_t *colors, int e) {
    RawEdge r;
    const double *p = points+8*(size_t) e;
    for (int i = 0; i < 4; ++i)
        r.p[i] = mk(p[2*i], p[2*i+1]);
    r.type = types[e];
    r.color = colors[e];
    return r;
}

MSDF_HD bool nonzero(V2 a) { return a.x != 0 || a.y != 0; }

MSDF_HD V2 rawDirection(const RawEdge &e, double t) {                         // edge-segments.cpp:121-139
    if (e.type == 1)
        return e.p[1]-e.p[0];
    if (e.type == 2) {
        V2 tangent = mixv(e.p[1]-e.p[0], e.p[2]-e.p[1], t);
        if (!nonzero(tangent))
            return e.p[2]-e.p[0];
        return tangent;
    }
    V2 tangent = mixv(mixv(e.p[1]-e.p[0], e.p[2]-e.p[1], t), mixv(e.p[2]-e.p[1], e.p[3]-e.p[2], t), t);
    if (!nonzero(tangent)) {
        if (t == 0) return e.p[2]-e.p[0];
        if (t == 1) return e.p[3]-e.p[1];
    }
    return tangent;
}

MSDF_HD V2 rawPoint(const RawEdge &e, double t) {                             // edge-segments.cpp:108-119
    if (e.type == 1)
        return mixv(e.p[0], e.p[1], t);
    if (e.type == 2)
        return mixv(mixv(e.p[0], e.p[1], t), mixv(e.p[1], e.p[2], t), t);
    V2 p12 = mixv(e.p[1], e.p[2], t);
    return mixv(mixv(mixv(e.p[0], e.p[1], t), p12, t), mixv(p12, mixv(e.p[2], e.p[3], t), t), t);
}

MSDF_HD void st(double *dst, V2 v) { dst[0] = v.x, dst[1] = v.y; }

// Builds the record of edge `cur` whose cyclic neighbours in its contour are `prev` and `next`.
__attribute__((always_inline)) MSDF_HD void buildRecord(EdgeRec &r, const RawEdge &prev, const RawEdge &cur, const RawEdge &next, int contour) {
    st(r.p0, cur.p[0]);
    st(r.pe, cur.type == 1 ? cur.p[1] : cur.type == 2 ? cur.p[2] : cur.p[3]);  // point(1): the last control point (selected, not indexed: the edge stays in registers)
    st(r.p1, cur.type >= 2 ? cur.p[1] : mk(0, 0));
    st(r.p2, cur.type == 3 ? cur.p[2] : mk(0, 0));
    r.pad_[0] = r.pad_[1] = 0;
    for (int i = 0; i < 6; ++i)
        r.k[i] = 0;
    const V2 ab = cur.p[1]-cur.p[0];
    V2 br = mk(0, 0), as = mk(0, 0);
    if (cur.type >= 2)
        br = cur.p[2]-cur.p[1]-ab;
    if (cur.type == 3)
        as = (cur.p[3]-cur.p[2])-(cur.p[2]-cur.p[1])-br;
    st(r.ab, ab);
    st(r.br, br);
    st(r.as_, as);
    const V2 ep0 = rawDirection(cur, 0), ep1 = rawDirection(cur, 1);
    st(r.ep0, ep0);
    st(r.ep1, ep1);
    r.e0dot = dot(ep0, ep0);
    r.e1dot = dot(ep1, ep1);
    const V2 aDir = normalize(ep0, true), bDir = normalize(ep1, true);        // edge-selectors.cpp:193-194
    const V2 prevDir = normalize(rawDirection(prev, 1), true);               // :195
    const V2 nextDir = normalize(rawDirection(next, 0), true);               // :196
    st(r.aDirN, aDir);
    st(r.bDirN, bDir);
    st(r.na, normalize(prevDir+aDir, true));                                  // :197
    st(r.nb, normalize(bDir+nextDir, true));                                  // :198
    int flags = 0;
    if (vlen(ep0) == 0)
        flags |= REC_A_ZERO;
    if (vlen(ep1) == 0)
        flags |= REC_B_ZERO;
    if (cur.type == 1) {                                                      // edge-segments.cpp:173-185
        r.k[0] = dot(ab, ab);
        const V2 on = orthonormalFalse(ab), abn = normalize(ab, false);
        r.k[1] = on.x, r.k[2] = on.y;
        r.k[3] = abn.x, r.k[4] = abn.y;
    } else if (cur.type == 2) {                                               // edge-segments.cpp:191-193, equation-solver.cpp:63-70, 34-39
        const double a = dot(br, br);
        const double b = 3*dot(ab, br);
        r.k[0] = a;
        r.k[1] = b;
        r.k[2] = 2*dot(ab, ab);
        if (a != 0) {
            const double bn = b/a;
            if (fabs(bn) < 1e6) {
                flags |= REC_NORMED;
                r.k[3] = bn;
                r.k[4] = bn*bn;
                r.k[5] = bn*(1/3.);
            }
        }
    } else {                                                                  // edge-segments.cpp:247-249
        st(r.k, 3*ab);
        st(r.k+2, 6*br);
    }
    V2 lo = cur.p[0], hi = cur.p[0];                                          // culling aids: control-point box, an on-curve sample
    MSDF_UNROLL
    for (int i = 1; i <= 3; ++i)
        if (i <= cur.type) {
            lo = mk(dmin(lo.x, cur.p[i].x), dmin(lo.y, cur.p[i].y));
            hi = mk(dmax(hi.x, cur.p[i].x), dmax(hi.y, cur.p[i].y));
        }
    st(r.lo, lo);
    st(r.hi, hi);
    st(r.mid, rawPoint(cur, .5));
    for (int i = 0; i < 4; ++i)                                              // reciprocals of the pixel-independent divisors (divExact)
        r.rcp[i] = 0;
    bool fast;
    if (cur.type == 1) {
        fast = divSafe(r.k[0]);
        r.k[5] = 1/r.k[0];                                                    // (a linear edge is complete with the R and E0 blocks)
    } else {
        fast = divSafe(r.e0dot) && divSafe(r.e1dot) && (cur.type == 3 || !(flags&REC_NORMED) || divSafe(r.k[0]));
        r.rcp[0] = cur.type == 2 ? 1/r.k[0] : 0;
        r.rcp[1] = 1/r.e0dot;
        r.rcp[2] = 1/r.e1dot;
    }
    if (fast)
        flags |= REC_FASTDIV;
    const int common = prev.color&cur.color;                                  // MSDFErrorCorrection.cpp:127-129
    if (!(common&(common-1)))
        flags |= REC_CORNER;
    r.type = cur.type;
    r.color = cur.color;
    r.flags = flags;
    r.contour = contour;
}

// Record slot r (global, = contour start + visit position) -> natural edge index. Visit order: last, first, ..., last-1.
MSDF_HD int visitToEdge(int start, int n, int v) { return start+(v == 0 ? n-1 : v-1); }

// (always inlined: as a call it cost its callers an 84-168 B stack frame in scratch memory -- the only scratch of the digest kernels)
__attribute__((always_inline)) MSDF_HD void prepRecord(EdgeRec *recs, int slot, int contour, const int32_t *contourOffsets, const double *points, const uint8_t *types, const uint8_t *colors) {
    const int start = contourOffsets[contour], n = contourOffsets[contour+1]-start;
    const int e = visitToEdge(start, n, slot-start);
    const int prev = start+(e-start+n-1)%n, next = start+(e-start+1)%n;
    buildRecord(recs[slot], loadRaw(points, types, colors, prev), loadRaw(points, types, colors, e), loadRaw(points, types, colors, next), contour);
}

MSDF_HD double shoelace(V2 a, V2 b) { return (b.x-a.x)*(a.y+b.y); }           // Contour.cpp:7-9

MSDF_HD int contourWinding(int contour, const int32_t *contourOffsets, const double *points, const uint8_t *types, const uint8_t *colors) { // Contour.cpp:57-81
    const int begin = contourOffsets[contour], end = contourOffsets[contour+1];
    const int n = end-begin;
    if (n <= 0)
        return 0;
    double total = 0;
    if (n == 1) {
        RawEdge e = loadRaw(points, types, colors, begin);
        V2 a = rawPoint(e, 0), b = rawPoint(e, 1/3.), c = rawPoint(e, 2/3.);
        total += shoelace(a, b);
        total += shoelace(b, c);
        total += shoelace(c, a);
    } else if (n == 2) {
        RawEdge e0 = loadRaw(points, types, colors, begin), e1 = loadRaw(points, types, colors, begin+1);
        V2 a = rawPoint(e0, 0), b = rawPoint(e0, .5), c = rawPoint(e1, 0), d = rawPoint(e1, .5);
        total += shoelace(a, b);
        total += shoelace(b, c);
        total += shoelace(c, d);
        total += shoelace(d, a);
    } else {
        V2 prev = rawPoint(loadRaw(points, types, colors, end-1), 0);
        for (int i = begin; i < end; ++i) {
            V2 cur = rawPoint(loadRaw(points, types, colors, i), 0);
            total += shoelace(prev, cur);
            prev = cur;
        }
    }
    return (0 < total)-(total < 0);
}

// Contour::winding of the contours [cBegin, cEnd) by ONE wavefront (round 4). The shoelace sum of a contour (Contour.cpp:72-79) is a serial
// floating-point sum whose sign is the result, so its ORDER is kept; what a lane per contour made serial as well were the LOADS -- one dependent
// memory round trip per edge. The fused single-shape kernel walks all of a shape's contours this way (its digest phase: 8.6 -> 6.1 us), the batch
// digest only its long contours (k_prep_records). Here the terms shoelace(point_{i-1}(0), point_i(0)) are computed with lanes = edges, 64 per round over the edges of all the
// contours at once, and one wave-uniform pass adds them in edge order, closing a contour whenever its last edge has been added (`terms`: 64 doubles
// the lanes share). Contours of fewer than three edges use other sample points (:59-71): they keep contourWinding, a lane each.
// Written against a wave context (lanes / leader / sync) like colourContourWave (msdf_shapeprep.hpp): tests/hostemu runs the same source.
enum { PREP_WINDING_WAVE_MIN_EDGES = 48 };   // k_prep_records: contours from this length on are walked by their wavefront together

template <class Ctx>
MSDF_HD void contourWindingsWave(const Ctx &ctx, double *terms, int cBegin, int cEnd, const int32_t *contourOffsets, const double *points, const uint8_t *types,
                                 const uint8_t *colors, int8_t *windings) {
    if (cBegin >= cEnd)
        return;
    const int e0 = contourOffsets[cBegin], e1 = contourOffsets[cEnd];
    int c = cBegin, cFirstEdge = e0, cEndEdge = contourOffsets[cBegin+1];             // the contour the ordered pass is in
    double total = 0;
    for (int base = e0; base < e1; base += 64) {
        ctx.lanes([&](int lane) {
            const int i = base+lane;
            double term = 0;
            if (i < e1) {
                int lo = cBegin, hi = cEnd-1;                                         // last contour with contourOffsets[c] <= i (skips empty contours)
                while (lo < hi) {
                    const int mid = (lo+hi+1)>>1;
                    if (contourOffsets[mid] <= i)
                        lo = mid;
                    else
                        hi = mid-1;
                }
                const int b = contourOffsets[lo], e = contourOffsets[lo+1];
                if (e-b >= 3)
                    term = shoelace(rawPoint(loadRaw(points, types, colors, i == b ? e-1 : i-1), 0), rawPoint(loadRaw(points, types, colors, i), 0));
            }
            terms[lane] = term;
        });
        ctx.sync();
        const int count = e1-base < 64 ? e1-base : 64;
        for (int k = 0; k < count; ++k) {
            while (cEndEdge <= base+k) {                                              // the contours that ended before this edge (empty ones among them)
                if (cEndEdge-cFirstEdge >= 3) {
                    const int8_t w = (int8_t) ((0 < total)-(total < 0));
                    const int at = c;
                    ctx.leader([&]() { windings[at] = w; });
                }
                total = 0;
                ++c;
                cFirstEdge = cEndEdge, cEndEdge = contourOffsets[c+1];                // (c < cEnd: the edge base+k belongs to a contour of the range)
            }
            total += terms[k];
        }
        ctx.sync();
    }
    if (cEndEdge-cFirstEdge >= 3) {                                                   // the contour the last edge belonged to
        const int8_t w = (int8_t) ((0 < total)-(total < 0));
        const int at = c;
        ctx.leader([&]() { windings[at] = w; });
    }
    ctx.lanes([&](int lane) {                                                         // fewer than three edges (none included): a lane each
        for (int k = cBegin+lane; k < cEnd; k += 64)
            if (contourOffsets[k+1]-contourOffsets[k] < 3)
                windings[k] = (int8_t) contourWinding(k, contourOffsets, points, types, colors);
    });
}

} // namespace msdfhip
