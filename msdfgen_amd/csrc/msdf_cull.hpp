// msdf_cull.hpp -- conservative, EXACT per-tile edge culling for the distance-field kernel.
//
// The reference prunes edges per pixel with a serpentine-order cache (edge-selectors.cpp:64-79); a GPU wavefront instead
// culls once per 8x8 texel tile with lanes = edges, compacts the survivors, and then runs lanes = texels over the short list.
// An edge may be dropped for a tile only if it cannot change the result of ANY texel P of the tile. With
//   c, r      tile centre / radius in shape space (every texel centre P of the tile has |P-c| <= r)
//   U[ch]     an upper bound of the distance from c to the nearest edge of this CONTOUR that carries channel ch
//             (so the contour-channel's final minimum true distance at P is <= U[ch]+r)
// an edge e with channel mask m is irrelevant if, for Umax = max over ch in m of U[ch]:
//   (1) its true distance is too large everywhere:  LB(c, e) - r > Umax + r          -> never the nearest edge, not even on ties
//   (2) neither end can contribute a perpendicular distance that matters (edge-selectors.cpp:199-224, :108-117): for end A,
//       either the whole tile is outside the wedge { add > 0, ts > 0 } in which the reference considers it, or the distance of
//       the tile to the tangent line exceeds Umax + r -- a perpendicular distance larger than the final minimum true distance
//       never survives computeDistance(), because the nearest edge's own (pseudo-)distance is smaller.  Same for end B.
// All bounds are inflated by a relative 1e-6 so that rounding in the bounds themselves can never flip an exact tie.
#pragma once

#include <string.h>
#include "msdf_device.hpp"

namespace msdfhip {

#define MSDF_CULL_SLACK (1.+1e-6)

// Upper bound of the distance from c to edge e: the nearest of three on-curve points (both ends and point(0.5)).
MSDF_HD double cullUpperDistance(const EdgeRec &e, V2 c) {
    const V2 a = c-ld(e.p0), b = c-ld(e.pe), m = c-ld(e.mid);
    const double da = dot(a, a), db = dot(b, b), dm = dot(m, m);
    return sqrt(cmin(cmin(da, db), dm));
}

// The same bound as it travels in the kernel: fp32, rounded UP, without the fp64 square root (90 cycles on gfx950 against 8 for the
// fp32 one). sample2 = RN((float) min squared distance) is also the walk-order key below, so phase 1 computes it once per (tile, edge).
MSDF_HD float cullNearestSample2(const EdgeRec &e, V2 c) {
    const V2 a = c-ld(e.p0), b = c-ld(e.pe), m = c-ld(e.mid);
    return (float) cmin(cmin(dot(a, a), dot(b, b)), dot(m, m));
}
MSDF_HD float cullBumpUp(float f) {                              // next float above a non-negative finite one (inf / nan stay)
    unsigned bits;
    memcpy(&bits, &f, sizeof(bits));
    if (bits < 0x7f800000u)
        ++bits;
    memcpy(&f, &bits, sizeof(bits));
    return f;
}
MSDF_HD float cullUpperFromSample2(float sample2) {              // >= sqrt(the fp64 value sample2 was rounded from)
    return cullBumpUp(sqrtf(cullBumpUp(sample2)));               // RN(d2) >= d2*(1-2^-24): one ulp up covers it; sqrtf is correctly rounded: one more
}

// Lower bound of the SQUARED distance from c to edge e: distance to the control-point bounding box (a Bezier lies in its control hull).
MSDF_HD double cullLowerDistance2(const EdgeRec &e, V2 c) {
    const double dx = cmax(cmax(e.lo[0]-c.x, c.x-e.hi[0]), 0.);
    const double dy = cmax(cmax(e.lo[1]-c.y, c.y-e.hi[1]), 0.);
    return dx*dx+dy*dy;
}

// Can edge e matter for some texel of the tile (c, r), given Umax (see header)? PERP: the selector uses perpendicular distances.
template <bool PERP>
MSDF_HD bool cullEdgeSurvives(const EdgeRec &e, V2 c, double r, double Umax) {
    const double R = r*MSDF_CULL_SLACK;
    const double reach = (Umax+R)*MSDF_CULL_SLACK;               // >= final minimum true distance of the contour-channel at any texel
    const double beyond = reach+R;
    if (!(cullLowerDistance2(e, c) > beyond*beyond))                   // LB-R > reach, squared (both sides non-negative; the slack dwarfs the rounding)
        return true;
    if (PERP) {
        const V2 ap = c-ld(e.p0), aDir = ld(e.aDirN);
        const bool outsideA = dot(ap, ld(e.na))+R <= 0 || -dot(ap, aDir)+R <= 0;   // add <= 0 or ts <= 0 on the whole tile
        if (!outsideA && !(fabs(cross(ap, aDir))-R > reach))
            return true;
        const V2 bp = c-ld(e.pe), bDir = ld(e.bDirN);
        const bool outsideB = -dot(bp, ld(e.nb))+R <= 0 || dot(bp, bDir)+R <= 0;    // bdd <= 0 or ts <= 0 on the whole tile
        if (!outsideB && !(fabs(cross(bp, bDir))-R > reach))
            return true;
    }
    return false;
}

// Walk order of a contour's survivors: nearest first (ascending key), so that the per-texel relevance test (selEdgeRelevant) can
// drop most of the others -- the selector's result does not depend on the order (Selector::idx). The key is the squared distance
// from the tile centre to the nearest of three on-curve points, as non-negative float bits with the low four mantissa bits replaced
// by the edge's position within its group of 16 (keys are then unique within a group: a plain rank is a permutation).
MSDF_HD unsigned cullOrderKeyOfSample2(float d2, int slot) {
    unsigned bits;
    memcpy(&bits, &d2, sizeof(bits));
    if (!(bits < 0x7f800000u))
        bits = 0x7f7ffff0u;                                      // inf / nan (degenerate input): still ahead of the non-survivors
    return (bits&~15u)|(unsigned) (slot&15);
}
MSDF_HD unsigned cullOrderKey(const EdgeRec &e, V2 c, int slot) { return cullOrderKeyOfSample2(cullNearestSample2(e, c), slot); }
#define MSDF_CULL_KEY_DROPPED 0x7f800000u                        // non-survivors: behind every survivor
#define MSDF_CULL_KEY_DROPPED_SEGMENTED 0xfffffff0u              // the same for keys that lead with a contour segment (k_distance, overlapping combiner)

// Channel mask with which an edge takes part in selector SEL (1: all edges one "channel"; 2: ditto; 3/4: its colour bits).
template <int SEL>
MSDF_HD int cullMask(const EdgeRec &e) { return SEL <= 2 ? 1 : (e.color&7); }

} // namespace msdfhip
