// TEST INFRASTRUCTURE ONLY -- never linked, imported or called by the product path.
//
// A flat C ABI over the *compiled, unmodified* reference (Chlumsky/msdfgen v1.13.0, sources read
// in place from /root/reference by oracle/Makefile; none of them are copied into this repo).
// It exists so that Python tests / fixture generators can drive the real reference:
//   * build a msdfgen::Shape from shape-description text or from the flat CSR edge buffer,
//   * run the caller-side shape prep (Shape::normalize core/Shape.cpp:65, edgeColoringSimple
//     core/edge-coloring.cpp:68) that precedes the hot path,
//   * flatten a Shape back to the CSR edge buffer (exact doubles),
//   * run generateSDF/PSDF/MSDF/MTSDF (msdfgen.h:46-56) and msdfErrorCorrection
//     (core/msdf-error-correction.h:15-16),
//   * expose per-function known-answer hooks (EdgeSegment::signedDistance core/edge-segments.h:39,
//     ShapeDistanceFinder::oneShotDistance core/ShapeDistanceFinder.hpp:36, the MSDFErrorCorrection
//     stencil stages core/MSDFErrorCorrection.h:11-53, solveCubic core/equation-solver.h:12).
//
// Output: oracle/_ref/libmsdfgen_ref.so (git-ignored; travels to the GPU box with gpurun).

#include <cstdio>
#include <cstring>
#include <cstdint>
#include <vector>
#include <thread>
#include <atomic>
#include <chrono>

#include "msdfgen.h"
#include "core/ShapeDistanceFinder.h"
#include "core/MSDFErrorCorrection.h"
#include "core/pixel-conversion.hpp"
#include "core/render-sdf.h"
#include "core/sdf-error-estimation.h"
#include "core/equation-solver.h"

using namespace msdfgen;

namespace {

struct FlatView {
    int nContours;
    const int32_t *contourOffsets; // nContours+1
    const double *points;          // E*8
    const int32_t *types;          // E
    const int32_t *colors;         // E
};

Shape *shapeFromFlat(int nContours, const int32_t *contourOffsets, const double *points, const int32_t *types, const int32_t *colors, int inverseY) {
    Shape *shape = new Shape;
    for (int c = 0; c < nContours; ++c) {
        Contour &contour = shape->addContour();
        for (int e = contourOffsets[c]; e < contourOffsets[c+1]; ++e) {
            const double *p = points+8*e;
            EdgeColor color = (EdgeColor) colors[e];
            // Direct constructors: EdgeSegment::create() would simplify degenerate curves.
            switch (types[e]) {
                case 1:
                    contour.addEdge(EdgeHolder(new LinearSegment(Point2(p[0], p[1]), Point2(p[2], p[3]), color)));
                    break;
                case 2:
                    contour.addEdge(EdgeHolder(new QuadraticSegment(Point2(p[0], p[1]), Point2(p[2], p[3]), Point2(p[4], p[5]), color)));
                    break;
                case 3:
                    contour.addEdge(EdgeHolder(new CubicSegment(Point2(p[0], p[1]), Point2(p[2], p[3]), Point2(p[4], p[5]), Point2(p[6], p[7]), color)));
                    break;
                default:
                    delete shape;
                    return NULL;
            }
        }
    }
    shape->setYAxisOrientation(inverseY ? Y_DOWNWARD : Y_UPWARD);
    return shape;
}

MSDFGeneratorConfig makeConfig(int overlap, int ecMode, int ecDist, double minDev, double minImp, unsigned char *buffer) {
    ErrorCorrectionConfig ec((ErrorCorrectionConfig::Mode) ecMode, (ErrorCorrectionConfig::DistanceCheckMode) ecDist, minDev, minImp, buffer);
    return MSDFGeneratorConfig(overlap != 0, ec);
}

void generateOne(const Shape &shape, int mode, float *pixels, int w, int h, int rowStride, int yDown,
                 const double *xf /* sx sy tx ty rangeLower rangeUpper */, const MSDFGeneratorConfig &cfg) {
    SDFTransformation t(Projection(Vector2(xf[0], xf[1]), Vector2(xf[2], xf[3])), DistanceMapping(Range(xf[4], xf[5])));
    YAxisOrientation yo = yDown ? Y_DOWNWARD : Y_UPWARD;
    switch (mode) {
        case 1: generateSDF(BitmapSection<float, 1>(pixels, w, h, rowStride, yo), shape, t, cfg); break;
        case 2: generatePSDF(BitmapSection<float, 1>(pixels, w, h, rowStride, yo), shape, t, cfg); break;
        case 3: generateMSDF(BitmapSection<float, 3>(pixels, w, h, rowStride, yo), shape, t, cfg); break;
        case 4: generateMTSDF(BitmapSection<float, 4>(pixels, w, h, rowStride, yo), shape, t, cfg); break;
    }
}

}

extern "C" {

const char *ref_version() { return "msdfgen 1.13.0 core (compiled from /root/reference, -O2 -std=c++11)"; }

void *ref_shape_from_desc(const char *text) {
    Shape *shape = new Shape;
    if (!readShapeDescription(text, *shape)) {
        delete shape;
        return NULL;
    }
    return shape;
}

void *ref_shape_from_flat(int nContours, const int32_t *contourOffsets, const double *points, const int32_t *types, const int32_t *colors, int inverseY) {
    return shapeFromFlat(nContours, contourOffsets, points, types, colors, inverseY);
}

void ref_shape_free(void *s) { delete (Shape *) s; }
int ref_shape_validate(void *s) { return ((Shape *) s)->validate() ? 1 : 0; }
void ref_shape_normalize(void *s) { ((Shape *) s)->normalize(); }
void ref_shape_orient_contours(void *s) { ((Shape *) s)->orientContours(); }
void ref_shape_color_simple(void *s, double angle, unsigned long long seed) { edgeColoringSimple(*(Shape *) s, angle, seed); }
void ref_shape_color_inktrap(void *s, double angle, unsigned long long seed) { edgeColoringInkTrap(*(Shape *) s, angle, seed); }
int ref_shape_inverse_y(void *s) { return ((Shape *) s)->getYAxisOrientation() == Y_DOWNWARD; }
void ref_shape_set_inverse_y(void *s, int inv) { ((Shape *) s)->setYAxisOrientation(inv ? Y_DOWNWARD : Y_UPWARD); }

void ref_shape_counts(void *s, int32_t *nContours, int32_t *nEdges) {
    const Shape &shape = *(Shape *) s;
    *nContours = (int32_t) shape.contours.size();
    *nEdges = shape.edgeCount();
}

void ref_shape_bounds(void *s, double *lbrt) {
    Shape::Bounds b = ((Shape *) s)->getBounds();
    lbrt[0] = b.l, lbrt[1] = b.b, lbrt[2] = b.r, lbrt[3] = b.t;
}

void ref_shape_flatten(void *s, int32_t *contourOffsets, double *points, int32_t *types, int32_t *colors, int32_t *windings) {
    const Shape &shape = *(Shape *) s;
    int e = 0, c = 0;
    for (std::vector<Contour>::const_iterator contour = shape.contours.begin(); contour != shape.contours.end(); ++contour, ++c) {
        contourOffsets[c] = e;
        if (windings)
            windings[c] = contour->winding();
        for (std::vector<EdgeHolder>::const_iterator edge = contour->edges.begin(); edge != contour->edges.end(); ++edge, ++e) {
            int type = (*edge)->type();
            const Point2 *cp = (*edge)->controlPoints();
            memset(points+8*e, 0, 8*sizeof(double));
            for (int i = 0; i <= type; ++i)
                points[8*e+2*i] = cp[i].x, points[8*e+2*i+1] = cp[i].y;
            types[e] = type;
            colors[e] = (int) (*edge)->color;
        }
    }
    contourOffsets[c] = e;
}

/// generateSDF/PSDF/MSDF/MTSDF (msdfgen.h:46-56). mode 1..4; xf = {sx, sy, tx, ty, rangeLower, rangeUpper}.
void ref_generate(void *s, int mode, float *pixels, int w, int h, int rowStride, int yDown, const double *xf,
                  int overlap, int ecMode, int ecDist, double minDev, double minImp, unsigned char *stencilBuffer) {
    generateOne(*(Shape *) s, mode, pixels, w, h, rowStride, yDown, xf, makeConfig(overlap, ecMode, ecDist, minDev, minImp, stencilBuffer));
}

/// msdfErrorCorrection (core/msdf-error-correction.h:15-16) on an existing 3- or 4-channel bitmap.
void ref_error_correction(void *s, int channels, float *pixels, int w, int h, int rowStride, int yDown, const double *xf,
                          int overlap, int ecMode, int ecDist, double minDev, double minImp, unsigned char *stencilBuffer) {
    SDFTransformation t(Projection(Vector2(xf[0], xf[1]), Vector2(xf[2], xf[3])), DistanceMapping(Range(xf[4], xf[5])));
    YAxisOrientation yo = yDown ? Y_DOWNWARD : Y_UPWARD;
    MSDFGeneratorConfig cfg = makeConfig(overlap, ecMode, ecDist, minDev, minImp, stencilBuffer);
    if (channels == 3)
        msdfErrorCorrection(BitmapSection<float, 3>(pixels, w, h, rowStride, yo), *(Shape *) s, t, cfg);
    else
        msdfErrorCorrection(BitmapSection<float, 4>(pixels, w, h, rowStride, yo), *(Shape *) s, t, cfg);
}

/// msdfFastDistanceErrorCorrection / msdfFastEdgeErrorCorrection (core/msdf-error-correction.h:21-34): the shapeless passes.
void ref_fast_error_correction(int channels, float *pixels, int w, int h, int rowStride, const double *xf, double minDev, int protectAll) {
    SDFTransformation t(Projection(Vector2(xf[0], xf[1]), Vector2(xf[2], xf[3])), DistanceMapping(Range(xf[4], xf[5])));
    if (channels == 3) {
        BitmapSection<float, 3> sdf(pixels, w, h, rowStride);
        if (protectAll) msdfFastEdgeErrorCorrection(sdf, t, minDev); else msdfFastDistanceErrorCorrection(sdf, t, minDev);
    } else {
        BitmapSection<float, 4> sdf(pixels, w, h, rowStride);
        if (protectAll) msdfFastEdgeErrorCorrection(sdf, t, minDev); else msdfFastDistanceErrorCorrection(sdf, t, minDev);
    }
}

/// Stencil after each stage of the default pipeline (core/msdf-error-correction.cpp:12-48), for stage-by-stage diffs.
/// stages: w*h bytes each: [0] after protectCorners, [1] after protectEdges, [2] after findErrors(sdf), [3] after protectAll+findErrors(sdf,shape)
void ref_ec_stages(void *s, int channels, const float *pixels, int w, int h, const double *xf, int overlap, double minDev, double minImp, unsigned char *stages) {
    const Shape &shape = *(Shape *) s;
    SDFTransformation t(Projection(Vector2(xf[0], xf[1]), Vector2(xf[2], xf[3])), DistanceMapping(Range(xf[4], xf[5])));
    std::vector<unsigned char> buf(size_t(w)*h);
    BitmapSection<byte, 1> stencil(buf.data(), w, h);
    MSDFErrorCorrection ec(stencil, t);
    ec.setMinDeviationRatio(minDev);
    ec.setMinImproveRatio(minImp);
    size_t n = size_t(w)*h;
    ec.protectCorners(shape);
    memcpy(stages, buf.data(), n);
    if (channels == 3) ec.protectEdges<3>(BitmapConstSection<float, 3>(pixels, w, h)); else ec.protectEdges<4>(BitmapConstSection<float, 4>(pixels, w, h));
    memcpy(stages+n, buf.data(), n);
    if (channels == 3) ec.findErrors<3>(BitmapConstSection<float, 3>(pixels, w, h)); else ec.findErrors<4>(BitmapConstSection<float, 4>(pixels, w, h));
    memcpy(stages+2*n, buf.data(), n);
    ec.protectAll();
    if (channels == 3) {
        if (overlap) ec.findErrors<OverlappingContourCombiner, 3>(BitmapConstSection<float, 3>(pixels, w, h), shape);
        else ec.findErrors<SimpleContourCombiner, 3>(BitmapConstSection<float, 3>(pixels, w, h), shape);
    } else {
        if (overlap) ec.findErrors<OverlappingContourCombiner, 4>(BitmapConstSection<float, 4>(pixels, w, h), shape);
        else ec.findErrors<SimpleContourCombiner, 4>(BitmapConstSection<float, 4>(pixels, w, h), shape);
    }
    memcpy(stages+3*n, buf.data(), n);
}

/// distanceSignCorrection (core/rasterization.cpp:19-88; declared core/rasterization.h:17-19), in place. xf = {sx, sy, tx, ty}.
void ref_sign_correction(void *s, int channels, float *pixels, int w, int h, int rowStride, int yDown, const double *xf, float sdfZeroValue, int fillRule) {
    const Shape &shape = *(Shape *) s;
    Projection proj(Vector2(xf[0], xf[1]), Vector2(xf[2], xf[3]));
    YAxisOrientation yo = yDown ? Y_DOWNWARD : Y_UPWARD;
    switch (channels) {
        case 1: distanceSignCorrection(BitmapSection<float, 1>(pixels, w, h, rowStride, yo), shape, proj, sdfZeroValue, (FillRule) fillRule); break;
        case 3: distanceSignCorrection(BitmapSection<float, 3>(pixels, w, h, rowStride, yo), shape, proj, sdfZeroValue, (FillRule) fillRule); break;
        case 4: distanceSignCorrection(BitmapSection<float, 4>(pixels, w, h, rowStride, yo), shape, proj, sdfZeroValue, (FillRule) fillRule); break;
    }
}

void ref_pixel_float_to_byte(const float *in, unsigned char *out, long n) {
    for (long i = 0; i < n; ++i)
        out[i] = pixelFloatToByte(in[i]);
}

int ref_render_sdf(float *out, int ow, int oh, int No, const float *sdf, int sw, int sh, int Ns, double rangeLower, double rangeUpper, float sdThreshold) {
    Range range(rangeLower, rangeUpper);
    #define RENDER(NO, NS) renderSDF(BitmapSection<float, NO>(out, ow, oh), BitmapConstSection<float, NS>(sdf, sw, sh), range, sdThreshold)
    if (No == 1 && Ns == 1) RENDER(1, 1);
    else if (No == 3 && Ns == 1) RENDER(3, 1);
    else if (No == 1 && Ns == 3) RENDER(1, 3);
    else if (No == 3 && Ns == 3) RENDER(3, 3);
    else if (No == 1 && Ns == 4) RENDER(1, 4);
    else if (No == 4 && Ns == 4) RENDER(4, 4);
    else return -1;
    #undef RENDER
    return 0;
}

double ref_estimate_sdf_error(void *s, const float *px, int w, int h, int N, const double *xf, int scanlinesPerRow, int fillRule) {
    const Shape &shape = *(Shape *) s;
    Projection proj(Vector2(xf[0], xf[1]), Vector2(xf[2], xf[3]));
    switch (N) {
        case 1: return estimateSDFError(BitmapConstSection<float, 1>(px, w, h), shape, proj, scanlinesPerRow, (FillRule) fillRule);
        case 3: return estimateSDFError(BitmapConstSection<float, 3>(px, w, h), shape, proj, scanlinesPerRow, (FillRule) fillRule);
        default: return estimateSDFError(BitmapConstSection<float, 4>(px, w, h), shape, proj, scanlinesPerRow, (FillRule) fillRule);
    }
}

void ref_simulate_8bit(float *px, int w, int h, int N) {
    if (N == 1) simulate8bit(BitmapSection<float, 1>(px, w, h));
    else if (N == 3) simulate8bit(BitmapSection<float, 3>(px, w, h));
    else simulate8bit(BitmapSection<float, 4>(px, w, h));
}

void ref_rasterize(void *s, float *pixels, int w, int h, int rowStride, int yDown, const double *xf, int fillRule) {
    rasterize(BitmapSection<float, 1>(pixels, w, h, rowStride, yDown ? Y_DOWNWARD : Y_UPWARD), *(Shape *) s, Projection(Vector2(xf[0], xf[1]), Vector2(xf[2], xf[3])), (FillRule) fillRule);
}

/// EdgeSegment::scanlineIntersections (core/edge-segments.cpp:279-403) on one edge: returns n, fills x[3], dy[3].
int ref_scanline_intersections(int type, const double *p, double y, double *x, int32_t *dy) {
    EdgeSegment *edge = NULL;
    switch (type) {
        case 1: edge = new LinearSegment(Point2(p[0], p[1]), Point2(p[2], p[3])); break;
        case 2: edge = new QuadraticSegment(Point2(p[0], p[1]), Point2(p[2], p[3]), Point2(p[4], p[5])); break;
        default: edge = new CubicSegment(Point2(p[0], p[1]), Point2(p[2], p[3]), Point2(p[4], p[5]), Point2(p[6], p[7])); break;
    }
    int d[3] = { 0, 0, 0 };
    int n = edge->scanlineIntersections(x, d, y);
    for (int i = 0; i < 3; ++i)
        dy[i] = d[i];
    delete edge;
    return n;
}

/// EdgeSegment::signedDistance (core/edge-segments.cpp:173/187/228) on one edge. out = {distance, dot, param}
void ref_signed_distance(int type, const double *p, double ox, double oy, double *out) {
    EdgeSegment *edge = NULL;
    switch (type) {
        case 1: edge = new LinearSegment(Point2(p[0], p[1]), Point2(p[2], p[3])); break;
        case 2: edge = new QuadraticSegment(Point2(p[0], p[1]), Point2(p[2], p[3]), Point2(p[4], p[5])); break;
        case 3: edge = new CubicSegment(Point2(p[0], p[1]), Point2(p[2], p[3]), Point2(p[4], p[5]), Point2(p[6], p[7])); break;
    }
    double param = 0;
    SignedDistance sd = edge->signedDistance(Point2(ox, oy), param);
    out[0] = sd.distance, out[1] = sd.dot, out[2] = param;
    delete edge;
}

int ref_solve_cubic(double *x, double a, double b, double c, double d) { return solveCubic(x, a, b, c, d); }
int ref_solve_quadratic(double *x, double a, double b, double c) { return solveQuadratic(x, a, b, c); }

/// ShapeDistanceFinder<CC>::oneShotDistance (core/ShapeDistanceFinder.hpp:36-58). selector 1..4 = true/perp/multi/multi+true.
void ref_oneshot_distance(void *s, int selector, int overlap, int n, const double *pts, double *out /* n*4 */) {
    const Shape &shape = *(Shape *) s;
    for (int i = 0; i < n; ++i) {
        Point2 p(pts[2*i], pts[2*i+1]);
        double *o = out+4*i;
        o[0] = o[1] = o[2] = o[3] = 0;
        switch (selector) {
            case 1:
                o[0] = overlap ? ShapeDistanceFinder<OverlappingContourCombiner<TrueDistanceSelector> >::oneShotDistance(shape, p)
                               : ShapeDistanceFinder<SimpleContourCombiner<TrueDistanceSelector> >::oneShotDistance(shape, p);
                break;
            case 2:
                o[0] = overlap ? ShapeDistanceFinder<OverlappingContourCombiner<PerpendicularDistanceSelector> >::oneShotDistance(shape, p)
                               : ShapeDistanceFinder<SimpleContourCombiner<PerpendicularDistanceSelector> >::oneShotDistance(shape, p);
                break;
            case 3: {
                MultiDistance d = overlap ? ShapeDistanceFinder<OverlappingContourCombiner<MultiDistanceSelector> >::oneShotDistance(shape, p)
                                          : ShapeDistanceFinder<SimpleContourCombiner<MultiDistanceSelector> >::oneShotDistance(shape, p);
                o[0] = d.r, o[1] = d.g, o[2] = d.b;
                break;
            }
            case 4: {
                MultiAndTrueDistance d = overlap ? ShapeDistanceFinder<OverlappingContourCombiner<MultiAndTrueDistanceSelector> >::oneShotDistance(shape, p)
                                                 : ShapeDistanceFinder<SimpleContourCombiner<MultiAndTrueDistanceSelector> >::oneShotDistance(shape, p);
                o[0] = d.r, o[1] = d.g, o[2] = d.b, o[3] = d.a;
                break;
            }
        }
    }
}

/// CPU baseline: generate nGlyphs tiles (each w*h*N floats, contiguous) through the reference with a glyph-parallel
/// std::thread pool (one glyph per task, per-thread EC stencil buffer -- SURVEY.md 8d). Returns elapsed seconds.
double ref_generate_batch_timed(void **shapes, int nGlyphs, int mode, float *pixels, int w, int h, const double *xfs /* nGlyphs*6 */,
                                int overlap, int ecMode, int ecDist, double minDev, double minImp, int threads) {
    int N = mode <= 2 ? 1 : mode == 3 ? 3 : 4;
    std::atomic<int> next(0);
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&]() {
        std::vector<unsigned char> stencil(size_t(w)*h);
        MSDFGeneratorConfig cfg = makeConfig(overlap, ecMode, ecDist, minDev, minImp, stencil.data());
        for (int g; (g = next++) < nGlyphs;)
            generateOne(*(Shape *) shapes[g], mode, pixels+size_t(g)*w*h*N, w, h, w*N, 0, xfs+6*g, cfg);
    };
    if (threads <= 1)
        worker();
    else {
        std::vector<std::thread> pool;
        for (int i = 0; i < threads; ++i)
            pool.push_back(std::thread(worker));
        for (size_t i = 0; i < pool.size(); ++i)
            pool[i].join();
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
}

}
