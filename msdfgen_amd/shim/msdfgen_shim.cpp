// msdfgen_shim.cpp -- C++ drop-in: msdfgen's OWN generator signatures (msdfgen.h:46-69, core/msdf-error-correction.h:15-18,
// core/rasterization.h:13-27, core/render-sdf.h:12-22),
// implemented on the MI355X through the C ABI of libmsdfgen_hip.so.
//
// Build against the user's msdfgen checkout (headers only; this file includes <msdfgen.h>) and link it INSTEAD of the reference's
// definitions of the same functions (core/msdfgen.cpp:78-162, core/msdf-error-correction.cpp:61-72, core/rasterization.cpp) -- see
// INTEGRATION.md.
// Callers such as msdf-atlas-gen's glyph generators then run the HIP path unchanged.
//
// Differences from the reference that a caller can observe: none in the texels (see tests); the functions can now fail (no
// device, HIP error) and there is deliberately NO CPU fallback -- failures throw std::runtime_error with the library's message.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "msdfgen.h"
#include "msdfgen_hip.h"
#include "msdfgen_hip_batch.hpp"

namespace msdfgen {

namespace {

struct FlatShape {
    std::vector<int32_t> contourOffsets;
    std::vector<double> points;
    std::vector<uint8_t> types, colors;
};

// const Shape & -> CSR edge buffer (read-only traversal: contours[i].edges[j]->type()/controlPoints()/color, core/Shape.h:24, core/edge-segments.h:28-31).
// (`flat` is the calling thread's reusable buffer: after the first calls of a thread the four vectors have their capacity and a call allocates nothing)
void flatten(const Shape &shape, FlatShape &flat) {
    flat.contourOffsets.clear(), flat.points.clear(), flat.types.clear(), flat.colors.clear();
    const int edges = shape.edgeCount();
    flat.contourOffsets.reserve(shape.contours.size()+1);
    flat.points.reserve(8*(size_t) edges);
    flat.types.reserve(edges);
    flat.colors.reserve(edges);
    flat.contourOffsets.push_back(0);
    for (std::vector<Contour>::const_iterator contour = shape.contours.begin(); contour != shape.contours.end(); ++contour) {
        for (std::vector<EdgeHolder>::const_iterator edge = contour->edges.begin(); edge != contour->edges.end(); ++edge) {
            const int type = (*edge)->type();
            const Point2 *p = (*edge)->controlPoints();
            for (int i = 0; i < 4; ++i) {
                flat.points.push_back(i <= type ? p[i].x : 0.);
                flat.points.push_back(i <= type ? p[i].y : 0.);
            }
            flat.types.push_back((uint8_t) type);
            flat.colors.push_back((uint8_t) (*edge)->color);
        }
        flat.contourOffsets.push_back((int32_t) flat.types.size());
    }
    if (flat.types.empty()) {            // keep the pointers valid for an empty shape
        flat.points.resize(8);
        flat.types.push_back(1);
        flat.colors.push_back(0);
    }
}

// Projection and DistanceMapping keep their members private (core/Projection.h:31-33, core/DistanceMapping.h:28-32).
// Projection is recovered exactly through its public API; DistanceMapping by its object representation (two doubles, in the
// order {scale, translate}), verified against its public operator() before use.
void transformationToXf(const SDFTransformation &t, double xf[6]) {
    const Vector2 scale = t.projectVector(Vector2(1, 1));          // scale*1, exact (core/Projection.cpp:18-20)
    const Point2 origin = t.unproject(Point2(0, 0));               // 0/scale-translate = -translate, exact (core/Projection.cpp:14-16)
    xf[0] = scale.x, xf[1] = scale.y, xf[2] = -origin.x, xf[3] = -origin.y;
    static_assert(sizeof(DistanceMapping) == 2*sizeof(double), "DistanceMapping layout changed; update the shim");
    double raw[2];
    std::memcpy(raw, &t.distanceMapping, sizeof(raw));
    const double probes[3] = { 0., 1., -.375 };
    for (int i = 0; i < 3; ++i) {
        // bit patterns, not ==: a degenerate Range(0) maps to inf*(x-0) = nan on both sides, which the reference simply writes out
        const double mine = raw[0]*(probes[i]+raw[1]), theirs = t.distanceMapping(probes[i]);
        if (std::memcmp(&mine, &theirs, sizeof(double)) != 0 && !(mine != mine && theirs != theirs))
            throw std::runtime_error("msdfgen_hip shim: DistanceMapping layout does not match {scale, translate}");
    }
    xf[4] = raw[0], xf[5] = raw[1];
}

MsdfHipConfig makeConfig(bool overlapSupport, const ErrorCorrectionConfig *ec) {
    MsdfHipConfig cfg;
    msdfhip_default_config(&cfg);
    cfg.overlap_support = overlapSupport ? 1 : 0;
    if (ec) {
        cfg.ec_mode = (int) ec->mode;                              // same enumerator order as core/generator-config.h:20-38
        cfg.ec_distance_check = (int) ec->distanceCheckMode;
        cfg.min_deviation_ratio = ec->minDeviationRatio;
        cfg.min_improve_ratio = ec->minImproveRatio;
    }
    return cfg;
}

// The reference's functions return void and cannot fail; these can (no device, HIP error, shape too complex). By default a failure
// throws std::runtime_error -- also out of a caller's worker thread. A host that cannot take exceptions there switches to the status
// mode (msdfgen_hip_shim_set_nothrow(1)): a failed call leaves the output untouched and records the status for the calling thread.
std::atomic<int> gNoThrow(0);
thread_local int tlsStatus = MSDFHIP_OK;
thread_local std::string tlsMessage;

void check(int rc, const char *what) {
    tlsStatus = rc;
    if (rc == MSDFHIP_OK)
        return;
    tlsMessage = std::string("msdfgen_hip: ")+what+" failed: "+msdfhip_last_error();
    if (!gNoThrow.load())
        throw std::runtime_error(tlsMessage);
}

template <int N>
void generate(int mode, const BitmapSection<float, N> &output, const Shape &shape, const SDFTransformation &transformation, bool overlapSupport, const ErrorCorrectionConfig *ec) {
    static thread_local FlatShape flat;
    flatten(shape, flat);
    double xf[6];
    transformationToXf(transformation, xf);
    MsdfHipConfig cfg = makeConfig(overlapSupport, ec);
    cfg.stencil_y_down = output.yOrientation == Y_DOWNWARD;                // row order of ErrorCorrectionConfig::buffer, see msdfgen_hip.h
    const int flip = shape.getYAxisOrientation() != output.yOrientation;   // output.reorient(shape.getYAxisOrientation()), core/msdfgen.cpp:55
    check(msdfhip_generate(mode, output.pixels, output.width, output.height, output.rowStride, flip, flat.contourOffsets.data(),
                           (int) shape.contours.size(), flat.points.data(), flat.types.data(), flat.colors.data(), xf, &cfg, ec ? ec->buffer : NULL),
          "generate");
}

template <int N>
void correct(const BitmapSection<float, N> &sdf, const Shape &shape, const SDFTransformation &transformation, const MSDFGeneratorConfig &config) {
    static thread_local FlatShape flat;
    flatten(shape, flat);
    double xf[6];
    transformationToXf(transformation, xf);
    MsdfHipConfig cfg = makeConfig(config.overlapSupport, &config.errorCorrection);
    cfg.stencil_y_down = sdf.yOrientation == Y_DOWNWARD;
    const int flip = shape.getYAxisOrientation() != sdf.yOrientation;
    check(msdfhip_error_correction(N, sdf.pixels, sdf.width, sdf.height, sdf.rowStride, flip, flat.contourOffsets.data(), (int) shape.contours.size(),
                                   flat.points.data(), flat.types.data(), flat.colors.data(), xf, &cfg, config.errorCorrection.buffer),
          "msdfErrorCorrection");
}

template <int N>
void signCorrect(const BitmapSection<float, N> &sdf, const Shape &shape, const Projection &projection, float sdfZeroValue, FillRule fillRule) {
    static thread_local FlatShape flat;
    flatten(shape, flat);
    const Vector2 one = projection.projectVector(Vector2(1, 1)), origin = projection.unproject(Point2(0, 0));
    const double xf[6] = { one.x, one.y, -origin.x, -origin.y, 1, 0 };       // scale, translate (Projection.cpp:12-36)
    const int flip = shape.getYAxisOrientation() != sdf.yOrientation;        // sdf.reorient(...), rasterization.cpp:22,39
    check(msdfhip_distance_sign_correction(N, sdf.pixels, sdf.width, sdf.height, sdf.rowStride, flip, flat.contourOffsets.data(),
                                           (int) shape.contours.size(), flat.points.data(), flat.types.data(), flat.colors.data(), xf,
                                           sdfZeroValue, (int) fillRule),
          "distanceSignCorrection");
}

}

// ---- msdfgen.h:46-53
void generateSDF(const BitmapSection<float, 1> &output, const Shape &shape, const SDFTransformation &transformation, const GeneratorConfig &config) {
    generate<1>(MSDFHIP_MODE_SDF, output, shape, transformation, config.overlapSupport, NULL);
}
void generatePSDF(const BitmapSection<float, 1> &output, const Shape &shape, const SDFTransformation &transformation, const GeneratorConfig &config) {
    generate<1>(MSDFHIP_MODE_PSDF, output, shape, transformation, config.overlapSupport, NULL);
}
void generateMSDF(const BitmapSection<float, 3> &output, const Shape &shape, const SDFTransformation &transformation, const MSDFGeneratorConfig &config) {
    generate<3>(MSDFHIP_MODE_MSDF, output, shape, transformation, config.overlapSupport, &config.errorCorrection);
}
void generateMTSDF(const BitmapSection<float, 4> &output, const Shape &shape, const SDFTransformation &transformation, const MSDFGeneratorConfig &config) {
    generate<4>(MSDFHIP_MODE_MTSDF, output, shape, transformation, config.overlapSupport, &config.errorCorrection);
}

// ---- msdfgen.h:59-63 (Projection + Range; the form msdf-atlas-gen calls)
void generateSDF(const BitmapSection<float, 1> &output, const Shape &shape, const Projection &projection, Range range, const GeneratorConfig &config) {
    generateSDF(output, shape, SDFTransformation(projection, range), config);
}
void generatePSDF(const BitmapSection<float, 1> &output, const Shape &shape, const Projection &projection, Range range, const GeneratorConfig &config) {
    generatePSDF(output, shape, SDFTransformation(projection, range), config);
}
void generatePseudoSDF(const BitmapSection<float, 1> &output, const Shape &shape, const Projection &projection, Range range, const GeneratorConfig &config) {
    generatePSDF(output, shape, SDFTransformation(projection, range), config);
}
void generateMSDF(const BitmapSection<float, 3> &output, const Shape &shape, const Projection &projection, Range range, const MSDFGeneratorConfig &config) {
    generateMSDF(output, shape, SDFTransformation(projection, range), config);
}
void generateMTSDF(const BitmapSection<float, 4> &output, const Shape &shape, const Projection &projection, Range range, const MSDFGeneratorConfig &config) {
    generateMTSDF(output, shape, SDFTransformation(projection, range), config);
}

// ---- msdfgen.h:65-69 (legacy Range/scale/translate)
void generateSDF(const BitmapSection<float, 1> &output, const Shape &shape, Range range, const Vector2 &scale, const Vector2 &translate, bool overlapSupport) {
    generateSDF(output, shape, Projection(scale, translate), range, GeneratorConfig(overlapSupport));
}
void generatePSDF(const BitmapSection<float, 1> &output, const Shape &shape, Range range, const Vector2 &scale, const Vector2 &translate, bool overlapSupport) {
    generatePSDF(output, shape, Projection(scale, translate), range, GeneratorConfig(overlapSupport));
}
void generatePseudoSDF(const BitmapSection<float, 1> &output, const Shape &shape, Range range, const Vector2 &scale, const Vector2 &translate, bool overlapSupport) {
    generatePSDF(output, shape, Projection(scale, translate), range, GeneratorConfig(overlapSupport));
}
void generateMSDF(const BitmapSection<float, 3> &output, const Shape &shape, Range range, const Vector2 &scale, const Vector2 &translate, const ErrorCorrectionConfig &errorCorrectionConfig, bool overlapSupport) {
    generateMSDF(output, shape, Projection(scale, translate), range, MSDFGeneratorConfig(overlapSupport, errorCorrectionConfig));
}
void generateMTSDF(const BitmapSection<float, 4> &output, const Shape &shape, Range range, const Vector2 &scale, const Vector2 &translate, const ErrorCorrectionConfig &errorCorrectionConfig, bool overlapSupport) {
    generateMTSDF(output, shape, Projection(scale, translate), range, MSDFGeneratorConfig(overlapSupport, errorCorrectionConfig));
}

// ---- core/msdf-error-correction.h:15-18
void msdfErrorCorrection(const BitmapSection<float, 3> &sdf, const Shape &shape, const SDFTransformation &transformation, const MSDFGeneratorConfig &config) {
    correct<3>(sdf, shape, transformation, config);
}
void msdfErrorCorrection(const BitmapSection<float, 4> &sdf, const Shape &shape, const SDFTransformation &transformation, const MSDFGeneratorConfig &config) {
    correct<4>(sdf, shape, transformation, config);
}
void msdfErrorCorrection(const BitmapSection<float, 3> &sdf, const Shape &shape, const Projection &projection, Range range, const MSDFGeneratorConfig &config) {
    correct<3>(sdf, shape, SDFTransformation(projection, range), config);
}
void msdfErrorCorrection(const BitmapSection<float, 4> &sdf, const Shape &shape, const Projection &projection, Range range, const MSDFGeneratorConfig &config) {
    correct<4>(sdf, shape, SDFTransformation(projection, range), config);
}

// ---- core/msdf-error-correction.h:21-34: the shapeless passes (findErrors(sdf) + apply, core/msdf-error-correction.cpp:50-59)
namespace {
template <int N>
void correctShapeless(const BitmapSection<float, N> &sdf, const SDFTransformation &transformation, double minDeviationRatio, bool protectAll) {
    double xf[6];
    transformationToXf(transformation, xf);
    check(msdfhip_error_correction_shapeless(N, sdf.pixels, sdf.width, sdf.height, sdf.rowStride, xf, minDeviationRatio, protectAll ? 1 : 0),
          protectAll ? "msdfFastEdgeErrorCorrection" : "msdfFastDistanceErrorCorrection");
}
}
void msdfFastDistanceErrorCorrection(const BitmapSection<float, 3> &sdf, const SDFTransformation &transformation, double minDeviationRatio) { correctShapeless<3>(sdf, transformation, minDeviationRatio, false); }
void msdfFastDistanceErrorCorrection(const BitmapSection<float, 4> &sdf, const SDFTransformation &transformation, double minDeviationRatio) { correctShapeless<4>(sdf, transformation, minDeviationRatio, false); }
void msdfFastDistanceErrorCorrection(const BitmapSection<float, 3> &sdf, const Projection &projection, Range range, double minDeviationRatio) { correctShapeless<3>(sdf, SDFTransformation(projection, range), minDeviationRatio, false); }
void msdfFastDistanceErrorCorrection(const BitmapSection<float, 4> &sdf, const Projection &projection, Range range, double minDeviationRatio) { correctShapeless<4>(sdf, SDFTransformation(projection, range), minDeviationRatio, false); }
void msdfFastDistanceErrorCorrection(const BitmapSection<float, 3> &sdf, Range pxRange, double minDeviationRatio) { correctShapeless<3>(sdf, SDFTransformation(Projection(), pxRange), minDeviationRatio, false); }
void msdfFastDistanceErrorCorrection(const BitmapSection<float, 4> &sdf, Range pxRange, double minDeviationRatio) { correctShapeless<4>(sdf, SDFTransformation(Projection(), pxRange), minDeviationRatio, false); }
void msdfFastEdgeErrorCorrection(const BitmapSection<float, 3> &sdf, const SDFTransformation &transformation, double minDeviationRatio) { correctShapeless<3>(sdf, transformation, minDeviationRatio, true); }
void msdfFastEdgeErrorCorrection(const BitmapSection<float, 4> &sdf, const SDFTransformation &transformation, double minDeviationRatio) { correctShapeless<4>(sdf, transformation, minDeviationRatio, true); }
void msdfFastEdgeErrorCorrection(const BitmapSection<float, 3> &sdf, const Projection &projection, Range range, double minDeviationRatio) { correctShapeless<3>(sdf, SDFTransformation(projection, range), minDeviationRatio, true); }
void msdfFastEdgeErrorCorrection(const BitmapSection<float, 4> &sdf, const Projection &projection, Range range, double minDeviationRatio) { correctShapeless<4>(sdf, SDFTransformation(projection, range), minDeviationRatio, true); }
void msdfFastEdgeErrorCorrection(const BitmapSection<float, 3> &sdf, Range pxRange, double minDeviationRatio) { correctShapeless<3>(sdf, SDFTransformation(Projection(), pxRange), minDeviationRatio, true); }
void msdfFastEdgeErrorCorrection(const BitmapSection<float, 4> &sdf, Range pxRange, double minDeviationRatio) { correctShapeless<4>(sdf, SDFTransformation(Projection(), pxRange), minDeviationRatio, true); }

// ---- core/render-sdf.h:12-22: together these replace the whole of core/render-sdf.cpp
namespace {
template <int NO, int NS>
void render(const BitmapSection<float, NO> &output, const BitmapConstSection<float, NS> &sdf, Range sdfPxRange, float sdThreshold) {
    check(msdfhip_render_sdf_host(output.pixels, output.width, output.height, output.rowStride, NO, sdf.pixels, sdf.width, sdf.height, sdf.rowStride, NS,
                                  sdfPxRange.lower, sdfPxRange.upper, sdThreshold),
          "renderSDF");
}
}
void renderSDF(const BitmapSection<float, 1> &output, const BitmapConstSection<float, 1> &sdf, Range sdfPxRange, float sdThreshold) { render<1, 1>(output, sdf, sdfPxRange, sdThreshold); }
void renderSDF(const BitmapSection<float, 3> &output, const BitmapConstSection<float, 1> &sdf, Range sdfPxRange, float sdThreshold) { render<3, 1>(output, sdf, sdfPxRange, sdThreshold); }
void renderSDF(const BitmapSection<float, 1> &output, const BitmapConstSection<float, 3> &sdf, Range sdfPxRange, float sdThreshold) { render<1, 3>(output, sdf, sdfPxRange, sdThreshold); }
void renderSDF(const BitmapSection<float, 3> &output, const BitmapConstSection<float, 3> &sdf, Range sdfPxRange, float sdThreshold) { render<3, 3>(output, sdf, sdfPxRange, sdThreshold); }
void renderSDF(const BitmapSection<float, 1> &output, const BitmapConstSection<float, 4> &sdf, Range sdfPxRange, float sdThreshold) { render<1, 4>(output, sdf, sdfPxRange, sdThreshold); }
void renderSDF(const BitmapSection<float, 4> &output, const BitmapConstSection<float, 4> &sdf, Range sdfPxRange, float sdThreshold) { render<4, 4>(output, sdf, sdfPxRange, sdThreshold); }
// (the reference walks N*width*height contiguous floats and ignores rowStride, core/render-sdf.cpp:172-188; rows are honoured here)
void simulate8bit(const BitmapSection<float, 1> &bitmap) { check(msdfhip_simulate_8bit_host(bitmap.pixels, bitmap.width, bitmap.height, bitmap.rowStride, 1), "simulate8bit"); }
void simulate8bit(const BitmapSection<float, 3> &bitmap) { check(msdfhip_simulate_8bit_host(bitmap.pixels, bitmap.width, bitmap.height, bitmap.rowStride, 3), "simulate8bit"); }
void simulate8bit(const BitmapSection<float, 4> &bitmap) { check(msdfhip_simulate_8bit_host(bitmap.pixels, bitmap.width, bitmap.height, bitmap.rowStride, 4), "simulate8bit"); }

// ---- core/rasterization.h:13-27: together these replace the whole of core/rasterization.cpp
void rasterize(BitmapSection<float, 1> output, const Shape &shape, const Projection &projection, FillRule fillRule) {
    static thread_local FlatShape flat;
    flatten(shape, flat);
    const Vector2 one = projection.projectVector(Vector2(1, 1)), origin = projection.unproject(Point2(0, 0));
    const double xf[6] = { one.x, one.y, -origin.x, -origin.y, 1, 0 };
    const int flip = shape.getYAxisOrientation() != output.yOrientation;     // output.reorient(...), rasterization.cpp:9
    check(msdfhip_rasterize(output.pixels, output.width, output.height, output.rowStride, flip, flat.contourOffsets.data(), (int) shape.contours.size(),
                            flat.points.data(), flat.types.data(), flat.colors.data(), xf, (int) fillRule),
          "rasterize");
}
void rasterize(const BitmapSection<float, 1> &output, const Shape &shape, const Vector2 &scale, const Vector2 &translate, FillRule fillRule) {
    rasterize(output, shape, Projection(scale, translate), fillRule);
}
void distanceSignCorrection(BitmapSection<float, 1> sdf, const Shape &shape, const Projection &projection, float sdfZeroValue, FillRule fillRule) {
    signCorrect<1>(sdf, shape, projection, sdfZeroValue, fillRule);
}
void distanceSignCorrection(BitmapSection<float, 3> sdf, const Shape &shape, const Projection &projection, float sdfZeroValue, FillRule fillRule) {
    signCorrect<3>(sdf, shape, projection, sdfZeroValue, fillRule);
}
void distanceSignCorrection(BitmapSection<float, 4> sdf, const Shape &shape, const Projection &projection, float sdfZeroValue, FillRule fillRule) {
    signCorrect<4>(sdf, shape, projection, sdfZeroValue, fillRule);
}
void distanceSignCorrection(BitmapSection<float, 1> sdf, const Shape &shape, const Projection &projection, FillRule fillRule) {
    signCorrect<1>(sdf, shape, projection, .5f, fillRule);
}
void distanceSignCorrection(BitmapSection<float, 3> sdf, const Shape &shape, const Projection &projection, FillRule fillRule) {
    signCorrect<3>(sdf, shape, projection, .5f, fillRule);
}
void distanceSignCorrection(BitmapSection<float, 4> sdf, const Shape &shape, const Projection &projection, FillRule fillRule) {
    signCorrect<4>(sdf, shape, projection, .5f, fillRule);
}
void distanceSignCorrection(const BitmapSection<float, 1> &sdf, const Shape &shape, const Vector2 &scale, const Vector2 &translate, FillRule fillRule) {
    signCorrect<1>(sdf, shape, Projection(scale, translate), .5f, fillRule);
}
void distanceSignCorrection(const BitmapSection<float, 3> &sdf, const Shape &shape, const Vector2 &scale, const Vector2 &translate, FillRule fillRule) {
    signCorrect<3>(sdf, shape, Projection(scale, translate), .5f, fillRule);
}
void distanceSignCorrection(const BitmapSection<float, 4> &sdf, const Shape &shape, const Vector2 &scale, const Vector2 &translate, FillRule fillRule) {
    signCorrect<4>(sdf, shape, Projection(scale, translate), .5f, fillRule);
}

}

// ---- msdfgen_hip_batch.hpp: a LIST of shapes in one pipelined call (msdfhip_generate_stream) ------------------------------------------------------
namespace msdfgen_hip {

namespace {

using namespace msdfgen;

// The library's view of `const Shape *const *` (MsdfHipShapeSource, include/msdfgen_hip.h): called from its host threads, read-only.
struct ShapeList {
    const Shape *const *shapes;
    const int *index;                                                      // the glyphs of this launch group (sections of one size), in list order
    static void count(void *user, int g, int32_t *nContours, int32_t *nEdges) {
        const ShapeList &list = *static_cast<const ShapeList *>(user);
        const Shape &shape = *list.shapes[list.index[g]];
        *nContours = (int32_t) shape.contours.size();
        *nEdges = (int32_t) shape.edgeCount();
    }
    // core/Shape.h:24 (contours), core/Contour.h:17 (edges), core/edge-segments.h:28-31 (type / controlPoints), core/EdgeHolder.h (one heap object per edge)
    static void fill(void *user, int g, int32_t edgeBase, int32_t *contourEnd, double *points, uint8_t *types, uint8_t *colors) {
        const ShapeList &list = *static_cast<const ShapeList *>(user);
        const Shape &shape = *list.shapes[list.index[g]];
        int32_t at = 0;
        for (std::vector<Contour>::const_iterator contour = shape.contours.begin(); contour != shape.contours.end(); ++contour) {
            for (std::vector<EdgeHolder>::const_iterator edge = contour->edges.begin(); edge != contour->edges.end(); ++edge, ++at) {
                const EdgeSegment &segment = **edge;
                const int type = segment.type();
                const Point2 *p = segment.controlPoints();
                double *dst = points+(size_t) at*8;
                for (int i = 0; i < 4; ++i)
                    dst[2*i] = i <= type ? p[i].x : 0., dst[2*i+1] = i <= type ? p[i].y : 0.;
                types[at] = (uint8_t) type;
                colors[at] = (uint8_t) segment.color;
            }
            *contourEnd++ = edgeBase+at;
        }
    }
};

template <typename T, int N>
void generateBatch(int mode, const BitmapSection<T, N> *outputs, const Shape *const *shapes, const SDFTransformation *transformations, int count, bool overlapSupport,
                   const ErrorCorrectionConfig *ec) {
    if (count <= 0)
        return;
    if (!outputs || !shapes || !transformations) {
        msdfgen::check(MSDFHIP_ERR_INVALID, "generateBatch (NULL argument)");
        return;
    }
    const bool bytes = sizeof(T) == 1;
    // one pipelined call per bitmap size (a launch parameter); an atlas run has one
    std::vector<int> order((size_t) count);
    for (int i = 0; i < count; ++i)
        order[(size_t) i] = i;
    std::stable_sort(order.begin(), order.end(), [outputs](int a, int b) {
        return outputs[a].width != outputs[b].width ? outputs[a].width < outputs[b].width : outputs[a].height < outputs[b].height;
    });
    std::vector<MsdfHipGlyph> glyphs;
    for (size_t first = 0; first < order.size(); ) {
        size_t last = first;
        const int w = outputs[order[first]].width, h = outputs[order[first]].height;
        while (last < order.size() && outputs[order[last]].width == w && outputs[order[last]].height == h)
            ++last;
        const int n = (int) (last-first);
        if (w > 0 && h > 0) {
            // all sections of the group as rectangles of ONE address range [base, end): the lowest and the highest element any of them touches
            const T *base = NULL, *end = NULL;
            for (size_t k = first; k < last; ++k) {
                const BitmapSection<T, N> &o = outputs[order[k]];
                const T *lo = o.pixels+(o.rowStride < 0 ? (ptrdiff_t) o.rowStride*(h-1) : 0), *hi = o.pixels+(o.rowStride < 0 ? 0 : (ptrdiff_t) o.rowStride*(h-1))+(ptrdiff_t) N*w;
                base = !base || lo < base ? lo : base, end = !end || hi > end ? hi : end;
            }
            glyphs.resize((size_t) n);
            for (int g = 0; g < n; ++g) {
                const int i = order[first+(size_t) g];
                msdfgen::transformationToXf(transformations[i], glyphs[(size_t) g].xf);
                glyphs[(size_t) g].out_offset = (int64_t) (outputs[i].pixels-base);
                glyphs[(size_t) g].row_stride = outputs[i].rowStride;
                glyphs[(size_t) g].flip = shapes[i]->getYAxisOrientation() != outputs[i].yOrientation;   // output.reorient(shape.getYAxisOrientation()), core/msdfgen.cpp:55
            }
            MsdfHipConfig cfg = msdfgen::makeConfig(overlapSupport, ec);
            ShapeList list = { shapes, order.data()+first };
            MsdfHipShapeSource source = { &list, ShapeList::count, ShapeList::fill };
            const int rc = msdfhip_generate_stream(-1, mode, w, h, n, &source, glyphs.data(), bytes ? NULL : (float *) base, bytes ? 0 : (size_t) (end-base),
                                                   bytes ? (uint8_t *) base : NULL, bytes ? (size_t) (end-base) : 0, NULL, &cfg);
            msdfgen::check(rc, "generateBatch");
            if (rc != MSDFHIP_OK)
                return;
        }
        first = last;
    }
}

}

void generateSDFBatch(const BitmapSection<float, 1> *outputs, const Shape *const *shapes, const SDFTransformation *transformations, int count, const GeneratorConfig &config) {
    generateBatch<float, 1>(MSDFHIP_MODE_SDF, outputs, shapes, transformations, count, config.overlapSupport, NULL);
}
void generatePSDFBatch(const BitmapSection<float, 1> *outputs, const Shape *const *shapes, const SDFTransformation *transformations, int count, const GeneratorConfig &config) {
    generateBatch<float, 1>(MSDFHIP_MODE_PSDF, outputs, shapes, transformations, count, config.overlapSupport, NULL);
}
void generateMSDFBatch(const BitmapSection<float, 3> *outputs, const Shape *const *shapes, const SDFTransformation *transformations, int count, const MSDFGeneratorConfig &config) {
    generateBatch<float, 3>(MSDFHIP_MODE_MSDF, outputs, shapes, transformations, count, config.overlapSupport, &config.errorCorrection);
}
void generateMTSDFBatch(const BitmapSection<float, 4> *outputs, const Shape *const *shapes, const SDFTransformation *transformations, int count, const MSDFGeneratorConfig &config) {
    generateBatch<float, 4>(MSDFHIP_MODE_MTSDF, outputs, shapes, transformations, count, config.overlapSupport, &config.errorCorrection);
}
void generateSDFBatch(const BitmapSection<byte, 1> *outputs, const Shape *const *shapes, const SDFTransformation *transformations, int count, const GeneratorConfig &config) {
    generateBatch<byte, 1>(MSDFHIP_MODE_SDF, outputs, shapes, transformations, count, config.overlapSupport, NULL);
}
void generatePSDFBatch(const BitmapSection<byte, 1> *outputs, const Shape *const *shapes, const SDFTransformation *transformations, int count, const GeneratorConfig &config) {
    generateBatch<byte, 1>(MSDFHIP_MODE_PSDF, outputs, shapes, transformations, count, config.overlapSupport, NULL);
}
void generateMSDFBatch(const BitmapSection<byte, 3> *outputs, const Shape *const *shapes, const SDFTransformation *transformations, int count, const MSDFGeneratorConfig &config) {
    generateBatch<byte, 3>(MSDFHIP_MODE_MSDF, outputs, shapes, transformations, count, config.overlapSupport, &config.errorCorrection);
}
void generateMTSDFBatch(const BitmapSection<byte, 4> *outputs, const Shape *const *shapes, const SDFTransformation *transformations, int count, const MSDFGeneratorConfig &config) {
    generateBatch<byte, 4>(MSDFHIP_MODE_MTSDF, outputs, shapes, transformations, count, config.overlapSupport, &config.errorCorrection);
}

}

// ---- measurement hook (not part of msdfgen's API): what the shim does to a Shape before the C ABI sees it -----------------------------
// Flattens the n shapes (const Shape & -> CSR edge buffer, flatten() above) on `threads` host threads, `reps` times; returns the median
// milliseconds of one pass over all n. bench.py reports it as end_to_end.ms_flatten (SURVEY 8d counts the host flatten in the metric).
extern "C" double msdfgen_hip_shim_flatten_ms(const msdfgen::Shape *const *shapes, int n, int threads, int reps, long long *edges_out) {
    std::vector<double> ms;
    std::atomic<long long> edges(0);
    for (int r = 0; r < reps; ++r) {
        edges.store(0);
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t)
            pool.push_back(std::thread([&, t]() {
                long long mine = 0;
                msdfgen::FlatShape flat;
                for (int g = t; g < n; g += threads) {
                    flat.contourOffsets.clear(), flat.points.clear(), flat.types.clear(), flat.colors.clear();
                    msdfgen::flatten(*shapes[g], flat);
                    mine += (long long) flat.contourOffsets.back();
                }
                edges += mine;
            }));
        for (size_t t = 0; t < pool.size(); ++t)
            pool[t].join();
        ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now()-t0).count());
    }
    std::sort(ms.begin(), ms.end());
    if (edges_out)
        *edges_out = edges.load();
    return ms.empty() ? 0. : ms[ms.size()/2];
}

// ---- failure reporting of the shim (not part of msdfgen's API) -----------------------------------------------------------------
extern "C" {
// 1: failed calls no longer throw; query msdfgen_hip_shim_last_status() / _last_error() on the same thread after a call.
void msdfgen_hip_shim_set_nothrow(int enable) { msdfgen::gNoThrow.store(enable ? 1 : 0); }
// MSDFHIP_OK (0) or the MSDFHIP_ERR_* code of the calling thread's last shim call.
int msdfgen_hip_shim_last_status(void) { return msdfgen::tlsStatus; }
const char *msdfgen_hip_shim_last_error(void) { return msdfgen::tlsStatus == MSDFHIP_OK ? "" : msdfgen::tlsMessage.c_str(); }
}
