// TEST TOOLING: a tiny msdfgen *client* written against the reference's public headers exactly like the README example
// (README.md:114-142): read a shape description, normalize, colour, generateMSDF / generateMTSDF / generateSDF, dump raw floats.
// It is linked against msdfgen_amd's C++ shim (which provides msdfgen::generate*) plus the reference's remaining objects
// (shape description parser, Shape, edge colouring ...) -- never against the reference's own msdfgen.o / msdf-error-correction.o.
//   usage: shim_check <shapedesc-file> <out.bin> <mode 1|2|3|4> <w> <h> <scale> <tx> <ty> <range> [ydown [scanline-fill-rule+1]]
// With a fill rule the client follows main.cpp's -scanline flow (main.cpp:1233-1298): generate without correction, simple
// combiner; distanceSignCorrection; msdfErrorCorrection with DO_NOT_CHECK_DISTANCE.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#include "msdfgen.h"

using namespace msdfgen;

extern "C" double msdfgen_hip_shim_flatten_ms(const msdfgen::Shape *const *shapes, int n, int threads, int reps, long long *edges_out);

// shim_check flatten <csr.bin> <threads> <reps>: builds msdfgen::Shape objects from a flat CSR dump (int32 nGlyphs, nContours, nEdges;
// gco[nGlyphs+1]; co[nContours+1]; f64 points[nEdges][8]; u8 types[nEdges]; u8 colors[nEdges]) the way a font loader would (heap-allocated
// edge segments behind EdgeHolders), then times the shim's Shape -> CSR flattening over all of them. Prints one JSON line.
static int flattenBench(const char *path, int threads, int reps) {
    FILE *f = fopen(path, "rb");
    if (!f)
        return 3;
    int32_t hdr[3];
    if (fread(hdr, sizeof(int32_t), 3, f) != 3)
        return 3;
    const int nG = hdr[0], nC = hdr[1], nE = hdr[2];
    std::vector<int32_t> gco((size_t) nG+1), co((size_t) nC+1);
    std::vector<double> pts((size_t) nE*8);
    std::vector<unsigned char> types((size_t) nE), colors((size_t) nE);
    if (fread(gco.data(), 4, gco.size(), f) != gco.size() || fread(co.data(), 4, co.size(), f) != co.size() || fread(pts.data(), 8, pts.size(), f) != pts.size() ||
        fread(types.data(), 1, types.size(), f) != types.size() || fread(colors.data(), 1, colors.size(), f) != colors.size())
        return 3;
    fclose(f);
    std::vector<Shape> shapes((size_t) nG);
    for (int g = 0; g < nG; ++g)
        for (int c = gco[g]; c < gco[g+1]; ++c) {
            Contour &contour = shapes[g].addContour();
            for (int e = co[c]; e < co[c+1]; ++e) {
                const double *p = &pts[(size_t) e*8];
                const EdgeColor col = (EdgeColor) colors[e];
                if (types[e] == 1)
                    contour.addEdge(EdgeHolder(Point2(p[0], p[1]), Point2(p[2], p[3]), col));
                else if (types[e] == 2)
                    contour.addEdge(EdgeHolder(Point2(p[0], p[1]), Point2(p[2], p[3]), Point2(p[4], p[5]), col));
                else
                    contour.addEdge(EdgeHolder(Point2(p[0], p[1]), Point2(p[2], p[3]), Point2(p[4], p[5]), Point2(p[6], p[7]), col));
            }
        }
    std::vector<const Shape *> ptrs((size_t) nG);
    for (int g = 0; g < nG; ++g)
        ptrs[g] = &shapes[g];
    long long edges = 0;
    const double ms = msdfgen_hip_shim_flatten_ms(ptrs.data(), nG, threads, reps, &edges);
    printf("{\"ms_flatten\": %.4f, \"shapes\": %d, \"edges\": %lld, \"threads\": %d, \"reps\": %d}\n", ms, nG, edges, threads, reps);
    return edges == nE ? 0 : 5;
}

int main(int argc, char **argv) {
    if (argc >= 5 && !strcmp(argv[1], "flatten"))
        return flattenBench(argv[2], atoi(argv[3]), atoi(argv[4]));
    if (argc < 10)
        return 2;
    FILE *f = fopen(argv[1], "r");
    Shape shape;
    if (!f || !readShapeDescription(f, shape))
        return 3;
    fclose(f);
    shape.normalize();
    edgeColoringSimple(shape, 3.0);
    const int mode = atoi(argv[3]), w = atoi(argv[4]), h = atoi(argv[5]);
    const double scale = atof(argv[6]), tx = atof(argv[7]), ty = atof(argv[8]), range = atof(argv[9]);
    const YAxisOrientation yo = argc > 10 && atoi(argv[10]) ? Y_DOWNWARD : Y_UPWARD;
    const int N = mode <= 2 ? 1 : mode;
    std::vector<float> px((size_t) w*h*N);
    SDFTransformation t(Projection(scale, Vector2(tx, ty)), Range(range));
    const int scanline = argc > 11 ? atoi(argv[11]) : 0;
    try {
        if (scanline == 9) {                                                // renderSDF + simulate8bit (core/render-sdf.h) of the generated field
            std::vector<float> field((size_t) w*h*N);
            switch (mode) {
                case 1: generateSDF(BitmapSection<float, 1>(field.data(), w, h, yo), shape, t); break;
                case 3: generateMSDF(BitmapSection<float, 3>(field.data(), w, h, yo), shape, t); break;
                default: generateMTSDF(BitmapSection<float, 4>(field.data(), w, h, yo), shape, t); break;
            }
            px.assign((size_t) 2*w*2*h, 0.f);                               // 1-channel preview at twice the size
            BitmapSection<float, 1> preview(px.data(), 2*w, 2*h);
            switch (mode) {
                case 1: renderSDF(preview, BitmapConstSection<float, 1>(field.data(), w, h), Range(range), .5f); break;
                case 3: renderSDF(preview, BitmapConstSection<float, 3>(field.data(), w, h), Range(range), .5f); break;
                default: renderSDF(preview, BitmapConstSection<float, 4>(field.data(), w, h), Range(range), .5f); break;
            }
            simulate8bit(preview);
        } else if (scanline == 7 || scanline == 8) {                          // the shapeless passes (core/msdf-error-correction.h:21-34) on an uncorrected field
            MSDFGeneratorConfig gen(true, ErrorCorrectionConfig(ErrorCorrectionConfig::DISABLED));
            if (mode == 3) {
                BitmapSection<float, 3> b(px.data(), w, h, yo);
                generateMSDF(b, shape, t, gen);
                if (scanline == 7) msdfFastDistanceErrorCorrection(b, t); else msdfFastEdgeErrorCorrection(b, Projection(scale, Vector2(tx, ty)), Range(range), 1.5);
            } else {
                BitmapSection<float, 4> b(px.data(), w, h, yo);
                generateMTSDF(b, shape, t, gen);
                if (scanline == 7) msdfFastDistanceErrorCorrection(b, Projection(scale, Vector2(tx, ty)), Range(range)); else msdfFastEdgeErrorCorrection(b, t, 1.5);
            }
        } else if (scanline) {
            const FillRule rule = (FillRule) (scanline-1);
            const Projection proj(scale, Vector2(tx, ty));
            MSDFGeneratorConfig gen(false, ErrorCorrectionConfig(ErrorCorrectionConfig::DISABLED)), post;
            post.overlapSupport = false;
            post.errorCorrection.distanceCheckMode = ErrorCorrectionConfig::DO_NOT_CHECK_DISTANCE;
            switch (mode) {
                case 1: {
                    BitmapSection<float, 1> b(px.data(), w, h, yo);
                    generateSDF(b, shape, t, GeneratorConfig(false));
                    distanceSignCorrection(b, shape, proj, .5f, rule);
                    break;
                }
                case 3: {
                    BitmapSection<float, 3> b(px.data(), w, h, yo);
                    generateMSDF(b, shape, t, gen);
                    distanceSignCorrection(b, shape, proj, .5f, rule);
                    msdfErrorCorrection(b, shape, t, post);
                    break;
                }
                case 4: {
                    BitmapSection<float, 4> b(px.data(), w, h, yo);
                    generateMTSDF(b, shape, t, gen);
                    distanceSignCorrection(b, shape, proj, rule);           // legacy overload (rasterization.h:24)
                    msdfErrorCorrection(b, shape, t, post);
                    break;
                }
                default: {                                                  // mode 2 here: plain coverage
                    rasterize(BitmapSection<float, 1>(px.data(), w, h, yo), shape, proj, rule);
                    break;
                }
            }
        } else switch (mode) {
            case 1: generateSDF(BitmapSection<float, 1>(px.data(), w, h, yo), shape, t); break;
            case 2: generatePSDF(BitmapSection<float, 1>(px.data(), w, h, yo), shape, t); break;
            case 3: generateMSDF(BitmapSection<float, 3>(px.data(), w, h, yo), shape, t); break;
            default: generateMTSDF(BitmapSection<float, 4>(px.data(), w, h, yo), shape, Projection(scale, Vector2(tx, ty)), Range(range)); break;
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 4;
    }
    f = fopen(argv[2], "wb");
    fwrite(px.data(), sizeof(float), px.size(), f);
    fclose(f);
    return 0;
}
