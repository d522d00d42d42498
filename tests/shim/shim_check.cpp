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
#include <algorithm>
#include <chrono>
#include "msdfgen.h"
#include "msdfgen_hip_batch.hpp"

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

// ---- the batch entry of the shim (include/msdfgen_hip_batch.hpp) ------------------------------------------------------------------------------------
// A dump as above followed by f64 xf[nGlyphs][6] = { scale.x, scale.y, translate.x, translate.y, range lower, range upper } and u8 inverseY[nGlyphs].
struct Dump {
    std::vector<Shape> shapes;
    std::vector<const Shape *> ptrs;
    std::vector<SDFTransformation> xf;
    long long edges;
};
static bool loadDump(const char *path, Dump &d) {
    FILE *f = fopen(path, "rb");
    if (!f)
        return false;
    int32_t hdr[3];
    if (fread(hdr, sizeof(int32_t), 3, f) != 3)
        return false;
    const int nG = hdr[0], nC = hdr[1], nE = hdr[2];
    std::vector<int32_t> gco((size_t) nG+1), co((size_t) nC+1);
    std::vector<double> pts((size_t) nE*8), xf((size_t) nG*6);
    std::vector<unsigned char> types((size_t) nE), colors((size_t) nE), inverse((size_t) nG);
    if (fread(gco.data(), 4, gco.size(), f) != gco.size() || fread(co.data(), 4, co.size(), f) != co.size() || fread(pts.data(), 8, pts.size(), f) != pts.size() ||
        fread(types.data(), 1, types.size(), f) != types.size() || fread(colors.data(), 1, colors.size(), f) != colors.size() ||
        fread(xf.data(), 8, xf.size(), f) != xf.size() || fread(inverse.data(), 1, inverse.size(), f) != inverse.size())
        return false;
    fclose(f);
    d.shapes.resize((size_t) nG);
    for (int g = 0; g < nG; ++g) {
        for (int c = gco[g]; c < gco[g+1]; ++c) {
            Contour &contour = d.shapes[g].addContour();
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
        d.shapes[g].setYAxisOrientation(inverse[g] ? Y_DOWNWARD : Y_UPWARD);
        const double *x = &xf[(size_t) g*6];
        d.xf.push_back(SDFTransformation(Projection(Vector2(x[0], x[1]), Vector2(x[2], x[3])), Range(x[4], x[5])));
    }
    d.ptrs.resize((size_t) nG);
    for (int g = 0; g < nG; ++g)
        d.ptrs[g] = &d.shapes[g];
    d.edges = nE;
    return true;
}

extern "C" int msdfhip_host_alloc(void **p, size_t bytes);
extern "C" int msdfhip_host_free(void *p);

static double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int N>
static void perCall(int mode, const BitmapSection<float, N> &out, const Shape &shape, const SDFTransformation &t) {
    switch (mode) {
        case 1: generateSDF(BitmapSection<float, 1>(out.pixels, out.width, out.height, out.rowStride, out.yOrientation), shape, t); break;
        case 2: generatePSDF(BitmapSection<float, 1>(out.pixels, out.width, out.height, out.rowStride, out.yOrientation), shape, t); break;
        case 3: generateMSDF(BitmapSection<float, 3>(out.pixels, out.width, out.height, out.rowStride, out.yOrientation), shape, t); break;
        default: generateMTSDF(BitmapSection<float, 4>(out.pixels, out.width, out.height, out.rowStride, out.yOrientation), shape, t); break;
    }
}
template <typename T, int N>
static void batchCall(int mode, const std::vector<BitmapSection<T, N> > &outs, const Dump &d, int first, int count);
template <> void batchCall<float, 1>(int mode, const std::vector<BitmapSection<float, 1> > &o, const Dump &d, int first, int n) {
    if (mode == 1) msdfgen_hip::generateSDFBatch(o.data(), d.ptrs.data()+first, d.xf.data()+first, n); else msdfgen_hip::generatePSDFBatch(o.data(), d.ptrs.data()+first, d.xf.data()+first, n);
}
template <> void batchCall<float, 3>(int, const std::vector<BitmapSection<float, 3> > &o, const Dump &d, int first, int n) { msdfgen_hip::generateMSDFBatch(o.data(), d.ptrs.data()+first, d.xf.data()+first, n); }
template <> void batchCall<float, 4>(int, const std::vector<BitmapSection<float, 4> > &o, const Dump &d, int first, int n) { msdfgen_hip::generateMTSDFBatch(o.data(), d.ptrs.data()+first, d.xf.data()+first, n); }
template <> void batchCall<byte, 1>(int mode, const std::vector<BitmapSection<byte, 1> > &o, const Dump &d, int first, int n) {
    if (mode == 1) msdfgen_hip::generateSDFBatch(o.data(), d.ptrs.data()+first, d.xf.data()+first, n); else msdfgen_hip::generatePSDFBatch(o.data(), d.ptrs.data()+first, d.xf.data()+first, n);
}
template <> void batchCall<byte, 3>(int, const std::vector<BitmapSection<byte, 3> > &o, const Dump &d, int first, int n) { msdfgen_hip::generateMSDFBatch(o.data(), d.ptrs.data()+first, d.xf.data()+first, n); }
template <> void batchCall<byte, 4>(int, const std::vector<BitmapSection<byte, 4> > &o, const Dump &d, int first, int n) { msdfgen_hip::generateMTSDFBatch(o.data(), d.ptrs.data()+first, d.xf.data()+first, n); }

// shim_check batch <dump> <mode> <w> <h> [max glyphs [file for the packed float tiles]]: every glyph through its OWN generate*() call (the per-call drop-in) and the whole list through
// generate*Batch() -- packed float tiles, float rectangles of an atlas with gaps / both orientations / two bitmap sizes, and an 8-bit atlas; the batch results must equal
// the per-call results byte for byte (the 8-bit ones: pixelFloatToByte of them, core/pixel-conversion.hpp:8-10). Prints one JSON line.
template <int N>
static int batchCheck(const Dump &d, int mode, int w, int h, int limit, const char *packedOut = NULL) {
    const int n = limit > 0 && limit < (int) d.shapes.size() ? limit : (int) d.shapes.size();
    const size_t tile = (size_t) w*h*N;
    std::vector<float> want((size_t) n*tile), packed((size_t) n*tile, -7.f);
    for (int g = 0; g < n; ++g)
        perCall<N>(mode, BitmapSection<float, N>(want.data()+(size_t) g*tile, w, h, g%3 == 1 ? Y_DOWNWARD : Y_UPWARD), d.shapes[g], d.xf[g]);
    std::vector<BitmapSection<float, N> > outs;
    for (int g = 0; g < n; ++g)
        outs.push_back(BitmapSection<float, N>(packed.data()+(size_t) g*tile, w, h, g%3 == 1 ? Y_DOWNWARD : Y_UPWARD));
    batchCall<float, N>(mode, outs, d, 0, n);
    long long packedDiff = 0;
    for (size_t i = 0; i < want.size(); ++i)
        packedDiff += memcmp(&want[i], &packed[i], 4) != 0;
    if (packedOut) {                                                     // the batch entry's packed tiles as they are: the test hashes them against the REFERENCE's tiles
        FILE *f = fopen(packedOut, "wb");
        if (!f || fwrite(packed.data(), sizeof(float), packed.size(), f) != packed.size())
            return 5;
        fclose(f);
    }
    // an atlas with a one-texel gutter around every cell, cells of every fifth glyph four texels smaller in both directions (a second launch group),
    // every other cell addressed bottom-up through a negative row stride
    const int cols = 16, cw = w+2, ch = h+2, rows = (n+cols-1)/cols;
    const int aw = cols*cw, ah = rows*ch;
    std::vector<float> atlas((size_t) aw*ah*N, -3.f), atlasWant((size_t) aw*ah*N, -3.f);
    std::vector<BitmapSection<float, N> > cells, cellsWant;
    for (int g = 0; g < n; ++g) {
        const int x = (g%cols)*cw+1, y = (g/cols)*ch+1, gw = g%5 == 4 && w > 8 && h > 8 ? w-4 : w, gh = g%5 == 4 && w > 8 && h > 8 ? h-4 : h;
        for (int pass = 0; pass < 2; ++pass) {
            float *base = (pass ? atlasWant : atlas).data();
            BitmapSection<float, N> cell(base+((size_t) y*aw+x)*N, gw, gh, aw*N, g%2 ? Y_DOWNWARD : Y_UPWARD);
            if (g%4 == 2)
                cell = BitmapSection<float, N>(base+((size_t) (y+gh-1)*aw+x)*N, gw, gh, -aw*N, Y_UPWARD);
            (pass ? cellsWant : cells).push_back(cell);
        }
        SDFTransformation t = d.xf[g];
        perCall<N>(mode, cellsWant.back(), d.shapes[g], t);
    }
    batchCall<float, N>(mode, cells, d, 0, n);
    long long atlasDiff = 0;
    for (size_t i = 0; i < atlas.size(); ++i)
        atlasDiff += memcmp(&atlas[i], &atlasWant[i], 4) != 0;
    // 8-bit atlas, cells of one size
    std::vector<byte> atlas8((size_t) aw*ah*N, 77);
    std::vector<BitmapSection<byte, N> > cells8;
    for (int g = 0; g < n; ++g) {
        const int x = (g%cols)*cw+1, y = (g/cols)*ch+1;
        cells8.push_back(BitmapSection<byte, N>(atlas8.data()+((size_t) y*aw+x)*N, w, h, aw*N, g%3 == 1 ? Y_DOWNWARD : Y_UPWARD));
    }
    batchCall<byte, N>(mode, cells8, d, 0, n);
    long long byteDiff = 0, gutterDiff = 0;
    std::vector<unsigned char> covered((size_t) aw*ah, 0);
    for (int g = 0; g < n; ++g)
        for (int yy = 0; yy < h; ++yy)
            for (int xx = 0; xx < w; ++xx) {
                const size_t at = (size_t) ((g/cols)*ch+1+yy)*aw+(g%cols)*cw+1+xx;
                covered[at] = 1;
                for (int c = 0; c < N; ++c)
                    byteDiff += atlas8[at*N+c] != pixelFloatToByte(want[(size_t) g*tile+((size_t) yy*w+xx)*N+c]);
            }
    for (size_t at = 0; at < covered.size(); ++at)
        if (!covered[at])
            for (int c = 0; c < N; ++c)
                gutterDiff += atlas8[at*N+c] != 77;
    printf("{\"glyphs\": %d, \"mode\": %d, \"size\": [%d, %d], \"packed_values_differing\": %lld, \"atlas_values_differing\": %lld, \"byte_values_differing\": %lld, "
           "\"gutter_bytes_touched\": %lld}\n", n, mode, w, h, packedDiff, atlasDiff, byteDiff, gutterDiff);
    return packedDiff || atlasDiff || byteDiff || gutterDiff ? 6 : 0;
}

// shim_check e2e <dump> <w> <h> <reps>: SURVEY.md 8(d)'s end-to-end metric from REAL msdfgen::Shape objects: generateMSDFBatch over the whole list into pinned
// packed float tiles, and into a pinned 8-bit atlas of 128 cells per row. Median wall milliseconds per call over `reps` calls after one warm-up.
static int e2eBench(const Dump &d, int w, int h, int reps) {
    const int n = (int) d.shapes.size(), cols = 128, rows = (n+cols-1)/cols;
    const size_t tile = (size_t) w*h*3;
    float *tiles = NULL;
    byte *atlas = NULL;
    if (msdfhip_host_alloc((void **) &tiles, sizeof(float)*tile*n) || msdfhip_host_alloc((void **) &atlas, (size_t) rows*h*cols*w*3))
        return 7;
    std::vector<BitmapSection<float, 3> > outs;
    std::vector<BitmapSection<byte, 3> > cells;
    for (int g = 0; g < n; ++g) {
        outs.push_back(BitmapSection<float, 3>(tiles+(size_t) g*tile, w, h));
        cells.push_back(BitmapSection<byte, 3>(atlas+((size_t) (g/cols)*h*cols*w+(size_t) (g%cols)*w)*3, w, h, cols*w*3));
    }
    std::vector<double> tf, tb;
    for (int r = 0; r <= reps; ++r) {
        const double t0 = nowMs();
        msdfgen_hip::generateMSDFBatch(outs.data(), d.ptrs.data(), d.xf.data(), n);
        const double t1 = nowMs();
        msdfgen_hip::generateMSDFBatch(cells.data(), d.ptrs.data(), d.xf.data(), n);
        const double t2 = nowMs();
        if (r)
            tf.push_back(t1-t0), tb.push_back(t2-t1);
    }
    std::sort(tf.begin(), tf.end()), std::sort(tb.begin(), tb.end());
    unsigned long long sum = 0;                                          // (something read from both outputs)
    for (size_t i = 0; i < tile*n; i += 97) {
        uint32_t bits;
        memcpy(&bits, tiles+i, 4);
        sum += bits+atlas[i/4];
    }
    printf("{\"glyphs\": %d, \"edges\": %lld, \"size\": [%d, %d], \"reps\": %d, \"float_tiles_ms\": %.3f, \"float_tiles_ms_min\": %.3f, \"uint8_atlas_ms\": %.3f, \"uint8_atlas_ms_min\": %.3f, "
           "\"float_tiles_glyphs_per_s\": %.0f, \"uint8_atlas_glyphs_per_s\": %.0f, \"checksum\": %llu}\n", n, d.edges, w, h, reps, tf[tf.size()/2], tf[0], tb[tb.size()/2], tb[0],
           n/(tf[tf.size()/2]*1e-3), n/(tb[tb.size()/2]*1e-3), sum);
    msdfhip_host_free(tiles);
    msdfhip_host_free(atlas);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 5 && !strcmp(argv[1], "flatten"))
        return flattenBench(argv[2], atoi(argv[3]), atoi(argv[4]));
    if (argc >= 6 && (!strcmp(argv[1], "batch") || !strcmp(argv[1], "e2e"))) {
        Dump d;
        if (!loadDump(argv[2], d))
            return 3;
        try {
            if (!strcmp(argv[1], "e2e"))
                return e2eBench(d, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]));
            const int mode = atoi(argv[3]), w = atoi(argv[4]), h = atoi(argv[5]), limit = argc > 6 ? atoi(argv[6]) : 0;
            const char *packedOut = argc > 7 ? argv[7] : NULL;
            return mode <= 2 ? batchCheck<1>(d, mode, w, h, limit, packedOut) : mode == 3 ? batchCheck<3>(d, mode, w, h, limit, packedOut) : batchCheck<4>(d, mode, w, h, limit, packedOut);
        } catch (const std::exception &e) {
            fprintf(stderr, "%s\n", e.what());
            return 4;
        }
    }
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
