// msdf_kernels.hpp -- gfx950 kernels of the MSDF hot path.
//
//   k_prep_records / k_windings   once per batch upload: raw CSR edge buffer -> EdgeRec records (+ contour windings) in HBM
//   k_distance<SEL,OVERLAP,LDSREC>  one wavefront = one 8x8 texel tile of one glyph; lane = texel. The glyph's edge records are
//                                 staged through LDS once per workgroup and then read with wave-uniform (broadcast) ds_reads while
//                                 every lane runs the full per-contour nearest-edge selection in fp64 registers. The overlapping
//                                 contour combiner keeps its per-contour distances in LDS, laid out [contour][channel][lane] so that
//                                 lane-consecutive 8-byte accesses are bank-conflict free.
//   k_error_correction<N,OVERLAP,LDSREC>  the whole stencil pipeline as a per-texel gather over the pre-correction field
//   k_shape_distance<SEL,OVERLAP> distance queries at arbitrary points (known-answer tests)
//
// Workgroup -> work mapping is XCD-aware: hardware places block b on XCD b%8 (MI355X_MICROARCH.md), so all tiles of glyph g are
// issued with the same b%8 and share that XCD's L2 for the glyph's records.
#pragma once

#include <hip/hip_runtime.h>
#include "msdf_device.hpp"
#include "msdf_prep.hpp"
#include "msdf_ec.hpp"
#include "../../include/msdfgen_hip.h"

namespace msdfhip {

constexpr int TILE = 8;          // 8x8 texels per wavefront
constexpr int WAVE = 64;
constexpr int REC_DOUBLES = sizeof(EdgeRec)/sizeof(double);

struct BatchView {
    int nGlyphs;
    const int32_t *glyphContourOffsets; // [G+1]
    const int32_t *contourOffsets;      // [C+1] global edge indices
    const EdgeRec *recs;                // [E] contour by contour, visit order
    const int8_t *windings;             // [C]
};

// ------------------------------------------------------------------------------------------------------------- prep

__global__ void k_prep_records(EdgeRec *recs, int nEdges, int nContours, const int32_t *contourOffsets,
                               const double *points, const uint8_t *types, const uint8_t *colors) {
    int slot = blockIdx.x*blockDim.x+threadIdx.x;
    if (slot >= nEdges)
        return;
    int lo = 0, hi = nContours-1;               // last contour c with contourOffsets[c] <= slot (skips empty contours)
    while (lo < hi) {
        int mid = (lo+hi+1)>>1;
        if (contourOffsets[mid] <= slot)
            lo = mid;
        else
            hi = mid-1;
    }
    prepRecord(recs, slot, lo, contourOffsets, points, types, colors);
}

__global__ void k_windings(int8_t *windings, int nContours, const int32_t *contourOffsets,
                           const double *points, const uint8_t *types, const uint8_t *colors) {
    int c = blockIdx.x*blockDim.x+threadIdx.x;
    if (c < nContours)
        windings[c] = (int8_t) contourWinding(c, contourOffsets, points, types, colors);
}

// ---------------------------------------------------------------------------------------------------------- helpers

struct GlyphWork {
    int g, tile;
    bool valid;
};

// XCD-aware decode of blockIdx -> (glyph, tile): glyphs are dealt round-robin to the 8 XCDs, every tile of a glyph to the same XCD.
__device__ inline GlyphWork decodeBlock(int nGlyphs, int tilesPerGlyph) {
    GlyphWork w;
    const unsigned b = blockIdx.x;
    const unsigned xcd = b&7u, slot = b>>3;
    w.g = (int) ((slot/(unsigned) tilesPerGlyph)*8u+xcd);
    w.tile = (int) (slot%(unsigned) tilesPerGlyph);
    w.valid = w.g < nGlyphs;
    return w;
}

__device__ inline Xform loadXform(const MsdfHipGlyph &gd) {
    Xform t;
    t.sx = gd.xf[0], t.sy = gd.xf[1], t.tx = gd.xf[2], t.ty = gd.xf[3];
    t.mapScale = gd.xf[4], t.mapTranslate = gd.xf[5];
    return t;
}

// Cooperative copy of n records global -> LDS (8-byte lanes, coalesced).
__device__ inline void stageRecords(double *dst, const EdgeRec *src, int n) {
    const double *s = reinterpret_cast<const double *>(src);
    const int total = n*REC_DOUBLES;
    for (int i = threadIdx.x; i < total; i += blockDim.x)
        dst[i] = s[i];
}

// ----------------------------------------------------------------------------------------------------- distance field

// dst: tile-major destination. If toScratch, texels go to the tightly packed pre-correction buffer [g][h][w][N] (native rows),
// else straight to the caller's bitmap at out_offset/row_stride (generateDistanceField, core/msdfgen.cpp:52-76).
template <int SEL, bool OVERLAP, bool LDSREC>
__global__ void __launch_bounds__(WAVE)
k_distance(BatchView batch, const MsdfHipGlyph *glyphs, int width, int height, int tilesX, int tilesPerGlyph, float *dst, int toScratch) {
    enum { NCH = SelTraits<SEL>::NCH };
    extern __shared__ double smem[];
    const GlyphWork wk = decodeBlock(batch.nGlyphs, tilesPerGlyph);
    if (!wk.valid)
        return;
    const int c0 = batch.glyphContourOffsets[wk.g], C = batch.glyphContourOffsets[wk.g+1]-c0;
    const int32_t *coff = batch.contourOffsets+c0;
    const int e0 = coff[0], nE = coff[C]-e0;
    const int lane = threadIdx.x;

    double *res = smem;                                             // [C][NCH][64] (overlap only)
    const EdgeRec *rec = batch.recs+e0;
    if (LDSREC) {
        double *recLds = smem+(OVERLAP ? (size_t) C*NCH*WAVE : 0);
        stageRecords(recLds, rec, nE);
        rec = reinterpret_cast<const EdgeRec *>(recLds);
        __syncthreads();
    }

    const int tx = wk.tile%tilesX, ty = wk.tile/tilesX;
    const int x = tx*TILE+(lane&(TILE-1)), y = ty*TILE+(lane>>3);
    if (x >= width || y >= height)
        return;
    const MsdfHipGlyph gd = glyphs[wk.g];
    const Xform t = loadXform(gd);
    const V2 p = unproject(t, mk(x+.5, y+.5));                      // msdfgen.cpp:68
    double d[NCH];
    if (OVERLAP)
        shapeDistanceOverlap<SEL>(rec, coff, batch.windings+c0, C, p, res+lane, WAVE, d);
    else
        shapeDistanceSimple<SEL>(rec, coff, C, p, d);
    const int yn = gd.flip ? height-1-y : y;                        // output.reorient(shape orientation), msdfgen.cpp:55
    float *px = toScratch ? dst+(((size_t) wk.g*height+yn)*width+x)*NCH
                          : dst+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) NCH*x;
    for (int ch = 0; ch < NCH; ++ch)
        px[ch] = mapDistance(t, d[ch]);                             // msdfgen.cpp:20-48
}

// --------------------------------------------------------------------------------------------------- error correction

template <bool OVERLAP>
struct PsdfQuery {                                                  // ShapeDistanceFinder<CC<PerpendicularDistanceSelector>>, MSDFErrorCorrection.cpp:97
    const EdgeRec *rec;
    const int32_t *coff;
    const int8_t *windings;
    int C;
    double *res;
    __device__ MSDF_NOINLINE double operator()(V2 q) const {
        double out[1];
        if (OVERLAP)
            shapeDistanceOverlap<2>(rec, coff, windings, C, q, res, WAVE, out);
        else
            shapeDistanceSimple<2>(rec, coff, C, q, out);
        return out[0];
    }
};

// src: pre-correction field, packed [g][h][w][N] in native row order. Writes corrected texels to the caller's bitmap
// (msdfErrorCorrectionInner, core/msdf-error-correction.cpp:12-48) and, if stencilOut, the final stencil byte [g][h][w] (native rows).
template <int N, bool OVERLAP, bool LDSREC>
__global__ void __launch_bounds__(WAVE)
k_error_correction(BatchView batch, const MsdfHipGlyph *glyphs, int width, int height, int tilesX, int tilesPerGlyph,
                   const float *src, float *out, uint8_t *stencilOut, MsdfHipConfig cfg) {
    extern __shared__ double smem[];
    const GlyphWork wk = decodeBlock(batch.nGlyphs, tilesPerGlyph);
    if (!wk.valid)
        return;
    const int c0 = batch.glyphContourOffsets[wk.g], C = batch.glyphContourOffsets[wk.g+1]-c0;
    const int32_t *coff = batch.contourOffsets+c0;
    const int e0 = coff[0], nE = coff[C]-e0;
    const int lane = threadIdx.x;

    double *res = smem;                                             // [C][1][64] (overlap only)
    const EdgeRec *rec = batch.recs+e0;
    if (LDSREC) {
        double *recLds = smem+(OVERLAP ? (size_t) C*WAVE : 0);
        stageRecords(recLds, rec, nE);
        rec = reinterpret_cast<const EdgeRec *>(recLds);
        __syncthreads();
    }

    const int tx = wk.tile%tilesX, ty = wk.tile/tilesX;
    const int x = tx*TILE+(lane&(TILE-1)), yn = ty*TILE+(lane>>3);
    if (x >= width || yn >= height)
        return;
    const MsdfHipGlyph gd = glyphs[wk.g];
    EcParams p;
    p.t = loadXform(gd);
    p.minDeviationRatio = cfg.min_deviation_ratio;
    p.minImproveRatio = cfg.min_improve_ratio;
    p.mode = cfg.ec_mode, p.distanceCheck = cfg.ec_distance_check, p.overlap = OVERLAP, p.stageLimit = cfg.ec_stage_limit;
    ecDerive(p);
    SdfView sdf;
    sdf.px = src+(size_t) wk.g*height*width*N;
    sdf.w = width, sdf.h = height, sdf.N = N, sdf.flip = gd.flip;
    PsdfQuery<OVERLAP> query;
    query.rec = rec, query.coff = coff, query.windings = batch.windings+c0, query.C = C, query.res = res+lane;

    const int st = ecTexelStencil(sdf, p, rec, nE, x, yn, &query);
    const float *in = sdf.native(x, yn);
    float v[N];
    for (int i = 0; i < N; ++i)
        v[i] = in[i];
    if ((st&EC_ERROR) && cfg.ec_stage_limit == 0) {                 // apply, MSDFErrorCorrection.cpp:459-479 (alpha untouched)
        const float m = medianf(v[0], v[1], v[2]);
        v[0] = m, v[1] = m, v[2] = m;
    }
    float *px = out+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) N*x;
    for (int i = 0; i < N; ++i)
        px[i] = v[i];
    if (stencilOut)
        stencilOut[((size_t) wk.g*height+yn)*width+x] = (uint8_t) st;
}

// ------------------------------------------------------------------------------------------------- distance queries

template <int SEL, bool OVERLAP>
__global__ void __launch_bounds__(WAVE)
k_shape_distance(BatchView batch, int nPoints, const double *pts, double *out) {
    enum { NCH = SelTraits<SEL>::NCH };
    extern __shared__ double smem[];
    const int i = blockIdx.x*WAVE+threadIdx.x;
    if (i >= nPoints)
        return;
    const int C = batch.glyphContourOffsets[1]-batch.glyphContourOffsets[0];
    double d[4] = { 0, 0, 0, 0 };
    const V2 p = mk(pts[2*i], pts[2*i+1]);
    if (OVERLAP)
        shapeDistanceOverlap<SEL>(batch.recs, batch.contourOffsets, batch.windings, C, p, smem+threadIdx.x, WAVE, d);
    else
        shapeDistanceSimple<SEL>(batch.recs, batch.contourOffsets, C, p, d);
    for (int ch = 0; ch < 4; ++ch)
        out[4*(size_t) i+ch] = ch < NCH ? d[ch] : 0.;
}

} // namespace msdfhip
