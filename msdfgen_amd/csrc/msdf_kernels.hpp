// msdf_kernels.hpp -- gfx950 kernels of the MSDF hot path.
//
//   k_prep_records (+ windings)   once per batch upload: raw CSR edge buffer -> EdgeRec records (+ contour windings) in HBM
//   k_distance<SEL,OVERLAP,GRES>  one wavefront = four 8x8 texel tiles of one glyph. Phase 1 culls the glyph's edges for the four
//                                 tiles at once (16 lanes per tile, lanes = edges), phase 2 runs the per-contour nearest-edge
//                                 selection tile after tile (lanes = texels) in fp64 registers; the surviving edge records are read
//                                 with wave-uniform scalar loads. The overlapping contour combiner keeps its per-contour distances in
//                                 LDS, laid out [contour][channel][lane] (lane-consecutive 8-byte accesses: bank-conflict free).
//   k_ec_fast<N>                  error correction, lean single sweep over all texels (gather form, no atomics on the stencil);
//                                 texels whose verdict needs an exact shape-distance query (~0.1 %) are appended to a list
//   k_ec_slow<N,OVERLAP>          full per-texel pipeline incl. the PSDF distance query for the listed texels
//   k_shape_distance<SEL,OVERLAP> distance queries at arbitrary points (known-answer tests)
//
// Workgroup -> work mapping is XCD-aware: hardware places block b on XCD b%8 (MI355X_MICROARCH.md), so all tiles of glyph g are
// issued with the same b%8 and share that XCD's L2 for the glyph's records.
#pragma once

#include <hip/hip_runtime.h>
#include "msdf_device.hpp"
#include "msdf_prep.hpp"
#include "msdf_ec.hpp"
#include "msdf_ec_fast.hpp"
#include "msdf_cull.hpp"
#include "msdf_scanline.hpp"
#include "msdf_shapeprep.hpp"
#include "../../include/msdfgen_hip.h"

namespace msdfhip {

#ifndef MSDF_DISTANCE_WAVES_PER_SIMD
#define MSDF_DISTANCE_WAVES_PER_SIMD 4   // overlapping-combiner instantiations of k_distance: 128 VGPRs, FOUR wavefronts per SIMD. Round 2 measured "1 wave 8.7 ms per 8 192
                                         // glyphs, 2 waves 4.4, 3 waves 3.7, 4 waves 3.9" and kept three (154-168 VGPRs, no scratch) -- but at the 13 KB of LDS per wavefront of
                                         // that build a CU holds 12 wavefronts whatever the registers allow, so "4" had measured the spills without the occupancy. Round 5
                                         // (profiles/r05_ab_notes.md): the 94-164 dwords the 128-VGPR build spills all land OUTSIDE the edge loop (tools/isa_loop_depth.py: 0
                                         // scratch operations and 0 lane moves at its depth; 44-58 per tile, 49-74 per phase-1 round), and with the LDS budget of the class
                                         // at 10 KB (msdf_capi.hip: ldsBudget) a CU really holds 16: bench step 5.62 -> 5.39 ms, CJK-like set 13.65 -> 12.45, logo 7.54 -> 7.31.
                                         // 5 / 6 wavefronts (96 / 80 VGPRs) put scratch traffic into the contour and edge loops: 7.5 / 8.9 ms.
#endif
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

// The wave context of contourWindingsWave (msdf_prep.hpp) and colourContourWave (msdf_shapeprep.hpp) on the device: one lane each; tests/hostemu's
// loops over 64.
struct WaveCtx {
    int lane;
    template <class F> __device__ void lanes(F f) const { f(lane); }
    template <class P> __device__ unsigned long long ballot(P pred) const { return __ballot(pred(lane)); }
    template <class F> __device__ void leader(F f) const { if (lane == 0) f(); }
    __device__ void sync() const {                                  // tables in LDS or (contours beyond PREP_WAVE_MAX_EDGES) in global memory
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};

enum { PREP_WINDING_LDS_EDGES = 768 };                              // edges whose point(0) a winding wavefront parks in LDS at a time (12 KB; contours beyond: contourWindingsWave)
#ifndef MSDF_PREP_ABLATE
#define MSDF_PREP_ABLATE 0                                          // measurement builds only: 1 no windings, 3 no records (2, no contour search, hangs: a wrong contour makes visitToEdge loop)
#endif
// The first edgeBlocks workgroups digest one edge per thread; the workgroups after them compute the contour windings (k_windings' job, folded
// into the same launch: the single-shape entry points are launch-latency bound): a lane per contour, except that contours of at least
// PREP_WINDING_WAVE_MIN_EDGES edges -- whose serial walk by one lane would be the tail of the whole launch -- are taken by their wavefront together,
// lanes = edges (contourWindingsWave). (All of a wavefront's 64 contours that way was measured SLOWER than a lane each: 0.154 vs 0.097 ms on the bench
// workload -- ten rounds of loads and ~600 ordered additions per wavefront where most contours have ten edges.) blockDim.x = 256.
__global__ void k_prep_records(EdgeRec *recs, int nEdges, int nContours, const int32_t *contourOffsets,
                               const double *points, const uint8_t *types, const uint8_t *colors, int8_t *windings, int edgeBlocks) {
    if ((int) blockIdx.x >= edgeBlocks) {
#if MSDF_PREP_ABLATE == 1                                            // measurement only: no windings
        return;
#endif
        __shared__ double terms[4][64];
        const int wave = threadIdx.x>>6, lane = threadIdx.x&63;
        const int cBegin = ((int) blockIdx.x-edgeBlocks)*256+64*wave;
        if (cBegin >= nContours)
            return;
        const int c = cBegin+lane;
        const int cb = c < nContours ? contourOffsets[c] : 0, ce = c < nContours ? contourOffsets[c+1] : 0;
#if !defined(MSDF_PREP_LANE_WINDINGS)
        const bool isLong = c < nContours && ce-cb > PREP_WINDING_LDS_EDGES;    // (beyond the LDS area: contourWindingsWave, a wavefront per contour, as before)
#else
        const bool isLong = c < nContours && ce-cb >= PREP_WINDING_WAVE_MIN_EDGES;
#endif
#if !defined(MSDF_PREP_LANE_WINDINGS)
        // Round 6: a lane per contour made one DEPENDENT, first-touch memory round trip per edge -- up to 47 of them, 0.073 of the digest's 0.114 ms with
        // the edge workgroups long gone (variants/pab1: the launch without windings). Now in two steps per group of consecutive contours whose edges fit the
        // wavefront's LDS area: lanes = EDGES load the rows (coalesced, independent) and park point(0) of every edge (edge-segments.cpp:108-119, the
        // reference's own expression); lanes = CONTOURS then add their shoelace terms in the reference's order (Contour.cpp:72-79) from LDS.
        // Contours of fewer than three edges (other sample points, :59-71) keep contourWinding; a contour beyond the LDS area takes contourWindingsWave below.
        // (The contours of 48 edges and more used to take that path, one after the other per wavefront, 64 ordered LDS additions per round: they were the launch's tail.)
        __shared__ double startPoints[4][2*PREP_WINDING_LDS_EDGES];
        double *pts = startPoints[wave];
        WaveCtx sync;                                                // (its sync(): LDS hand-off between the lanes of one wavefront)
        sync.lane = lane;
        const bool mid = c < nContours && !isLong && ce-cb >= 3;
        if (c < nContours && ce-cb < 3)
            windings[c] = (int8_t) contourWinding(c, contourOffsets, points, types, colors);
        for (int done = 0; done < WAVE;) {
            if (!__shfl((int) mid, done)) {                          // (done is wave-uniform)
                ++done;
                continue;
            }
            const int base = __shfl(cb, done);
            const bool fits = lane >= done && c < nContours && ce-base <= PREP_WINDING_LDS_EDGES;   // a run of lanes from `done` on (offsets do not decrease); the first fits
            const int count = __popcll(__ballot(fits));
            const int top = __shfl(ce, done+count-1);
            for (int i = base+lane; i < top; i += WAVE) {
                const V2 p0 = rawPoint(loadRaw(points, types, colors, i), 0);
                pts[2*(i-base)] = p0.x, pts[2*(i-base)+1] = p0.y;
            }
            sync.sync();
            if (mid && lane >= done && lane < done+count) {
                double total = 0;
                V2 prev = mk(pts[2*(ce-1-base)], pts[2*(ce-1-base)+1]);
                int i = cb;
                for (; i+4 <= ce; i += 4) {                          // (four points requested together: the chain is the additions, not the LDS reads)
                    const double *q = pts+2*(i-base);
                    const V2 c0 = mk(q[0], q[1]), c1 = mk(q[2], q[3]), c2 = mk(q[4], q[5]), c3 = mk(q[6], q[7]);
                    const double t0 = shoelace(prev, c0), t1 = shoelace(c0, c1), t2 = shoelace(c1, c2), t3 = shoelace(c2, c3);
                    total += t0, total += t1, total += t2, total += t3;
                    prev = c3;
                }
                for (; i < ce; ++i) {
                    const V2 cur = mk(pts[2*(i-base)], pts[2*(i-base)+1]);
                    total += shoelace(prev, cur);
                    prev = cur;
                }
                windings[c] = (int8_t) ((0 < total)-(total < 0));
            }
            sync.sync();
            done += count;
        }
#else
        if (c < nContours && !isLong)
            windings[c] = (int8_t) contourWinding(c, contourOffsets, points, types, colors);
#endif
        unsigned long long longMask = __ballot(isLong);
        WaveCtx ctx;
        ctx.lane = lane;
        while (longMask) {
            const int k = __builtin_ctzll(longMask);
            longMask &= longMask-1;
            contourWindingsWave(ctx, terms[wave], cBegin+k, cBegin+k+1, contourOffsets, points, types, colors, windings);
        }
        return;
    }
    int slot = blockIdx.x*blockDim.x+threadIdx.x;
    if (slot >= nEdges)
        return;
    int lo = 0, hi = nContours-1;               // last contour c with contourOffsets[c] <= slot (skips empty contours)
#if MSDF_PREP_ABLATE == 3                                            // measurement only: no records
    return;
#endif
    while (lo < hi) {
        int mid = (lo+hi+1)>>1;
        if (contourOffsets[mid] <= slot)
            lo = mid;
        else
            hi = mid-1;
    }
    prepRecord(recs, slot, lo, contourOffsets, points, types, colors);
}

// Small host-to-device uploads from PINNED host memory as a kernel (the device reads the pinned source over PCIe): descriptors and class
// lists of a pipeline chunk. As hipMemcpyAsync they go to an SDMA engine and queue up BEHIND the 100 MB device-to-host copy of the previous
// chunk -- the next chunk's kernels then start only when that copy is done (MSDFHIP_PIPELINE_TRACE showed exactly that).
// (Round 5: 16 bytes per lane. A lane's load from host memory is a PCIe read of its own; as 4-byte loads the 128 KB descriptor block of a 2 048-glyph
// chunk took 0.2-0.37 ms -- twice per chunk of the 8-bit pipeline, at the head of the chunk's launch chain (rocprofv3 timeline, profiles/r05_ab_notes.md).)
__global__ void k_upload_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ srcPinned, size_t nWords) {
    const size_t nQuads = ((reinterpret_cast<size_t>(dst)|reinterpret_cast<size_t>(srcPinned))&15) == 0 ? nWords/4 : 0;
    const uint4 *src4 = reinterpret_cast<const uint4 *>(srcPinned);
    uint4 *dst4 = reinterpret_cast<uint4 *>(dst);
    for (size_t i = (size_t) blockIdx.x*blockDim.x+threadIdx.x; i < nQuads; i += (size_t) gridDim.x*blockDim.x)
        dst4[i] = src4[i];
    for (size_t i = nQuads*4+(size_t) blockIdx.x*blockDim.x+threadIdx.x; i < nWords; i += (size_t) gridDim.x*blockDim.x)
        dst[i] = srcPinned[i];
}

// ---------------------------------------------------------------------------------------------------------- helpers

struct GlyphWork {
    int g, tile;
    bool valid;
};

// XCD-aware decode of blockIdx -> (glyph, tile): glyphs are dealt round-robin to the 8 XCDs, every tile of a glyph to the same XCD
// (its records stay in one L2). The last, partial group of nGlyphs%8 glyphs is dealt tile by tile instead -- a single 1024x1024 shape
// would otherwise run on one XCD, an eighth of the device. The launch has exactly nGlyphs*tilesPerGlyph workgroups.
__device__ inline GlyphWork decodeItem(unsigned b, int nGlyphs, int tilesPerGlyph) {
    GlyphWork w;
    const unsigned fullBlocks = ((unsigned) nGlyphs&~7u)*(unsigned) tilesPerGlyph;
    if (b >= fullBlocks) {
        const unsigned j = b-fullBlocks;
        w.g = (int) (((unsigned) nGlyphs&~7u)+j/(unsigned) tilesPerGlyph);
        w.tile = (int) (j%(unsigned) tilesPerGlyph);
    } else {
        const unsigned xcd = b&7u, slot = b>>3;
        w.g = (int) ((slot/(unsigned) tilesPerGlyph)*8u+xcd);
        w.tile = (int) (slot%(unsigned) tilesPerGlyph);
    }
    w.valid = w.g < nGlyphs;
    return w;
}
__device__ inline GlyphWork decodeBlock(int nGlyphs, int tilesPerGlyph, unsigned blockBase = 0) {
    return decodeItem(blockIdx.x+blockBase, nGlyphs, tilesPerGlyph);
}

// Work queue of a PERSISTENT launch (fewer workgroups than items: each keeps its slice of the global workspace and draws items until
// none are left). counters[8]: one per XCD, so that a workgroup on XCD x (blockIdx%8, MI355X_MICROARCH.md) first drains the items
// decodeItem() deals to x -- the glyphs whose records its L2 already holds -- and only then helps the others (steal: 1..7 XCDs further).
__device__ inline unsigned nextItem(unsigned *counters, unsigned total, int &steal) {
    const unsigned x = blockIdx.x&7u;
    while (steal < 8) {
        const unsigned xx = (x+(unsigned) steal)&7u;
        unsigned k = 0;
        if (threadIdx.x == 0)
            k = atomicAdd(&counters[xx], 1u);
        k = (unsigned) __builtin_amdgcn_readfirstlane((int) k);
        const unsigned b = k*8u+xx;
        if (b < total)
            return b;
        ++steal;
    }
    return ~0u;
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

// LDS hand-off between lanes of ONE wavefront: a wavefront's LDS operations complete in issue order, so only the compiler has to be
// kept from reordering them across this point; no workgroup barrier is involved (the wavefronts of a workgroup are independent).
__device__ inline void waveSync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-wide minimum of non-negative floats via DPP row shifts / broadcasts (VALU speed; a shuffle-based reduction costs an LDS
// round trip per step). Non-negative IEEE floats order like their bit patterns, so the reduction runs on ints.
// All 64 lanes must be active. Result is broadcast to every lane.
__device__ inline float waveMinNonNegative(float v) {
    int x = __float_as_int(v);
    const int inf = 0x7f800000;
#define MSDF_DPP_MIN(ctrl, rowmask) { const int y = __builtin_amdgcn_update_dpp(inf, x, ctrl, rowmask, 0xf, false); x = y < x ? y : x; }
    MSDF_DPP_MIN(0x111, 0xf)   // row_shr:1
    MSDF_DPP_MIN(0x112, 0xf)   // row_shr:2
    MSDF_DPP_MIN(0x114, 0xf)   // row_shr:4
    MSDF_DPP_MIN(0x118, 0xf)   // row_shr:8   -> lane 15 of every row holds the row minimum
    MSDF_DPP_MIN(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
    MSDF_DPP_MIN(0x143, 0xc)   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave minimum
#undef MSDF_DPP_MIN
    return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}

// Smallest float >= d (d >= 0): the culling bounds only need to be conservative, so they travel as fp32 rounded up.
__device__ inline float floatAbove(double d) {
    float f = (float) d;
    if ((double) f < d)
        f = __int_as_float(__float_as_int(f)+1);
    return f;
}

// dst: tile-major destination. If toScratch, texels go to the tightly packed pre-correction buffer [g][h][w][N] (native rows),
// else straight to the caller's bitmap at out_offset/row_stride (generateDistanceField, core/msdfgen.cpp:52-76).
//
// Phase 1 (lanes = edges, one 16-lane row per tile of the quad): per contour, bound the tile's distance to each channel, cull edges
// that cannot matter for any texel of the tile (msdf_cull.hpp), and compact the survivors -- in visit order -- into LDS.  Phase 2
// (lanes = texels, one tile at a time): every lane runs the reference's per-contour nearest-edge selection over the survivors with
// wave-uniform (scalar) record reads.
// LDS: [res: C*NCH*64 doubles (overlap)] [lists: 4 x maxEdges ints] [cstarts: 4 x (C+1) ints].
// GRES: the combiner scratch of glyphs with very many contours does not fit the CU's LDS; it then lives in a global workspace
// (gres, one slice per workgroup of the launch chunk) and the launch is chunked (blockBase) to bound that workspace.
// Row-wise (16 lanes = one DPP row) minimum of non-negative floats, broadcast to the row.
__device__ inline float rowMinNonNegative(float v, int lane) {
    int x = __float_as_int(v);
    const int inf = 0x7f800000;
#define MSDF_DPP_MIN(ctrl) { const int y = __builtin_amdgcn_update_dpp(inf, x, ctrl, 0xf, 0xf, false); x = y < x ? y : x; }
    MSDF_DPP_MIN(0x111)   // row_shr:1
    MSDF_DPP_MIN(0x112)   // row_shr:2
    MSDF_DPP_MIN(0x114)   // row_shr:4
    MSDF_DPP_MIN(0x118)   // row_shr:8   -> lane 15 of every row holds the row minimum
#undef MSDF_DPP_MIN
    return __int_as_float(__shfl(x, lane|15));
}

// Rank of this lane's key among the 16 keys of its DPP row (keys unique within a row): number of smaller keys. VALU only.
__device__ inline int rowRank(unsigned key) {
    int rank = 0;
#define MSDF_DPP_ROR(n) rank += (unsigned) __builtin_amdgcn_update_dpp((int) key, (int) key, 0x120+n, 0xf, 0xf, false) < key ? 1 : 0;
    MSDF_DPP_ROR(1) MSDF_DPP_ROR(2) MSDF_DPP_ROR(3) MSDF_DPP_ROR(4) MSDF_DPP_ROR(5) MSDF_DPP_ROR(6) MSDF_DPP_ROR(7) MSDF_DPP_ROR(8)
    MSDF_DPP_ROR(9) MSDF_DPP_ROR(10) MSDF_DPP_ROR(11) MSDF_DPP_ROR(12) MSDF_DPP_ROR(13) MSDF_DPP_ROR(14) MSDF_DPP_ROR(15)
#undef MSDF_DPP_ROR
    return rank;
}

// ---- survivor records in SCALAR REGISTERS ---------------------------------------------------------------------------------------
// Phase 2 walks a tile's survivor list with wave-uniform record reads. Left to the compiler, every field of a record is loaded where it
// is first used: one DEPENDENT scalar-cache round trip per field group -- flags, control box, p0 + bisector, tangent, type, ... about ten
// per evaluated edge, a third of them missing the 16 KB scalar cache (profiles/r02_scalar_cache.txt) -- and a wavefront spent 44 % of its
// life in s_waitcnt. Here the walk is laid out by hand instead:
//   * the survivor list holds PACKED entries (record index | type | colour | flags): nothing has to be loaded to know what an edge is;
//   * EdgeRec is stored in 64-byte blocks in the order of use (msdf_device.hpp); the relevance test + a whole linear evaluation need
//     R + E0 = three s_load_dwordx16, issued together with the LDS read of the NEXT list entry and waited for ONCE;
//   * a curve fetches E1 + E2 after the wave vote said the edge matters (second round trip).
// SMEM returns out of order, so the only usable wait is lgkmcnt(0): every asm statement issues its loads AND waits for them -- no load is
// ever in flight across C++ code, whose register allocation could move or reuse a pending destination.
typedef double d8 __attribute__((ext_vector_type(8)));

enum { ENTRY_INDEX_BITS = 20, ENTRY_INDEX_MASK = (1<<ENTRY_INDEX_BITS)-1 };   // 1 M edges per glyph (the LDS lists hold far fewer)
__device__ inline unsigned packEntry(int i, const EdgeRec &e) {
    return (unsigned) (i&ENTRY_INDEX_MASK)|(unsigned) e.type<<20|(unsigned) (e.color&7)<<22|(unsigned) (e.flags&7)<<25|(unsigned) ((e.flags>>4)&1)<<28;   // (index bits: k_distance's lists only; k_ec_query passes the index separately)
}

struct EdgeRegs {
    d8 r0, r1, e0, e1, e2;              // blocks R (two halves), E0, E1, E2 of EdgeRec
    unsigned meta;                      // the packed list entry
    __device__ V2 Lo() const { return mk(r0[0], r0[1]); }
    __device__ V2 Hi() const { return mk(r0[2], r0[3]); }
    __device__ V2 P0() const { return mk(r0[4], r0[5]); }
    __device__ V2 PE() const { return mk(r0[6], r0[7]); }
    __device__ V2 NA() const { return mk(r1[0], r1[1]); }
    __device__ V2 NB() const { return mk(r1[2], r1[3]); }
    __device__ V2 ADirN() const { return mk(r1[4], r1[5]); }
    __device__ V2 BDirN() const { return mk(r1[6], r1[7]); }
    __device__ V2 AB() const { return mk(e0[0], e0[1]); }
    __device__ double K(int i) const { return e0[2+i]; }
    __device__ V2 BR() const { return mk(e1[0], e1[1]); }
    __device__ V2 EP0() const { return mk(e1[2], e1[3]); }
    __device__ V2 EP1() const { return mk(e1[4], e1[5]); }
    __device__ double E0dot() const { return e1[6]; }
    __device__ double E1dot() const { return e1[7]; }
    __device__ double Rcp(int i) const { return e2[i]; }
    __device__ V2 P1() const { return mk(e2[4], e2[5]); }
    __device__ V2 AS() const { return mk(e2[6], e2[7]); }
    __device__ int Type() const { return (int) (meta>>20)&3; }
    __device__ int Color() const { return (int) (meta>>22)&7; }
    __device__ int Flags() const { return (int) ((meta>>25)&7)|(int) ((meta>>28)&1)<<4; }
};
static_assert(REC_A_ZERO == 1 && REC_B_ZERO == 2 && REC_NORMED == 4 && REC_FASTDIV == 16, "packEntry / EdgeRegs::Flags");

// The same register image of a record for a lane that evaluates ITS OWN edge (k_ec_query: lanes = edges): the five blocks and the packed
// type / colour / flags arrive with 21 independent 16-byte loads -- one round trip. Evaluated straight from memory (selAddEdge on an
// EdgeRec &), every field is loaded where it is first used: ~80 DEPENDENT vector-load round trips per edge, which is what a distance
// check cost (profiles/r03_profile_query.jsonl: 300 k cycles per round of 64 edges, 60 times an evaluation of k_distance).
__device__ inline EdgeRegs loadEdgeRegs(const EdgeRec *rp, int i) {
    const d8 *blocks = reinterpret_cast<const d8 *>(rp);
    EdgeRegs r;
    r.r0 = blocks[0], r.r1 = blocks[1], r.e0 = blocks[2], r.e1 = blocks[3], r.e2 = blocks[4];
    r.meta = packEntry(i, *rp);
    return r;
}

#if defined(MSDF_PROFILE_WAITS)
// Measurement build only (tools/profile_waits.sh): where a k_distance wavefront's cycles go. s_memtime stamps inside the hand-placed load
// batches; per-wave sums are added to this table by lane 0 when the wavefront ends. [0] waves [1] wave cycles [2] phase 1 [3] cycles in
// the R+E0 batch (issue -> landed) [4] batches [5] cycles in the E1+E2 batch [6] batches [7] cycles evaluating edges (selAddEdge) [8]
// evaluations [9] cycles in relevance tests + vote [10] phase 2 total [11] batches that took > 1000 cycles [12] > 3000 cycles
__device__ unsigned long long gWaitProfile[24];
#define MSDF_STAMP(x) const unsigned long long x = __builtin_readcyclecounter()
#endif

// Contour::winding of a glyph's contours as two bit masks in scalar registers (bit c: winding > 0 / < 0), built once per wavefront from one
// vector load + two ballots; contours beyond 63 (global-scratch class only) are read from memory.
struct WindingMasks {
    unsigned long long pos, neg;
    const int8_t *mem;
    __device__ int operator[](int c) const {
        if (c < 64)
            return (int) ((pos>>c)&1ull)-(int) ((neg>>c)&1ull);
        return mem[c];
    }
};

struct EdgesCulledPacked {              // survivors of the per-tile cull as packed entries, grouped by contour, nearest-first within a row
    const int *cstart;                  // C+1 compacted offsets
    const int *list;                    // packed entries (LDS)
    int total;                          // cstart[C]
    int first, step;                    // this wavefront walks entries first, first+step, ... of every contour's segment (0, 1: all of them; a team of wavefronts: its rank, the team's size)
#if defined(MSDF_PROFILE_WAITS)
    mutable unsigned long long prof[16]; // [8] contour walks of pass 0 [9] per-contour bookkeeping after a walk [10] second walks [11] combiner epilogue [12] tile prologue [13] tile stores | [0] batch cycles [1] batches [2] E batch cycles [3] E batches [4] eval cycles [5] evals [6] relevance cycles [7] slow batches (>1000) | (>3000)<<32
#endif
    // (holding the offsets in lanes 0..C and fetching them with readlane -- no LDS round trip per contour -- was measured: no gain, and
    // one more VGPR in kernels that sit at their register cap)
    __device__ int begin(int c) const { return MSDF_UNIFORM(cstart[c]); }
    __device__ int end(int c) const { return MSDF_UNIFORM(cstart[c+1]); }
};

#if defined(MSDF_PROFILE_WAITS)
__device__ inline void profAdd(const EdgesCulledPacked &edges, int i, unsigned long long dt) { edges.prof[i] += dt; }
__device__ inline unsigned long long profNow(const EdgesCulledPacked &) { return __builtin_readcyclecounter(); }
#endif

template <int SEL>
__device__ inline void selAddContour(Selector<SEL> &sel, const EdgeRec *rec, const EdgesCulledPacked &edges, int c, V2 o) {
    const int e = edges.end(c);
    int k = edges.begin(c)+edges.first;
    if (k >= e)
        return;
    unsigned cur = (unsigned) MSDF_UNIFORM(edges.list[k]);
    MSDF_NOUNROLL
    for (; k < e; k += edges.step) {
        const EdgeRec *rp = rec+(cur&ENTRY_INDEX_MASK);
        EdgeRegs r;
        r.meta = cur;
        unsigned nextV;
        // LDS byte address of the next entry = the LOW 32 bits of its flat address: on gfx9-family devices (gfx950 included) a flat address
        // inside the shared aperture is { src_shared_base (high dword) | LDS offset (low dword) } -- the aperture is 4 GB-aligned, so the
        // truncation is the address-space cast the compiler itself emits for flat -> LDS. This library is built for gfx950 only (build.py).
        const unsigned nextAddr = (unsigned) (size_t) (edges.list+k+edges.step);
#if defined(MSDF_PROFILE_WAITS)
        MSDF_STAMP(t0);
#endif
        asm volatile("ds_read_b32 %0, %4\n\ts_load_dwordx16 %1, %5, 0x0\n\ts_load_dwordx16 %2, %5, 0x40\n\ts_load_dwordx16 %3, %5, 0x80\n\ts_waitcnt lgkmcnt(0)"
                     : "=v"(nextV), "=&s"(r.r0), "=&s"(r.r1), "=&s"(r.e0) : "v"(nextAddr), "s"(rp));
#if defined(MSDF_PROFILE_WAITS)
        MSDF_STAMP(t1);
        edges.prof[0] += t1-t0, edges.prof[1] += 1;
        if (t1-t0 > 1000) edges.prof[7] += 1;
        if (t1-t0 > 3000) edges.prof[7] += 1ull<<32;
#endif
        const unsigned next = k+edges.step < edges.total ? (unsigned) __builtin_amdgcn_readfirstlane((int) nextV) : cur;
        // (Requesting the NEXT survivor's lines here without waiting for them -- a scalar-cache warm-up -- was measured: no change, 6.26 vs
        // 6.28 ms per step; a batch lands in 380 cycles on average, 4 % of a wavefront's life. It also cannot be made safe in C++: the
        // dummy destination of an unwaited s_load may be copied or spilled by the register allocator and its register reused while the
        // load is still in flight. Gone.)
#if defined(MSDF_ONE_STAGE_RELEVANCE)                               // A/B only
        const bool relevant = MSDF_WAVE_ANY(selEdgeRelevant(sel, r, o));
#else
        double bound2;
        bool relevant = MSDF_WAVE_ANY(selEdgeRelevantBox(sel, r, o, bound2));      // some lane within reach of the control box: evaluate, whatever the wedges say
        if (SEL >= 2 && !relevant)
            relevant = MSDF_WAVE_ANY(selEdgeRelevantWedges<SEL>(r, o, bound2));
#endif
#if defined(MSDF_PROFILE_WAITS)
        MSDF_STAMP(t2);
        edges.prof[6] += t2-t1;
#endif
        if (relevant) {
            if (r.Type() >= 2)                                       // (fetching a curve's E1 + E2 with the first batch: measured, no gain)
                asm volatile("s_load_dwordx16 %0, %2, 0xc0\n\ts_load_dwordx16 %1, %2, 0x100\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r.e1), "=&s"(r.e2) : "s"(rp));
#if defined(MSDF_PROFILE_WAITS)
            MSDF_STAMP(t3);
            if (r.Type() >= 2) edges.prof[2] += t3-t2, edges.prof[3] += 1;
#endif
            selAddEdge(sel, r, (int) (cur&ENTRY_INDEX_MASK), o);
#if defined(MSDF_PROFILE_WAITS)
            {
                double keep = sel.c[0].td;                          // the stamp must not move above the evaluation
                asm volatile("" : "+v"(keep));
                sel.c[0].td = keep;
            }
            MSDF_STAMP(t4);
            edges.prof[4] += t4-t3, edges.prof[5] += 1;
#endif
        }
        cur = next;
    }
}

// Every edge of a glyph for a per-lane query point (k_ec_query, lane-per-candidate chunks), the records fetched like the survivor walk
// above: ONE batch of scalar loads per edge (R + E0 + the type / colour / flags words of block T) instead of a dependent scalar-cache
// round trip per field group -- a chunk of a 44-edge glyph was the ~0.2 ms critical path of the whole launch.
typedef int i4 __attribute__((ext_vector_type(4)));
struct EdgesAllBatched {
    const int32_t *coff;
    __device__ int begin(int c) const { return coff[c]-coff[0]; }
    __device__ int end(int c) const { return coff[c+1]-coff[0]; }
};

template <int SEL>
__device__ inline void selAddContour(Selector<SEL> &sel, const EdgeRec *rec, const EdgesAllBatched &edges, int c, V2 o) {
    const int e = MSDF_UNIFORM(edges.end(c));
    MSDF_NOUNROLL
    for (int i = MSDF_UNIFORM(edges.begin(c)); i < e; ++i) {
        const EdgeRec *rp = rec+i;
        EdgeRegs r;
        i4 words;                                                    // type, color, flags, contour
        asm volatile("s_load_dwordx16 %0, %4, 0x0\n\ts_load_dwordx16 %1, %4, 0x40\n\ts_load_dwordx16 %2, %4, 0x80\n\ts_load_dwordx4 %3, %4, 0x160\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(r.r0), "=&s"(r.r1), "=&s"(r.e0), "=&s"(words) : "s"(rp));
        r.meta = (unsigned) (i&ENTRY_INDEX_MASK)|(unsigned) words[0]<<20|(unsigned) (words[1]&7)<<22|(unsigned) (words[2]&7)<<25|(unsigned) ((words[2]>>4)&1)<<28;   // packEntry
        double bound2;
        bool relevant = MSDF_WAVE_ANY(selEdgeRelevantBox(sel, r, o, bound2));
        if (SEL >= 2 && !relevant)
            relevant = MSDF_WAVE_ANY(selEdgeRelevantWedges<SEL>(r, o, bound2));
        if (relevant) {
            if (r.Type() >= 2)
                asm volatile("s_load_dwordx16 %0, %2, 0xc0\n\ts_load_dwordx16 %1, %2, 0x100\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r.e1), "=&s"(r.e2) : "s"(rp));
            selAddEdge(sel, r, i, o);
        }
    }
}

// Grid form of the distance checks (k_ec_query, round 6): lanes = (candidate, slice). A lane walks edges slice, slice+S, ... of the contour for its own
// candidate -- its own record in registers (loadEdgeRegs: one round trip of independent loads), the edge types diverge like in the cooperative form --
// and the S lanes of a candidate then merge their partial selectors in log2(S) shuffle steps. selMergePartial selects by (|d|, dot, visit index): a total
// order, so the result does not depend on how the contour's edges were dealt (the same argument, and the same function, as the team of k_single_call).
// All S lanes of a candidate hold identical values afterwards and take identical branches; xor-shuffles below S never leave the (aligned) group, so
// divergence BETWEEN candidates (second walks of the combiner) cannot make a lane read an inactive one.
#ifndef MSDF_QGRID_ABLATE
#define MSDF_QGRID_ABLATE 0                                         // measurement builds only (profiles/r06_ab_notes.md 12): 1 loads without evaluation, 2 one record for all lanes, 3 no items, 4 no distance query, 5 no loads / evaluation, 6 no shuffles either, 7 grid items only, 8 all but the grid items, 9 cooperative items without their query, 10 cooperative rounds without loads / evaluation
#endif
struct EdgesGrid {
    const int32_t *coff;
    int slice, S;
    __device__ int begin(int c) const { return coff[c]-coff[0]; }
    __device__ int end(int c) const { return coff[c+1]-coff[0]; }
};

__device__ inline void selAddContour(Selector<2> &sel, const EdgeRec *rec, const EdgesGrid &edges, int c, V2 o) {
    Selector<2> mine;
    selInit(mine);
    const int e = MSDF_UNIFORM(edges.end(c));
#if MSDF_QGRID_ABLATE >= 5                                          // measurement only: no loads, no evaluation (6: no shuffles either)
    mine.c[0].td = o.x, mine.c[0].tdot = o.y+e;
#endif
    MSDF_NOUNROLL
    for (int i = MSDF_UNIFORM(edges.begin(c))+edges.slice; i < (MSDF_QGRID_ABLATE >= 5 ? 0 : e); i += edges.S) {
#if MSDF_QGRID_ABLATE == 2                                          // measurement only: every lane evaluates the contour's first record (loads coalesce and hit)
        const EdgeRegs r = loadEdgeRegs(rec+MSDF_UNIFORM(edges.begin(c)), i);
#else
        const EdgeRegs r = loadEdgeRegs(rec+i, i);
#endif
#if MSDF_QGRID_ABLATE == 1                                          // measurement only: the loads without the evaluation
        mine.c[0].neg += r.r0[0]+r.r1[1]+r.e0[2]+r.e1[3]+r.e2[4];
#else
        selAddEdge(mine, r, i, o);
#endif
    }
    MSDF_NOUNROLL
    for (int off = 1; off < (MSDF_QGRID_ABLATE == 6 ? 1 : edges.S); off <<= 1) {
        Selector<2> other;
        selInit(other);
        PB &m = mine.c[0], &t = other.c[0];
        t.td = __shfl_xor(m.td, off), t.tdot = __shfl_xor(m.tdot, off), t.perp = __shfl_xor(m.perp, off);
        t.neg = __shfl_xor(m.neg, off), t.pos = __shfl_xor(m.pos, off);
        other.idx[0] = __shfl_xor(mine.idx[0], off);
        selMergePartial(mine, other);
    }
    selMergePartial(sel, mine);                                     // (sel: the caller's running selector -- the initial state in the overlapping combiner, earlier contours in the simple one)
}

template <bool OVERLAP, class Wind>
struct PsdfQueryGrid {                                              // PsdfQuery with the lane's slice
    const EdgeRec *rec;
    const int32_t *coff;
    Wind windings;
    int C, slice, S;
    double *res;
    __device__ double operator()(V2 q) const {
        double out[1];
#if MSDF_QGRID_ABLATE == 4                                          // measurement only: candidate, texels, interpolation, stores -- no distance query
        return q.x+q.y;
#endif
        EdgesGrid edges;
        edges.coff = coff, edges.slice = slice, edges.S = S;
        if (OVERLAP)
            shapeDistanceOverlap<2>(rec, edges, windings, C, q, res, WAVE, out);
        else
            shapeDistanceSimple<2>(rec, edges, C, q, out);
        return out[0];
    }
};

// ---- a TEAM of wavefronts on one tile (k_single_call, round 6) -----------------------------------------------------------------------------------
// A single-shape call runs 64 tiles on 1 024 SIMDs: a tile's wavefront is alone on its SIMD and what the call waits for is the serial walk of its busiest
// tile (33 of the launch's 67 us). With TEAM wavefronts per tile, member r walks survivors r, r+TEAM, ... of every contour's (nearest-first) segment into a
// partial selector; the helpers park theirs in LDS, the leader merges them -- selMergePartial: selection by (|d|, dot, visit index) does not depend on how the
// edges were dealt -- and carries on alone with the contour's distance, the combiner and the stores. Two workgroup barriers per contour.
// x: TEAM-1 areas of TEAM_XCHG_DOUBLES x 64 doubles, field-major (lane-consecutive 8-byte accesses).
enum { TEAM_XCHG_DOUBLES = 17 };                                    // 3 channels x (td, tdot, perp, neg, pos) + 3 visit indices (two doubles' worth)
template <int SEL> __device__ inline void teamPut(const Selector<SEL> &s, double *x, int lane) {
    if (SEL == 1) {
        x[lane] = s.m.d, x[WAVE+lane] = s.m.dot;
        reinterpret_cast<int *>(x+15*WAVE)[lane] = s.idx[0];
        return;
    }
    for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i) {
        x[(5*i+0)*WAVE+lane] = s.c[i].td, x[(5*i+1)*WAVE+lane] = s.c[i].tdot, x[(5*i+2)*WAVE+lane] = s.c[i].perp;
        x[(5*i+3)*WAVE+lane] = s.c[i].neg, x[(5*i+4)*WAVE+lane] = s.c[i].pos;
        reinterpret_cast<int *>(x+15*WAVE)[i*WAVE+lane] = s.idx[i];
    }
}
template <int SEL> __device__ inline void teamGet(Selector<SEL> &s, const double *x, int lane) {
    selInit(s);
    if (SEL == 1) {
        s.m.d = x[lane], s.m.dot = x[WAVE+lane];
        s.idx[0] = reinterpret_cast<const int *>(x+15*WAVE)[lane];
        return;
    }
    for (int i = 0; i < (int) SelTraits<SEL>::NPB; ++i) {
        s.c[i].td = x[(5*i+0)*WAVE+lane], s.c[i].tdot = x[(5*i+1)*WAVE+lane], s.c[i].perp = x[(5*i+2)*WAVE+lane];
        s.c[i].neg = x[(5*i+3)*WAVE+lane], s.c[i].pos = x[(5*i+4)*WAVE+lane];
        s.idx[i] = reinterpret_cast<const int *>(x+15*WAVE)[i*WAVE+lane];
    }
}
template <int TEAM>
struct TeamExchange {
    double *x;
    int lane, rank;
    template <int SEL> __device__ bool afterWalk(Selector<SEL> &s) const {
        if (rank != 0)
            teamPut(s, x+(size_t) (rank-1)*TEAM_XCHG_DOUBLES*WAVE, lane);
        __syncthreads();
        if (rank == 0)
            for (int r = 1; r < TEAM; ++r) {
                Selector<SEL> other;
                teamGet(other, x+(size_t) (r-1)*TEAM_XCHG_DOUBLES*WAVE, lane);
                selMergePartial(s, other);
            }
        __syncthreads();                                            // the areas are free for the next contour
        return rank == 0;
    }
    __device__ bool isHelper() const { return rank != 0; }
};

enum { QUAD = 4 };   // tiles per wavefront of the LDS-scratch variant (the global-scratch variant, GRES, takes one: measured faster there)

#ifndef MSDF_SIMPLE_WAVES_PER_SIMD
#define MSDF_SIMPLE_WAVES_PER_SIMD 5     // the simple-combiner instantiations need ~96 VGPRs: five wavefronts per SIMD
#endif
struct DistanceArgs {
    BatchView batch;
    const MsdfHipGlyph *glyphs;
    int width, height, tilesX, tilesPerGlyph, maxEdges;
    float *dst;
    int toScratch;
    unsigned blockBase;
    double *gres;
    size_t gresStride;
    const int *glyphMap;
    int nMapped;
    unsigned *workQueue;               // persistent launch (global-scratch form only): 8 per-XCD item counters, zero when the launch starts (the launch before it zeroed them: queues alternate)
    unsigned workItems;
};

// The body of k_distance as a device function: k_distance is its only caller per instantiation (inlined: the kernel's code is what it was);
// k_single_call (msdf_single.hpp) runs the same body as one phase of a fused launch. blockId = the workgroup's index in the launch.
template <int SEL, bool OVERLAP, bool GRES, int TPW_ = (GRES ? 1 : (int) QUAD), int TEAM = 1>
__device__ __forceinline__ void distanceBody(int nGlyphs, const int32_t *__restrict__ glyphContourOffsets, const int32_t *__restrict__ contourOffsets, const EdgeRec *__restrict__ recs,
           const int8_t *__restrict__ windings, const MsdfHipGlyph *__restrict__ glyphs, int width, int height, int tilesX, int tilesPerGlyph, int maxEdges,
           float *__restrict__ dst, int toScratch, unsigned blockBase, double *__restrict__ gres, size_t gresStride, const int *__restrict__ glyphMap, int nMapped,
           unsigned *__restrict__ workQueue, unsigned workItems, const unsigned blockId, double *smem, const int knownContours = -1, const int knownEdges = -1,
           double *teamXchg = NULL) {
    // (TEAM > 1: a workgroup of TEAM wavefronts on ONE tile -- see TeamExchange; the leader culls, every member walks its share, the leader stores)
    static_assert(TEAM == 1 || (TPW_ == 1 && !GRES), "teams take one tile, combiner scratch in LDS");
    const int teamRank = TEAM > 1 ? MSDF_UNIFORM((int) (threadIdx.x>>6)) : 0;   // (wave-uniform, and said so: the walk's record pointers must be scalar)
    // (knownContours / knownEdges >= 0: a launch over ONE glyph whose counts are kernel arguments -- k_single_call -- skips the dependent loads of the offsets)
    // (every pointer __restrict__: the survivor records are read with SCALAR loads only while the compiler can prove that none of the
    // kernel's own stores -- tiles, workspace, and in the persistent form those of the previous item -- may have clobbered them)
    enum { NCH = SelTraits<SEL>::NCH, TPW = TPW_, ROW = WAVE/TPW };   // tiles per wavefront (4; 1 in the global-scratch form and in the latency-shaped launches), lanes per tile in phase 1
    // One wavefront = TPW consecutive tiles of one glyph. Phase 1 culls for all of them at once -- the edges of a contour rarely fill
    // 64 lanes, so each 16-lane row takes one tile -- and phase 2 then walks the tiles one after the other, lanes = texels.
    // workQueue (global-scratch form only): a persistent launch -- one workgroup per resident wavefront slot, each drawing items from
    // the queue and reusing ITS slice of the workspace, which then stays in L2 / Infinity Cache instead of streaming through HBM.
    BatchView batch;
    batch.nGlyphs = nGlyphs, batch.glyphContourOffsets = glyphContourOffsets, batch.contourOffsets = contourOffsets, batch.recs = recs, batch.windings = windings;
    unsigned item = blockId+blockBase;
    int steal = 0;
    const bool persistent = GRES && workQueue != NULL;
    // The queues of a batch come in PAIRS 64 bytes apart and consecutive launches alternate between them: this launch draws from `workQueue` and leaves the
    // OTHER one zeroed for the next launch -- no memset in front of every launch (msdf_capi.hip: ensureWorkQueue). Done here, at the start, where nothing else
    // is live: the same reset at a workgroup's exit (a count of the workgroups that left against gridDim.x) cost this kernel 592 SGPR-spill lane moves INSIDE
    // its edge loop (tools/isa_loop_depth.py) and 40 % of its speed.
    for (;;) {
    if (persistent) {
        item = nextItem(workQueue, workItems, steal);
        if (item == ~0u) {
            if (blockId == 0 && threadIdx.x < 8)
                reinterpret_cast<unsigned *>(reinterpret_cast<size_t>(workQueue)^64u)[threadIdx.x] = 0u;
            return;
        }
    }
    const int quadsPerGlyph = (tilesPerGlyph+TPW-1)/TPW;
    GlyphWork wk = decodeItem(item, glyphMap ? nMapped : batch.nGlyphs, quadsPerGlyph);
    if (!wk.valid)
        return;                                                         // (direct mapping only: every queued item is valid)
    if (glyphMap)
        wk.g = MSDF_UNIFORM(glyphMap[wk.g]);                            // this launch covers a subset of the batch (bucketed by contour count)
    // (wave-uniform values; made scalar explicitly: inside k_single_call the offsets are memory the launch itself wrote, which the compiler
    // reads with vector loads -- the record pointer of the hand-placed s_load batches has to live in SGPRs)
    const int c0 = knownContours >= 0 ? 0 : MSDF_UNIFORM(batch.glyphContourOffsets[wk.g]), C = knownContours >= 0 ? knownContours : MSDF_UNIFORM(batch.glyphContourOffsets[wk.g+1])-c0;
    const int32_t *coff = batch.contourOffsets+c0;
    const int e0 = knownContours >= 0 ? 0 : MSDF_UNIFORM(coff[0]);
    const int lane = TEAM > 1 ? (int) (threadIdx.x&(WAVE-1)) : (int) threadIdx.x;
    const EdgeRec *rec = batch.recs+e0;

    double *res = GRES ? gres+(size_t) blockId*gresStride : smem; // [C][NCH][64] (overlap only)
    int *lists = reinterpret_cast<int *>(smem+(OVERLAP && !GRES ? (size_t) C*NCH*WAVE : 0));   // [QUAD][maxEdges] survivor indices
    int *cstarts = lists+(size_t) TPW*maxEdges;                                                 // [TPW][C+1] offsets per contour

    const MsdfHipGlyph gd = glyphs[wk.g];
    const Xform t = loadXform(gd);
    WindingMasks wind;
    wind.mem = batch.windings+c0;
    if (OVERLAP) {
        const int w = lane < C ? (int) wind.mem[lane] : 0;
        wind.pos = __ballot(w > 0), wind.neg = __ballot(w < 0);
    }
#if defined(MSDF_PROFILE_WAITS)
    MSDF_STAMP(pStart);
    unsigned long long pAcc[16] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#endif

    // ---- phase 1: cull + compact (row q of 16 lanes = tile q of the quad; lanes of a row = edges)
    // (two divisions per wavefront; texel positions below use divExact. The quotients come out of the VALU in VGPRs although they are wave-uniform, and the
    // 128-VGPR builds keep them in scratch: four dwords reloaded per tile. Forcing them into SGPRs was tried in round 6 -- the readfirstlane builtin is dropped
    // as redundant by the compiler, an asm statement with SGPR outputs miscompiled the mtsdf instantiations -- and left alone: loads that hit L2.)
    const double rsx = 1/t.sx, rsy = 1/t.sy;
    const bool fastXf = divSafe(t.sx) && divSafe(t.sy);
    const double hx = (.5*TILE-.5)*fabs(rsx), hy = (.5*TILE-.5)*fabs(rsy);
    const double tr = sqrt(hx*hx+hy*hy);
    if (TEAM == 1 || teamRank == 0) {
        const int q = lane/ROW, col = lane%ROW;
        const int tileQ = wk.tile*TPW+q;
        const bool tileValid = tileQ < tilesPerGlyph;
        const int txq = tileQ%tilesX, tyq = tileQ/tilesX;
        const V2 tc = mk((txq*TILE+.5*TILE)*rsx-t.tx, (tyq*TILE+.5*TILE)*rsy-t.ty);   // tile centre: only has to be accurate to within the cull slack
        int *list = lists+(size_t) q*maxEdges, *cstart = cstarts+(size_t) q*(C+1);
        int nSurv = 0;
        // The bounds U[ch] are per contour for the overlapping combiner (one selector per contour, every contour's own distance is
        // needed) and over the whole shape for the simple combiner (a single selector).
        double U[3] = { DBL_MAX, DBL_MAX, DBL_MAX };
#if defined(MSDF_ABLATE_PHASE1)                                     // measurement only (with MSDF_ABLATE_PHASE2): launch + header loads + stores
        for (int c = col; c < C; c += ROW)
            cstart[c] = 0;
#else
        const int nE = knownEdges >= 0 ? knownEdges : coff[C]-e0;
#if defined(MSDF_PROFILE_WAITS)
        unsigned long long pHdr;
        {
            int sink = MSDF_UNIFORM(nE+(int) gd.flip);              // the header loads have landed when these are usable
            asm volatile("" : "+s"(sink));
            pHdr = __builtin_readcyclecounter()+(unsigned long long) (sink&0);
        }
#endif
        // Segment-parallel: the lanes of a row take ROW CONSECUTIVE edges of the glyph whatever their contour (a loop per contour left
        // most lanes idle and cost two dependent gathers per contour: 27 rounds for a 14-contour glyph, now 4).
        // Pass A: the bounds. Overlapping combiner: every edge lowers the bound of ITS contour and channels with an LDS atomic minimum
        // (non-negative floats order like their bit patterns); the bounds live in the region of the combiner scratch, which phase 2 only
        // writes after the barrier below (LDS form), or behind the list offsets (global-scratch form). Simple combiner: one bound per
        // channel for the whole shape, a DPP minimum per round.
        unsigned *bounds = (GRES || !OVERLAP ? reinterpret_cast<unsigned *>(cstarts+(size_t) TPW*(C+1)) : reinterpret_cast<unsigned *>(smem))+(size_t) q*C*3;   // [TPW][C][3]
        if (OVERLAP) {
            for (int k = col; k < C*3; k += ROW)
                bounds[k] = 0x7f800000u;
            waveSync();
        }
        for (int base = 0; base < nE; base += ROW) {
            const int i = base+col;
            float sample2 = 0;
            int mask = 0;
            if (i < nE && tileValid) {
                mask = cullMask<SEL>(rec[i]);
                if (mask || !OVERLAP)
                    sample2 = cullNearestSample2(rec[i], tc);
                list[i] = __float_as_int(sample2);                  // pass B's walk-order key; survivors only ever land at or below their own index
            }
            const float ubf = cullUpperFromSample2(sample2);
            if (OVERLAP) {
                if (mask) {
                    unsigned *mine = bounds+(size_t) (rec[i].contour-c0)*3;
                    for (int ch = 0; ch < (SEL <= 2 ? 1 : 3); ++ch)
                        if ((mask>>ch)&1)
                            atomicMin(mine+ch, (unsigned) __float_as_int(ubf));
                }
            } else
                for (int ch = 0; ch < (SEL <= 2 ? 1 : 3); ++ch)
                    U[ch] = dmin(U[ch], (double) (TPW == 1 ? waveMinNonNegative((mask>>ch)&1 ? ubf : __int_as_float(0x7f800000))
                                                           : rowMinNonNegative((mask>>ch)&1 ? ubf : __int_as_float(0x7f800000), lane)));
        }
        if (OVERLAP)
            waveSync();
#if defined(MSDF_PROFILE_WAITS)
        MSDF_STAMP(pPassA);
        pAcc[14] = pHdr-pStart, pAcc[15] = pPassA-pHdr;
#endif
        // Pass B: the cull test against the bounds; survivors go to the list grouped by contour (edges are stored contour by contour, so
        // list order = lane order does it), within a DPP row nearest-first inside each contour's segment.
        for (int base = 0; base < nE; base += ROW) {
            const int i = base+col;
            bool keep = false;
            int c = 0;
            float sample2 = 0;
            if (i < nE) {
                if (tileValid)
                    sample2 = __int_as_float(list[i]);              // read before this round's survivors are written (same wavefront: LDS operations stay in order)
                c = rec[i].contour-c0;
                const int mask = cullMask<SEL>(rec[i]);
                if (mask && tileValid) {
                    double umax = 0;
                    for (int ch = 0; ch < (SEL <= 2 ? 1 : 3); ++ch)
                        if ((mask>>ch)&1)
                            umax = dmax(umax, OVERLAP ? (double) __int_as_float((int) bounds[(size_t) c*3+ch]) : U[ch]);
#if defined(MSDF_NO_TILE_CULL)
                    keep = true;
#else
                    keep = cullEdgeSurvives<(SEL >= 2)>(rec[i], tc, tr, umax);
#endif
                }
            }
            // key = (contour segment within the 16-lane row | distance | slot): contours stay grouped, nearest first inside each. The
            // segment is the contour's offset from the row's first contour -- below 16 unless EMPTY contours (valid input: they own no edge)
            // sit between the row's edges; such a row keeps plain lane order (which is contour order) instead of nearest-first.
            const int segment = c-__shfl(c, lane&~15);
            const bool wideRow = ((__ballot(i < nE && segment > 15)>>(lane&~15))&0xffffull) != 0;
            unsigned key = MSDF_CULL_KEY_DROPPED_SEGMENTED|(unsigned) (col&15);
            if (keep) {
                const unsigned d = (cullOrderKeyOfSample2(sample2, 0)>>3)&0x0ffffff0u;
                key = wideRow ? (unsigned) (col&15) : ((unsigned) segment<<28)|(d < 0x0ffffff0u ? d : 0x0fffffe0u)|(unsigned) (col&15);
            }
            const int rank = rowRank(key);
            const unsigned long long ballot = __ballot(keep);
            const unsigned long long rowBallot = TPW == 1 ? ballot : (ballot>>(ROW*q))&0xffffull;
            const int before = TPW == 1 ? __popcll(ballot&((1ull<<(col&~15))-1ull)) : 0;     // survivors in the earlier rows of a 64-lane chunk
            if (keep)
                list[nSurv+before+rank] = (int) packEntry(i, rec[i]);
            if (i < nE && coff[c]-e0 == i) {                        // first edge of its contour: the contour's list starts here ...
                const int at = nSurv+__popcll(rowBallot&((1ull<<col)-1ull));
                cstart[c] = at;
                for (int cc = c-1; cc >= 0 && coff[cc]-e0 == i; --cc)
                    cstart[cc] = at;                                // ... and so do the (empty) contours right before it
            }
            nSurv += __popcll(rowBallot);
        }
        for (int c = col; c < C; c += ROW)                          // empty contours at the end
            if (coff[c]-e0 == nE)
                cstart[c] = nSurv;
#endif
        if (col == 0)
            cstart[C] = nSurv;
    }
    waveSync();
    if (TEAM > 1)
        __syncthreads();                                            // the leader's lists are the team's
#if defined(MSDF_PROFILE_WAITS)
    MSDF_STAMP(pPhase2);
#endif

    // A latency-shaped launch (k_single_call: this wavefront is alone on its SIMD, nothing hides a miss) first pulls the record lines of ALL the
    // tile's survivors into the scalar cache, four records per wait -- the walk below then pays a cache hit per batch instead of an L2 round
    // trip per evaluated edge, one after the other. (In the batched launches three wavefronts per SIMD hide those: measured, no gain there.)
    if (TPW == 1 && knownContours >= 0) {
        const int total = MSDF_UNIFORM(cstarts[C]);
        for (int base = 0; base < total; base += WAVE) {
            const int entry = base+lane < total ? lists[base+lane] : 0;
            const int n = total-base < WAVE ? total-base : WAVE;
            for (int k = 0; k < n; k += 4) {
                const EdgeRec *r0 = rec+(__builtin_amdgcn_readlane(entry, k)&ENTRY_INDEX_MASK);
                const EdgeRec *r1 = rec+(__builtin_amdgcn_readlane(entry, k+1 < n ? k+1 : k)&ENTRY_INDEX_MASK);
                const EdgeRec *r2 = rec+(__builtin_amdgcn_readlane(entry, k+2 < n ? k+2 : k)&ENTRY_INDEX_MASK);
                const EdgeRec *r3 = rec+(__builtin_amdgcn_readlane(entry, k+3 < n ? k+3 : k)&ENTRY_INDEX_MASK);
                int sink;
                asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_load_dword %0, %1, 0x80\n\ts_load_dword %0, %1, 0xc0\n\ts_load_dword %0, %1, 0x100\n\t"
                             "s_load_dword %0, %2, 0x0\n\ts_load_dword %0, %2, 0x40\n\ts_load_dword %0, %2, 0x80\n\ts_load_dword %0, %2, 0xc0\n\ts_load_dword %0, %2, 0x100\n\t"
                             "s_load_dword %0, %3, 0x0\n\ts_load_dword %0, %3, 0x40\n\ts_load_dword %0, %3, 0x80\n\ts_load_dword %0, %3, 0xc0\n\ts_load_dword %0, %3, 0x100\n\t"
                             "s_load_dword %0, %4, 0x0\n\ts_load_dword %0, %4, 0x40\n\ts_load_dword %0, %4, 0x80\n\ts_load_dword %0, %4, 0xc0\n\ts_load_dword %0, %4, 0x100\n\t"
                             "s_waitcnt lgkmcnt(0)" : "=&s"(sink) : "s"(r0), "s"(r1), "s"(r2), "s"(r3));
            }
        }
    }

    // ---- phase 2: per-texel selection over the survivors (lanes = texels), tile after tile
    MSDF_NOUNROLL
    for (int q = 0; q < TPW; ++q) {
        const int tile = wk.tile*TPW+q;
        if (tile >= tilesPerGlyph)
            break;
        const int tx = tile%tilesX, ty = tile/tilesX;
        const int x = tx*TILE+(lane&(TILE-1)), y = ty*TILE+(lane>>3);
        if (x >= width || y >= height)
            continue;
        const V2 p = fastXf ? mk(divExact(x+.5, t.sx, rsx)-t.tx, divExact(y+.5, t.sy, rsy)-t.ty)
                            : unproject(t, mk(x+.5, y+.5));         // msdfgen.cpp:68 (coord/scale-translate, correctly rounded either way)
#if defined(MSDF_LAZY_RECORDS)                                      // A/B only: compiler-placed field loads, one dependent round trip each
        EdgesCulled edges;
#else
        EdgesCulledPacked edges;
        edges.total = MSDF_UNIFORM(cstarts[(size_t) q*(C+1)+C]);
#if defined(MSDF_PROFILE_WAITS)
        for (int i = 0; i < 16; ++i)
            edges.prof[i] = 0;
        MSDF_STAMP(tTile0);
#endif
#endif
        edges.cstart = cstarts+(size_t) q*(C+1);
        edges.list = lists+(size_t) q*maxEdges;
#if !defined(MSDF_LAZY_RECORDS)
        edges.first = teamRank, edges.step = TEAM;
#endif
        TeamExchange<TEAM> team;
        team.x = teamXchg, team.lane = lane, team.rank = teamRank;
        double d[NCH];
#if defined(MSDF_ABLATE_PHASE2)                                     // measurement only: what phase 1 + the launch cost alone
        for (int ch = 0; ch < NCH; ++ch)
            d[ch] = (double) edges.cstart[C];
#else
        if (OVERLAP) {
#if defined(MSDF_LAZY_RECORDS) || defined(MSDF_ONE_PASS_LOOP)    // A/B: the single rolled pass loop of rounds 2-5
            shapeDistanceOverlap<SEL>(rec, edges, wind, C, p, res+lane, WAVE, d);
#else
            EdgesCulled cold;                                       // the rare second walks: the same survivor lists, record fields loaded where they are used
            cold.cstart = edges.cstart, cold.list = edges.list;
            if (TEAM > 1)
                shapeDistanceOverlapSplit<SEL>(rec, edges, cold, wind, C, p, res+lane, WAVE, d, team);
            else
                shapeDistanceOverlapSplit<SEL>(rec, edges, cold, wind, C, p, res+lane, WAVE, d);
#endif
        } else if (TEAM > 1)
            shapeDistanceSimple<SEL>(rec, edges, C, p, d, team);
        else
            shapeDistanceSimple<SEL>(rec, edges, C, p, d);
#endif
        if (TEAM > 1 && teamRank != 0)
            continue;                                               // a helper: the leader holds the tile's distances and stores them
        // The texel's coordinates are derived AGAIN from the lane index here, through a copy the compiler cannot see through: kept live across the walk,
        // x and y were three dwords of scratch stores per tile in the 128-VGPR builds (half of the pass's spill traffic, 0.3 GB per 8 192 glyphs; round 6).
        int laneAfter = lane;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(laneAfter));
#endif
        const int xo = tx*TILE+(laneAfter&(TILE-1)), yo = ty*TILE+(laneAfter>>3);
        const int yn = gd.flip ? height-1-yo : yo;                  // output.reorient(shape orientation), msdfgen.cpp:55
        float *px = toScratch ? dst+(((size_t) wk.g*height+yn)*width+xo)*NCH
                              : dst+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) NCH*xo;
        for (int ch = 0; ch < NCH; ++ch) {
#if !defined(MSDF_PLAIN_TILE_STORES)
            // streaming (nontemporal) stores: 400 MB of tiles per pass otherwise push the 128-VGPR kernels' scratch lines out of L2 (round 5, A/B: the pass's
            // FETCH_SIZE 361 -> 160 MB, WRITE_SIZE 2.98 -> 2.80 GB, step 5.47 -> 5.39 ms)
            __builtin_nontemporal_store(mapDistance(t, d[ch]), &px[ch]);   // msdfgen.cpp:20-48
#else
            px[ch] = mapDistance(t, d[ch]);
#endif
        }
#if defined(MSDF_PROFILE_WAITS) && !defined(MSDF_LAZY_RECORDS)
        {
            MSDF_STAMP(tTile1);
            edges.prof[13] += tTile1-tTile0;                        // whole tile: prologue + distance + stores
        }
        for (int i = 0; i < 16; ++i)
            pAcc[i] += edges.prof[i];
#endif
    }
#if defined(MSDF_PROFILE_WAITS)
    {
        MSDF_STAMP(pEnd);
        if (lane == 0) {
            atomicAdd(&gWaitProfile[0], 1ull), atomicAdd(&gWaitProfile[1], pEnd-pStart), atomicAdd(&gWaitProfile[2], pPhase2-pStart);
            atomicAdd(&gWaitProfile[3], pAcc[0]), atomicAdd(&gWaitProfile[4], pAcc[1]), atomicAdd(&gWaitProfile[5], pAcc[2]), atomicAdd(&gWaitProfile[6], pAcc[3]);
            atomicAdd(&gWaitProfile[7], pAcc[4]), atomicAdd(&gWaitProfile[8], pAcc[5]), atomicAdd(&gWaitProfile[9], pAcc[6]), atomicAdd(&gWaitProfile[10], pEnd-pPhase2);
            atomicAdd(&gWaitProfile[11], pAcc[7]&0xffffffffull), atomicAdd(&gWaitProfile[12], pAcc[7]>>32);
            atomicAdd(&gWaitProfile[13], pAcc[8]), atomicAdd(&gWaitProfile[14], pAcc[9]), atomicAdd(&gWaitProfile[15], pAcc[10]);
            atomicAdd(&gWaitProfile[16], pAcc[11]), atomicAdd(&gWaitProfile[17], pAcc[13]);
            atomicAdd(&gWaitProfile[18], pAcc[14]), atomicAdd(&gWaitProfile[19], pAcc[15]);
        }
    }
#endif
    if (!persistent)
        return;
    waveSync();                                                     // the survivor lists in LDS are rebuilt for the next item
    }
}

template <int SEL, bool OVERLAP, bool GRES = false, int TPW_ = (GRES ? 1 : (int) QUAD)>
__global__ void __launch_bounds__(WAVE, OVERLAP ? MSDF_DISTANCE_WAVES_PER_SIMD : GRES ? MSDF_SIMPLE_WAVES_PER_SIMD-1 : MSDF_SIMPLE_WAVES_PER_SIMD)   // (simple combiner, one tile per wavefront = latency-bound launches only: 128 VGPRs, no spills)
k_distance(int nGlyphs, const int32_t *__restrict__ glyphContourOffsets, const int32_t *__restrict__ contourOffsets, const EdgeRec *__restrict__ recs,
           const int8_t *__restrict__ windings, const MsdfHipGlyph *__restrict__ glyphs, int width, int height, int tilesX, int tilesPerGlyph, int maxEdges,
           float *__restrict__ dst, int toScratch, unsigned blockBase, double *__restrict__ gres, size_t gresStride, const int *__restrict__ glyphMap, int nMapped,
           unsigned *__restrict__ workQueue, unsigned workItems) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    distanceBody<SEL, OVERLAP, GRES, TPW_>(nGlyphs, glyphContourOffsets, contourOffsets, recs, windings, glyphs, width, height, tilesX, tilesPerGlyph, maxEdges, dst, toScratch, blockBase,
                                     gres, gresStride, glyphMap, nMapped, workQueue, workItems, blockIdx.x, smem);
}

// --------------------------------------------------------------------------------------------------- error correction

// Wind: Contour::winding per contour -- the int8 array in memory, or (k_ec_query) WindingMasks: the combiner reads it up to four times per
// contour and query, and a wave-uniform byte load from global memory is a ~1.5 us round trip each (they were most of a work item's 90 us).
template <bool OVERLAP, class Wind = const int8_t *, class Edges = EdgesAll>
struct PsdfQuery {                                                  // ShapeDistanceFinder<CC<PerpendicularDistanceSelector>>, MSDFErrorCorrection.cpp:97
    const EdgeRec *rec;
    const int32_t *coff;
    Wind windings;
    int C;
    double *res;
    __device__ double operator()(V2 q) const {
        double out[1];
        Edges edges;
        edges.coff = coff;
        if (OVERLAP)
            shapeDistanceOverlap<2>(rec, edges, windings, C, q, res, WAVE, out);
        else
            shapeDistanceSimple<2>(rec, edges, C, q, out);
        return out[0];
    }
};

// Edge policy for a WAVE-UNIFORM query point: the 64 lanes evaluate 64 edges of the contour at once, then the single-edge selector
// states are merged across lanes in visit order. pbMerge keeps the earlier state on ties exactly like the sequential addEdge
// (strict SignedDistance <, edge-selectors.cpp:81-87, :96-106) and max/min of the perpendicular distances do not depend on the
// order, so the merged state -- identical in every lane afterwards -- equals the sequential one bit for bit.
typedef PB PBSlot;                                                  // a single-edge selector state parked in LDS (40 B)

struct EdgesCooperative {
    const int32_t *coff;
    int lane;
    PBSlot *slots;                      // LDS: one slot per edge of the glyph, or NULL (glyph too large: per-contour lane merge instead)
    PBSlot *merged;                     // LDS: one slot per contour (the contour's edges merged in visit order), valid with slots
    int nE, C;
    mutable unsigned long long cached;  // (no slots) contours whose merged state already sits in merged[c]: the combiner's second walks do not evaluate a contour's edges again
    int cacheCap;                       // contours merged[] has room for in that case (0: no caching)
#if defined(MSDF_PROFILE_QUERY)
    mutable unsigned long long prof[16]; // [8] contour walks of pass 0 [9] bookkeeping [10] second walks [11] epilogue | [0] all-edge evaluation [1] per-contour slot merges
#endif
    MSDF_HD int begin(int c) const { return coff[c]-coff[0]; }
    MSDF_HD int end(int c) const { return coff[c+1]-coff[0]; }
};
#if defined(MSDF_PROFILE_QUERY)
__device__ inline unsigned long long qNow() {
#if MSDF_PROFILE_QUERY >= 2
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
    return __builtin_readcyclecounter();
}
__device__ inline void profAdd(const EdgesCooperative &edges, int i, unsigned long long dt) { edges.prof[i] += dt; }
__device__ inline unsigned long long profNow(const EdgesCooperative &) { return qNow(); }
__device__ unsigned long long gQueryDetail[16];
#endif

__device__ inline void selAddContour(Selector<2> &sel, const EdgeRec *rec, const EdgesCooperative &edges, int c, V2 o) {
    if (edges.slots) {
        // All edges of the glyph at once (lanes = edges, whatever their contour), states parked in LDS; every contour then merges its
        // own slots in visit order -- a handful of uniform LDS reads per edge instead of a cross-lane reduction per contour.
        // (Slots are filled when contour 0 is walked: the first walk of every query starts there, later walks reuse them.)
        if (c == 0) {
#if defined(MSDF_PROFILE_QUERY)
            const unsigned long long qa = qNow();
#endif
            for (int base = 0; base < edges.nE; base += WAVE) {
                const int i = base+edges.lane;
                if (i < edges.nE) {
                    Selector<2> mine;
                    selInit(mine);
#if MSDF_QGRID_ABLATE != 10                                         // (10, measurement only: cooperative rounds without loads / evaluation)
                    const EdgeRegs r = loadEdgeRegs(rec+i, i);
                    selAddEdge(mine, r, i, o);
#else
                    mine.c[0].td = o.x+i, mine.c[0].tdot = o.y;
#endif
                    edges.slots[i] = mine.c[0];
                }
            }
            waveSync();
#if defined(MSDF_PROFILE_QUERY)
            const unsigned long long qb = qNow();
            edges.prof[0] += qb-qa;
#endif
            // lanes = contours: each merges ITS contour's slots in visit order (the serial part is the longest contour, not the glyph)
            for (int cc = edges.lane; cc < edges.C; cc += WAVE) {
                PB acc;
                pbInit(acc);
                const int e = edges.end(cc);
                for (int i = edges.begin(cc); i < e; ++i) {
                    const PBSlot one = edges.slots[i];
                    pbMerge(acc, one);
                }
                edges.merged[cc] = acc;
            }
            waveSync();
#if defined(MSDF_PROFILE_QUERY)
            edges.prof[1] += qNow()-qb;
#endif
        }
        // merge(sel, merge(...merge(initial, e_first)..., e_last)) == the sequential merges: the earlier state survives ties either way
        const PBSlot whole = edges.merged[c];
        pbMerge(sel.c[0], whole);
        return;
    }
    // A glyph without slots walks a contour in rounds of 64 edges -- and walked it AGAIN in every second walk of the combiner (members of the inner / outer selector,
    // contour-combiners.cpp:88-93): the contour's merged state is kept in merged[c] after the first walk (round 6; wave-uniform query point, so one bit per contour).
    const bool cacheable = c < 64 && c < edges.cacheCap;
    if (cacheable && ((edges.cached>>c)&1ull)) {
        const PBSlot whole = edges.merged[c];
        pbMerge(sel.c[0], whole);
        return;
    }
    PB contour;
    pbInit(contour);
    const int e = edges.end(c);
    for (int base = edges.begin(c); base < e; base += WAVE) {
        Selector<2> mine;
        selInit(mine);
        const int i = base+edges.lane;
        if (i < e) {
#if MSDF_QGRID_ABLATE != 10
            const EdgeRegs r = loadEdgeRegs(rec+i, i);
            selAddEdge(mine, r, i, o);
#else
            mine.c[0].td = o.x+i, mine.c[0].tdot = o.y;
#endif
        }
        PB &m = mine.c[0];
        MSDF_UNROLL
        for (int off = 1; off < WAVE; off <<= 1) {                  // lane l <- merge(l, l+off): the lower lane is the earlier edge
            PB other;
            other.td = __shfl_down(m.td, off), other.tdot = __shfl_down(m.tdot, off), other.perp = __shfl_down(m.perp, off);
            other.neg = __shfl_down(m.neg, off), other.pos = __shfl_down(m.pos, off);
            if (edges.lane+off < WAVE)
                pbMerge(m, other);
        }
        PB all;                                                     // lane 0 holds the chunk's state: broadcast
        all.td = __shfl(m.td, 0), all.tdot = __shfl(m.tdot, 0), all.perp = __shfl(m.perp, 0), all.neg = __shfl(m.neg, 0), all.pos = __shfl(m.pos, 0);
        pbMerge(contour, all);                                      // chunks in order; the running state is the earlier one
    }
    pbMerge(sel.c[0], contour);                                     // (merge(sel, merge(chunks in order)) == the sequential merges: the earlier state survives ties either way)
    if (cacheable) {
        edges.merged[c] = contour;
        edges.cached |= 1ull<<c;
        waveSync();
    }
}

template <bool OVERLAP, class Wind = const int8_t *>
struct PsdfQueryCooperative {                                       // same query, all 64 lanes working on ONE point
    const EdgeRec *rec;
    const int32_t *coff;
    Wind windings;
    int C, lane;
    double *res;
    PBSlot *slots, *merged;
    int cacheCap = 0;                                               // contours merged[] holds when the glyph has no slots (k_ec_query: the launch's mergedCap)
    __device__ double operator()(V2 q) const {
        double out[1];
        EdgesCooperative edges;
        edges.coff = coff, edges.lane = lane, edges.slots = slots, edges.merged = merged, edges.nE = coff[C]-coff[0], edges.C = C;
        edges.cached = 0, edges.cacheCap = slots ? 0 : cacheCap;
#if defined(MSDF_PROFILE_QUERY)
        for (int i = 0; i < 16; ++i)
            edges.prof[i] = 0;
        const unsigned long long qq0 = qNow();
#endif
#if MSDF_QGRID_ABLATE == 9                                          // measurement only: cooperative items without their distance query
        return q.x+q.y;
#endif
        if (slots)
            waveSync();                                             // the previous query's slot reads are done
        if (OVERLAP)
            shapeDistanceOverlap<2>(rec, edges, windings, C, q, res, 1, out);   // wave-uniform point: every lane holds the same values, one slot per contour
        else
            shapeDistanceSimple<2>(rec, edges, C, q, out);
#if defined(MSDF_PROFILE_QUERY)
        if (lane == 0) {
            atomicAdd(&gQueryDetail[0], 1ull), atomicAdd(&gQueryDetail[1], qNow()-qq0);
            atomicAdd(&gQueryDetail[2], edges.prof[0]), atomicAdd(&gQueryDetail[3], edges.prof[1]), atomicAdd(&gQueryDetail[4], edges.prof[8]);
            atomicAdd(&gQueryDetail[5], edges.prof[9]), atomicAdd(&gQueryDetail[6], edges.prof[10]), atomicAdd(&gQueryDetail[7], edges.prof[11]);
            atomicAdd(&gQueryDetail[8], (unsigned long long) C), atomicAdd(&gQueryDetail[9], (unsigned long long) (slots ? 1 : 0));
        }
#endif
        return out[0];
    }
};

// One deferred distance check: texel (packed index), neighbour direction and interpolation parameter.
// The list is segmented per glyph so that the query kernel runs one glyph per wavefront (uniform edge loop, scalar record loads):
//   header  unsigned[1+G]   [0] = overflow flag (some glyph's segment was full), [1+g] = candidates pushed for glyph g
//   records EcCandidate[G][seg], starting ecHeaderRecords(G) records into the buffer.
// The host zeroes the header before k_ec_fast.
struct EcCandidate {
    unsigned texel;
    int dir;          // (dx+1) | (dy+1)<<2
    double t;
};
static_assert(sizeof(EcCandidate) == 16, "EcCandidate layout");

MSDF_HD size_t ecHeaderRecords(int nGlyphs) { return ((size_t) (nGlyphs+1)*sizeof(unsigned)+sizeof(EcCandidate)-1)/sizeof(EcCandidate); }
MSDF_HD unsigned ecSegment(size_t texelsPerGlyph) { return (unsigned) (texelsPerGlyph/8 > 64 ? (texelsPerGlyph/8+63)/64*64 : 64); }

struct CandidateSink {
    unsigned *header;           // [0] overflow, [1+g] counts
    EcCandidate *segment;       // this glyph's records
    unsigned seg, texel;
    int g;
    __device__ void operator()(double t, int dx, int dy) {
        const unsigned slot = atomicAdd(&header[1+g], 1u);
        if (slot < seg) {
            EcCandidate c;
            c.texel = texel, c.dir = (dx+1)|((dy+1)<<2), c.t = t;
            segment[slot] = c;
        } else
            header[0] = 1u;
    }
};

// Where the stencil byte of the texel in the bitmap's memory row yn goes: the reference's stencil keeps its rows in upward order
// whatever the bitmap's orientation (core/msdf-error-correction.cpp:19, core/MSDFErrorCorrection.cpp:122,192,415); texel = index of
// the texel in the packed field [g][h][w] (memory rows).
MSDF_HD size_t stencilIndex(size_t texel, int yn, int width, int height, int stencilYDown) {
    return stencilYDown ? texel+(size_t) (height-1-2*yn)*width : texel;
}

// Per-glyph constants of the error-correction pass (protectEdges radii, findErrors spans, texel size: MSDFErrorCorrection.cpp:90,
// 194-226, 387-389): 3 sqrt + 8 divisions that every texel of a glyph shares, computed once per glyph instead of once per wavefront.
struct EcGlyphParams {
    double hSpan, vSpan, dSpan, texelX, texelY;
    float radiusH, radiusV, radiusD;
    int cornerBegin, nCorners;          // the glyph's colour-change corners in the launch's corner list (k_ec_params)
    int pad;
};
static_assert(sizeof(EcGlyphParams) == 64, "EcGlyphParams layout");

// Per-glyph constants of the error-correction pass, one wavefront per glyph: the spans / radii of MSDFErrorCorrection.cpp:180-187 and
// the texel corners protectCorners marks (:121-151; lanes = edges, the (l, b) texel pair of every colour-change corner under this
// call's transform, appended at corners[2*(e0+slot)]). k_ec_fast then needs ONE dependent load per tile (params -> its corner list)
// instead of walking glyphContourOffsets -> contourOffsets -> the records' flags in each of the glyph's tiles.
__device__ __forceinline__ void ecParamsBody(EcGlyphParams *out, const BatchView &batch, const MsdfHipGlyph *glyphs, const MsdfHipConfig &cfg, unsigned *candidateHeader,
                                             int *corners, int *sizes, const int g, const int lane) {
    if (candidateHeader && lane == 0) {                             // zero the candidate counters for k_ec_fast (saves a memset launch)
        candidateHeader[1+g] = 0;
        if (g == 0)
            candidateHeader[0] = 0;
    }
    EcParams p;
    p.t = loadXform(glyphs[g]);
    p.minDeviationRatio = cfg.min_deviation_ratio;
    p.minImproveRatio = cfg.min_improve_ratio;
    ecDerive(p);
    const int c0 = batch.glyphContourOffsets[g], C = batch.glyphContourOffsets[g+1]-c0;
    const int e0 = batch.contourOffsets[c0], nE = batch.contourOffsets[c0+C]-e0;
    int nCorners = 0;
    if (cfg.ec_mode == EC_MODE_EDGE_PRIORITY && corners) {
        const EdgeRec *rec = batch.recs+e0;
        for (int base = 0; base < nE; base += WAVE) {
            const int e = base+lane;
            const bool corner = e < nE && (rec[e].flags&REC_CORNER);
            const unsigned long long mask = __ballot(corner);
            if (corner) {
                const V2 pp = project(p.t, ld(rec[e].p0));
                const int slot = e0+nCorners+__popcll(mask&((1ull<<lane)-1ull));
                corners[2*slot] = (int) floor(pp.x-.5);
                corners[2*slot+1] = (int) floor(pp.y-.5);
            }
            nCorners += __popcll(mask);
        }
    }
    if (lane == 0) {
        EcGlyphParams o;
        o.hSpan = p.hSpan, o.vSpan = p.vSpan, o.dSpan = p.dSpan, o.texelX = p.texelX, o.texelY = p.texelY;
        o.radiusH = p.radiusH, o.radiusV = p.radiusV, o.radiusD = p.radiusD, o.cornerBegin = e0, o.nCorners = nCorners, o.pad = 0;
        out[g] = o;
        if (sizes)
            sizes[2*g] = nE, sizes[2*g+1] = C;
    }
}
__global__ void __launch_bounds__(WAVE)
k_ec_params(EcGlyphParams *out, BatchView batch, const MsdfHipGlyph *glyphs, MsdfHipConfig cfg, unsigned *candidateHeader, int *corners, int *sizes) {
    ecParamsBody(out, batch, glyphs, cfg, candidateHeader, corners, sizes, (int) blockIdx.x, (int) threadIdx.x);
}

// Error correction, fast sweep over ALL texels (msdf_ec_fast.hpp). src: pre-correction field, packed [g][h][w][N] in native row order.
// Writes corrected texels to the caller's bitmap (msdfErrorCorrectionInner, core/msdf-error-correction.cpp:12-48) and, if stencilOut,
// the stencil byte [g][h][w] (native rows). Candidates whose verdict needs an exact shape-distance query are appended to `cands`
// and judged by k_ec_query; a texel with a cheaply decided ERROR never needs them (the flag is an OR).
#ifndef MSDF_EC_FAST_WAVES_PER_SIMD
#define MSDF_EC_FAST_WAVES_PER_SIMD 7   // 72 VGPRs, no spills; the kernel's 5.6 KB of LDS allow 7 wavefronts per SIMD anyway. Correction pass on the distinct-glyph set: 8 -> 1.94 ms, 7 -> 1.90, 6 -> 1.90
#endif
// LDS of k_ec_fast per wavefront: 10x10 halo tile of the field | per-texel verdict words | item count | item queue (5.6 KB for msdf).
enum { EC_HALO = TILE+2, EC_QUEUE_CAP = WAVE*24, EC_PROTECT_QUEUE_CAP = WAVE*8 };
__host__ __device__ inline size_t ecFastLdsBytes(int maxEdges, int n) {
    return (size_t) EC_HALO*EC_HALO*n*sizeof(float)+(WAVE+4)*sizeof(int)
           +(EC_QUEUE_CAP+EC_PROTECT_QUEUE_CAP)*sizeof(unsigned short);
}

template <int N>
__device__ __forceinline__ void ecFastBody(const BatchView &batch, const MsdfHipGlyph *glyphs, int width, int height, int tilesX, int tilesPerGlyph,
          const float *src, float *out, uint8_t *stencilOut, const MsdfHipConfig &cfg, const EcGlyphParams *glyphParams, EcCandidate *cands, unsigned seg,
          int maxEdges, const int *corners, const unsigned blockId, int *smemFast, const int sinkSlots = 0, const int sinkSlot = 0) {
    // (sinkSlots > 0: the distance-check candidates go to segment sinkSlot of sinkSlots instead of the glyph's -- k_single_call gives every TILE its own
    // segment and lets the tile's workgroup judge them itself)
    const GlyphWork wk = decodeItem(blockId, batch.nGlyphs, tilesPerGlyph);
    if (!wk.valid)
        return;
    float *halo = reinterpret_cast<float *>(smemFast);                                      // [EC_HALO*EC_HALO][N]
    int *verdictLds = reinterpret_cast<int *>(halo+EC_HALO*EC_HALO*N);                      // [WAVE]: bits 0-1 judge(), bit 8 protected
    int *itemCount = verdictLds+WAVE;
    unsigned short *queue = reinterpret_cast<unsigned short *>(itemCount+4);               // [EC_QUEUE_CAP]: lane | k<<6 | j<<9
    unsigned short *protectQueue = queue+EC_QUEUE_CAP;                                      // [EC_PROTECT_QUEUE_CAP]: lane | m<<6
    // |median - .5| of every halo texel, computed once while the halo is staged (msdf_ec_fast.hpp: Neighbourhood::dev). The table overlays the TAIL of the
    // item queue: every lane has its nine values in registers before the wavefront pushes its first item (one wavefront, LDS operations in program order, a
    // wave barrier between the reads and the pushes) -- no LDS is added, the kernel keeps its seven wavefronts per SIMD.
    float *devLds = reinterpret_cast<float *>(queue+EC_QUEUE_CAP)-EC_HALO*EC_HALO;
    static_assert((EC_QUEUE_CAP*sizeof(unsigned short))%sizeof(float) == 0 && EC_QUEUE_CAP*sizeof(unsigned short) >= EC_HALO*EC_HALO*sizeof(float), "devLds overlays the queue's tail");
    const int lane = threadIdx.x;
    const MsdfHipGlyph gd = glyphs[wk.g];
    EcParams p;
    p.t = loadXform(gd);
    p.minDeviationRatio = cfg.min_deviation_ratio;
    p.minImproveRatio = cfg.min_improve_ratio;
    p.mode = cfg.ec_mode, p.distanceCheck = cfg.ec_distance_check, p.overlap = cfg.overlap_support, p.stageLimit = 0;
    const EcGlyphParams gp = glyphParams[wk.g];
    p.hSpan = gp.hSpan, p.vSpan = gp.vSpan, p.dSpan = gp.dSpan, p.texelX = gp.texelX, p.texelY = gp.texelY;
    p.radiusH = gp.radiusH, p.radiusV = gp.radiusV, p.radiusD = gp.radiusD;
    const int tx = wk.tile%tilesX, ty = wk.tile/tilesX;

    // protectCorners (MSDFErrorCorrection.cpp:121-151): lanes = the glyph's corner texel pairs (k_ec_params); the few that touch this
    // tile are broadcast one by one and every texel lane tests itself -- no list in LDS (it had been sized by the batch's largest glyph
    // and capped the kernel at 4 wavefronts per SIMD on a batch with one 543-edge symbol)
    const int lxT = lane&(TILE-1), lyT = lane>>3;
    const int xT = tx*TILE+lxT, ysT = gd.flip ? height-1-(ty*TILE+lyT) : ty*TILE+lyT;
    bool cornerTexel = false;
    if (p.mode == EC_MODE_EDGE_PRIORITY) {
        const int x0 = tx*TILE, x1 = tx*TILE+TILE-1;
        const int ya = gd.flip ? height-1-(ty*TILE+TILE-1) : ty*TILE, yb = gd.flip ? height-1-ty*TILE : ty*TILE+TILE-1;   // shape-space rows of the tile
        const int *list = corners+2*(size_t) gp.cornerBegin;
        for (int base = 0; base < gp.nCorners; base += WAVE) {
            const int k = base+lane;
            int l = 0, b = 0;
            bool near = false;
            if (k < gp.nCorners) {
                l = list[2*k], b = list[2*k+1];
                near = l+1 >= x0 && l <= x1 && b+1 >= ya && b <= yb;
            }
            for (unsigned long long mask = __ballot(near); mask; mask &= mask-1) {
                const int src = __ffsll((long long) mask)-1;
                const int cl = __builtin_amdgcn_readlane(l, src), cb = __builtin_amdgcn_readlane(b, src);
                cornerTexel = cornerTexel || ((xT == cl || xT == cl+1) && (ysT == cb || ysT == cb+1));
            }
        }
    }

    // ---- the tile and its one-texel halo go to LDS once (native rows); texels outside the bitmap are never read back
    const float *field = src+(size_t) wk.g*height*width*N;
    for (int idx = lane; idx < EC_HALO*EC_HALO; idx += WAVE) {
        const int hx = tx*TILE+idx%EC_HALO-1, hy = ty*TILE+idx/EC_HALO-1;
        if (hx >= 0 && hy >= 0 && hx < width && hy < height) {
            const float *t = field+((size_t) hy*width+hx)*N;
            float tv[N];
            for (int ch = 0; ch < N; ++ch)
                tv[ch] = t[ch], halo[idx*N+ch] = tv[ch];
            devLds[idx] = ecTexelDeviation(tv);
        }
    }
    if (lane < 2)
        itemCount[lane] = 0;
    waveSync();

    // ---- phase A (lane = texel): corner protection and the sign / radius stages of findErrors and protectEdges; survivors are queued
    const int lx = lane&(TILE-1), ly = lane>>3;
    const int x = tx*TILE+lx, yn = ty*TILE+ly;
    const bool inside = x < width && yn < height;
    int st = 0;
#if defined(MSDF_EC_ABLATE) && MSDF_EC_ABLATE == 1                          // measurement only: halo load + store, no classification
    const bool classify = false;
#else
    const bool classify = inside;
#endif
    Neighbourhood nb;
    nb.valid = 0;
    if (classify) {
        MSDF_UNROLL
        for (int dy = -1; dy <= 1; ++dy) {
            MSDF_UNROLL
            for (int dx = -1; dx <= 1; ++dx) {
                const int nx = x+dx, ny = yn+dy;
                const bool in = nx >= 0 && ny >= 0 && nx < width && ny < height;
                const float *t = halo+((ly+1+(in ? dy : 0))*EC_HALO+lx+1+(in ? dx : 0))*N;
                nb.v[dy+1][dx+1][0] = t[0], nb.v[dy+1][dx+1][1] = t[1], nb.v[dy+1][dx+1][2] = t[2];
                nb.dev[dy+1][dx+1] = devLds[(ly+1+(in ? dy : 0))*EC_HALO+lx+1+(in ? dx : 0)];
                if (in)
                    nb.valid |= 1u<<((dy+1)*3+(dx+1));
            }
        }
    }
    waveSync();                                                         // devLds is dead from here on: the queue may grow into it
    if (classify) {
        if (p.mode == EC_MODE_EDGE_PRIORITY) {
            if (cornerTexel)
                st |= EC_PROTECTED;
            if (!(st&EC_PROTECTED)) {
                struct PushProtect {
                    unsigned short *queue;
                    int *count;
                    int lane;
                    __device__ void operator()(int m) { queue[atomicAdd(count, 1)] = (unsigned short) (lane|m<<6); }
                } pushProtect = { protectQueue, itemCount+1, lane };
                texelProtectPairs(nb, p, pushProtect);
            }
        } else if (p.mode == EC_MODE_EDGE_ONLY)
            st |= EC_PROTECTED;
        struct Push {
            unsigned short *queue;
            int *count;
            int lane;
            __device__ void operator()(int k, int j) { queue[atomicAdd(count, 1)] = (unsigned short) (lane|k<<6|j<<9); }
        } push = { queue, itemCount, lane };
        texelCandidatePairs(nb, push);
    }
    verdictLds[lane] = (st&EC_PROTECTED) ? 0x100 : 0;
    waveSync();

    // ---- phase B (lane = queued pair): protectEdges' edgeBetweenTexels tests, densely; a hit protects the owning texel
#if defined(MSDF_EC_ABLATE) && MSDF_EC_ABLATE <= 2
    const int nProtect = 0;
#else
    const int nProtect = itemCount[1];
#endif
    for (int base = 0; base < nProtect; base += WAVE) {
        const int it = base+lane;
        if (it >= nProtect)
            continue;
        const unsigned item = protectQueue[it];
        const int s = item&63, m = item>>6;
        const int sx = s&(TILE-1), sy = s>>3;
        const float *self = halo+((sy+1)*EC_HALO+sx+1)*N, *other = halo+((sy+m/3)*EC_HALO+sx+m%3)*N;
        if (evaluateProtectPair(self, other, m))
            atomicOr(&verdictLds[s], 0x100);
    }
    waveSync();

    // ---- phase C (lane = queued test): stage 2, densely; verdicts are OR-ed into the owning texel's word
#if defined(MSDF_EC_ABLATE) && MSDF_EC_ABLATE <= 3
    const int nItems = 0;
#else
    const int nItems = itemCount[0];
#endif
    for (int base = 0; base < nItems; base += WAVE) {
        const int it = base+lane;
        if (it >= nItems)
            continue;
        const unsigned item = queue[it];
        const int s = item&63, k = (item>>6)&7, jp = item>>9;
        const int sx = s&(TILE-1), sy = s>>3;
        const int dx = ecNeighbourDx(k), dy = ecNeighbourDy(k);
        const float *c = halo+((sy+1)*EC_HALO+sx+1)*N, *n = halo+((sy+1+dy)*EC_HALO+sx+1+dx)*N;
        const float *hb = halo+((sy+1)*EC_HALO+sx+1+dx)*N, *vc = halo+((sy+1+dy)*EC_HALO+sx+1)*N;
        CandidateSink sink;
        sink.header = reinterpret_cast<unsigned *>(cands);
        sink.segment = sinkSlots > 0 ? cands+ecHeaderRecords(sinkSlots)+(size_t) sinkSlot*seg : cands+ecHeaderRecords(batch.nGlyphs)+(size_t) wk.g*seg;
        sink.seg = seg, sink.g = sinkSlots > 0 ? sinkSlot : wk.g;
        sink.texel = (unsigned) (((size_t) wk.g*height+ty*TILE+sy)*width+tx*TILE+sx);
        const int v = evaluatePair(c, n, hb, vc, p, (verdictLds[s]&0x100) != 0, gd.flip, k, jp, sink);
        if (v)
            atomicOr(&verdictLds[s], v);
    }
    waveSync();

    if (!inside)
        return;
    const int verdict = verdictLds[lane];
    if ((verdict&0x100) || (ecHasBasePass(p) && p.distanceCheck == EC_CHECK_AT_EDGE))
        st |= EC_PROTECTED;                                         // protectEdges hit (phase B) / protectAll (:38-39)
    if (verdict&1)
        st |= EC_ERROR;
    const size_t texel = ((size_t) wk.g*height+yn)*width+x;
    const float *in = halo+((ly+1)*EC_HALO+lx+1)*N;
    float v[N];
    for (int i = 0; i < N; ++i)
        v[i] = in[i];
    if (st&EC_ERROR) {                                              // apply, MSDFErrorCorrection.cpp:459-479 (alpha untouched)
        const float m = medianf(v[0], v[1], v[2]);
        v[0] = m, v[1] = m, v[2] = m;
    }
    float *px = out+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) N*x;
    for (int i = 0; i < N; ++i)
        px[i] = v[i];
    if (stencilOut)
        stencilOut[stencilIndex(texel, yn, width, height, cfg.stencil_y_down)] = (uint8_t) st;
}
template <int N>
__global__ void __launch_bounds__(WAVE, MSDF_EC_FAST_WAVES_PER_SIMD)
k_ec_fast(BatchView batch, const MsdfHipGlyph *glyphs, int width, int height, int tilesX, int tilesPerGlyph,
          const float *src, float *out, uint8_t *stencilOut, MsdfHipConfig cfg, const EcGlyphParams *glyphParams, EcCandidate *cands, unsigned seg,
          int maxEdges, const int *corners) {
    extern __shared__ int smemFast[];
    ecFastBody<N>(batch, glyphs, width, height, tilesX, tilesPerGlyph, src, out, stencilOut, cfg, glyphParams, cands, seg, maxEdges, corners, blockIdx.x, smemFast);
}

// Two ways to spend a wavefront on deferred distance checks of a glyph with nE edges:
//   cooperative        one candidate at a time, lanes = edges (one round of edge evaluations per 64 edges, then a serial merge of the
//                      single-edge states per contour): cost per candidate ~ 340*ceil(nE/64) + 10*nE + 1000 instructions;
//   lane per candidate 64 candidates at a time, every lane walks all edges (uniform control flow, scalar record loads -- the loop of
//                      k_distance): cost per chunk ~ 340*nE + 1000, whatever the number of live lanes.
// The first wins for the usual handful of candidates, the second for many (an icon with 186, a 1024x1024 logo with thousands).
//   grid (round 6)     lanes = (candidate, slice): the 64 lanes are J = 64/S candidates x S slices; a lane walks edges slice, slice+S, ... of every contour
//                      for ITS candidate (its own record in registers, like the cooperative form) and the S lanes of a candidate merge their partial
//                      selectors with lane shuffles (selMergePartial: a total order, the union's winner is the better of the partial winners). The usual
//                      glyph (7 candidates, 23 edges: J = 8, S = 8) is ONE item of ~3 evaluation steps per lane, where the forms above spend either a serial
//                      walk of all 23 edges with 7 of 64 lanes alive, or 7 items with 23 of 64 lanes alive: k_ec_query is a chain of item latencies.
//                      S = max(64 / pow2(count), pow2(nE / gridSteps)); S = 64 is the cooperative form, S = 1 the lane-per-candidate form -- they stay.
struct EcQueryPolicy { int lpcMaxContours, lpcEdgeCost, lpcMaxEdges, lpcMinCount, wideMaxEdges; float wideLoad, wideMeanCount; int gridSteps; };
MSDF_HD int ecQueryGridSlices(unsigned count, int nE, int C, EcQueryPolicy q) {     // S of the grid form, or 0: one of the other two forms
    if (q.gridSteps <= 0 || C > q.lpcMaxContours || count == 0)      // (the [contour][lane] scratch of the lane-per-candidate form is the grid form's too)
        return 0;
    int J = 1;
    while (J < (int) count && J < WAVE)
        J <<= 1;
    int S = WAVE/J, need = 1;
    while (need*q.gridSteps < nE && need < WAVE)
        need <<= 1;
    if (need > S)
        S = need;
    return S >= 2 && S <= WAVE/2 ? S : 0;
}
MSDF_HD bool ecQueryLanePerCandidate(unsigned count, int nE, int C, EcQueryPolicy q) {
    if (C > q.lpcMaxContours)                                        // its [contour][lane] scratch would not fit the LDS the launch reserved
        return false;
    if ((nE > q.lpcMaxEdges && (int) count < q.lpcMinCount))         // a chunk is one long serial walk: only worth it when there are many chunks
        return false;
    const long long coop = (long long) count*(340ll*((nE+WAVE-1)/WAVE)+10ll*nE+1000ll);
    const long long lpc = (long long) ((count+WAVE-1)/WAVE)*((long long) q.lpcEdgeCost*nE+1000ll);
    return lpc < coop;
}
MSDF_HD int ecQueryItems(unsigned count, unsigned seg, int nE, int C, EcQueryPolicy lpcMaxContours) {     // work items of one glyph; a glyph whose segment overflowed has none (k_ec_slow redoes it)
    if (count > seg)
        return 0;
    const int S = ecQueryGridSlices(count, nE, C, lpcMaxContours);
    if (S)
        return (int) ((count+WAVE/S-1)/(WAVE/S));
    return ecQueryLanePerCandidate(count, nE, C, lpcMaxContours) ? (int) ((count+WAVE-1)/WAVE) : (int) count;
}
// ... and whether they are listed with the chunks (the long items, drawn first) or with the cooperative items
// (round 6: with the grid form the longest single items are the cooperative ones of glyphs with several ROUNDS of edges -- 186 candidates x 9 rounds for the 543-edge
// symbol of the DejaVu set, 0.15 ms for its 34 largest glyphs alone -- and they were drawn LAST: now they go first, with their glyph's position in the heavy-first order)
MSDF_HD bool ecQueryListedFirst(unsigned count, int nE, int C, EcQueryPolicy q) {
    return ecQueryGridSlices(count, nE, C, q) != 0 || ecQueryLanePerCandidate(count, nE, C, q) || (q.gridSteps > 0 && nE > WAVE);
}

// Prefix sums of the per-glyph work items of k_ec_query (one workgroup). The list is ordered HEAVY FIRST: the lane-per-candidate chunks
// (one long serial walk each) of all glyphs, then the cooperative items -- drawn late, a chunk was what the launch ended up waiting
// for. offsets[0..G] = chunks of glyphs < g (offsets[G] = all chunks); offsets[G+1..2G+1] = the same for the cooperative items,
// offsets[2G+1] = all of them; offsets[2G+2] = the work counter of k_ec_query (zeroed here). A glyph has items of one kind only.
// Which glyphs take the lane-per-candidate form depends on the LOAD of the launch: a chunk costs the fewest instructions but is one
// serial walk of all the glyph's edges by one wavefront. With little work (55 k candidates of the DejaVu set: the kernel is a latency
// chain) only small glyphs (<= lpcMaxEdges) use it; when the cooperative form of everything would cost more than wideLoad instructions
// (440 k candidates of the CJK-like set: throughput bound) glyphs up to wideMaxEdges do. offsets[2G+3] = the bound in force.
// ... followed by (edges, contours) per glyph, written by k_ec_params (k_ec_scan then reads two ints instead of walking two offset arrays).
MSDF_HD size_t ecQueuesAt(int nGlyphs) { return 2*((size_t) nGlyphs+1)+2+2*(size_t) nGlyphs; }      // eight ticket counters, 64 bytes apart (k_ec_query, queryFlags & 4)
MSDF_HD size_t ecOffsetInts(int nGlyphs) { return ecQueuesAt(nGlyphs)+8*16; }
MSDF_HD size_t ecSizesAt(int nGlyphs) { return 2*((size_t) nGlyphs+1)+2; }

// Positions, not glyphs: `order` (or NULL = identity) lists the glyphs HEAVIEST FIRST (edges x contours, sorted once per batch by the host),
// offsets[] is indexed by position and k_ec_query maps a position back through the same array -- the launch is as long as the work it
// starts last, like the distance pass (msdf_capi.hip: ensureBuckets).
__global__ void __launch_bounds__(1024)
k_ec_scan(int nGlyphs, const unsigned *__restrict__ header, unsigned seg, int *__restrict__ offsets, EcQueryPolicy lpcMaxContours, const int *__restrict__ order) {
    // Sixteen wavefronts: sums and prefix sums run inside a wavefront with lane shuffles and meet once per step in LDS (the first form met at
    // ~30 workgroup barriers -- a tree in LDS -- and took 47 us of a 1.8 ms correction pass).
    __shared__ int waveTotals[2][16];
    __shared__ float waveLoad[16], waveCount[16];
    enum { PER = 8 };                                               // positions per thread and round: their loads are issued together
    const int t = threadIdx.x, lane = t&(WAVE-1), wave = t/WAVE;
    const int *sizes = offsets+ecSizesAt(nGlyphs);
    float mine = 0, mineCount = 0;
    for (int base = 0; base < nGlyphs; base += 1024*PER) {
        unsigned count[PER];
        int nE[PER];
        MSDF_UNROLL
        for (int k = 0; k < PER; ++k) {
            const int p = base+t*PER+k;
            const int g = p < nGlyphs ? (order ? order[p] : p) : -1;
            count[k] = g >= 0 ? header[1+g] : 0u, nE[k] = g >= 0 ? sizes[2*g] : 0;
        }
        MSDF_UNROLL
        for (int k = 0; k < PER; ++k)
            if (count[k] <= seg)
                mine += (float) count[k]*(340.f*((nE[k]+WAVE-1)/WAVE)+10.f*nE[k]+1000.f), mineCount += (float) count[k];
    }
    for (int off = WAVE/2; off > 0; off >>= 1)
        mine += __shfl_down(mine, off), mineCount += __shfl_down(mineCount, off);
    if (lane == 0)
        waveLoad[wave] = mine, waveCount[wave] = mineCount;
    __syncthreads();
    float total = 0, totalCount = 0;
    for (int w = 0; w < 16; ++w)
        total += waveLoad[w], totalCount += waveCount[w];           // (the same sums in every thread; they only pick a policy, no result depends on them)
    // ... or when the launch's glyphs have MANY candidates each (round 4): a chunk of a glyph with ~50 candidates fills its lanes whatever the size of
    // the launch -- a 1 024-glyph shard of the CJK-like set (load 9e7, like ALL of the DejaVu set with its 7 candidates per glyph) was left with
    // the cooperative form: config-4 shards 8.29 -> 7.13 ms (2-way), 4.39 -> 3.94 (4-way), 2.45 -> 2.32 (8-way)
    const bool dense = lpcMaxContours.wideMeanCount > 0 && totalCount >= lpcMaxContours.wideMeanCount*(float) nGlyphs;
    if ((total > lpcMaxContours.wideLoad || dense) && lpcMaxContours.wideMaxEdges > lpcMaxContours.lpcMaxEdges)
        lpcMaxContours.lpcMaxEdges = lpcMaxContours.wideMaxEdges;
    // ... and such a launch is throughput bound: the grid form (fewer live edges per instruction than a chunk's uniform walk) is for latency chains only
    // (CJK-like set with the grid form: correction 2.08 -> 3.17 ms). Published to k_ec_query with the edge bound (bit 30).
    const bool gridOff = total > lpcMaxContours.wideLoad || dense || lpcMaxContours.gridSteps <= 0;
    if (gridOff)
        lpcMaxContours.gridSteps = 0;
    int *coop = offsets+nGlyphs+1;
    int carry[2] = { 0, 0 };                                        // items of the earlier rounds (batches of more than 8 192 glyphs)
    for (int base = 0; base < nGlyphs; base += 1024*PER) {
        int items[PER];
        bool chunky[PER];
        int sum[2] = { 0, 0 };
        MSDF_UNROLL
        for (int k = 0; k < PER; ++k) {
            const int p = base+t*PER+k;
            const int g = p < nGlyphs ? (order ? order[p] : p) : -1;
            const unsigned count = g >= 0 ? header[1+g] : 0u;
            const int nE = g >= 0 ? sizes[2*g] : 0, C = g >= 0 ? sizes[2*g+1] : 0;
            chunky[k] = g >= 0 && ecQueryListedFirst(count, nE, C, lpcMaxContours);
            items[k] = g >= 0 ? ecQueryItems(count, seg, nE, C, lpcMaxContours) : 0;
            sum[chunky[k] ? 0 : 1] += items[k];
        }
        int incl[2] = { sum[0], sum[1] };                           // inclusive scan over the lanes of this wavefront
        for (int off = 1; off < WAVE; off <<= 1) {
            const int n0 = __shfl_up(incl[0], off), n1 = __shfl_up(incl[1], off);
            if (lane >= off)
                incl[0] += n0, incl[1] += n1;
        }
        if (lane == WAVE-1)
            waveTotals[0][wave] = incl[0], waveTotals[1][wave] = incl[1];
        __syncthreads();
        int before[2] = { 0, 0 }, all[2] = { 0, 0 };
        for (int w = 0; w < 16; ++w) {
            const int v0 = waveTotals[0][w], v1 = waveTotals[1][w];
            if (w < wave)
                before[0] += v0, before[1] += v1;
            all[0] += v0, all[1] += v1;
        }
        int at[2] = { carry[0]+before[0]+incl[0]-sum[0], carry[1]+before[1]+incl[1]-sum[1] };
        MSDF_UNROLL
        for (int k = 0; k < PER; ++k) {
            const int p = base+t*PER+k;
            if (p < nGlyphs) {
                offsets[p] = at[0], coop[p] = at[1];
                at[chunky[k] ? 0 : 1] += items[k];
            }
        }
        carry[0] += all[0], carry[1] += all[1];
        __syncthreads();                                            // waveTotals are rewritten by the next round
    }
    if (t == 1023) {
        offsets[nGlyphs] = carry[0];
        coop[nGlyphs] = carry[1];
        coop[nGlyphs+1] = 0;
        for (int q = 0; q < 8; ++q)
            offsets[ecQueuesAt(nGlyphs)+16*q] = 0;
        coop[nGlyphs+2] = lpcMaxContours.lpcMaxEdges|(gridOff ? 1<<30 : 0);
    }
}

// The deferred distance checks. The work items of the whole batch form one flat list (offsets[], k_ec_scan); a wavefront grabs one
// item at a time from an atomic counter -- glyphs differ by orders of magnitude in candidates x edges (one symbol of the DejaVu set:
// 186 candidates x 543 edges), a static split per glyph left the launch waiting for a single wavefront. An item is one candidate
// (cooperative: the 64 lanes evaluate the glyph's edges in parallel, EdgesCooperative; the query point is wave-uniform, so the
// combiner scratch is one double per contour) or a chunk of 64 candidates of one glyph (lane per candidate, scratch [contour][lane]).
// A candidate that turns out to be an artifact flags its texel: rgb := median (apply, MSDFErrorCorrection.cpp:459-479),
// stencil |= ERROR. Several candidates of one texel write identical values.
#ifndef MSDF_EC_QUERY_WAVES_PER_SIMD
#define MSDF_EC_QUERY_WAVES_PER_SIMD 3   // 168 VGPRs (13 spilled) with the record images of the cooperative path; 2 -> CJK-like set 2.90 ms, 3 -> 2.61, 4 (119 spilled) -> 2.94
#endif
#if defined(MSDF_PROFILE_QUERY)
// Measurement build (tools/profile_query.py): where a wavefront of k_ec_query spends its cycles. [0] waves [1] wave cycles [2] waves with work
// [3] first start [4] last end (atomic min / max) | chunk items: [5] count [6] cycles | cooperative items: [7] count [8] cycles |
// [9] ticket wait [10] lookup [11] glyph state [12] evaluation (candidate load, interpolation, distance query, comparison) [13] store + barrier
// [14] longest item [15] cooperative rounds (64 edges each) [16] end of the last wavefront that still found work
__device__ unsigned long long gQueryProfile[24];
// -DMSDF_PROFILE_QUERY=2: every stamp waits for all outstanding memory operations first (attributes the time to the segments, slows the
// kernel); =1: plain s_memtime stamps -- per-item totals, the longest item and a histogram stay meaningful, the kernel runs at its own pace.
#if MSDF_PROFILE_QUERY >= 2
#define MSDF_QSTAMP(x) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long x = __builtin_readcyclecounter()
#else
#define MSDF_QSTAMP(x) const unsigned long long x = __builtin_readcyclecounter()
#endif
#endif
template <int N, bool OVERLAP>
__global__ void __launch_bounds__(WAVE, MSDF_EC_QUERY_WAVES_PER_SIMD)
k_ec_query(int nGlyphs, const int32_t *__restrict__ glyphContourOffsets, const int32_t *__restrict__ contourOffsets, const EdgeRec *__restrict__ recs,
           const int8_t *__restrict__ windings, const MsdfHipGlyph *__restrict__ glyphs, int width, int height, const float *__restrict__ src, float *__restrict__ out,
           uint8_t *__restrict__ stencilOut, MsdfHipConfig cfg, const EcGlyphParams *__restrict__ glyphParams, const EcCandidate *__restrict__ cands, unsigned seg,
           const int *__restrict__ offsets, int *__restrict__ counter, int itemsPerTicket, int slotCap, int slotOffset, EcQueryPolicy lpcMaxContours,
           unsigned *__restrict__ overflowOut, const int *__restrict__ order, int queryFlags) {   // queryFlags: 1 tickets dealt statically | 4 ... only the first, the others drawn from eight counters
    // overflowOut (single-shape host calls): the candidate-overflow count is mirrored next to the results, so that the host sees it with the
    // copy back instead of a k_ec_slow launch that does nothing in all but pathological cases (one launch less on a latency-bound path)
    if (overflowOut && blockIdx.x == 0 && threadIdx.x == 0)
        *overflowOut = reinterpret_cast<const unsigned *>(cands)[0];
    // (__restrict__ throughout, as in k_distance: the lane-per-candidate walk reads the records with scalar loads only if the kernel's
    // own stores -- corrected texels, stencil bytes, the work counter -- provably do not touch them)
    BatchView batch;
    batch.nGlyphs = nGlyphs, batch.glyphContourOffsets = glyphContourOffsets, batch.contourOffsets = contourOffsets, batch.recs = recs, batch.windings = windings;
    extern __shared__ double smemLds[];                             // combiner scratch ([maxContours][64] or [maxContours]) | slotCap PBSlots at slotOffset
    PBSlot *slotBuf = reinterpret_cast<PBSlot *>(smemLds+slotOffset);
    const unsigned *header = reinterpret_cast<const unsigned *>(cands);
    const size_t texelsPerGlyph = (size_t) width*height;
    // offsets[] is read-only here and the work counter (k_ec_scan keeps it in the same array) arrives as a pointer of its own: the glyph
    // lookup below is then 13 SCALAR loads instead of 13 vector-load round trips to L2 -- together with the atomic they were half of the
    // ~27 us an item took (55 k items of the DejaVu set over 3 072 resident wavefronts in 0.48 ms: the kernel is this latency chain).
    const int chunks = offsets[batch.nGlyphs], coopItems = offsets[2*batch.nGlyphs+1];
    const int tickets = chunks+(coopItems+itemsPerTicket-1)/itemsPerTicket;   // a ticket = one chunk, or itemsPerTicket consecutive cooperative items
    lpcMaxContours.lpcMaxEdges = offsets[2*batch.nGlyphs+3]&((1<<30)-1);   // as k_ec_scan decided for this launch
    if (offsets[2*batch.nGlyphs+3]&(1<<30))
        lpcMaxContours.gridSteps = 0;
#if defined(MSDF_PROFILE_QUERY)
    MSDF_QSTAMP(qStart);
    unsigned long long qAcc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, qItems[4] = { 0, 0, 0, 0 }, qLongest = 0, qLastWork = 0, qLongestWho = 0;
#endif
    // Which tickets a wavefront takes (round 6). They used to be drawn from ONE atomic counter: with the evaluation of every item ablated the launch still took
    // 0.20 of its 0.28 ms (variants/qab3, profiles/r06_ab_notes.md 12) -- ~17 000 device-scope atomics on one address (8 600 tickets + one failing draw per
    // workgroup) serialize at ~12 ns each, across eight XCDs. The list is ordered longest items first (k_ec_scan), so a STATIC deal does nearly as well as
    // the dynamic one without a single atomic: workgroup b of W takes tickets b, 2W-1-b, 2W+b, 4W-1-b, ... (a serpentine: whoever drew the longest items of
    // one round gets the shortest of the next); W is what the device holds of this kernel at once (msdf_capi.hip: queryResidentBlocks).
    const int dealt = (queryFlags&1) ? (int) gridDim.x : 0;
    for (int round = 0;; ++round) {
#if defined(MSDF_PROFILE_QUERY)
        MSDF_QSTAMP(q0);
#endif
        int ticket = 0;
        if (dealt && (queryFlags&4) && round > 0) {
            // (variant: the first ticket dealt, the others drawn from eight counters -- workgroup b from counter b % 8, which hands out the tickets = b (mod 8))
            const int q = (int) blockIdx.x&7;
            if (threadIdx.x == 0)
                ticket = atomicAdd(counter+(ecQueuesAt(batch.nGlyphs)-(2*(size_t) batch.nGlyphs+2))+16*q, 1);
            ticket = dealt+__builtin_amdgcn_readfirstlane(ticket)*8+q;
            if (ticket >= tickets)
                break;
        } else if (dealt) {
            if ((long long) round*dealt >= tickets)
                break;
            ticket = (round&1) ? (round+1)*dealt-1-(int) blockIdx.x : round*dealt+(int) blockIdx.x;
            if (ticket >= tickets)
                continue;
        } else {
            if (threadIdx.x == 0)
                ticket = atomicAdd(counter, 1);
            ticket = __builtin_amdgcn_readfirstlane(ticket);
            if (ticket >= tickets)
                break;
        }
#if defined(MSDF_PROFILE_QUERY)
        MSDF_QSTAMP(q1);
        qAcc[0] += q1-q0;
#endif
        const int *part = ticket < chunks ? offsets : offsets+batch.nGlyphs+1;   // heavy first: the lane-per-candidate chunks, then the cooperative items
        const int first = ticket < chunks ? ticket : (ticket-chunks)*itemsPerTicket;
        const int last = ticket < chunks ? first+1 : (first+itemsPerTicket < coopItems ? first+itemsPerTicket : coopItems);
        int lo = 0, hi = batch.nGlyphs-1;                           // the glyph g with part[g] <= first < part[g+1]
        while (lo < hi) {
            const int mid = (lo+hi)>>1;
            if (part[mid+1] > first)
                hi = mid;
            else
                lo = mid+1;
        }
        int pos = lo;
        MSDF_NOUNROLL
        for (int i = first; i < last; ++i) {
#if defined(MSDF_PROFILE_QUERY)
        MSDF_QSTAMP(q2a);
#endif
        while (part[pos+1] <= i)                                    // consecutive items: the same position or one of the next few
            ++pos;
        const int item = i-part[pos];
        const int g = order ? order[pos] : pos;                     // positions list the glyphs heaviest first (k_ec_scan)
#if defined(MSDF_PROFILE_QUERY)
        MSDF_QSTAMP(q2);
        qAcc[1] += q2-(i == first ? q1 : q2a);
#endif
        const unsigned count = header[1+g];
        const EcCandidate *segment = cands+ecHeaderRecords(batch.nGlyphs)+(size_t) g*seg;
        const int c0 = batch.glyphContourOffsets[g], C = batch.glyphContourOffsets[g+1]-c0;
        const int32_t *coff = batch.contourOffsets+c0;
        const MsdfHipGlyph gd = glyphs[g];
        EcParams p;
        p.t = loadXform(gd);
        p.minDeviationRatio = cfg.min_deviation_ratio;
        p.minImproveRatio = cfg.min_improve_ratio;
        p.mode = cfg.ec_mode, p.distanceCheck = cfg.ec_distance_check, p.overlap = OVERLAP, p.stageLimit = 0;
        const EcGlyphParams gp = glyphParams[g];
        p.hSpan = gp.hSpan, p.vSpan = gp.vSpan, p.dSpan = gp.dSpan, p.texelX = gp.texelX, p.texelY = gp.texelY;
        p.radiusH = gp.radiusH, p.radiusV = gp.radiusV, p.radiusD = gp.radiusD;
        SdfView sdf;
        sdf.px = src+(size_t) g*texelsPerGlyph*N;
        sdf.w = width, sdf.h = height, sdf.N = N, sdf.flip = gd.flip;
        const int nE = coff[C]-coff[0];
        WindingMasks wind;                                          // one vector load + two ballots per item instead of up to 4 C dependent byte loads
        wind.mem = batch.windings+c0;
        wind.pos = wind.neg = 0;
        if (OVERLAP) {
            const int w = (int) threadIdx.x < C ? (int) wind.mem[threadIdx.x] : 0;
            wind.pos = __ballot(w > 0), wind.neg = __ballot(w < 0);
        }
#if defined(MSDF_PROFILE_QUERY)
        MSDF_QSTAMP(q3);
        qAcc[2] += q3-q2;
        unsigned long long q4 = q3;
        const bool qChunk = ecQueryListedFirst(count, nE, C, lpcMaxContours);
#endif
        const int gridS = ecQueryGridSlices(count, nE, C, lpcMaxContours);
#if MSDF_QGRID_ABLATE == 3                                          // measurement only: ticket, lookup and glyph state of every item, nothing else
        if (count != 0xffffffffu)
            continue;
#endif
#if MSDF_QGRID_ABLATE == 7                                          // measurement only: the grid items alone (no cooperative, no lane-per-candidate items)
        if (!gridS)
            continue;
#endif
#if MSDF_QGRID_ABLATE == 8                                          // measurement only: without the grid items
        if (gridS)
            continue;
#endif
        if (gridS) {                                                // lanes = (candidate, slice): 64 / gridS candidates per item
            PsdfQueryGrid<OVERLAP, WindingMasks> query;
            query.rec = batch.recs+coff[0], query.coff = coff, query.windings = wind, query.C = C, query.res = smemLds+threadIdx.x;
            query.S = gridS, query.slice = (int) threadIdx.x&(gridS-1);
            const unsigned k = (unsigned) item*(unsigned) (WAVE/gridS)+threadIdx.x/(unsigned) gridS;
            if (k < count) {
                const EcCandidate cand = segment[k];
                const size_t texel = cand.texel;
                const int rem = (int) (texel-(size_t) g*texelsPerGlyph);
                const int yn = rem/width, x = rem%width;
                const int ys = gd.flip ? height-1-yn : yn;
                const bool artifact = ecEvaluateCandidate(sdf, p, x, ys, cand.t, (cand.dir&3)-1, ((cand.dir>>2)&3)-1, query);
                if (artifact && query.slice == 0) {
                    const float *in = sdf.native(x, yn);
                    const float m = medianf(in[0], in[1], in[2]);
                    float *px = out+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) N*x;
                    px[0] = m, px[1] = m, px[2] = m;
                    if (stencilOut)
                        stencilOut[stencilIndex(texel, yn, width, height, cfg.stencil_y_down)] |= (uint8_t) EC_ERROR;
                }
            }
        } else if (ecQueryLanePerCandidate(count, nE, C, lpcMaxContours)) {
            PsdfQuery<OVERLAP, WindingMasks, EdgesAllBatched> query;
            query.rec = batch.recs+coff[0], query.coff = coff, query.windings = wind, query.C = C, query.res = smemLds+threadIdx.x;
            const unsigned k = (unsigned) item*WAVE+threadIdx.x;
            if (k < count) {
                const EcCandidate cand = segment[k];
                const size_t texel = cand.texel;
                const int rem = (int) (texel-(size_t) g*texelsPerGlyph);
                const int yn = rem/width, x = rem%width;
                const int ys = gd.flip ? height-1-yn : yn;
                const bool artifact = ecEvaluateCandidate(sdf, p, x, ys, cand.t, (cand.dir&3)-1, ((cand.dir>>2)&3)-1, query);
                if (artifact) {
                    const float *in = sdf.native(x, yn);
                    const float m = medianf(in[0], in[1], in[2]);
                    float *px = out+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) N*x;
                    px[0] = m, px[1] = m, px[2] = m;
                    if (stencilOut)
                        stencilOut[stencilIndex(texel, yn, width, height, cfg.stencil_y_down)] |= (uint8_t) EC_ERROR;
                }
            }
        } else {
            const EcCandidate cand = segment[item];
            PsdfQueryCooperative<OVERLAP, WindingMasks> query;
            query.rec = batch.recs+coff[0], query.coff = coff, query.windings = wind, query.C = C, query.lane = threadIdx.x;
            query.res = smemLds;
            // larger glyphs: per-contour cross-lane merge instead of the slots. `merged` holds min(maxContours, slotCap) states: a glyph of
            // <= slotCap edges normally has no more contours than that, but EMPTY contours (valid input) do not count as edges -- such a
            // glyph takes the other path too.
            query.slots = nE <= slotCap && C <= slotCap ? slotBuf : NULL;
            query.merged = slotBuf+slotCap;
            query.cacheCap = OVERLAP ? (slotCap < slotOffset ? slotCap : slotOffset) : 0;   // (= mergedCap of the launch, msdf_capi.hip: launchEc)
            const size_t texel = cand.texel;
            const int rem = (int) (texel-(size_t) g*texelsPerGlyph);
            const int yn = rem/width, x = rem%width;
            const int ys = gd.flip ? height-1-yn : yn;
            const bool artifact = ecEvaluateCandidate(sdf, p, x, ys, cand.t, (cand.dir&3)-1, ((cand.dir>>2)&3)-1, query);
#if defined(MSDF_PROFILE_QUERY)
            {
                MSDF_QSTAMP(q4c);
                q4 = q4c;
            }
#endif
            if (artifact && threadIdx.x == 0) {
                const float *in = sdf.native(x, yn);
                const float m = medianf(in[0], in[1], in[2]);
                float *px = out+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) N*x;
                px[0] = m, px[1] = m, px[2] = m;
                if (stencilOut)
                    stencilOut[stencilIndex(texel, yn, width, height, cfg.stencil_y_down)] |= (uint8_t) EC_ERROR;
            }
        }
        waveSync();                                                 // the next item rewrites the LDS scratch
#if defined(MSDF_PROFILE_QUERY)
        {
            MSDF_QSTAMP(q5);
            if (qChunk)
                q4 = q5;                                            // (chunk: evaluation and stores are not told apart)
            qAcc[3] += q4-q3, qAcc[4] += q5-q4;
            qItems[qChunk ? 0 : 2] += 1, qItems[qChunk ? 1 : 3] += q5-q2a;
            qItems[2] += 0;
            if (!qChunk)
                qAcc[5] += (unsigned long long) ((nE+WAVE-1)/WAVE);
            if (q5-q2a > qLongest)
                qLongest = q5-q2a, qLongestWho = (unsigned long long) g<<1|(qChunk ? 1ull : 0ull);
            qLastWork = q5;
            if (threadIdx.x == 0) {                                 // histogram of item durations: < 16 k cycles, < 32 k, ... >= 512 k
                int bucket = 0;
                for (unsigned long long d = (q5-q2a)>>14; d && bucket < 6; d >>= 1)
                    ++bucket;
                atomicAdd(&gQueryDetail[10+(bucket > 5 ? 5 : bucket)], 1ull);
            }
        }
#endif
        }
    }
#if defined(MSDF_PROFILE_QUERY)
    {
        MSDF_QSTAMP(qEnd);
        if (threadIdx.x == 0) {
            atomicAdd(&gQueryProfile[0], 1ull), atomicAdd(&gQueryProfile[1], qEnd-qStart);
            if (qLastWork)
                atomicAdd(&gQueryProfile[2], 1ull), atomicMax(&gQueryProfile[16], qLastWork);
            atomicMin(&gQueryProfile[3], qStart), atomicMax(&gQueryProfile[4], qEnd);
            atomicAdd(&gQueryProfile[5], qItems[0]), atomicAdd(&gQueryProfile[6], qItems[1]), atomicAdd(&gQueryProfile[7], qItems[2]), atomicAdd(&gQueryProfile[8], qItems[3]);
            atomicAdd(&gQueryProfile[9], qAcc[0]), atomicAdd(&gQueryProfile[10], qAcc[1]), atomicAdd(&gQueryProfile[11], qAcc[2]), atomicAdd(&gQueryProfile[12], qAcc[3]);
            atomicAdd(&gQueryProfile[13], qAcc[4]), atomicMax(&gQueryProfile[14], qLongest), atomicAdd(&gQueryProfile[15], qAcc[5]);
            atomicMax(&gQueryProfile[17], (qLongest>>8)<<24|qLongestWho);   // longest item: cycles/256 | glyph<<1 | chunk
        }
    }
#endif
}

// Full per-texel pipeline incl. the exact shape-distance check (msdf_ec.hpp) for EVERY texel of the batch: used for the stage
// snapshots of the tests (overflowOnly == 0) and as the safety net when the candidate list overflowed (overflowOnly != 0).
template <int N, bool OVERLAP, bool GRES = false>
__global__ void __launch_bounds__(WAVE)
k_ec_slow(BatchView batch, const MsdfHipGlyph *glyphs, int width, int height, const float *src, float *out, uint8_t *stencilOut,
          MsdfHipConfig cfg, const EcCandidate *cands, unsigned seg, int overflowOnly, double *gres, size_t gresStride) {
    extern __shared__ double smemLds[];                             // [maxContours][64] combiner scratch (overlap only)
    double *smem = GRES ? gres+(size_t) blockIdx.x*gresStride : smemLds;
    const size_t texelsPerGlyph = (size_t) width*height;
    const size_t allTexels = texelsPerGlyph*batch.nGlyphs;
    const unsigned *header = reinterpret_cast<const unsigned *>(cands);
    if (overflowOnly && header[0] == 0)
        return;                                                     // the candidate segments held everything: k_ec_query did the job
    for (size_t i = (size_t) blockIdx.x*WAVE+threadIdx.x; i < allTexels; i += (size_t) gridDim.x*WAVE) {
        const size_t texel = i;
        const int g = (int) (texel/texelsPerGlyph);
        if (overflowOnly && header[1+g] <= seg)
            continue;                                               // only the glyphs whose segment overflowed are redone
        const int rem = (int) (texel%texelsPerGlyph);
        const int yn = rem/width, x = rem%width;
        const int c0 = batch.glyphContourOffsets[g], C = batch.glyphContourOffsets[g+1]-c0;
        const int32_t *coff = batch.contourOffsets+c0;
        const int e0 = coff[0], nE = coff[C]-e0;
        const EdgeRec *rec = batch.recs+e0;
        const MsdfHipGlyph gd = glyphs[g];
        EcParams p;
        p.t = loadXform(gd);
        p.minDeviationRatio = cfg.min_deviation_ratio;
        p.minImproveRatio = cfg.min_improve_ratio;
        p.mode = cfg.ec_mode, p.distanceCheck = cfg.ec_distance_check, p.overlap = OVERLAP, p.stageLimit = cfg.ec_stage_limit;
        ecDerive(p);
        SdfView sdf;
        sdf.px = src+(size_t) g*texelsPerGlyph*N;
        sdf.w = width, sdf.h = height, sdf.N = N, sdf.flip = gd.flip;
        PsdfQuery<OVERLAP> query;
        query.rec = rec, query.coff = coff, query.windings = batch.windings+c0, query.C = C, query.res = smem+threadIdx.x;
        const int st = ecTexelStencil(sdf, p, rec, nE, x, yn, &query);
        const float *in = sdf.native(x, yn);
        float v[N];
        for (int k = 0; k < N; ++k)
            v[k] = in[k];
        if ((st&EC_ERROR) && cfg.ec_stage_limit == 0) {
            const float m = medianf(v[0], v[1], v[2]);
            v[0] = m, v[1] = m, v[2] = m;
        }
        float *px = out+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) N*x;
        for (int k = 0; k < N; ++k)
            px[k] = v[k];
        if (stencilOut)
            stencilOut[stencilIndex(texel, yn, width, height, cfg.stencil_y_down)] = (uint8_t) st;
    }
}

// ------------------------------------------------------------------------------------------- distance sign correction

// distanceSignCorrection (core/rasterization.cpp:19-88); one wavefront = `span` horizontally adjacent 8x8 tiles of one tile row (they
// share the row lists; the host picks span so that the launch still has thousands of wavefronts); src -> out (distinct buffers).
// Phase 1 (lanes = (row, edge) tasks): intersections of every edge with the tile row's texel rows, appended per row in LDS (order is
// irrelevant: the fill test is a sum). Multi-channel fields also list the row above and the row below the tile row, so that the
// neighbour vote of an ambiguous texel (:69-88) finds all four neighbours' fill bits in LDS.
// Phase 2 (lanes = texels): fill bit = fill rule of the sum of directions left of the texel centre; flip the distances whose sign
// disagrees; texels whose median equals the zero value exactly take the reference's neighbour vote.
// Shapes with more than cap/3 edges (cap = list capacity per row, bounded by the host to keep the lists in LDS) run phase 1 in edge
// chunks and accumulate the five sums of every texel chunk by chunk (slower, only for such shapes).
// src: packed [g][h][w][N] native rows (unused when rasterizeOnly: the output is the fill bit itself, rasterization.cpp:8-16).
// dstPacked != 0: out is packed the same way (a further pass follows), else the caller's bitmap.
// LDS: [ROWS][cap] doubles (x) + [ROWS][cap] ints (direction) + ROWS counters, ROWS = 10.
enum { SIGN_ROWS = TILE+2 };

struct RowLists {
    double *x;      // [SIGN_ROWS][cap]
    int *dir;       // [SIGN_ROWS][cap]
    int *count;     // [SIGN_ROWS]
    int cap;
    __device__ int sum(int row, double px) const {                       // Scanline::sumIntersections, Scanline.cpp:98-104
        const double *rx = x+(size_t) row*cap;
        const int *rd = dir+(size_t) row*cap;
        const int cnt = count[row];
        int s = 0;
        for (int i = 0; i < cnt; ++i)
            s += px >= rx[i] ? rd[i] : 0;
        return s;
    }
};

// Lists of edges [eBegin, eEnd) for list rows rowLo..rowLo+rows-1 (list row rr <-> bitmap row ty*TILE+rr-1). All lanes take part.
__device__ inline void buildRowLists(const RowLists &L, const EdgeRec *rec, int eBegin, int eEnd, int rowLo, int rows, int ty, int height,
                                     const Xform &t, int lane) {
    waveSync();                                                          // earlier readers of the lists are done
    if (lane < SIGN_ROWS)
        L.count[lane] = 0;
    waveSync();
    for (int task = lane; task < rows*(eEnd-eBegin); task += WAVE) {
        const int k = task/rows, rr = rowLo+task-k*rows;
        const int e = eBegin+k;
        const int ys = ty*TILE+rr-1;
        if (ys < 0 || ys >= height)
            continue;
        const double y = (ys+.5)/t.sy-t.ty;                              // projection.unprojectY(y+.5), Projection.cpp:38-40
        if (!rowMayIntersect(rec[e], y))
            continue;
        double x[3];
        int dy[3];
        const int n = scanlineIntersections(rec[e], x, dy, y);
        if (n > 0) {
            const int slot = atomicAdd(&L.count[rr], n);
            for (int q = 0; q < 3; ++q)
                if (q < n) {
                    L.x[(size_t) rr*L.cap+slot+q] = x[q];
                    L.dir[(size_t) rr*L.cap+slot+q] = dy[q];
                }
        }
    }
    waveSync();
}

template <int N>
__global__ void __launch_bounds__(WAVE, 3)
k_sign_correction(BatchView batch, const MsdfHipGlyph *glyphs, int width, int height, int spansX, int span, int spansPerGlyph, int cap,
                  const float *src, float *out, int dstPacked, float zero, int fillRule, int rasterizeOnly) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const GlyphWork wk = decodeBlock(batch.nGlyphs, spansPerGlyph);
    if (!wk.valid)
        return;
    const int c0 = batch.glyphContourOffsets[wk.g], C = batch.glyphContourOffsets[wk.g+1]-c0;
    const int32_t *coff = batch.contourOffsets+c0;
    const int e0 = coff[0], nE = coff[C]-e0;
    const EdgeRec *rec = batch.recs+e0;
    const int lane = threadIdx.x;
    RowLists L;
    L.cap = cap;
    L.x = smem;
    L.dir = reinterpret_cast<int *>(L.x+(size_t) SIGN_ROWS*cap);
    L.count = L.dir+(size_t) SIGN_ROWS*cap;
    const MsdfHipGlyph gd = glyphs[wk.g];
    const Xform t = loadXform(gd);
    const int tx0 = (wk.tile%spansX)*span, ty = wk.tile/spansX;          // this wavefront: tiles tx0..tx0+span-1 of tile row ty
    const int rowLo = N == 1 ? 1 : 0, rows = N == 1 ? TILE : SIGN_ROWS;
    const bool chunked = 3*nE > cap;                                     // wave-uniform
    const int chunkE = chunked ? cap/3 : (nE > 0 ? nE : 1);

    const int lx = lane&(TILE-1), ly = lane>>3;
    const int ys = ty*TILE+ly;                                           // shape orientation (sdf.reorient, rasterization.cpp:39)
    const int yn = gd.flip ? height-1-ys : ys;
    const float *tile = src+(size_t) wk.g*height*width*N;
    const float twice = zero+zero;
    for (int s = 0; s < span; ++s) {                                     // one 8x8 tile per step
        if ((tx0+s)*TILE >= width)
            break;                                                       // wave-uniform
        const int x = (tx0+s)*TILE+lx;
        const bool valid = x < width && ys < height;
        const double px = (x+.5)/t.sx-t.tx;                              // projection.unprojectX(x+.5)
        const double pxl = (x-.5)/t.sx-t.tx, pxr = (x+1.5)/t.sx-t.tx;    // the horizontal neighbours' centres
        int sum = 0, sumL = 0, sumR = 0, sumU = 0, sumD = 0;
        for (int k = 0; k == 0 || (chunked && k < nE); k += chunkE) {     // a single pass unless the shape is chunked
            if (chunked || s == 0)
                buildRowLists(L, rec, k, k+chunkE < nE ? k+chunkE : nE, rowLo, rows, ty, height, t, lane);
            sum += L.sum(ly+1, px);
            if (N > 1 && chunked) {
                sumL += L.sum(ly+1, pxl), sumR += L.sum(ly+1, pxr);
                sumU += L.sum(ly, px), sumD += L.sum(ly+2, px);
            }
        }
        if (!valid)
            continue;
        const bool fill = interpretFillRule(sum, fillRule);
        float v[N];
        if (rasterizeOnly) {                                             // rasterize(), rasterization.cpp:8-16 (N == 1)
            for (int i = 0; i < N; ++i)
                v[i] = (float) fill;
        } else {
            const float *in = tile+((size_t) yn*width+x)*N;
            for (int i = 0; i < N; ++i)
                v[i] = in[i];
            if (N == 1) {                                                // :19-33
                if ((v[0] > zero) != fill)
                    v[0] = twice-v[0];
            } else {                                                     // :35-88
                const int match = signMatch(v, fill, zero);
                bool flipRgb = match < 0;
                if (match == 0) {                                        // ambiguous texel: neighbour vote (:69-86)
                    if (!chunked) {
                        sumL = L.sum(ly+1, pxl), sumR = L.sum(ly+1, pxr);
                        sumU = L.sum(ly, px), sumD = L.sum(ly+2, px);
                    }
                    int vote = 0;
                    if (x > 0)
                        vote += signMatch(tile+((size_t) yn*width+x-1)*N, interpretFillRule(sumL, fillRule), zero);
                    if (x < width-1)
                        vote += signMatch(tile+((size_t) yn*width+x+1)*N, interpretFillRule(sumR, fillRule), zero);
                    if (ys > 0)
                        vote += signMatch(tile+((size_t) (gd.flip ? height-1-(ys-1) : ys-1)*width+x)*N, interpretFillRule(sumU, fillRule), zero);
                    if (ys < height-1)
                        vote += signMatch(tile+((size_t) (gd.flip ? height-1-(ys+1) : ys+1)*width+x)*N, interpretFillRule(sumD, fillRule), zero);
                    flipRgb = vote < 0;
                }
                if (flipRgb)
                    for (int i = 0; i < (N < 3 ? N : 3); ++i)
                        v[i] = twice-v[i];
                if (N >= 4 && (v[N >= 4 ? 3 : 0] > zero) != fill)
                    v[N >= 4 ? 3 : 0] = twice-v[N >= 4 ? 3 : 0];
            }
        }
        float *px_ = dstPacked ? out+(((size_t) wk.g*height+yn)*width+x)*N
                               : out+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) N*x;
        for (int i = 0; i < N; ++i)
            px_[i] = v[i];
    }
}

// ------------------------------------------------------------------------------------------- shape preparation (row f3)

// Shape::normalize, lanes = OUTPUT edges of the whole batch (msdf_shapeprep.hpp: normalizeEdgeFlat). co: raw contour offsets, co1: offsets after
// normalize, nOut = co1[nContours]. cusp[c] (zeroed by the caller) is raised for a contour with a convergent junction: k_prep_normalize_cusps redoes it.
__global__ void k_prep_normalize_flat(EdgeArrays raw, const int32_t *co, const int32_t *co1, int nContours, int nOut, int doNormalize, EdgeArrays out, int32_t *cusp) {
    const int slot = blockIdx.x*blockDim.x+threadIdx.x;
    if (slot >= nOut)
        return;
    int lo = 0, hi = nContours-1;                                   // last contour c with co1[c] <= slot (skips empty contours)
    while (lo < hi) {
        const int mid = (lo+hi+1)>>1;
        if (co1[mid] <= slot)
            lo = mid;
        else
            hi = mid-1;
    }
    const int ib = co[lo], ob = co1[lo];
    if (normalizeEdgeFlat(raw, ib, co[lo+1]-ib, out, ob, slot-ob, doNormalize != 0))
        cusp[lo] = 1;                                               // (every writer stores the same value)
}

// The sequential cusp repair (Shape.cpp:74-90) of the contours the pass above flagged -- none in any font we have; one thread per contour.
__global__ void k_prep_normalize_cusps(EdgeArrays raw, const int32_t *co, const int32_t *co1, int nContours, EdgeArrays out, const int32_t *cusp) {
    const int c = blockIdx.x*blockDim.x+threadIdx.x;
    if (c < nContours && cusp[c])
        normalizeContour(raw, co[c], co[c+1]-co[c], out, co1[c]);
}

// Edges per contour after edgeColoringSimple (a one-corner contour with fewer than three edges is split), one thread per contour.
__global__ void k_prep_count(EdgeArrays norm, const int32_t *co1, int nContours, double crossThreshold, int32_t *count) {
    const int c = blockIdx.x*blockDim.x+threadIdx.x;
    if (c < nContours)
        count[c] = colouredCount(norm, co1[c], co1[c+1]-co1[c], crossThreshold);
}

// co2 = exclusive prefix of the counts above (co2[nContours] = total): one workgroup, a contiguous chunk per thread -- the offsets stay on the
// device between the passes (until round 4 the host took the counts back in the middle of the call and returned the prefix).
__global__ void __launch_bounds__(256) k_prep_offsets(const int32_t *count, int nContours, int32_t *co2) {
    __shared__ int sums[256];
    const int t = threadIdx.x, chunk = (nContours+255)/256;
    const int lo = t*chunk < nContours ? t*chunk : nContours, hi = lo+chunk < nContours ? lo+chunk : nContours;
    int sum = 0;
    for (int i = lo; i < hi; ++i)
        sum += count[i];
    sums[t] = sum;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int k = 0; k < 256; ++k) {
            const int v = sums[k];
            sums[k] = run;
            run += v;
        }
        co2[nContours] = run;
    }
    __syncthreads();
    int run = sums[t];
    for (int i = lo; i < hi; ++i) {
        co2[i] = run;
        run += count[i];
    }
}

// edgeColoringSimple / edgeColoringInkTrap, one wavefront per glyph, lanes = edges / corners (colourContourWave): the colour / seed state runs
// through the glyph's contours (edge-coloring.cpp:68-72, :151-155) as wave-uniform values. Contours of up to PREP_WAVE_MAX_EDGES edges keep
// their tables in LDS; longer ones in `big` (global, indexed like the edges; allocated only when the batch has such a contour).
// LDS_EDGES: the tier of the launch -- PREP_WAVE_SMALL_EDGES when no contour of the batch is longer (the usual case: font contours have tens of
// edges; the ink-trap tables then take 5.4 KB per one-wavefront workgroup instead of 45 KB, which had capped a CU at 3 wavefronts), else
// PREP_WAVE_MAX_EDGES. The host picks it from the longest contour (msdfhip_batch_create_prepared).
enum { PREP_WAVE_MAX_EDGES = 2048, PREP_WAVE_SMALL_EDGES = 256 };
template <bool INKTRAP, int LDS_EDGES = PREP_WAVE_MAX_EDGES>
__global__ void __launch_bounds__(WAVE)
k_prep_colour_wave(EdgeArrays norm, const int32_t *gco, const int32_t *co1, const int32_t *co2, int nGlyphs, double crossThreshold,
                   const unsigned long long *seeds, unsigned long long seedAll, EdgeArrays out, ColourTables big) {
    static_assert(LDS_EDGES%WAVE == 0, "one ballot word per 64 edges");
    enum { INK_EDGES = INKTRAP ? (int) LDS_EDGES : 1 };
    __shared__ unsigned long long cornerMask[LDS_EDGES/WAVE];
    __shared__ unsigned char splineColor[LDS_EDGES];
    __shared__ double edgeLength[INK_EDGES], cornerLength[INK_EDGES];
    __shared__ int cornerIndex[INK_EDGES];
    __shared__ unsigned char minor[INK_EDGES];
    const int g = blockIdx.x;
    if (g >= nGlyphs)
        return;
    WaveCtx ctx;
    ctx.lane = threadIdx.x;
    unsigned long long seed = seeds ? seeds[g] : seedAll;          // (wave-uniform state: every lane advances its own copy identically)
    int color = initColor(seed);
    for (int c = gco[g]; c < gco[g+1]; ++c) {
        const int ib = co1[c], n = co1[c+1]-ib, ob = co2[c];
        if (n <= LDS_EDGES) {
            ColourTables lds;
            lds.cornerMask = cornerMask, lds.splineColor = splineColor, lds.edgeLength = edgeLength, lds.cornerLength = cornerLength;
            lds.cornerIndex = cornerIndex, lds.minor = minor;
            colourContourWave<INKTRAP>(ctx, lds, norm, ib, n, out, ob, crossThreshold, color, seed);
        } else {
            ColourTables t;                                         // (ib/64 + c: the ballot words of consecutive contours never overlap)
            t.cornerMask = big.cornerMask+ib/WAVE+c, t.splineColor = big.splineColor+ib, t.edgeLength = big.edgeLength+ib;
            t.cornerLength = big.cornerLength+ib, t.cornerIndex = big.cornerIndex+ib, t.minor = big.minor+ib;
            colourContourWave<INKTRAP>(ctx, t, norm, ib, n, out, ob, crossThreshold, color, seed);
        }
    }
}

// ------------------------------------------------------------------------------------------- 8-bit atlas output (row f2)

// pixelFloatToByte (core/pixel-conversion.hpp:8-10): byte(~int(255.5f-255.f*clamp(x))), fp32 arithmetic, clamp(NaN) = 0.
__device__ inline uint8_t pixelFloatToByte(float x) {
    const float c = x >= 0.f && x <= 1.f ? x : (float) (x > 0.f);            // arithmetics.hpp:35-37
    return (uint8_t) ~(int) (255.5f-255.f*c);
}

// tiles: packed fp32 [g][h][w][N] (what batch_generate writes with default descriptors); every texel is converted and stored at
// atlas + out_offset[g] + row_stride[g]*y + N*x (bytes): conversion and atlas-rectangle blit in one pass, D2H shrinks 4x.
template <int N>
__global__ void k_tiles_to_bytes(const float *tiles, const MsdfHipGlyph *glyphs, int nGlyphs, int width, int height, uint8_t *atlas) {
    const size_t texelsPerGlyph = (size_t) width*height;
    const size_t total = texelsPerGlyph*nGlyphs;
    for (size_t i = (size_t) blockIdx.x*blockDim.x+threadIdx.x; i < total; i += (size_t) gridDim.x*blockDim.x) {
        const int g = (int) (i/texelsPerGlyph);
        const int rem = (int) (i-(size_t) g*texelsPerGlyph);
        const int y = rem/width, x = rem-y*width;
        const float *in = tiles+i*N;
        uint8_t *o = atlas+glyphs[g].out_offset+(ptrdiff_t) glyphs[g].row_stride*y+(ptrdiff_t) N*x;
        for (int ch = 0; ch < N; ++ch)
            o[ch] = pixelFloatToByte(in[ch]);
    }
}

// ------------------------------------------------------------------------------------------- renderSDF / simulate8bit (row f4)

// renderSDF (core/render-sdf.cpp:14-170) of G tiles: out[g][oh][ow][NO] from sdf[g][sh][sw][NS]; one thread per output texel.
// threshold != 0: sdfPxRange.lower == upper (binary output); else mapScale/mapTranslate = DistanceMapping::inverse(range scaled to
// the output size), computed by the host in fp64 exactly as render-sdf.cpp:24-25. Rows are memory rows (the reference does not reorient).
template <int NO, int NS>
__global__ void k_render_sdf(const float *sdf, int nGlyphs, int sw, int sh, float *out, int ow, int oh, double scaleX, double scaleY, int threshold,
                             double mapScale, double mapTranslate, float sdThreshold, float sdBias) {
    const size_t perGlyph = (size_t) ow*oh, total = perGlyph*nGlyphs;
    for (size_t i = (size_t) blockIdx.x*blockDim.x+threadIdx.x; i < total; i += (size_t) gridDim.x*blockDim.x) {
        const int g = (int) (i/perGlyph);
        const int rem = (int) (i-(size_t) g*perGlyph);
        const int y = rem/ow, x = rem-y*ow;
        const float *px = sdf+(size_t) g*sw*sh*NS;
        V2 pos = mk(scaleX*(x+.5), scaleY*(y+.5));                  // scale*Point2(x+.5, y+.5)
        pos.x = clampd(pos.x, (double) sw);                         // interpolate, bitmap-interpolation.hpp:10-25
        pos.y = clampd(pos.y, (double) sh);
        pos.x -= .5, pos.y -= .5;
        int l = (int) floor(pos.x), b = (int) floor(pos.y);
        int r = l+1, t = b+1;
        const double lr = pos.x-l, bt = pos.y-b;
        l = clampi(l, sw-1), r = clampi(r, sw-1);
        b = clampi(b, sh-1), t = clampi(t, sh-1);
        float sd[NS];
        for (int c = 0; c < NS; ++c)
            sd[c] = mixf(mixf(px[((size_t) b*sw+l)*NS+c], px[((size_t) b*sw+r)*NS+c], lr), mixf(px[((size_t) t*sw+l)*NS+c], px[((size_t) t*sw+r)*NS+c], lr), bt);
        float in[NO];
        if (NO == 1 && NS >= 3)
            in[0] = medianf(sd[0], sd[NS >= 3 ? 1 : 0], sd[NS >= 3 ? 2 : 0]);
        else
            for (int c = 0; c < NO; ++c)
                in[c] = sd[c < NS ? c : 0];
        float *o = out+i*NO;
        for (int c = 0; c < NO; ++c) {
            if (threshold)
                o[c] = (float) (in[c] >= sdThreshold);
            else {                                                  // distVal, render-sdf.cpp:10-12
                const double v = mapScale*((double) (in[c]+sdBias)+mapTranslate)+.5;
                o[c] = (float) (v >= 0 && v <= 1 ? v : (double) (v > 0));
            }
        }
    }
}

// simulate8bit (core/render-sdf.cpp:172-188): p = pixelByteToFloat(pixelFloatToByte(p)).
__global__ void k_simulate_8bit(float *px, size_t n) {
    for (size_t i = (size_t) blockIdx.x*blockDim.x+threadIdx.x; i < n; i += (size_t) gridDim.x*blockDim.x)
        px[i] = 1.f/255.f*(float) pixelFloatToByte(px[i]);
}

// estimateSDFError (core/sdf-error-estimation.cpp:134-154), one lane per scanline (glyph, row, subRow). items: scanlines of this
// launch chunk [itemBase, itemBase+nItems) in (glyph, row, subRow) order; lists: workspace, lane-strided (msdf_scanline.hpp).
// lines[item] = 1 - overlapFactor*overlap. glyphs[g].flip is read as "the shape's Y axis points down" (shape.getYAxisOrientation()).
template <int N>
__global__ void k_sdf_error_lines(BatchView batch, const MsdfHipGlyph *glyphs, const float *tiles, int width, int height, int scanlinesPerRow, int fillRule,
                                  size_t itemBase, size_t nItems, double *listX, int *listDir, int refCap, double *lines) {
    const size_t local = (size_t) blockIdx.x*blockDim.x+threadIdx.x;
    if (local >= nItems)
        return;
    const size_t item = itemBase+local;
    const size_t perGlyph = (size_t) (height-1)*scanlinesPerRow;
    const int g = (int) (item/perGlyph);
    const int rem = (int) (item-(size_t) g*perGlyph);
    const int row = rem/scanlinesPerRow, subRow = rem-row*scanlinesPerRow;
    const int c0 = batch.glyphContourOffsets[g], C = batch.glyphContourOffsets[g+1]-c0;
    const int32_t *coff = batch.contourOffsets+c0;
    const int e0 = coff[0], nE = coff[C]-e0;
    const MsdfHipGlyph gd = glyphs[g];
    StridedList refList, sdfList;
    refList.x = listX+local, refList.dir = listDir+local, refList.stride = nItems, refList.n = 0;
    sdfList.x = listX+(size_t) refCap*nItems+local, sdfList.dir = listDir+(size_t) refCap*nItems+local, sdfList.stride = nItems, sdfList.n = 0;
    lines[item] = sdfErrorOfLine<N>(batch.recs+e0, nE, tiles+(size_t) g*width*height*N, width, height, gd.xf[0], gd.xf[1], gd.xf[2], gd.xf[3], gd.flip != 0,
                                    row, subRow, scanlinesPerRow, fillRule, refList, sdfList);
}

// The reference sums the scanlines of a glyph in (row, subRow) order and divides (:141-153): one thread per glyph, same order.
__global__ void k_sdf_error_sum(const double *lines, int nGlyphs, int height, int scanlinesPerRow, double *errors) {
    const int g = blockIdx.x*blockDim.x+threadIdx.x;
    if (g >= nGlyphs)
        return;
    const size_t perGlyph = (size_t) (height-1)*scanlinesPerRow;
    double error = 0;
    for (size_t i = 0; i < perGlyph; ++i)
        error += lines[(size_t) g*perGlyph+i];
    errors[g] = error/((height-1)*scanlinesPerRow);
}

// ------------------------------------------------------------------------------------------------- distance queries

// The form that NEVER refuses a shape (core/msdfgen.cpp:78-106 returns void for any Shape): no survivor lists, no LDS at all. A glyph whose
// lists would not fit a CU's LDS (> ~40 000 edges or > ~10 000 contours; k_distance's phase 1 needs one list entry per edge) is rendered
// by walking ALL its edges per tile, straight from the CSR offsets, with the per-texel relevance vote as the only cull and the combiner's
// per-contour distances in a global workspace slice per workgroup. Persistent: a bounded pool of workgroups (the workspace is
// contours x 1.5 KB per workgroup) draws tiles from one counter. Orders of magnitude slower per edge than the culled path -- it exists so
// that a valid input is rendered instead of returning MSDFHIP_ERR_TOO_COMPLEX.
template <int SEL, bool OVERLAP>
__global__ void __launch_bounds__(WAVE, 2)
k_distance_unculled(int nGlyphs, const int32_t *__restrict__ glyphContourOffsets, const int32_t *__restrict__ contourOffsets, const EdgeRec *__restrict__ recs,
                    const int8_t *__restrict__ windings, const MsdfHipGlyph *__restrict__ glyphs, int width, int height, int tilesX, int tilesPerGlyph,
                    float *__restrict__ dst, int toScratch, double *__restrict__ gres, size_t gresStride, unsigned *__restrict__ counter, unsigned items,
                    const int *__restrict__ glyphMap) {
    // glyphMap: the launch covers these glyphs of the batch (the oversized ones of a mixed batch: msdf_capi.hip, ensureBuckets), or NULL = all
    enum { NCH = SelTraits<SEL>::NCH };
    const int lane = threadIdx.x;
    for (;;) {
        unsigned item = 0;
        if (lane == 0)
            item = atomicAdd(counter, 1u);
        item = (unsigned) __builtin_amdgcn_readfirstlane((int) item);
        if (item >= items)
            return;
        const int slot = (int) (item/(unsigned) tilesPerGlyph), tile = (int) (item%(unsigned) tilesPerGlyph);
        const int g = glyphMap ? glyphMap[slot] : slot;
        const int c0 = glyphContourOffsets[g], C = glyphContourOffsets[g+1]-c0;
        const int32_t *coff = contourOffsets+c0;
        const EdgeRec *rec = recs+coff[0];
        const MsdfHipGlyph gd = glyphs[g];
        const Xform t = loadXform(gd);
        const int tx = tile%tilesX, ty = tile/tilesX;
        const int x = tx*TILE+(lane&(TILE-1)), y = ty*TILE+(lane>>3);
        if (x >= width || y >= height)
            continue;
        const V2 p = unproject(t, mk(x+.5, y+.5));                   // msdfgen.cpp:68
        EdgesAll edges;
        edges.coff = coff;
        double d[NCH];
        if (OVERLAP)
            shapeDistanceOverlap<SEL>(rec, edges, windings+c0, C, p, gres+(size_t) blockIdx.x*gresStride+lane, WAVE, d);
        else
            shapeDistanceSimple<SEL>(rec, edges, C, p, d);
        const int yn = gd.flip ? height-1-y : y;
        float *px = toScratch ? dst+(((size_t) g*height+yn)*width+x)*NCH : dst+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) NCH*x;
        for (int ch = 0; ch < NCH; ++ch)
            px[ch] = mapDistance(t, d[ch]);
    }
}

// res: the overlapping combiner's per-contour scratch -- LDS (dynamic shared memory sized by the host) or, for shapes with more contours
// than a CU's LDS holds, a global workspace slice per workgroup (gres != NULL).
template <int SEL, bool OVERLAP>
__global__ void __launch_bounds__(WAVE)
k_shape_distance(BatchView batch, int nPoints, const double *pts, double *out, double *gres, size_t gresStride) {
    enum { NCH = SelTraits<SEL>::NCH };
    extern __shared__ double smemSd[];
    double *smem = gres ? gres+(size_t) blockIdx.x*gresStride : smemSd;
    const int i = blockIdx.x*WAVE+threadIdx.x;
    if (i >= nPoints)
        return;
    const int C = batch.glyphContourOffsets[1]-batch.glyphContourOffsets[0];
    double d[4] = { 0, 0, 0, 0 };
    const V2 p = mk(pts[2*i], pts[2*i+1]);
    EdgesAll edges;
    edges.coff = batch.contourOffsets;
    if (OVERLAP)
        shapeDistanceOverlap<SEL>(batch.recs, edges, batch.windings, C, p, smem+threadIdx.x, WAVE, d);
    else
        shapeDistanceSimple<SEL>(batch.recs, edges, C, p, d);
    for (int ch = 0; ch < 4; ++ch)
        out[4*(size_t) i+ch] = ch < NCH ? d[ch] : 0.;
}

} // namespace msdfhip
