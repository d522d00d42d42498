// msdf_single.hpp -- ONE launch per single-shape call (the literal drop-in: one generateSDF / PSDF / MSDF / MTSDF of one Shape into one bitmap,
// core/msdfgen.cpp:78-106 -- for the multi-channel types incl. msdfErrorCorrection, :92-98).
//
// The batched path spends 7 dependent launches on such a call (k_prep_records, k_distance, k_ec_params, k_ec_fast, k_ec_scan, k_ec_query, copies):
// ~60 us of launch / dependency latency around ~60 us of kernels, 116-128 us per call, 8.3 k glyphs/s from one caller thread. Here the same
// device code (distanceBody, ecParamsBody, ecFastBody and the cooperative distance checks -- the bodies of those kernels, msdf_kernels.hpp) runs as
// the PHASES of one launch of tiles+1 single-wavefront workgroups, separated by grid barriers:
//
//   phase 0  digest: raw CSR edges -> EdgeRec records + contour windings (threads = edges / contours over the whole grid)
//   phase 1  workgroup t < tiles: distance field of tile t (one tile per wavefront, the latency-shaped form of k_distance: combiner scratch
//            in a global slice); workgroup `tiles`: the glyph's error-correction constants and corner texels (k_ec_params' job)
//   phase 2  workgroup t: error correction of tile t (k_ec_fast's body: halo from the pre-correction field, corrected texels + stencil out,
//            distance-check candidates appended)
//   phase 3  the candidates, one per ticket, lanes = edges (k_ec_query's cooperative form)
//
// All workgroups of the launch are resident at once (<= 257 wavefronts on 1 024 SIMDs), so a barrier is an atomic counter in device memory:
// arrive with an agent-scope release (L2 write-back: the workgroups sit on different XCDs), spin, agent-scope acquire. The survivor walk of
// phase 1 reads the records written in phase 0 with hand-placed SCALAR loads, which the compiler's memory model does not cover (it never
// emits scalar loads of memory the kernel itself writes): the scalar data cache is invalidated explicitly after every barrier. The counter
// is never reset -- each call waits for values above the ones the previous call on the same arena left behind (barrierBase).
#pragma once

#include "msdf_kernels.hpp"

namespace msdfhip {

struct SingleArgs {
    // the shape: CSR arrays of ONE glyph (glyphContourOffsets = { 0, nContours })
    const int32_t *glyphContourOffsets, *contourOffsets;
    const double *points;
    const uint8_t *types, *colors;
    int nContours, nEdges;
    const MsdfHipGlyph *glyph;
    // derived on the device
    EdgeRec *recs;
    int8_t *windings;
    // the bitmap
    int width, height, tilesX, tiles, listStride;
    float *scratch;                    // pre-correction field [h][w][N] (only when the correction pass runs)
    float *out;                        // the caller's tile (packed rows)
    uint8_t *stencil;
    double *gres;                      // combiner scratch, one slice per tile
    size_t gresStride;
    // error correction
    MsdfHipConfig cfg;
    int correct;
    EcGlyphParams *ecParams;
    EcCandidate *cands;
    unsigned seg;
    int *corners, *sizes, *ticket;
    int slotCap, slotOffset;
    // launch bookkeeping
    unsigned *barrier;
    unsigned barrierBase;
    unsigned *status;                  // [0] = candidate overflow (the host then reruns the call through the batched path)
};

// (The spin is bounded -- ~0.3 s -- so that a launch whose workgroups could not all become resident ends with status[1] set instead of hanging
// the queue; the host then fails the call loudly. It has never been seen to happen: <= 257 wavefronts on a device with 3 072+ slots.)
__device__ inline void gridBarrier(unsigned *counter, unsigned target, unsigned *status) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");              // this wavefront's stores (all lanes) are written back before it arrives
    __builtin_amdgcn_wave_barrier();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int) (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)-target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u<<22)) {
                status[1] = 1u;
                break;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // the hand-placed s_load batches of the survivor walk (see above)
}

template <int SEL, bool OVERLAP>
__global__ void __launch_bounds__(WAVE)
k_single_call(SingleArgs a) {
    enum { N = SelTraits<SEL>::NCH };
    extern __shared__ __attribute__((aligned(16))) double smemSingle[];
    const int lane = threadIdx.x;
    const unsigned blk = blockIdx.x, groups = gridDim.x;
    const unsigned T = (unsigned) a.tiles;

    if (blk == 0 && lane == 0)
        a.status[0] = 0, a.status[1] = 0;                           // (phase 3 overwrites [0] with the candidate-overflow flag; [1]: a barrier gave up)
    // ---- phase 0: digest (k_prep_records)
    {
        const int total = (int) groups*WAVE, tid = (int) blk*WAVE+lane;
        for (int slot = tid; slot < a.nEdges; slot += total) {
            int lo = 0, hi = a.nContours-1;                         // last contour c with contourOffsets[c] <= slot (skips empty contours)
            while (lo < hi) {
                const int mid = (lo+hi+1)>>1;
                if (a.contourOffsets[mid] <= slot)
                    lo = mid;
                else
                    hi = mid-1;
            }
            prepRecord(a.recs, slot, lo, a.contourOffsets, a.points, a.types, a.colors);
        }
        for (int c = tid; c < a.nContours; c += total)
            a.windings[c] = (int8_t) contourWinding(c, a.contourOffsets, a.points, a.types, a.colors);
    }
    gridBarrier(a.barrier, a.barrierBase+groups, a.status);

    BatchView batch;
    batch.nGlyphs = 1, batch.glyphContourOffsets = a.glyphContourOffsets, batch.contourOffsets = a.contourOffsets, batch.recs = a.recs, batch.windings = a.windings;

    // ---- phase 1: distance field, one tile per workgroup; the extra workgroup prepares the correction pass
    if (blk < T)
        distanceBody<SEL, OVERLAP, true>(1, a.glyphContourOffsets, a.contourOffsets, a.recs, a.windings, a.glyph, a.width, a.height, a.tilesX, a.tiles, a.listStride,
                                         a.correct ? a.scratch : a.out, a.correct, 0u, a.gres, a.gresStride, (const int *) NULL, 0, (unsigned *) NULL, 0u, blk, smemSingle);
    else if constexpr (SEL >= 3) {
        ecParamsBody(a.ecParams, batch, a.glyph, a.cfg, reinterpret_cast<unsigned *>(a.cands), a.corners, a.sizes, 0, lane);
        if (lane == 0)
            a.ticket[0] = 0;
    }
    if (SEL < 3 || !a.correct)
        return;
    gridBarrier(a.barrier, a.barrierBase+2u*groups, a.status);

    // ---- phase 2: error correction sweep (k_ec_fast)
    if constexpr (SEL >= 3) {
        if (blk < T)
            ecFastBody<(int) N>(batch, a.glyph, a.width, a.height, a.tilesX, a.tiles, a.scratch, a.out, a.stencil, a.cfg, a.ecParams, a.cands, a.seg,
                                                 a.listStride, a.corners, blk, reinterpret_cast<int *>(smemSingle));
        gridBarrier(a.barrier, a.barrierBase+3u*groups, a.status);

        // ---- phase 3: deferred distance checks, cooperative (k_ec_query): a ticket = one candidate
        const unsigned *header = reinterpret_cast<const unsigned *>(a.cands);
        const unsigned count = header[1];
        if (blk == 0 && lane == 0)
            a.status[0] = count > a.seg ? 1u : header[0];
        if (count == 0 || count > a.seg)
            return;
        const EcCandidate *segment = a.cands+ecHeaderRecords(1);
        const MsdfHipGlyph gd = a.glyph[0];
        EcParams p;
        p.t = loadXform(gd);
        p.minDeviationRatio = a.cfg.min_deviation_ratio;
        p.minImproveRatio = a.cfg.min_improve_ratio;
        p.mode = a.cfg.ec_mode, p.distanceCheck = a.cfg.ec_distance_check, p.overlap = OVERLAP, p.stageLimit = 0;
        const EcGlyphParams gp = a.ecParams[0];
        p.hSpan = gp.hSpan, p.vSpan = gp.vSpan, p.dSpan = gp.dSpan, p.texelX = gp.texelX, p.texelY = gp.texelY;
        p.radiusH = gp.radiusH, p.radiusV = gp.radiusV, p.radiusD = gp.radiusD;
        SdfView sdf;
        sdf.px = a.scratch;
        sdf.w = a.width, sdf.h = a.height, sdf.N = N, sdf.flip = gd.flip;
        const int C = a.nContours;
        WindingMasks wind;
        wind.mem = a.windings;
        wind.pos = wind.neg = 0;
        if (OVERLAP) {
            const int w = lane < C ? (int) wind.mem[lane] : 0;
            wind.pos = __ballot(w > 0), wind.neg = __ballot(w < 0);
        }
        PBSlot *slotBuf = reinterpret_cast<PBSlot *>(smemSingle+a.slotOffset);
        for (;;) {
            int t = 0;
            if (lane == 0)
                t = atomicAdd(a.ticket, 1);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t >= (int) count)
                break;
            const EcCandidate cand = segment[t];
            PsdfQueryCooperative<OVERLAP, WindingMasks> query;
            query.rec = a.recs, query.coff = a.contourOffsets, query.windings = wind, query.C = C, query.lane = lane;
            query.res = smemSingle;
            query.slots = a.nEdges <= a.slotCap && C <= a.slotCap ? slotBuf : NULL;
            query.merged = slotBuf+a.slotCap;
            const int rem = (int) cand.texel;
            const int yn = rem/a.width, x = rem%a.width;
            const int ys = gd.flip ? a.height-1-yn : yn;
            const bool artifact = ecEvaluateCandidate(sdf, p, x, ys, cand.t, (cand.dir&3)-1, ((cand.dir>>2)&3)-1, query);
            if (artifact && lane == 0) {
                const float *in = sdf.native(x, yn);
                const float m = medianf(in[0], in[1], in[2]);
                float *px = a.out+gd.out_offset+(ptrdiff_t) gd.row_stride*yn+(ptrdiff_t) N*x;
                px[0] = m, px[1] = m, px[2] = m;
                if (a.stencil)
                    a.stencil[stencilIndex((size_t) rem, yn, a.width, a.height, a.cfg.stencil_y_down)] |= (uint8_t) EC_ERROR;
            }
            waveSync();                                             // the next candidate rewrites the LDS scratch
        }
    }
}

} // namespace msdfhip
