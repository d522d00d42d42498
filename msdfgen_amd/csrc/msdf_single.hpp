// msdf_single.hpp -- ONE launch per single-shape call (the literal drop-in: one generateSDF / PSDF / MSDF / MTSDF of one Shape into one bitmap,
// core/msdfgen.cpp:78-106 -- for the multi-channel types incl. msdfErrorCorrection, :92-98).
//
// The batched path spends 7 dependent launches on such a call (k_prep_records, k_distance, k_ec_params, k_ec_fast, k_ec_scan, k_ec_query, copies):
// ~60 us of launch / dependency latency around ~60 us of kernels, 116-133 us per call, 7.5-8.3 k glyphs/s from one caller thread. Here the same
// device code (distanceBody, ecParamsBody, ecFastBody and the cooperative distance checks -- the bodies of those kernels, msdf_kernels.hpp) runs as
// the PHASES of one launch of tiles+1 single-wavefront workgroups:
//
//   digest    EVERY workgroup digests the whole shape (raw CSR edges -> EdgeRec records + contour windings, lanes = edges) into an area of its
//             own: a 25-edge glyph is one round of lanes, and what the workgroup reads later it wrote itself -- no grid barrier, and the
//             records are warm in its XCD's L2. Small shapes arrive INSIDE the kernel arguments (device memory, written by the host through the
//             PCIe aperture), larger ones are read from the caller's pinned staging area (two dependent PCIe round trips).
//   distance  workgroup t < tiles: distance field of tile t (k_distance's body, one tile per wavefront, the scalar cache warmed with the
//             survivors' records first; the overlapping combiner's scratch in LDS where it fits); workgroup `tiles`: the glyph's
//             error-correction constants and corner texels (k_ec_params' job)
//   -- grid barrier (the correction sweep reads the neighbouring tiles' pre-correction texels) --
//   sweep     workgroup t: error correction of tile t (k_ec_fast's body: halo from the pre-correction field, corrected texels + stencil out),
//             then the distance checks of the candidates the tile left, one at a time with lanes = edges (k_ec_query's cooperative form): a
//             candidate only changes its own texel and reads the pre-correction field, so no work list and no further barrier are needed
//   done      results go straight to pinned host memory (posted writes); the workgroup that finishes last raises a flag there, which the host polls
//
// All workgroups of the launch are resident at once (<= 257 wavefronts on 1 024 SIMDs), so the barrier is an atomic counter in device memory:
// arrive with an agent-scope release (L2 write-back: the workgroups sit on different XCDs), spin, agent-scope acquire. The survivor walk reads
// the records written in the digest with hand-placed SCALAR loads, which the compiler's memory model does not cover (it never emits scalar loads of
// memory the kernel itself writes): the scalar data cache is invalidated explicitly after the digest and after the barrier. The counters are
// never reset -- each call waits for values above the ones the previous call on the same arena left behind (barrierBase / doneBase).
// Measured (profiles/r04_host_calls.jsonl): 127-135 -> 82-88 us per 64x64 generateMSDF from one host thread, 68-75 us of it inside the launch
// (digest 8, slowest tile's distance field 33, barrier + slowest sweep 25).
#pragma once

#include "msdf_kernels.hpp"

namespace msdfhip {

#ifndef MSDF_SINGLE_TEAM
#define MSDF_SINGLE_TEAM 4             // wavefronts per tile in the distance phase of k_single_call (msdf_kernels.hpp: TeamExchange); 1 = round 5's single wavefront
#endif
enum { SINGLE_TILE_SEGMENT = 64, SINGLE_TEAM = MSDF_SINGLE_TEAM };     // distance-check candidates a tile may leave (default config: ~0.1 per tile; beyond, the call is rerun through the batched path)

struct SingleArgs {
    // the shape as the CALLER staged it: CSR arrays of ONE glyph, read once in phase 0 -- device memory or pinned host memory (zero copy:
    // a few KB over PCIe, two dependent round trips, instead of an upload launch in front of this one)
    const int32_t *srcContourOffsets;
    const double *points;
    const uint8_t *types, *colors;
    const MsdfHipGlyph *srcGlyph;
    int nContours, nEdges;
    // Every workgroup digests the WHOLE shape into an area of its own (priv + blockIdx * privStride: records | windings | contour offsets |
    // { 0, nContours } | glyph descriptor): 25 edges are one round of lanes, and what a workgroup reads in the later phases it wrote itself --
    // no grid barrier between digest and distance field, and the records are warm in its XCD's L2 when the survivor walk asks for them
    char *priv;
    size_t privStride, privWindings, privOffsets, privGlyphOffsets, privGlyph;   // byte offsets inside an area (records at 0)
    // the bitmap
    int width, height, tilesX, tiles, listStride;
    float *scratch;                    // pre-correction field [h][w][N] (only when the correction pass runs)
    float *out;                        // the result tile, packed rows: device memory, or pinned host memory (written once per texel, posted writes)
    uint8_t *stencil;
    double *gres;                      // combiner scratch, one slice per tile
    size_t gresStride;
    // error correction
    MsdfHipConfig cfg;
    int correct;
    EcGlyphParams *ecParams;
    EcCandidate *cands;
    unsigned seg;
    int *corners, *sizes;
    int slotCap, slotOffset;
    // launch bookkeeping
    unsigned *barrier;                 // [0] grid-barrier counter, [16] finished-workgroup counter: device memory, only ever counted up
    unsigned barrierBase, doneBase;
    unsigned *status;                  // [0] candidate overflow (the host reruns the call through the batched path) [1] a barrier gave up
                                       // [2] completion flag (= doneValue, written last, system scope) [8..14] phase stamps of workgroup 0 (10 ns units) [15] its shader-clock cycles from the first to the last stamp
    unsigned doneValue;
    unsigned spinLimit;                // iterations a workgroup waits at the grid barrier before it gives up (status[1]; the host then reruns the call through the batched sequence)
    // Small shapes travel INSIDE the kernel arguments (the runtime keeps the argument buffer in device memory: the host writes it through the
    // PCIe aperture, posted, and the digest reads it locally -- reading the staged arrays from pinned host memory instead costs two dependent
    // PCIe round trips, ~8 of the ~11 us the digest took): payloadBytes != 0 -> contour offsets | points | types | colors | descriptor at
    // the given offsets of `payload`, and the src* pointers above are not used.
    unsigned payloadBytes, payOffsets, payPoints, payTypes, payColors, payGlyph;
    unsigned teamXchgOffset;           // byte offset of the team's exchange areas in the workgroup's LDS (behind everything the phases use)
    alignas(16) unsigned char payload[3440];
};
static_assert(sizeof(SingleArgs) <= 4096, "kernel arguments are limited to 4 KB");

// The launch is not cooperative: nothing GUARANTEES that its tiles+1 workgroups are resident together. The host only takes this path while the
// fused launches in flight fit the device (runGroup: fusedCapacity), but a persistent k_distance launch of another thread or process may still hold
// the slots. The spin is therefore bounded (spinLimit, scaled by the host with the shape's size: ~2 ms for a font glyph, at most ~0.3 s): a workgroup that
// gives up raises status[1], the rest of the launch drains, and the host reruns the call through the batched sequence -- generate*() never fails for it.
// Returns false (wave-uniform) when this workgroup gave up: what lies behind the barrier is then NOT complete -- the caller must skip its next phase (the
// other workgroups' fields, the candidate counters the extra workgroup zeroes: reading them would mean stale counts and out-of-range texel indices).
__device__ inline bool gridBarrier(unsigned *counter, unsigned target, unsigned *status, unsigned spinLimit) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");              // this wavefront's stores (all lanes) are written back before it arrives
    __builtin_amdgcn_wave_barrier();
    int ok = 1;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((int) (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)-target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > spinLimit) {
                __hip_atomic_store(status+1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                ok = 0;
                break;
            }
        }
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // the hand-placed s_load batches of the survivor walk (see above)
    return ok != 0;
}

__device__ __forceinline__ void stampPhase(const SingleArgs &a, int k) {
    if (blockIdx.x == 0 && threadIdx.x == 0)
        a.status[8+k] = (unsigned) __builtin_amdgcn_s_memrealtime();
}

// The distance checks of a tile's own candidates (k_ec_query's cooperative form: one candidate at a time, lanes = edges). A candidate only ever
// changes its own texel (ERROR: rgb := median of the pre-correction value, stencil |= ERROR) and reads the pre-correction field, which is
// complete once the grid is past the barrier -- so the workgroup that swept the tile judges them itself, no work list and no further barrier.
template <int N, bool OVERLAP>
__device__ __forceinline__ void singleCallChecks(const SingleArgs &a, const EdgeRec *recs, const int8_t *windings, const int32_t *contourOffsets, const MsdfHipGlyph *glyph,
                                                 double *smemSingle, const EcCandidate *segment, unsigned count, int lane) {
    const MsdfHipGlyph gd = glyph[0];
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
    wind.mem = windings;
    wind.pos = wind.neg = 0;
    if (OVERLAP) {
        const int w = lane < C ? (int) wind.mem[lane] : 0;
        wind.pos = __ballot(w > 0), wind.neg = __ballot(w < 0);
    }
    PBSlot *slotBuf = reinterpret_cast<PBSlot *>(smemSingle+a.slotOffset);
    for (unsigned t = 0; t < count; ++t) {
        const EcCandidate cand = segment[t];
        PsdfQueryCooperative<OVERLAP, WindingMasks> query;
        query.rec = recs, query.coff = contourOffsets, query.windings = wind, query.C = C, query.lane = lane;
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
        waveSync();                                                 // the next candidate rewrites the LDS scratch
    }
}

template <int SEL, bool OVERLAP>
__global__ void __launch_bounds__(WAVE*SINGLE_TEAM)
k_single_call(SingleArgs a) {
    enum { N = SelTraits<SEL>::NCH };
    extern __shared__ __attribute__((aligned(16))) double smemSingle[];
    // A workgroup is SINGLE_TEAM wavefronts: the helpers (rank > 0) only share the walk of the distance phase (distanceBody<..., TEAM>) and leave; digest,
    // correction sweep, distance checks and all bookkeeping are the leader's, lane = its thread index as before.
    const int lane = threadIdx.x&(WAVE-1), teamRank = MSDF_UNIFORM((int) (threadIdx.x>>6));
    const unsigned blk = blockIdx.x, groups = gridDim.x;
    const unsigned T = (unsigned) a.tiles;
    // (status[0] / [1] -- candidate overflow, a barrier that gave up -- are zeroed by the HOST before the launch and only ever raised here)
    stampPhase(a, 0);
    const unsigned long long cycles0 = __builtin_readcyclecounter();   // (shader clock: with the realtime stamps it tells at what clock the launch ran)

    // ---- phase 0: digest (k_prep_records), the whole shape by every workgroup into its own area. The contour offsets go to LDS first (every
    // lane searches them); the raw edges are read from the caller's staging (pinned host memory: two dependent PCIe round trips in all).
    char *mine = a.priv+(size_t) blk*a.privStride;
    EdgeRec *recs = reinterpret_cast<EdgeRec *>(mine);
    int8_t *windings = reinterpret_cast<int8_t *>(mine+a.privWindings);
    int32_t *contourOffsets = reinterpret_cast<int32_t *>(mine+a.privOffsets), *glyphContourOffsets = reinterpret_cast<int32_t *>(mine+a.privGlyphOffsets);
    MsdfHipGlyph *glyph = reinterpret_cast<MsdfHipGlyph *>(mine+a.privGlyph);
    if (teamRank == 0) {
        // (the payload is addressed through the kernarg segment pointer: taking the address of a member of `a` would make the compiler copy
        // the whole 4 KB struct to scratch)
        const unsigned char *pay = (const unsigned char *) __builtin_amdgcn_kernarg_segment_ptr()+offsetof(SingleArgs, payload);
        const bool inArgs = a.payloadBytes != 0;
        const int32_t *srcOffsets = inArgs ? reinterpret_cast<const int32_t *>(pay+a.payOffsets) : a.srcContourOffsets;
        const double *srcPoints = inArgs ? reinterpret_cast<const double *>(pay+a.payPoints) : a.points;
        const uint8_t *srcTypes = inArgs ? pay+a.payTypes : a.types, *srcColors = inArgs ? pay+a.payColors : a.colors;
        const MsdfHipGlyph *srcGlyph = inArgs ? reinterpret_cast<const MsdfHipGlyph *>(pay+a.payGlyph) : a.srcGlyph;
        int32_t *coLds = reinterpret_cast<int32_t *>(smemSingle);
        for (int i = lane; i <= a.nContours; i += WAVE)
            coLds[i] = srcOffsets[i];
        if (lane < (int) (sizeof(MsdfHipGlyph)/sizeof(double)))
            reinterpret_cast<double *>(glyph)[lane] = reinterpret_cast<const double *>(srcGlyph)[lane];
        if (lane < 2)
            glyphContourOffsets[lane] = lane ? a.nContours : 0;
        waveSync();
        for (int i = lane; i <= a.nContours; i += WAVE)
            contourOffsets[i] = coLds[i];
        for (int slot = lane; slot < a.nEdges; slot += WAVE) {
            int lo = 0, hi = a.nContours-1;                         // last contour c with contourOffsets[c] <= slot (skips empty contours)
            while (lo < hi) {
                const int mid = (lo+hi+1)>>1;
                if (coLds[mid] <= slot)
                    lo = mid;
                else
                    hi = mid-1;
            }
            prepRecord(recs, slot, lo, coLds, srcPoints, srcTypes, srcColors);
        }
        {                                                           // lanes = edges, the sums in edge order (msdf_prep.hpp); its 64 terms follow the offsets in LDS
            WaveCtx ctx;
            ctx.lane = lane;
            contourWindingsWave(ctx, smemSingle+(a.nContours+2+1)/2, 0, a.nContours, coLds, srcPoints, srcTypes, srcColors, windings);
        }
        // the area is this workgroup's own: its stores only have to have LEFT the wavefront (they are written through to the XCD's L2, which
        // also backs the scalar cache) before the phases below read them back; no other workgroup is involved
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        waveSync();
    }
    if (SINGLE_TEAM > 1)
        __syncthreads();                                            // the helpers read what the leader digested
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_dcache_inv\n\ts_waitcnt lgkmcnt(0)\n\tbuffer_inv sc0" ::: "memory");
    stampPhase(a, 1);

    BatchView batch;
    batch.nGlyphs = 1, batch.glyphContourOffsets = glyphContourOffsets, batch.contourOffsets = contourOffsets, batch.recs = recs, batch.windings = windings;

    // ---- phase 1: distance field, one tile per workgroup; the extra workgroup prepares the correction pass
    if (blk < T) {
        // the overlapping combiner's per-contour distances in LDS where they fit (a dependent L2 round trip per contour and pass otherwise)
        double *teamXchg = reinterpret_cast<double *>(reinterpret_cast<char *>(smemSingle)+a.teamXchgOffset);
        if (OVERLAP && a.gres == NULL)
            distanceBody<SEL, OVERLAP, false, 1, SINGLE_TEAM>(1, glyphContourOffsets, contourOffsets, recs, windings, glyph, a.width, a.height, a.tilesX, a.tiles, a.listStride,
                                                 a.correct ? a.scratch : a.out, a.correct, 0u, (double *) NULL, 0, (const int *) NULL, 0, (unsigned *) NULL, 0u, blk, smemSingle,
                                                 a.nContours, a.nEdges, teamXchg);
        else if (OVERLAP) {                                           // (combiner scratch in the global workspace: many contours -- the leader alone, as before)
            if (teamRank == 0)
                distanceBody<SEL, OVERLAP, true, 1>(1, glyphContourOffsets, contourOffsets, recs, windings, glyph, a.width, a.height, a.tilesX, a.tiles, a.listStride,
                                                    a.correct ? a.scratch : a.out, a.correct, 0u, a.gres, a.gresStride, (const int *) NULL, 0, (unsigned *) NULL, 0u, blk, smemSingle,
                                                    a.nContours, a.nEdges);
        } else
            distanceBody<SEL, OVERLAP, false, 1, SINGLE_TEAM>(1, glyphContourOffsets, contourOffsets, recs, windings, glyph, a.width, a.height, a.tilesX, a.tiles, a.listStride,
                                                a.correct ? a.scratch : a.out, a.correct, 0u, (double *) NULL, 0, (const int *) NULL, 0, (unsigned *) NULL, 0u, blk, smemSingle,
                                                a.nContours, a.nEdges, teamXchg);
    }
    else if (teamRank != 0)
        return;
    else if constexpr (SEL >= 3) {
        ecParamsBody(a.ecParams, batch, glyph, a.cfg, (unsigned *) NULL, a.corners, a.sizes, 0, lane);
        for (unsigned i = (unsigned) lane; i <= T; i += WAVE)       // the tiles' candidate counters ([0]: overflow flag)
            reinterpret_cast<unsigned *>(a.cands)[i] = 0;
    }
    if (teamRank != 0)
        return;                                                         // the helpers' part is over; everything below is per tile = per leader
    stampPhase(a, 2);
    if constexpr (SEL >= 3) {
        if (a.correct) {
            const bool together = gridBarrier(a.barrier, a.barrierBase+groups, a.status, a.spinLimit);
            stampPhase(a, 3);
            // ---- phase 2: error correction sweep (k_ec_fast) of the tile, then the distance checks of the candidates it left
            if (blk < T && together) {
                ecFastBody<(int) N>(batch, glyph, a.width, a.height, a.tilesX, a.tiles, a.scratch, a.out, a.stencil, a.cfg, a.ecParams, a.cands, a.seg,
                                    a.listStride, a.corners, blk, reinterpret_cast<int *>(smemSingle), (int) T, (int) blk);
                stampPhase(a, 4);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the candidates are this wavefront's own stores: they only have to have left it
                waveSync();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const unsigned count = __hip_atomic_load(reinterpret_cast<const unsigned *>(a.cands)+1+blk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                stampPhase(a, 5);
                if (count > a.seg) {
                    if (lane == 0)
                        a.status[0] = 1u;                           // more candidates than a tile's segment holds: the host reruns the call through the batched path
                } else if (count != 0)
                    singleCallChecks<(int) N, OVERLAP>(a, recs, windings, contourOffsets, glyph, smemSingle, a.cands+ecHeaderRecords((int) T)+(size_t) blk*a.seg, count, lane);
            }
        }
    }
    stampPhase(a, 6);
    if (blk == 0 && lane == 0)
        a.status[15] = (unsigned) (__builtin_readcyclecounter()-cycles0);

    // ---- completion: the workgroup that finishes last raises the flag the host polls (status[2]); every workgroup's results are visible at
    // system scope before it counts itself finished
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        const unsigned before = __hip_atomic_fetch_add(a.barrier+16, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before+1u == a.doneBase+groups) {
            a.status[7] = (unsigned) __builtin_amdgcn_s_memrealtime();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(a.status+2, a.doneValue, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

} // namespace msdfhip
