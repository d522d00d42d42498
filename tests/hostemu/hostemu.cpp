// TEST TOOLING ONLY -- compiles the *device* headers of the product (msdfgen_amd/csrc/msdf_{device,prep,ec}.hpp) for the host
// with g++ and walks them serially, texel by texel, exactly as the gfx950 kernels in msdf_kernels.hpp do per lane.
// Purpose: check the algorithmic restructuring of the device code (pre-digested edge records, visit order, LDS-resident
// combiner scratch, gather-form error correction) bit-for-bit against the oracle in a container that has no GPU.
// It is never loaded by the msdfgen_amd package and is not a fallback: the product path requires the HIP library.
#include <cstddef>
#include <stddef.h>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>

#include "../../msdfgen_amd/csrc/msdf_device.hpp"
#include "../../msdfgen_amd/csrc/msdf_prep.hpp"
#include "../../msdfgen_amd/csrc/msdf_ec.hpp"
#include "../../msdfgen_amd/csrc/msdf_ec_fast.hpp"
#include "../../msdfgen_amd/csrc/msdf_cull.hpp"
#include "../../msdfgen_amd/csrc/msdf_scanline.hpp"
#include "../../msdfgen_amd/csrc/msdf_shapeprep.hpp"

using namespace msdfhip;

static long g_lastDeferred = 0;

// The wave context of contourWindingsWave (msdf_prep.hpp) and colourContourWave (msdf_shapeprep.hpp) for the host: the 64 lanes one after the other between two sync points
// (the kernel's WaveCtx in msdf_kernels.hpp runs one lane each).
struct EmuWaveCtx {
    template <class F> void lanes(F f) const { for (int l = 0; l < 64; ++l) f(l); }
    template <class P> unsigned long long ballot(P pred) const {
        unsigned long long m = 0;
        for (int l = 0; l < 64; ++l)
            if (pred(l))
                m |= 1ull<<l;
        return m;
    }
    template <class F> void leader(F f) const { f(); }
    void sync() const { }
};


namespace {

struct Digest {
    std::vector<EdgeRec> recs;
    std::vector<int8_t> windings;
};

Digest digest(int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors) {
    Digest d;
    const int nE = co[nC];
    d.recs.resize(nE > 0 ? nE : 1);
    d.windings.resize(nC > 0 ? nC : 1);
    for (int slot = 0; slot < nE; ++slot) {
        int lo = 0, hi = nC-1;
        while (lo < hi) {
            int mid = (lo+hi+1)>>1;
            if (co[mid] <= slot) lo = mid; else hi = mid-1;
        }
        prepRecord(d.recs.data(), slot, lo, co, points, types, colors);
    }
    for (int c = 0; c < nC; ++c)
        d.windings[c] = (int8_t) contourWinding(c, co, points, types, colors);
    return d;
}

template <int SEL>
void distanceAt(const Digest &d, const int32_t *co, int nC, bool overlap, V2 p, double *res, double *out) {
    EdgesAll edges;
    edges.coff = co;
    if (overlap)
        shapeDistanceOverlap<SEL>(d.recs.data(), edges, d.windings.data(), nC, p, res, 1, out);
    else
        shapeDistanceSimple<SEL>(d.recs.data(), edges, nC, p, out);
}

// Phase 1 of k_distance for one 8x8 tile: per-contour channel bounds, cull, ordered compaction (lanes = edges, serialised here).
struct TileCull {
    std::vector<int> list, cstart;
};

// Phase 1 of k_distance for one tile, as the kernel does it (msdf_kernels.hpp): bounds per contour (overlapping combiner) or per shape
// (simple combiner) as floats rounded up, one cull pass over the glyph's edges in rows of 16 consecutive edges whatever their contour;
// the survivors of a row go to the list contour by contour, nearest-first inside each contour's segment of the row.
static float floatAboveHost(double d) {
    float f = (float) d;
    if ((double) f < d)
        f = std::nextafter(f, INFINITY);
    return f;
}

template <int SEL>
TileCull cullTile(const Digest &d, const int32_t *co, int nC, bool overlap, const Xform &t, int tx, int ty, int tile) {
    TileCull tcull;
    const V2 tc = unproject(t, mk(tx*tile+.5*tile, ty*tile+.5*tile));
    const double hx = (.5*tile-.5)/fabs(t.sx), hy = (.5*tile-.5)/fabs(t.sy);
    const double tr = sqrt(hx*hx+hy*hy);
    const int nE = co[nC]-co[0];
    std::vector<int> contourOf((size_t) nE, 0);
    for (int c = 0; c < nC; ++c)
        for (int i = co[c]-co[0]; i < co[c+1]-co[0]; ++i)
            contourOf[i] = c;
    std::vector<float> U((size_t) (nC > 0 ? nC : 1)*3, INFINITY);
    for (int i = 0; i < nE; ++i) {
        const int mask = cullMask<SEL>(d.recs[i]);
        const float ub = cullUpperFromSample2(cullNearestSample2(d.recs[i], tc));        // as the kernel: fp32, rounded up
        float *mine = &U[(size_t) (overlap ? contourOf[i] : 0)*3];
        for (int ch = 0; ch < 3; ++ch)
            if ((mask>>ch)&1)
                mine[ch] = ub < mine[ch] ? ub : mine[ch];
    }
    std::vector<int> perContour((size_t) nC+1, 0);
    for (int group = 0; group < nE; group += 16) {                            // one DPP row of phase 1
        std::vector<std::pair<std::pair<int, unsigned>, int> > keyed;
        for (int i = group; i < nE && i < group+16; ++i) {
            const int mask = cullMask<SEL>(d.recs[i]);
            if (!mask)
                continue;
            const float *mine = &U[(size_t) (overlap ? contourOf[i] : 0)*3];
            double umax = 0;
            for (int ch = 0; ch < 3; ++ch)
                if ((mask>>ch)&1)
                    umax = dmax(umax, (double) mine[ch]);
            if (cullEdgeSurvives<(SEL >= 2)>(d.recs[i], tc, tr, umax))
                keyed.push_back(std::make_pair(std::make_pair(contourOf[i], cullOrderKey(d.recs[i], tc, i-group)), i));
        }
        std::sort(keyed.begin(), keyed.end());
        for (size_t k = 0; k < keyed.size(); ++k) {
            tcull.list.push_back(keyed[k].second);
            ++perContour[keyed[k].first.first];
        }
    }
    int at = 0;
    for (int c = 0; c < nC; ++c) {                                            // the list is grouped by contour: offsets = running counts
        tcull.cstart.push_back(at);
        at += perContour[c];
    }
    tcull.cstart.push_back(at);
    return tcull;
}

static long g_cullKept = 0, g_cullTotal = 0;

// Which form of the overlapping combiner the tile walk below takes (emu_set_combiner_form): 0 the single rolled pass loop (shapeDistanceOverlap), 1 the two
// instances of the contour loop the kernels use (shapeDistanceOverlapSplit).
static int g_combinerForm = 0;

// k_distance for one texel of a tile whose cull result is given (phase 2, lanes = texels).
template <int SEL>
void distanceAtCulled(const Digest &d, const TileCull &tcull, int nC, bool overlap, V2 p, double *res, double *out) {
    EdgesCulled edges;
    edges.cstart = tcull.cstart.data();
    edges.list = tcull.list.data();
    if (overlap && g_combinerForm == 1)
        shapeDistanceOverlapSplit<SEL>(d.recs.data(), edges, edges, d.windings.data(), nC, p, res, 1, out);
    else if (overlap)
        shapeDistanceOverlap<SEL>(d.recs.data(), edges, d.windings.data(), nC, p, res, 1, out);
    else
        shapeDistanceSimple<SEL>(d.recs.data(), edges, nC, p, out);
}

template <int SEL>
void renderField(const Digest &d, const int32_t *co, int nC, bool overlap, const Xform &t, int w, int h, int flip, int N, double *res, float *tile) {
    const int T = 8;
    for (int ty = 0; ty*T < h; ++ty)
        for (int tx = 0; tx*T < w; ++tx) {
            TileCull tcull = cullTile<SEL>(d, co, nC, overlap, t, tx, ty, T);
            g_cullKept += (long) tcull.list.size();
            g_cullTotal += co[nC]-co[0];
            for (int y = ty*T; y < ty*T+T && y < h; ++y)
                for (int x = tx*T; x < tx*T+T && x < w; ++x) {
                    V2 p = unproject(t, mk(x+.5, y+.5));
                    double o[4];
                    distanceAtCulled<SEL>(d, tcull, nC, overlap, p, res, o);
                    const int yn = flip ? h-1-y : y;
                    for (int ch = 0; ch < N; ++ch)
                        tile[((size_t) yn*w+x)*N+ch] = mapDistance(t, o[ch]);
                }
        }
}

void distanceAtSel(int sel, const Digest &d, const int32_t *co, int nC, bool overlap, V2 p, double *res, double *out) {
    switch (sel) {
        case 1: distanceAt<1>(d, co, nC, overlap, p, res, out); break;
        case 2: distanceAt<2>(d, co, nC, overlap, p, res, out); break;
        case 3: distanceAt<3>(d, co, nC, overlap, p, res, out); break;
        default: distanceAt<4>(d, co, nC, overlap, p, res, out); break;
    }
}

struct HostQuery {
    const Digest *d;
    const int32_t *co;
    int nC;
    bool overlap;
    double *res;
    double operator()(V2 q) const {
        double out[1];
        distanceAt<2>(*d, co, nC, overlap, q, res, out);
        return out[0];
    }
};

}

extern "C" {

long emu_last_deferred() { return g_lastDeferred; }

// The lean transcendentals of the device build (cosKernel/sinKernel based cosThirds, cbrt based powThird), exported for accuracy tests.
void emu_lean_cos_thirds(const double *t, long n, double *out) {
    for (long i = 0; i < n; ++i) {
        const double a0 = 1/3.*t[i], a1 = 1/3.*(t[i]+2*M_PI), a2 = 1/3.*(t[i]-2*M_PI);
        out[3*i] = cosZeroToPi(a0);
        out[3*i+1] = cosZeroToPi(a1);
        out[3*i+2] = cosZeroToPi(fabs(a2));
    }
}
void emu_lean_pow_third(const double *x, long n, double *out) {
    for (long i = 0; i < n; ++i)
        out[i] = powThirdLean(x[i]);
}

// divExact (reciprocal + 2 FMA) against IEEE division: returns the number of (a, b) pairs with divSafe(b) whose quotients differ.
long emu_div_exact_violations(const double *a, const double *b, long n, long *safeCount) {
    long bad = 0, safe = 0;
    for (long i = 0; i < n; ++i)
        if (divSafe(b[i])) {
            ++safe;
            const double y = 1/b[i];
            const double q1 = divExact(a[i], b[i], y), q2 = a[i]/b[i];
            if (memcmp(&q1, &q2, sizeof(double)) && !(q1 != q1 && q2 != q2))
                ++bad;
        }
    if (safeCount)
        *safeCount = safe;
    return bad;
}
void emu_cull_stats(long *kept, long *total, int reset) { *kept = g_cullKept, *total = g_cullTotal; if (reset) g_cullKept = g_cullTotal = 0; }

// Fuzz of the exact-skip prefilter of the diagonal test: returns the number of coefficient triples for which
// quadraticMayHaveRootInRange() said "no" although solveQuadratic() yields a root inside (0.01, 0.99). Must be 0.
long emu_quadratic_prefilter_violations(const float *abc, long n, long *skipped) {
    long bad = 0, skip = 0;
    for (long i = 0; i < n; ++i) {
        const float dA = abc[3*i], dBC = abc[3*i+1], dD = abc[3*i+2];
        const double qa = dD-dBC+dA, qb = dBC-dA-dA, qc = dA;
        if (!bernsteinMayHaveRoot(dA, dBC, dD) || !quadraticMayHaveRootInRange(qa, qb, qc)) {
            ++skip;
            double t[2];
            int m = solveQuadratic(t, qa, qb, qc);
            for (int k = 0; k < m; ++k)
                if (t[k] > MSDF_ARTIFACT_T_EPSILON && t[k] < 1-MSDF_ARTIFACT_T_EPSILON)
                    ++bad;
        }
    }
    if (skipped)
        *skipped = skip;
    return bad;
}

void emu_windings(int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors, int32_t *out) {
    Digest d = digest(nC, co, points, types, colors);
    for (int c = 0; c < nC; ++c)
        out[c] = d.windings[c];
}

// The windings as the digest kernels compute them since round 4 (contourWindingsWave with the 64-lane context above). form 0: k_single_call -- all
// contours of the shape in one cooperative walk; form 1: k_prep_records -- 64 contours per "wavefront", a lane each (contourWinding) except the long
// ones, which the wavefront walks together one after the other.
void emu_windings_wave(int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors, int32_t *out, int form) {
    std::vector<int8_t> w((size_t) nC+1, (int8_t) 99);
    EmuWaveCtx ctx;
    double terms[64];
    if (form == 0)
        contourWindingsWave(ctx, terms, 0, nC, co, points, types, colors, w.data());
    else
        for (int cBegin = 0; cBegin < nC; cBegin += 64) {
            unsigned long long longMask = 0;
            for (int lane = 0; lane < 64 && cBegin+lane < nC; ++lane) {
                const int c = cBegin+lane;
                if (co[c+1]-co[c] >= PREP_WINDING_WAVE_MIN_EDGES)
                    longMask |= 1ull<<lane;
                else
                    w[c] = (int8_t) contourWinding(c, co, points, types, colors);
            }
            while (longMask) {
                const int k = __builtin_ctzll(longMask);
                longMask &= longMask-1;
                contourWindingsWave(ctx, terms, cBegin+k, cBegin+k+1, co, points, types, colors, w.data());
            }
        }
    for (int c = 0; c < nC; ++c)
        out[c] = w[c];
}

void emu_shape_distance(int sel, int overlap, int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors,
                        int n, const double *pts, double *out) {
    Digest d = digest(nC, co, points, types, colors);
    std::vector<double> res((size_t) (nC+1)*5);
    for (int i = 0; i < n; ++i) {
        double o[4] = { 0, 0, 0, 0 };
        distanceAtSel(sel, d, co, nC, overlap != 0, mk(pts[2*i], pts[2*i+1]), res.data(), o);
        memcpy(out+4*i, o, sizeof(o));
    }
}

void emu_set_combiner_form(int form) { g_combinerForm = form; }

// Mirrors msdfhip_generate / msdfhip_error_correction (correctionOnly): xf = {sx, sy, tx, ty, mapScale, mapTranslate}.
void emu_generate(int mode, int correctionOnly, float *pixels, int w, int h, int rowStride, int flip, int nC, const int32_t *co, const double *points,
                  const uint8_t *types, const uint8_t *colors, const double *xf, int overlap, int ecMode, int ecDist, int stageLimit,
                  double minDev, double minImp, uint8_t *stencil) {
    const int N = correctionOnly ? mode : (mode <= 2 ? 1 : mode);
    Digest d = digest(nC, co, points, types, colors);
    const int nE = co[nC];
    std::vector<double> res((size_t) (nC+1)*5);
    Xform t = { xf[0], xf[1], xf[2], xf[3], xf[4], xf[5] };
    std::vector<float> tile((size_t) w*h*N);
    if (correctionOnly) {
        for (int y = 0; y < h; ++y)
            memcpy(&tile[(size_t) y*w*N], pixels+(ptrdiff_t) rowStride*y, sizeof(float)*(size_t) w*N);
    } else {
        switch (mode) {
            case 1: renderField<1>(d, co, nC, overlap != 0, t, w, h, flip, N, res.data(), tile.data()); break;
            case 2: renderField<2>(d, co, nC, overlap != 0, t, w, h, flip, N, res.data(), tile.data()); break;
            case 3: renderField<3>(d, co, nC, overlap != 0, t, w, h, flip, N, res.data(), tile.data()); break;
            default: renderField<4>(d, co, nC, overlap != 0, t, w, h, flip, N, res.data(), tile.data()); break;
        }
    }
    const bool correct = N >= 3 && ecMode != EC_MODE_DISABLED;
    if (!correct) {
        for (int y = 0; y < h; ++y)
            memcpy(pixels+(ptrdiff_t) rowStride*y, &tile[(size_t) y*w*N], sizeof(float)*(size_t) w*N);
        return;
    }
    EcParams p;
    p.t = t;
    p.minDeviationRatio = minDev, p.minImproveRatio = minImp;
    p.mode = ecMode, p.distanceCheck = ecDist, p.overlap = overlap, p.stageLimit = stageLimit;
    ecDerive(p);
    SdfView sdf;
    sdf.px = tile.data(), sdf.w = w, sdf.h = h, sdf.N = N, sdf.flip = flip;
    HostQuery q = { &d, co, nC, overlap != 0, res.data() };
    // k_ec_fast prologue: protectCorners list (lanes = edges, ordered)
    std::vector<int> corners;
    if (ecMode == EC_MODE_EDGE_PRIORITY)
        for (int e = 0; e < nE; ++e)
            if (d.recs[e].flags&REC_CORNER) {
                V2 pp = project(t, ld(d.recs[e].p0));
                corners.push_back((int) floor(pp.x-.5));
                corners.push_back((int) floor(pp.y-.5));
            }
    long deferredCount = 0;
    for (int yn = 0; yn < h; ++yn)
        for (int x = 0; x < w; ++x) {
            int st;
            if (stageLimit != 0)
                st = ecTexelStencil(sdf, p, d.recs.data(), nE, x, yn, &q);           // k_ec_slow over all texels (stage snapshots)
            else {
                struct Cand { double t; int dx, dy; };
                struct VecSink {
                    std::vector<Cand> v;
                    void operator()(double t, int dx, int dy) { Cand c = { t, dx, dy }; v.push_back(c); }
                } sink;
                st = ecTexelFast(sdf, p, corners.data(), (int) corners.size()/2, x, yn, sink); // k_ec_fast
                st &= ~EC_DEFER;
                const int ys = flip ? h-1-yn : yn;
                for (size_t k = 0; k < sink.v.size(); ++k) {                              // k_ec_query, one candidate each
                    ++deferredCount;
                    if (ecEvaluateCandidate(sdf, p, x, ys, sink.v[k].t, sink.v[k].dx, sink.v[k].dy, q))
                        st |= EC_ERROR;
                }
            }
            const float *in = sdf.native(x, yn);
            float v[4];
            for (int i = 0; i < N; ++i)
                v[i] = in[i];
            if ((st&EC_ERROR) && stageLimit == 0) {
                float m = medianf(v[0], v[1], v[2]);
                v[0] = m, v[1] = m, v[2] = m;
            }
            float *px = pixels+(ptrdiff_t) rowStride*yn+(ptrdiff_t) N*x;
            for (int i = 0; i < N; ++i)
                px[i] = v[i];
            if (stencil)
                stencil[(size_t) yn*w+x] = (uint8_t) st;
        }
    g_lastDeferred = deferredCount;
}

}

// Mirrors k_sign_correction (msdf_kernels.hpp): per 8x8 tile, phase 1 builds the per-row intersection lists from the records
// (lanes = (row, edge) tasks, serialised here; the list order does not matter, only sums over it are taken) -- for multi-channel
// fields including the row above and the row below the tile; phase 2 is per texel.
static int scanlineSumEmu(const std::vector<double> &xs, const std::vector<int> &dirs, double px) {
    int sum = 0;
    for (size_t i = 0; i < xs.size(); ++i)
        sum += px >= xs[i] ? dirs[i] : 0;
    return sum;
}

template <int N>
static void signCorrectionEmu(const Digest &d, int nE, const Xform &t, int w, int h, int flip, const float *src, float *pixels, int rowStride,
                              float zero, int fillRule, int rasterizeOnly = 0) {
    const EdgeRec *rec = d.recs.data();
    const int tilesX = (w+7)/8, tilesY = (h+7)/8;
    for (int tile = 0; tile < tilesX*tilesY; ++tile) {
        const int tx = tile%tilesX, ty = tile/tilesX;
        std::vector<double> rowX[10];
        std::vector<int> rowDir[10];
        const int rowLo = N == 1 ? 1 : 0, rows = N == 1 ? 8 : 10;
        for (int task = 0; task < rows*nE; ++task) {
            const int e = task/rows, rr = rowLo+task-e*rows;
            const int ys = ty*8+rr-1;
            if (ys < 0 || ys >= h)
                continue;
            const double y = (ys+.5)/t.sy-t.ty;
            if (!rowMayIntersect(rec[e], y))
                continue;
            double x[3];
            int dy[3];
            const int n = scanlineIntersections(rec[e], x, dy, y);
            for (int k = 0; k < n; ++k)
                rowX[rr].push_back(x[k]), rowDir[rr].push_back(dy[k]);
        }
        for (int lane = 0; lane < 64; ++lane) {
            const int lx = lane&7, ly = lane>>3;
            const int x = tx*8+lx, ys = ty*8+ly;
            if (x >= w || ys >= h)
                continue;
            const double px = (x+.5)/t.sx-t.tx;
            const bool fill = interpretFillRule(scanlineSumEmu(rowX[ly+1], rowDir[ly+1], px), fillRule);
            const int yn = flip ? h-1-ys : ys;
            const float twice = zero+zero;
            float v[4] = { 0, 0, 0, 0 };
            if (rasterizeOnly) {
                v[0] = (float) fill;
            } else {
                const float *in = src+((size_t) yn*w+x)*N;
                for (int i = 0; i < N; ++i)
                    v[i] = in[i];
                if (N == 1) {
                    if ((v[0] > zero) != fill)
                        v[0] = twice-v[0];
                } else {
                    const int match = signMatch(v, fill, zero);
                    bool flipRgb = match < 0;
                    if (match == 0) {
                        int vote = 0;
                        if (x > 0)
                            vote += signMatch(src+((size_t) yn*w+x-1)*N, interpretFillRule(scanlineSumEmu(rowX[ly+1], rowDir[ly+1], (x-.5)/t.sx-t.tx), fillRule), zero);
                        if (x < w-1)
                            vote += signMatch(src+((size_t) yn*w+x+1)*N, interpretFillRule(scanlineSumEmu(rowX[ly+1], rowDir[ly+1], (x+1.5)/t.sx-t.tx), fillRule), zero);
                        if (ys > 0) {
                            const int ynb = flip ? h-1-(ys-1) : ys-1;
                            vote += signMatch(src+((size_t) ynb*w+x)*N, interpretFillRule(scanlineSumEmu(rowX[ly], rowDir[ly], px), fillRule), zero);
                        }
                        if (ys < h-1) {
                            const int ynb = flip ? h-1-(ys+1) : ys+1;
                            vote += signMatch(src+((size_t) ynb*w+x)*N, interpretFillRule(scanlineSumEmu(rowX[ly+2], rowDir[ly+2], px), fillRule), zero);
                        }
                        flipRgb = vote < 0;
                    }
                    if (flipRgb)
                        v[0] = twice-v[0], v[1] = twice-v[1], v[2] = twice-v[2];
                    if (N >= 4 && (v[3] > zero) != fill)
                        v[3] = twice-v[3];
                }
            }
            float *o = pixels+(ptrdiff_t) rowStride*yn+(ptrdiff_t) N*x;
            for (int i = 0; i < N; ++i)
                o[i] = v[i];
        }
    }
}

extern "C" void emu_sign_correction(int N, float *pixels, int w, int h, int rowStride, int flip, int nC, const int32_t *co, const double *points,
                                    const uint8_t *types, const uint8_t *colors, const double *xf, float zero, int fillRule) {
    Digest d = digest(nC, co, points, types, colors);
    Xform t = { xf[0], xf[1], xf[2], xf[3], 1, 0 };
    std::vector<float> src((size_t) w*h*N);
    for (int y = 0; y < h; ++y)
        memcpy(&src[(size_t) y*w*N], pixels+(ptrdiff_t) rowStride*y, sizeof(float)*(size_t) w*N);
    switch (N) {
        case 1: signCorrectionEmu<1>(d, co[nC], t, w, h, flip, src.data(), pixels, rowStride, zero, fillRule); break;
        case 3: signCorrectionEmu<3>(d, co[nC], t, w, h, flip, src.data(), pixels, rowStride, zero, fillRule); break;
        default: signCorrectionEmu<4>(d, co[nC], t, w, h, flip, src.data(), pixels, rowStride, zero, fillRule); break;
    }
}

extern "C" int emu_scanline_intersections(int type, const double *pts, double y, double *x, int *dy) {
    const int32_t co[2] = { 0, 1 };
    const uint8_t t8 = (uint8_t) type, c8 = 7;
    EdgeRec rec;
    prepRecord(&rec, 0, 0, co, pts, &t8, &c8);
    return scanlineIntersections(rec, x, dy, y);
}

extern "C" void emu_rasterize(float *pixels, int w, int h, int rowStride, int flip, int nC, const int32_t *co, const double *points,
                              const uint8_t *types, const uint8_t *colors, const double *xf, int fillRule) {
    Digest d = digest(nC, co, points, types, colors);
    Xform t = { xf[0], xf[1], xf[2], xf[3], 1, 0 };
    std::vector<float> src((size_t) w*h);
    signCorrectionEmu<1>(d, co[nC], t, w, h, flip, src.data(), pixels, rowStride, 0.f, fillRule, 1);
}

// Host rendition of EdgesCooperative (msdf_kernels.hpp): 64 "lanes" evaluate 64 edges each into a single-edge selector, the states
// are merged with the kernel's shuffle tree (lane l <- merge(l, l+off), off = 1, 2, ... 32), chunk results are merged in order.
namespace msdfhip {
struct EdgesCoopEmu {
    bool slotted;                       // true: the kernel's LDS-slot path (all edges first, then lanes = contours merge their own slots)
    const int32_t *coff;
    int begin(int c) const { return coff[c]-coff[0]; }
    int end(int c) const { return coff[c+1]-coff[0]; }
};
inline void selAddContour(Selector<2> &sel, const EdgeRec *rec, const EdgesCoopEmu &edges, int c, V2 o) {
    if (edges.slotted) {                 // single-edge states -> the contour's own merged state (from the initial one) -> merged into sel
        PB acc;
        pbInit(acc);
        for (int i = edges.begin(c); i < edges.end(c); ++i) {
            Selector<2> mine;
            selInit(mine);
            selAddEdge(mine, rec[i], i, o);
            pbMerge(acc, mine.c[0]);
        }
        pbMerge(sel.c[0], acc);
        return;
    }
    const int e = edges.end(c);
    for (int base = edges.begin(c); base < e; base += 64) {
        PB lanes[64];
        for (int l = 0; l < 64; ++l) {
            Selector<2> mine;
            selInit(mine);
            if (base+l < e)
                selAddEdge(mine, rec[base+l], base+l, o);
            lanes[l] = mine.c[0];
        }
        for (int off = 1; off < 64; off <<= 1) {
            PB next[64];
            for (int l = 0; l < 64; ++l) {
                next[l] = lanes[l];
                if (l+off < 64)
                    pbMerge(next[l], lanes[l+off]);
            }
            for (int l = 0; l < 64; ++l)
                lanes[l] = next[l];
        }
        pbMerge(sel.c[0], lanes[0]);
    }
}
}

extern "C" void emu_psdf_cooperative(int overlap, int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors,
                                     int n, const double *pts, double *out) {
    Digest d = digest(nC, co, points, types, colors);
    std::vector<double> res((size_t) (nC+1)*5);
    EdgesCoopEmu edges;
    edges.coff = co;
    edges.slotted = overlap >= 2;
    overlap &= 1;
    for (int i = 0; i < n; ++i) {
        double o[1] = { 0 };
        const V2 q = mk(pts[2*i], pts[2*i+1]);
        if (overlap)
            shapeDistanceOverlap<2>(d.recs.data(), edges, d.windings.data(), nC, q, res.data(), 1, o);
        else
            shapeDistanceSimple<2>(d.recs.data(), edges, nC, q, o);
        out[i] = o[0];
    }
}

// Mirrors the preparation passes of msdfhip_batch_create_prepared for ONE shape (msdf_shapeprep.hpp): normalizedCount ->
// normalizeContour (a "thread" per contour) -> colouredCount -> prefix -> colourContour along the shape (a "thread" per glyph).
// Output arrays sized for 3*E edges; returns the number of output edges.
#include <cmath>
extern "C" int emu_shape_prepare(int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors, int doNormalize,
                                 int coloring, double angleThreshold, unsigned long long seed, int32_t *outCo, double *outPoints,
                                 uint8_t *outTypes, uint8_t *outColors) {
    const int nE = co[nC];
    EdgeArrays raw = { const_cast<double *>(points), const_cast<uint8_t *>(types), const_cast<uint8_t *>(colors) };
    std::vector<int32_t> co1(nC+1, 0);
    for (int c = 0; c < nC; ++c)
        co1[c+1] = co1[c]+(doNormalize ? normalizedCount(co[c+1]-co[c]) : co[c+1]-co[c]);
    std::vector<double> p1((size_t) 8*(co1[nC]+1));
    std::vector<uint8_t> t1(co1[nC]+1), c1(co1[nC]+1);
    EdgeArrays norm = { p1.data(), t1.data(), c1.data() };
    for (int c = 0; c < nC; ++c) {
        if (doNormalize)
            normalizeContour(raw, co[c], co[c+1]-co[c], norm, co1[c]);
        else
            for (int i = 0; i < co[c+1]-co[c]; ++i)
                storeEdge(norm, co1[c]+i, loadEdge(raw, co[c]+i));
    }
    (void) nE;
    EdgeArrays out = { outPoints, outTypes, outColors };
    if (!coloring) {
        for (int c = 0; c <= nC; ++c)
            outCo[c] = co1[c];
        for (int e = 0; e < co1[nC]; ++e)
            storeEdge(out, e, loadEdge(norm, e));
        return co1[nC];
    }
    const double crossThreshold = sin(angleThreshold);
    outCo[0] = 0;
    for (int c = 0; c < nC; ++c)
        outCo[c+1] = outCo[c]+colouredCount(norm, co1[c], co1[c+1]-co1[c], crossThreshold);
    int color = initColor(seed);
    std::vector<int> ci(co1[nC]+1);
    std::vector<double> cl(co1[nC]+1);
    std::vector<uint8_t> cm(co1[nC]+1), cc(co1[nC]+1);
    CornerWork cw = { ci.data(), cl.data(), cm.data(), cc.data() };
    for (int c = 0; c < nC; ++c) {
        if (coloring == 2)
            colourContourInkTrap(norm, co1[c], co1[c+1]-co1[c], out, outCo[c], crossThreshold, color, seed, cw, co1[c]);
        else
            colourContour(norm, co1[c], co1[c+1]-co1[c], out, outCo[c], crossThreshold, color, seed);
    }
    return outCo[nC];
}

// Mirrors the round-4 form of msdfhip_batch_create_prepared for one shape: k_prep_normalize_flat (a "lane" per OUTPUT edge) + k_prep_normalize_cusps
// (flagged contours redone serially) -> k_prep_count -> k_prep_offsets -> k_prep_colour_wave<coloring == 2> (colourContourWave with the wave
// context above; tableCap = the kernel's PREP_WAVE_MAX_EDGES is irrelevant here, the tables are sized for the shape). cuspContours: flagged contours.
extern "C" int emu_shape_prepare_wave(int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors, int doNormalize,
                                      int coloring, double angleThreshold, unsigned long long seed, int32_t *outCo, double *outPoints,
                                      uint8_t *outTypes, uint8_t *outColors, int *cuspContours) {
    EdgeArrays raw = { const_cast<double *>(points), const_cast<uint8_t *>(types), const_cast<uint8_t *>(colors) };
    std::vector<int32_t> co1(nC+1, 0), cusp(nC+1, 0);
    for (int c = 0; c < nC; ++c)
        co1[c+1] = co1[c]+(doNormalize ? normalizedCount(co[c+1]-co[c]) : co[c+1]-co[c]);
    const int nE1 = co1[nC];
    std::vector<double> p1((size_t) 8*(nE1+1));
    std::vector<uint8_t> t1(nE1+1), c1(nE1+1);
    EdgeArrays norm = { p1.data(), t1.data(), c1.data() };
    for (int slot = 0; slot < nE1; ++slot) {                                          // k_prep_normalize_flat
        int lo = 0, hi = nC-1;
        while (lo < hi) {
            const int mid = (lo+hi+1)>>1;
            if (co1[mid] <= slot) lo = mid; else hi = mid-1;
        }
        if (normalizeEdgeFlat(raw, co[lo], co[lo+1]-co[lo], norm, co1[lo], slot-co1[lo], doNormalize != 0))
            cusp[lo] = 1;
    }
    int flagged = 0;
    for (int c = 0; c < nC; ++c)                                                      // k_prep_normalize_cusps
        if (doNormalize && cusp[c]) {
            normalizeContour(raw, co[c], co[c+1]-co[c], norm, co1[c]);
            ++flagged;
        }
    if (cuspContours)
        *cuspContours = flagged;
    EdgeArrays out = { outPoints, outTypes, outColors };
    if (!coloring) {
        for (int c = 0; c <= nC; ++c)
            outCo[c] = co1[c];
        for (int e = 0; e < nE1; ++e)
            storeEdge(out, e, loadEdge(norm, e));
        return nE1;
    }
    const double crossThreshold = sin(angleThreshold);
    outCo[0] = 0;
    for (int c = 0; c < nC; ++c)                                                      // k_prep_count + k_prep_offsets
        outCo[c+1] = outCo[c]+colouredCount(norm, co1[c], co1[c+1]-co1[c], crossThreshold);
    int longest = 1;
    for (int c = 0; c < nC; ++c)
        longest = std::max(longest, co1[c+1]-co1[c]);
    std::vector<unsigned long long> mask((size_t) longest/64+2);
    std::vector<unsigned char> splineColor(longest), minor(longest);
    std::vector<double> edgeLength(longest), cornerLength(longest);
    std::vector<int> cornerIndex(longest);
    ColourTables t = { mask.data(), splineColor.data(), edgeLength.data(), cornerLength.data(), cornerIndex.data(), minor.data() };
    EmuWaveCtx ctx;
    int color = initColor(seed);
    for (int c = 0; c < nC; ++c) {
        // (stale table contents from the previous contour stay in place, as in LDS)
        if (coloring == 2)
            colourContourWave<true>(ctx, t, norm, co1[c], co1[c+1]-co1[c], out, outCo[c], crossThreshold, color, seed);
        else
            colourContourWave<false>(ctx, t, norm, co1[c], co1[c+1]-co1[c], out, outCo[c], crossThreshold, color, seed);
    }
    return outCo[nC];
}

// Mirrors k_sdf_error_lines + k_sdf_error_sum: a "lane" per scanline (lists with stride 1 here), then the sequential sum per glyph.
extern "C" double emu_estimate_sdf_error(int N, const float *px, int w, int h, int yDown, int nC, const int32_t *co, const double *points,
                                         const uint8_t *types, const uint8_t *colors, const double *xf, int scanlinesPerRow, int fillRule) {
    if (w <= 1 || h <= 1 || scanlinesPerRow < 1)
        return 0;
    Digest d = digest(nC, co, points, types, colors);
    const int nE = co[nC];
    std::vector<double> rx(3*(size_t) nE+1), sx(3*(size_t) w+2);
    std::vector<int> rd(3*(size_t) nE+1), sd(3*(size_t) w+2);
    double error = 0;
    for (int row = 0; row < h-1; ++row)
        for (int subRow = 0; subRow < scanlinesPerRow; ++subRow) {
            StridedList refList = { rx.data(), rd.data(), 1, 0 }, sdfList = { sx.data(), sd.data(), 1, 0 };
            double v;
            if (N == 1)
                v = sdfErrorOfLine<1>(d.recs.data(), nE, px, w, h, xf[0], xf[1], xf[2], xf[3], yDown != 0, row, subRow, scanlinesPerRow, fillRule, refList, sdfList);
            else if (N == 3)
                v = sdfErrorOfLine<3>(d.recs.data(), nE, px, w, h, xf[0], xf[1], xf[2], xf[3], yDown != 0, row, subRow, scanlinesPerRow, fillRule, refList, sdfList);
            else
                v = sdfErrorOfLine<4>(d.recs.data(), nE, px, w, h, xf[0], xf[1], xf[2], xf[3], yDown != 0, row, subRow, scanlinesPerRow, fillRule, refList, sdfList);
            error += v;
        }
    return error/((h-1)*scanlinesPerRow);
}

// Diagnostics (tools only): wave-level cost model of phase 2. For one glyph: number of (tile, edge) evaluations a wavefront performs
// with the current tile cull and the per-texel wave vote, for the simple (overlap = 0) or overlapping combiner, optionally walking
// each contour's survivors nearest-first (order = 1: ascending cullUpperDistance at the tile centre) instead of in visit order.
// out[0] = evaluations, out[1] = survivors of the tile cull, out[2] = tiles, out[3] = contour walks (tile x contour with >= 1 survivor),
// out[4] = evaluations repeated by the second walks of the overlapping combiner, out[5] = tiles x passes that take a second walk.
// STUDY ONLY (order & 0x100): a tighter lower bound of the distance to a QUADRATIC edge -- the curve lies in its control box AND in the slab between
// its chord and the parallel through its apex (normal offset dot(p1-p0, n)/2), so the distance is at least the larger of the two bounds.
static bool relevantWithSlab(const Selector<3> &s, const EdgeRec &e, V2 o) {
    double bound2;
    bool box = selEdgeRelevantBox(s, e, o, bound2);
    if (box && e.type == 2) {
        const V2 ch = e.PE()-e.P0();
        const double len = sqrt(dot(ch, ch));
        if (len > 0) {
            const V2 n = mk(-ch.y/len, ch.x/len);
            const double sp = dot(o-e.P0(), n), apex = .5*dot(e.P1()-e.P0(), n);
            const double lo = apex < 0 ? apex : 0, hi = apex > 0 ? apex : 0;
            double ds = lo-sp > sp-hi ? lo-sp : sp-hi;
            ds = ds > 0 ? ds*(1-1e-9) : 0;
            if (ds*ds > bound2)
                box = false;
        }
    }
    return box || selEdgeRelevantWedges<3>(e, o, bound2);
}

extern "C" void emu_wave_cost(int w, int h, int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors,
                              const double *xf, int overlap, int order, long *out) {
    const bool slab = (order&0x100) != 0;
    order &= 0xff;
    Digest d = digest(nC, co, points, types, colors);
    Xform t = { xf[0], xf[1], xf[2], xf[3], xf[4], xf[5] };
    for (int ty = 0; ty < (h+7)/8; ++ty)
        for (int tx = 0; tx < (w+7)/8; ++tx) {
            TileCull tc = cullTile<3>(d, co, nC, overlap != 0, t, tx, ty, 8);
            const V2 centre = unproject(t, mk(tx*8+4., ty*8+4.));
            ++out[2];
            out[1] += (long) tc.list.size();
            Selector<3> sel[64];
            for (int l = 0; l < 64; ++l)
                selInit(sel[l]);
            std::vector<long> contourEvals((size_t) nC, 0);                   // overlap: evaluations of each contour's own walk ...
            std::vector<unsigned long long> innerOf((size_t) nC, 0), outerOf((size_t) nC, 0);   // ... and the lanes for which it is a member of the inner / outer selector
            int nIn[64] = { 0 }, nOut[64] = { 0 };
            for (int c = 0; c < nC; ++c) {
                if (overlap && c > 0)
                    for (int l = 0; l < 64; ++l) {
                        double dd[3];
                        selDistance(sel[l], dd);
                        const double m = resolve<3>(dd);
                        if (d.windings[c-1] > 0 && m >= 0)
                            innerOf[c-1] |= 1ull<<l, ++nIn[l];
                        if (d.windings[c-1] < 0 && m <= 0)
                            outerOf[c-1] |= 1ull<<l, ++nOut[l];
                    }
                if (overlap)
                    for (int l = 0; l < 64; ++l)
                        selInit(sel[l]);
                const long evalsBefore = out[0];
                struct Tally { long &slot, &total, before; ~Tally() { slot = total-before; } } tally = { contourEvals[c], out[0], evalsBefore };
                std::vector<int> walk(tc.list.begin()+tc.cstart[c], tc.list.begin()+tc.cstart[c+1]);
                if (!walk.empty())
                    ++out[3];
                auto nearer = [&](int a, int b) { return cullUpperDistance(d.recs[a], centre) < cullUpperDistance(d.recs[b], centre); };
                if (order == 1)
                    std::stable_sort(walk.begin(), walk.end(), nearer);
                else if (order == 2 && !walk.empty())                        // only the nearest moves to the front
                    std::rotate(walk.begin(), std::min_element(walk.begin(), walk.end(), nearer), walk.end());
                else if (order >= 16) {                                      // sorted within chunks of `order` consecutive EDGES of the contour (phase-1 lanes)
                    const int b0 = co[c]-co[0];
                    size_t from = 0;
                    while (from < walk.size()) {
                        size_t to = from;
                        while (to < walk.size() && (walk[to]-b0)/order == (walk[from]-b0)/order)
                            ++to;
                        std::stable_sort(walk.begin()+from, walk.begin()+to, nearer);
                        from = to;
                    }
                }
                for (size_t k = 0; k < walk.size(); ++k) {
                    const int i = walk[k];
                    int relevant = 0;
                    for (int l = 0; l < 64; ++l) {
                        const V2 p = unproject(t, mk(tx*8+(l&7)+.5, ty*8+(l>>3)+.5));
                        relevant += slab ? relevantWithSlab(sel[l], d.recs[i], p) : selEdgeRelevant(sel[l], d.recs[i], p);
                    }
                    if (!relevant)
                        continue;
                    ++out[0];
                    // out[6..]: how many evaluations change NO lane's state (what a tighter relevance test could still drop), by edge type
                    Selector<3> before[64];
                    memcpy(before, sel, sizeof(sel));
                    for (int l = 0; l < 64; ++l) {
                        const V2 p = unproject(t, mk(tx*8+(l&7)+.5, ty*8+(l>>3)+.5));
                        selAddEdge(sel[l], d.recs[i], i, p);
                    }
                    const int ty_ = d.recs[i].type;
                    ++out[7+ty_];
                    bool trueChanged = false, anyChanged = false;
                    for (int l = 0; l < 64; ++l)
                        for (int ch = 0; ch < 3; ++ch) {
                            trueChanged = trueChanged || sel[l].idx[ch] != before[l].idx[ch];
                            anyChanged = anyChanged || sel[l].idx[ch] != before[l].idx[ch] || sel[l].c[ch].neg != before[l].c[ch].neg || sel[l].c[ch].pos != before[l].c[ch].pos;
                        }
                    if (!anyChanged)
                        ++out[6], ++out[10+ty_];
                    else if (!trueChanged)
                        ++out[14];
                }
            }
            if (overlap && nC > 1) {                                          // out[4]: evaluations of the second walks (shapeDistanceOverlap, passes 1 and 2)
                for (int l = 0; l < 64; ++l) {
                    double dd[3];
                    selDistance(sel[l], dd);
                    const double m = resolve<3>(dd);
                    if (d.windings[nC-1] > 0 && m >= 0)
                        innerOf[nC-1] |= 1ull<<l, ++nIn[l];
                    if (d.windings[nC-1] < 0 && m <= 0)
                        outerOf[nC-1] |= 1ull<<l, ++nOut[l];
                }
                unsigned long long mineIn = 0, mineOut = 0;
                for (int l = 0; l < 64; ++l) {
                    if (nIn[l] >= 2) mineIn |= 1ull<<l;
                    if (nOut[l] >= 2) mineOut |= 1ull<<l;
                }
                for (int c = 0; c < nC; ++c) {
                    if (innerOf[c]&mineIn) out[4] += contourEvals[c];
                    if (outerOf[c]&mineOut) out[4] += contourEvals[c];
                }
                out[5] += (mineIn != 0)+(mineOut != 0);
            }
        }
}

// Diagnostics (tools only): lockstep walk of phase 2 of k_distance over one glyph, per-contour selectors as in the overlapping
// combiner: for every surviving edge that the wavefront evaluates (some lane finds it relevant), how many of the 64 lanes did?
// out[0] = edges evaluated by wavefronts, out[1] = lane-evaluations that were relevant, out[2] = edges skipped by the wave vote.
extern "C" void emu_lane_relevance_stats(int w, int h, int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors,
                                         const double *xf, long *out) {
    Digest d = digest(nC, co, points, types, colors);
    Xform t = { xf[0], xf[1], xf[2], xf[3], xf[4], xf[5] };
    for (int ty = 0; ty < (h+7)/8; ++ty)
        for (int tx = 0; tx < (w+7)/8; ++tx) {
            TileCull tc = cullTile<3>(d, co, nC, true, t, tx, ty, 8);
            for (int c = 0; c < nC; ++c) {
                Selector<3> sel[64];
                for (int l = 0; l < 64; ++l)
                    selInit(sel[l]);
                for (int k = tc.cstart[c]; k < tc.cstart[c+1]; ++k) {
                    const int i = tc.list[k];
                    int relevant = 0;
                    bool rel[64];
                    for (int l = 0; l < 64; ++l) {
                        const V2 p = unproject(t, mk(tx*8+(l&7)+.5, ty*8+(l>>3)+.5));
                        rel[l] = selEdgeRelevant(sel[l], d.recs[i], p);
                        relevant += rel[l];
                    }
                    if (!relevant) {
                        ++out[2];
                        continue;
                    }
                    ++out[0];
                    out[1] += relevant;
                    for (int l = 0; l < 64; ++l) {
                        const V2 p = unproject(t, mk(tx*8+(l&7)+.5, ty*8+(l>>3)+.5));
                        selAddEdge(sel[l], d.recs[i], i, p);
                    }
                }
            }
        }
}
