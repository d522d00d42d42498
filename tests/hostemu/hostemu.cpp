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

#include "../../msdfgen_amd/csrc/msdf_device.hpp"
#include "../../msdfgen_amd/csrc/msdf_prep.hpp"
#include "../../msdfgen_amd/csrc/msdf_ec.hpp"

using namespace msdfhip;

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
    if (overlap)
        shapeDistanceOverlap<SEL>(d.recs.data(), co, d.windings.data(), nC, p, res, 1, out);
    else
        shapeDistanceSimple<SEL>(d.recs.data(), co, nC, p, out);
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

void emu_windings(int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors, int32_t *out) {
    Digest d = digest(nC, co, points, types, colors);
    for (int c = 0; c < nC; ++c)
        out[c] = d.windings[c];
}

void emu_shape_distance(int sel, int overlap, int nC, const int32_t *co, const double *points, const uint8_t *types, const uint8_t *colors,
                        int n, const double *pts, double *out) {
    Digest d = digest(nC, co, points, types, colors);
    std::vector<double> res((size_t) (nC+1)*4);
    for (int i = 0; i < n; ++i) {
        double o[4] = { 0, 0, 0, 0 };
        distanceAtSel(sel, d, co, nC, overlap != 0, mk(pts[2*i], pts[2*i+1]), res.data(), o);
        memcpy(out+4*i, o, sizeof(o));
    }
}

// Mirrors msdfhip_generate / msdfhip_error_correction (correctionOnly): xf = {sx, sy, tx, ty, mapScale, mapTranslate}.
void emu_generate(int mode, int correctionOnly, float *pixels, int w, int h, int rowStride, int flip, int nC, const int32_t *co, const double *points,
                  const uint8_t *types, const uint8_t *colors, const double *xf, int overlap, int ecMode, int ecDist, int stageLimit,
                  double minDev, double minImp, uint8_t *stencil) {
    const int N = correctionOnly ? mode : (mode <= 2 ? 1 : mode);
    Digest d = digest(nC, co, points, types, colors);
    const int nE = co[nC];
    std::vector<double> res((size_t) (nC+1)*4);
    Xform t = { xf[0], xf[1], xf[2], xf[3], xf[4], xf[5] };
    std::vector<float> tile((size_t) w*h*N);
    if (correctionOnly) {
        for (int y = 0; y < h; ++y)
            memcpy(&tile[(size_t) y*w*N], pixels+(ptrdiff_t) rowStride*y, sizeof(float)*(size_t) w*N);
    } else {
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                V2 p = unproject(t, mk(x+.5, y+.5));
                double o[4];
                distanceAtSel(mode, d, co, nC, overlap != 0, p, res.data(), o);
                const int yn = flip ? h-1-y : y;
                for (int ch = 0; ch < N; ++ch)
                    tile[((size_t) yn*w+x)*N+ch] = mapDistance(t, o[ch]);
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
    for (int yn = 0; yn < h; ++yn)
        for (int x = 0; x < w; ++x) {
            const int st = ecTexelStencil(sdf, p, d.recs.data(), nE, x, yn, &q);
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
}

}
