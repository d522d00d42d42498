// san_driver.cpp -- exercises the HOST side of libmsdfgen_hip (arenas, group-commit micro-batcher, host-output pipeline, sharded
// generator, argument validation) from plain C++ so that it can run under AddressSanitizer + UBSan and under ThreadSanitizer on the
// GPU box (tests/sanitize/run.sh). Shapes are synthesised here (closed polygons with a few quadratic
// sides, 1..12 contours); results are cross-checked between the entry points (single call == batch == sharded), not against the oracle
// -- parity is the business of tests/test_gpu_*.py.
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <thread>
#include <vector>

#include "msdfgen_hip.h"

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
extern "C" void __lsan_do_leak_check(void);
#endif
#endif

struct Shapes {
    std::vector<int32_t> gco, co;
    std::vector<double> points;
    std::vector<uint8_t> types, colors;
    std::vector<double> xf;             // 6 per glyph
    int n() const { return (int) gco.size()-1; }
};

static unsigned rngState = 12345;
static double rnd() { rngState = rngState*1664525u+1013904223u; return (rngState>>8)/16777216.; }

static void addGlyph(Shapes &s, int contours, int size) {
    for (int c = 0; c < contours; ++c) {
        const int n = 3+(int) (rnd()*6);
        const double cx = .2+.6*rnd(), cy = .2+.6*rnd(), r = .05+.25*rnd(), a0 = 6.28*rnd();
        const bool hole = c > 0 && rnd() < .4;
        for (int i = 0; i < n; ++i) {
            const double a = a0+(hole ? -1 : 1)*6.28318530718*i/n, b = a0+(hole ? -1 : 1)*6.28318530718*(i+1)/n;
            double p[8] = { cx+r*cos(a), cy+r*sin(a), 0, 0, 0, 0, 0, 0 };
            const bool quad = rnd() < .5;
            if (quad) {
                const double m = .5*(a+b), rr = r*(1.05+.2*rnd());
                p[2] = cx+rr*cos(m), p[3] = cy+rr*sin(m), p[4] = cx+r*cos(b), p[5] = cy+r*sin(b);
            } else
                p[2] = cx+r*cos(b), p[3] = cy+r*sin(b);
            s.points.insert(s.points.end(), p, p+8);
            s.types.push_back(quad ? 2 : 1);
            static const uint8_t cyc[3] = { 6, 5, 3 };                // CYAN, MAGENTA, YELLOW
            s.colors.push_back(cyc[i%3]);
        }
        s.co.push_back((int32_t) s.types.size());
    }
    s.gco.push_back((int32_t) s.co.size()-1);
    const double range = 4./size;
    const double xf[6] = { (double) size, (double) size, 0, 0, 1/(2*range*.5), range*.5 };   // DistanceMapping(Range(-range/2, range/2))
    s.xf.insert(s.xf.end(), xf, xf+6);
}

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { ++failures; fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #cond, msdfhip_last_error()); } } while (0)

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 3;
    const int W = 32, N = 3;
    Shapes s;
    s.gco.push_back(0), s.co.push_back(0);
    const int G = 200;
    for (int g = 0; g < G; ++g)
        addGlyph(s, g%23 == 0 ? 12 : 1+g%4, W);
    MsdfHipConfig cfg;
    msdfhip_default_config(&cfg);
    EXPECT(msdfhip_init(0) == MSDFHIP_OK);
    if (failures)
        return 2;
    const size_t tile = (size_t) W*W*N;

    // 1. batch on the device -> host tiles (packed), with stencil; twice with different chunk sizes
    std::vector<MsdfHipGlyph> gd(G);
    for (int g = 0; g < G; ++g) {
        memcpy(gd[g].xf, &s.xf[6*g], sizeof(gd[g].xf));
        gd[g].out_offset = (int64_t) (g*tile), gd[g].row_stride = W*N, gd[g].flip = 0;
    }
    std::vector<float> want(G*tile), got(G*tile);
    std::vector<uint8_t> stencil((size_t) G*W*W);
    MsdfHipBatch *b = NULL;
    EXPECT(msdfhip_batch_create(&b, G, s.gco.data(), s.co.data(), s.points.data(), s.types.data(), s.colors.data()) == MSDFHIP_OK);
    EXPECT(msdfhip_batch_generate_host(b, MSDFHIP_MODE_MSDF, W, W, gd.data(), want.data(), want.size(), stencil.data(), &cfg) == MSDFHIP_OK);
    for (int chunk : { 7, 64, 1000 }) {
        EXPECT(msdfhip_set_pipeline_chunk(chunk) == MSDFHIP_OK);
        std::fill(got.begin(), got.end(), -1.f);
        EXPECT(msdfhip_batch_generate_host(b, MSDFHIP_MODE_MSDF, W, W, gd.data(), got.data(), got.size(), NULL, &cfg) == MSDFHIP_OK);
        EXPECT(memcmp(got.data(), want.data(), sizeof(float)*got.size()) == 0);
    }
    msdfhip_set_pipeline_chunk(0);

    // 2. rectangles of an atlas with gaps and a negative stride; 8-bit atlas; pinned memory
    {
        const int cols = 10, pad = 2, aw = cols*(W+pad), ah = (G+cols-1)/cols*(W+pad);
        void *pinned = NULL;
        EXPECT(msdfhip_host_alloc(&pinned, sizeof(float)*(size_t) aw*ah*N) == MSDFHIP_OK);
        float *atlas = (float *) pinned;
        for (size_t i = 0; i < (size_t) aw*ah*N; ++i)
            atlas[i] = -7.f;
        std::vector<MsdfHipGlyph> ad(gd);
        for (int g = 0; g < G; ++g) {
            const int x0 = (g%cols)*(W+pad)+1, y0 = (g/cols)*(W+pad)+1;
            ad[g].out_offset = g%2 ? (int64_t) (((size_t) (y0+W-1)*aw+x0)*N) : (int64_t) (((size_t) y0*aw+x0)*N);
            ad[g].row_stride = g%2 ? -aw*N : aw*N;
        }
        msdfhip_set_pipeline_chunk(33);
        EXPECT(msdfhip_batch_generate_host(b, MSDFHIP_MODE_MSDF, W, W, ad.data(), atlas, (size_t) aw*ah*N, NULL, &cfg) == MSDFHIP_OK);
        msdfhip_set_pipeline_chunk(0);
        size_t untouched = 0, bad = 0;
        for (int g = 0; g < G; ++g) {
            const int x0 = (g%cols)*(W+pad)+1, y0 = (g/cols)*(W+pad)+1;
            for (int y = 0; y < W; ++y) {
                const int ay = g%2 ? y0+W-1-y : y0+y;
                bad += memcmp(atlas+((size_t) ay*aw+x0)*N, want.data()+g*tile+(size_t) y*W*N, sizeof(float)*W*N) != 0;
            }
        }
        for (size_t i = 0; i < (size_t) aw*ah*N; ++i)
            untouched += atlas[i] == -7.f;
        EXPECT(bad == 0);
        EXPECT(untouched == (size_t) aw*ah*N-(size_t) G*tile);
        std::vector<uint8_t> bytes(G*tile, 0);
        std::vector<MsdfHipGlyph> bd(gd);
        EXPECT(msdfhip_batch_generate_bytes_host(b, MSDFHIP_MODE_MSDF, W, W, bd.data(), bytes.data(), bytes.size(), &cfg) == MSDFHIP_OK);
        size_t nonzero = 0;
        for (size_t i = 0; i < bytes.size(); ++i)
            nonzero += bytes[i] != 0;
        EXPECT(nonzero > bytes.size()/8);
        bd[G-1].out_offset = (int64_t) bytes.size();                                  // a rectangle outside the buffer must be refused
        EXPECT(msdfhip_batch_generate_bytes_host(b, MSDFHIP_MODE_MSDF, W, W, bd.data(), bytes.data(), bytes.size(), &cfg) == MSDFHIP_ERR_INVALID);
        EXPECT(msdfhip_host_free(pinned) == MSDFHIP_OK);
    }
    msdfhip_batch_destroy(b);

    // 3. the single-shape entry point from many short-lived threads (fresh team per round, like Workload::finish()), micro-batched
    for (int round = 0; round < rounds; ++round) {
        std::atomic<int> mismatches(0), errors(0);
        std::vector<std::thread> pool;
        for (int t = 0; t < 12; ++t)
            pool.emplace_back([&, t]() {
                std::vector<float> px(tile);
                std::vector<uint8_t> st((size_t) W*W);
                std::vector<int32_t> local;
                for (int i = 0; i < 25; ++i) {
                    const int g = (t*17+i*3+round)%G, c0 = s.gco[g], nC = s.gco[g+1]-c0, e0 = s.co[c0];
                    local.assign(nC+1, 0);
                    for (int k = 0; k <= nC; ++k)
                        local[k] = s.co[c0+k]-e0;
                    const int rc = msdfhip_generate(MSDFHIP_MODE_MSDF, px.data(), W, W, W*N, 0, local.data(), nC, &s.points[8*(size_t) e0], &s.types[e0], &s.colors[e0],
                                                    &s.xf[6*(size_t) g], &cfg, (i&1) ? st.data() : NULL);
                    if (rc != MSDFHIP_OK)
                        ++errors;
                    else if (memcmp(px.data(), want.data()+g*tile, sizeof(float)*tile) != 0)
                        ++mismatches;
                    else if ((i&1) && memcmp(st.data(), stencil.data()+(size_t) g*W*W, (size_t) W*W) != 0)
                        ++mismatches;
                }
            });
        for (auto &th : pool)
            th.join();
        EXPECT(errors.load() == 0);
        EXPECT(mismatches.load() == 0);
    }

    // 4. glyph-sharded over the same device listed several times (one host thread + batch + streams per entry)
    {
        const int devices[5] = { 0, 0, 0, 0, 0 };
        for (int n : { 1, 2, 5 }) {
            std::fill(got.begin(), got.end(), -1.f);
            EXPECT(msdfhip_generate_sharded(devices, n, MSDFHIP_MODE_MSDF, W, W, G, s.gco.data(), s.co.data(), s.points.data(), s.types.data(), s.colors.data(),
                                            gd.data(), got.data(), got.size(), NULL, 0, &cfg) == MSDFHIP_OK);
            EXPECT(memcmp(got.data(), want.data(), sizeof(float)*got.size()) == 0);
        }
    }

    // 4b. round 3: the front door spread over "devices" (MSDFHIP_DEVICES, device 0 listed three times -> six leaders), the shapeless
    // correction passes, overlapping rectangles in the pipeline (staging path), msdfhip_trim between bursts
    {
        setenv("MSDFHIP_DEVICES", "0,0,0", 1);
        EXPECT(msdfhip_reload_tuning() == MSDFHIP_OK);
        int devs[4] = { -1, -1, -1, -1 };
        EXPECT(msdfhip_front_door_devices(devs, 4) == 3 && devs[2] == 0);
        std::atomic<int> mismatches(0), errors(0);
        std::vector<std::thread> pool;
        for (int t = 0; t < 10; ++t)
            pool.emplace_back([&, t]() {
                std::vector<float> px(tile);
                std::vector<int32_t> local;
                for (int i = 0; i < 20; ++i) {
                    const int g = (t*13+i*7)%G, c0 = s.gco[g], nC = s.gco[g+1]-c0, e0 = s.co[c0];
                    local.assign(nC+1, 0);
                    for (int k = 0; k <= nC; ++k)
                        local[k] = s.co[c0+k]-e0;
                    const int rc = msdfhip_generate(MSDFHIP_MODE_MSDF, px.data(), W, W, W*N, 0, local.data(), nC, &s.points[8*(size_t) e0], &s.types[e0], &s.colors[e0],
                                                    &s.xf[6*(size_t) g], &cfg, NULL);
                    if (rc != MSDFHIP_OK) {
                        if (++errors == 1)
                            fprintf(stderr, "4b: msdfhip_generate -> %d: %s\n", rc, msdfhip_last_error());
                    } else if (memcmp(px.data(), want.data()+g*tile, sizeof(float)*tile) != 0)
                        ++mismatches;
                    if (i%5 == 0) {                                          // a shapeless pass on the fresh tile: must not fail, must stay finite
                        const int rc2 = msdfhip_error_correction_shapeless(N, px.data(), W, W, W*N, &s.xf[6*(size_t) g], 1.11111111111111111, i&1);
                        if (rc2 != MSDFHIP_OK && ++errors == 1)
                            fprintf(stderr, "4b: msdfhip_error_correction_shapeless -> %d: %s\n", rc2, msdfhip_last_error());
                    }
                }
            });
        for (auto &th : pool)
            th.join();
        EXPECT(errors.load() == 0);
        EXPECT(mismatches.load() == 0);
        unsetenv("MSDFHIP_DEVICES");
        EXPECT(msdfhip_reload_tuning() == MSDFHIP_OK);
        EXPECT(msdfhip_front_door_devices(NULL, 0) == 0);
        EXPECT(msdfhip_trim() == MSDFHIP_OK);                                // pools empty now; the next calls rebuild them
        MsdfHipBatch *b2 = NULL;
        EXPECT(msdfhip_batch_create(&b2, G, s.gco.data(), s.co.data(), s.points.data(), s.types.data(), s.colors.data()) == MSDFHIP_OK);
        std::vector<MsdfHipGlyph> od(gd);
        od[5].out_offset = od[4].out_offset;                                 // two glyphs on one tile, tile 5 free: not an exact tiling -> staging + scatter
        std::fill(got.begin(), got.end(), -9.f);
        EXPECT(msdfhip_batch_generate_host(b2, MSDFHIP_MODE_MSDF, W, W, od.data(), got.data(), got.size(), NULL, &cfg) == MSDFHIP_OK);
        size_t freeCellTouched = 0;
        for (size_t i = 5*tile; i < 6*tile; ++i)
            freeCellTouched += got[i] != -9.f;
        EXPECT(freeCellTouched == 0);
        EXPECT(memcmp(got.data()+6*tile, want.data()+6*tile, sizeof(float)*(G-6)*tile) == 0);
        msdfhip_batch_destroy(b2);
        EXPECT(msdfhip_trim() == MSDFHIP_OK);
    }

    // 5. argument validation never reads past what it was given
    {
        MsdfHipBatch *bad = NULL;
        std::vector<int32_t> gco2(s.gco);
        gco2[1] = -3;
        EXPECT(msdfhip_batch_create(&bad, G, gco2.data(), s.co.data(), s.points.data(), s.types.data(), s.colors.data()) == MSDFHIP_ERR_INVALID);
        EXPECT(msdfhip_batch_create(&bad, G, s.gco.data(), s.co.data(), NULL, s.types.data(), s.colors.data()) == MSDFHIP_ERR_INVALID);
        std::vector<uint8_t> t2(s.types);
        t2[3] = 9;
        EXPECT(msdfhip_batch_create(&bad, G, s.gco.data(), s.co.data(), s.points.data(), t2.data(), s.colors.data()) == MSDFHIP_ERR_INVALID);
        const int devs[2] = { 0, 77 };
        EXPECT(msdfhip_generate_sharded(devs, 2, MSDFHIP_MODE_MSDF, W, W, G, s.gco.data(), s.co.data(), s.points.data(), s.types.data(), s.colors.data(),
                                        gd.data(), got.data(), got.size(), NULL, 0, &cfg) == MSDFHIP_ERR_INVALID);
    }
    // 6. round 5: the STREAMED generator (msdfhip_generate_stream / _stream_csr): the library's host threads call the shape source concurrently and write
    // straight into pinned staging while earlier chunks are uploaded, rendered and copied back -- chunk sizes from one glyph per chunk ... one chunk,
    // float tiles + stencil and the 8-bit atlas, two caller threads at once (two pipes, one flatten pool), and a source that miscounts.
    {
        struct Src {
            const Shapes *s;
            std::atomic<long> counts, fills;
            int lieAt;
            static void count(void *u, int g, int32_t *nc, int32_t *ne) {
                Src *me = (Src *) u;
                ++me->counts;
                const Shapes &s = *me->s;
                *nc = s.gco[g+1]-s.gco[g], *ne = s.co[s.gco[g+1]]-s.co[s.gco[g]];
            }
            static void fill(void *u, int g, int32_t edgeBase, int32_t *contourEnd, double *points, uint8_t *types, uint8_t *colors) {
                Src *me = (Src *) u;
                ++me->fills;
                const Shapes &s = *me->s;
                const int c0 = s.gco[g], nC = s.gco[g+1]-c0, e0 = s.co[c0], nE = s.co[c0+nC]-e0;
                for (int c = 0; c < nC; ++c)
                    contourEnd[c] = edgeBase+s.co[c0+c+1]-e0-(g == me->lieAt && c == nC-1 ? 1 : 0);
                memcpy(points, &s.points[8*(size_t) e0], sizeof(double)*8*(size_t) nE);
                memcpy(types, &s.types[e0], (size_t) nE), memcpy(colors, &s.colors[e0], (size_t) nE);
            }
        } src;
        src.s = &s, src.counts = 0, src.fills = 0, src.lieAt = -1;
        MsdfHipShapeSource source = { &src, Src::count, Src::fill };
        std::vector<uint8_t> st2((size_t) G*W*W), bytesWant(G*tile), bytesGot(G*tile);
        for (size_t i = 0; i < bytesWant.size(); ++i) {                       // pixelFloatToByte of the float result (core/pixel-conversion.hpp:8-10)
            const float x = want[i], c = x >= 0.f && x <= 1.f ? x : (float) (x > 0.f);
            bytesWant[i] = (uint8_t) ~(int) (255.5f-255.f*c);
        }
        for (int chunk : { 1, 7, 64, 1000 }) {
            EXPECT(msdfhip_set_pipeline_chunk(chunk) == MSDFHIP_OK);
            std::fill(got.begin(), got.end(), -1.f), std::fill(st2.begin(), st2.end(), 0xee);
            EXPECT(msdfhip_generate_stream(-1, MSDFHIP_MODE_MSDF, W, W, G, &source, gd.data(), got.data(), got.size(), NULL, 0, st2.data(), &cfg) == MSDFHIP_OK);
            EXPECT(memcmp(got.data(), want.data(), sizeof(float)*got.size()) == 0);
            EXPECT(memcmp(st2.data(), stencil.data(), st2.size()) == 0);
            std::fill(bytesGot.begin(), bytesGot.end(), 0);
            EXPECT(msdfhip_generate_stream_csr(-1, MSDFHIP_MODE_MSDF, W, W, G, s.gco.data(), s.co.data(), s.points.data(), s.types.data(), s.colors.data(), gd.data(),
                                               NULL, 0, bytesGot.data(), bytesGot.size(), NULL, &cfg) == MSDFHIP_OK);
            EXPECT(memcmp(bytesGot.data(), bytesWant.data(), bytesGot.size()) == 0);
        }
        EXPECT(src.counts.load() == 4L*G && src.fills.load() == 4L*G);       // every glyph exactly once per pass
        msdfhip_set_pipeline_chunk(16);
        {
            std::atomic<int> bad(0);
            std::vector<std::thread> callers;
            for (int t = 0; t < 2; ++t)
                callers.emplace_back([&, t]() {
                    std::vector<float> mine(G*tile, -2.f);
                    for (int i = 0; i < 3; ++i) {
                        const int rc = t ? msdfhip_generate_stream(-1, MSDFHIP_MODE_MSDF, W, W, G, &source, gd.data(), mine.data(), mine.size(), NULL, 0, NULL, &cfg)
                                         : msdfhip_generate_stream_csr(-1, MSDFHIP_MODE_MSDF, W, W, G, s.gco.data(), s.co.data(), s.points.data(), s.types.data(), s.colors.data(),
                                                                       gd.data(), mine.data(), mine.size(), NULL, 0, NULL, &cfg);
                        if (rc != MSDFHIP_OK || memcmp(mine.data(), want.data(), sizeof(float)*mine.size()) != 0)
                            ++bad;
                    }
                });
            for (auto &th : callers)
                th.join();
            EXPECT(bad.load() == 0);
        }
        src.lieAt = 57;                                                       // fill delivers one edge fewer than count promised: refused, nothing read past the staging
        const long fillsBefore = src.fills.load();
        EXPECT(msdfhip_generate_stream(-1, MSDFHIP_MODE_MSDF, W, W, G, &source, gd.data(), got.data(), got.size(), NULL, 0, NULL, &cfg) == MSDFHIP_ERR_INVALID);
        {                                                                     // the flatten jobs of the chunks AHEAD of the failing one must not outlive the call --
            const long fillsAtReturn = src.fills.load();                      // no callback into the caller's objects after the API has returned (and, under ASan, no write
            std::this_thread::sleep_for(std::chrono::milliseconds(100));      // into a feeder / a staging area that is gone)
            EXPECT(src.fills.load() == fillsAtReturn);
            EXPECT(fillsAtReturn-fillsBefore < (long) G);                     // and the failing call stopped early (chunks of 16, glyph 57 lies): it did not flatten the whole list
        }
        src.lieAt = -1;
        EXPECT(msdfhip_generate_stream(-1, MSDFHIP_MODE_MSDF, W, W, G, &source, gd.data(), got.data(), got.size(), NULL, 0, NULL, &cfg) == MSDFHIP_OK);   // and the pipe is usable afterwards
        EXPECT(memcmp(got.data(), want.data(), sizeof(float)*got.size()) == 0);
        msdfhip_set_pipeline_chunk(0);
        EXPECT(msdfhip_trim() == MSDFHIP_OK);
    }
    printf("san_driver: %d failure(s)\n", failures);
    fflush(stdout);
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
    __lsan_do_leak_check();            // now, and not from an exit handler: the ROCm runtime's own teardown trips a CHECK in its ASan device allocator
#endif
#endif
    _Exit(failures ? 1 : 0);
}
