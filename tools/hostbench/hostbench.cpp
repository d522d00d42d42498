// hostbench.cpp -- a pool of C++ host threads calling the single-shape C ABI entry point the way msdf-atlas-gen's glyph workers call
// msdfgen::generateMSDF (one glyph, host pointers in and out, per call). Measures what the drop-in delivers to an UNMODIFIED caller:
// per-call latency from one thread and aggregate throughput from many (where the library's micro-batcher combines the calls).
// Driven from tools/host_call_latency.py through ctypes (one call; the GIL is not involved in the timed region).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "msdfgen_hip.h"

extern "C" double hostbench_run(int nThreads, int callsPerThread, int nGlyphs, const int32_t *gco, const int32_t *co, const double *points,
                                const uint8_t *types, const uint8_t *colors, const double *xfs, int mode, int w, int h, int *failures,
                                float *lastTile /* w*h*N floats of thread 0's last call, for a spot check */) {
    const int N = mode <= 2 ? 1 : mode;
    MsdfHipConfig cfg;
    msdfhip_default_config(&cfg);
    std::atomic<int> bad(0);
    std::vector<std::thread> pool;
    std::vector<std::vector<float> > tiles(nThreads, std::vector<float>((size_t) w*h*N));
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < nThreads; ++t)
        pool.emplace_back([&, t]() {
            std::vector<int32_t> local;
            for (int i = 0; i < callsPerThread; ++i) {
                const int g = (t*7+i)%nGlyphs;
                const int c0 = gco[g], nC = gco[g+1]-c0, e0 = co[c0];
                local.resize(nC+1);
                for (int k = 0; k <= nC; ++k)
                    local[k] = co[c0+k]-e0;
                const int rc = msdfhip_generate(mode, tiles[t].data(), w, h, w*N, 0, local.data(), nC, points+8*(size_t) e0, types+e0, colors+e0,
                                                xfs+6*(size_t) g, &cfg, NULL);
                if (rc != MSDFHIP_OK)
                    ++bad;
            }
        });
    for (auto &th : pool)
        th.join();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now()-t0).count();
    if (failures)
        *failures = bad.load();
    if (lastTile)
        memcpy(lastTile, tiles[0].data(), sizeof(float)*(size_t) w*h*N);
    return secs;
}
