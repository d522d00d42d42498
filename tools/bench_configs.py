"""Timings of the BASELINE configs that are NOT the headline bench line (bench.py owns that): each through GlyphBatch.generate on
one MI355X, inputs/outputs resident in HBM, torch events on the launch stream.  Prints one JSON object per line.

    python tools/bench_configs.py [--reps 10]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps, warmup=2):
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)/reps


def kernel_ms(fn):
    """HIP-event time of the distance kernel and of everything after it (sign pass + error correction) in one call."""
    import ctypes as C
    import torch
    from msdfgen_amd import lib as L
    lib = L.load()
    lib.msdfhip_set_kernel_timing(1)
    lib.msdfhip_kernel_timing(None, None, None, 1)
    fn()
    torch.cuda.synchronize()
    lib.msdfhip_set_kernel_timing(0)
    kd, kc, kn = C.c_double(), C.c_double(), C.c_int()
    lib.msdfhip_kernel_timing(C.byref(kd), C.byref(kc), C.byref(kn), 1)
    return [round(kd.value, 4), round(kc.value, 4)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="", help="run only the configs whose name contains this substring")
    args = ap.parse_args()
    import torch  # noqa: F401
    import msdfgen_amd as M
    from msdfgen_amd import synth
    from msdfgen_amd.shape import ShapeBatch, autoframe
    from bench import load_latin, tile_batch
    M.init(0)
    latin, xf64 = load_latin()

    def report(name, batch, mode, w, h, xfs, config=None, reps=args.reps, **kw):
        if args.only and not any(k in name for k in args.only.split(",")):
            return
        gb = M.GlyphBatch(batch)
        out = torch.empty((batch.n_glyphs, h, w, M.CHANNELS[mode]), dtype=torch.float32, device="cuda")
        desc = gb.descriptors(xfs, w, h, M.CHANNELS[mode])

        def step():
            gb.digest()
            gb.generate(mode, w, h, descriptors=desc, out=out, config=config, **kw)
        ms = timed(step, reps)
        km = kernel_ms(step)
        print(json.dumps({"config": name, "glyphs": batch.n_glyphs, "edges_per_glyph": round(batch.n_edges/batch.n_glyphs, 1),
                          "contours_per_glyph": round(batch.n_contours/batch.n_glyphs, 2), "tile": [w, h], "mode": mode,
                          "ms_per_step": round(ms, 3), "glyphs_per_s": round(batch.n_glyphs/ms*1e3), "mtexel_per_s": round(batch.n_glyphs*w*h/ms/1e3),
                          "kernel_ms_distance_and_post": km}), flush=True)
        gb.close()

    b, x = tile_batch(latin, xf64, 8192)
    report("cfg2/3 headline: Basic-Latin msdf 64x64 default EC", b, 3, 64, 64, x)
    report("cfg3: Basic-Latin mtsdf 64x64 default EC", b, 4, 64, 64, x)
    report("Basic-Latin sdf 64x64", b, 1, 64, 64, x)
    report("Basic-Latin psdf 64x64", b, 2, 64, 64, x)
    report("Basic-Latin msdf 64x64, simple combiner (overlapSupport=false)", b, 3, 64, 64, x, config=M.MSDFGeneratorConfig(False))
    report("Basic-Latin msdf 64x64, error correction disabled", b, 3, 64, 64, x, config=M.MSDFGeneratorConfig(True, M.ErrorCorrectionConfig(M.EC_DISABLED)))
    scan = M.MSDFGeneratorConfig(False, M.ErrorCorrectionConfig(M.EC_EDGE_PRIORITY, M.DO_NOT_CHECK_DISTANCE))
    report("scanline flow (main.cpp:1233-1298): msdf 64x64 simple combiner -> distanceSignCorrection -> EC without distance check", b, 3, 64, 64, x,
           config=scan, scanline_pass=True)
    base = [synth.cjk_like_shape(20000+i) for i in range(512)]
    cj = ShapeBatch.from_shapes([base[i % 512] for i in range(8192)])
    cx = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])[np.arange(8192) % 512]
    report("cfg4: 8192 CJK-like synthetic glyphs msdf 48x48 default EC", cj, 3, 48, 48, cx, reps=max(2, args.reps//3))
    report("cfg4 shapes, simple combiner (overlapSupport=false)", cj, 3, 48, 48, cx, config=M.MSDFGeneratorConfig(False), reps=max(2, args.reps//3))
    report("cfg4 shapes, sdf 48x48", cj, 1, 48, 48, cx, reps=max(2, args.reps//3))
    if not args.only or "quality" in args.only.split(","):                       # row f4: the reference's own quality metric, on the device
        gb = M.GlyphBatch(b)
        tiles = gb.generate(3, 64, 64, x)
        ms = timed(lambda: gb.estimate_sdf_error(tiles, x), max(2, args.reps//3))
        err = gb.estimate_sdf_error(tiles, x)
        view = M.render_sdf(tiles, 256, 256, 1, 4.)
        ms_render = timed(lambda: M.render_sdf(tiles, 256, 256, 1, 4., out=view), max(2, args.reps//3))
        print(json.dumps({"config": "quality: estimateSDFError (1 scanline per row) of the 8192 headline msdf tiles on the device", "ms": round(ms, 3),
                          "mean_error": float(err.mean()), "max_error": float(err.max()),
                          "renderSDF_8192_tiles_to_256x256_ms": round(ms_render, 3)}), flush=True)
        gb.close()
    if not args.only or "prep" in args.only.split(","):                          # row f3: raw outlines -> prepared, digested batch (host call, incl. copies)
        import time
        z = np.load(os.path.join(ROOT, "tests", "golden", "prep.npz"))
        raw = ShapeBatch(z["raw_gco"].astype(np.int32), z["raw_co"].astype(np.int32), z["raw_points"], z["raw_types"].astype(np.int32),
                         z["raw_colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
        big = raw.select([i % 283 for i in range(8192)])              # the font glyphs of the fixture, tiled
        M.GlyphBatch.from_raw(big).close()
        for coloring, name in ((1, "edgeColoringSimple"), (2, "edgeColoringInkTrap")):
            M.GlyphBatch.from_raw(big, True, coloring, 3.0, seed=0).close()
            t0 = time.perf_counter()
            for _ in range(5):
                M.GlyphBatch.from_raw(big, True, coloring, 3.0, seed=0).close()
            dt = (time.perf_counter()-t0)/5
            print(json.dumps({"config": "prep: Shape::normalize + %s + digest of 8192 raw glyphs (%d edges) via msdfhip_batch_create_prepared, host arrays in, "
                                        "prepared shapes read back" % (name, big.n_edges), "ms_per_call": round(1e3*dt, 3), "glyphs_per_s": round(8192/dt)}), flush=True)
    zd = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
    dj = ShapeBatch(zd["glyph_contour_offsets"].astype(np.int32), zd["contour_offsets"].astype(np.int32), zd["points"], zd["types"].astype(np.int32),
                    zd["colors"].astype(np.int32), np.zeros(len(zd["names"]), bool), [str(n) for n in zd["names"]])
    report("cfg4 real fonts: 8192 distinct DejaVuSans+Bold glyphs msdf 48x48 default EC", dj, 3, 48, 48, zd["xf48"], reps=max(2, args.reps//2))
    report("8192 distinct DejaVuSans+Bold glyphs msdf 64x64 default EC (bench workload)", dj, 3, 64, 64, zd["xf64"], reps=max(2, args.reps//2))
    report("8192 distinct DejaVu glyphs msdf 64x64, simple combiner", dj, 3, 64, 64, zd["xf64"], config=M.MSDFGeneratorConfig(False), reps=max(2, args.reps//2))
    report("8192 distinct DejaVu glyphs msdf 64x64, error correction disabled", dj, 3, 64, 64, zd["xf64"],
           config=M.MSDFGeneratorConfig(True, M.ErrorCorrectionConfig(M.EC_DISABLED)), reps=max(2, args.reps//2))
    report("scanline flow on the 8192 distinct DejaVu glyphs: msdf 64x64 simple combiner -> distanceSignCorrection -> EC without distance check", dj, 3, 64, 64,
           zd["xf64"], config=scan, scanline_pass=True, reps=max(2, args.reps//2))
    logo = synth.logo_shape(5)
    lb = ShapeBatch.from_shapes([logo])
    lx = np.stack([autoframe(logo.bounds(), 1024, 1024, 8)])
    report("cfg5: %d-edge cubic logo msdf 1024x1024 default EC" % logo.n_edges, lb, 3, 1024, 1024, lx, reps=max(2, args.reps//3))
    if not args.only or "cpu" in args.only.split(","):
        cpu_baselines(dj, zd["xf48"], base, logo, lx[0])


def cpu_baselines(dj, xf48, cjk_shapes, logo, logo_xf, budget_s=8.):
    """SURVEY 8(d) "CPU baseline timing" beside configs 4 and 5: the COMPILED REFERENCE (oracle/_ref, test infrastructure -- timed here as
    the baseline, never part of the product path) on the host cores of the same box. Config 4: glyph-parallel thread pool over a bounded
    sample (what msdf-atlas-gen does). Config 5: one 1024x1024 bitmap -- single thread, and the reference's own OpenMP row loops
    (MSDFGEN_USE_OPENMP, core/msdfgen.cpp:56-64, core/MSDFErrorCorrection.cpp:420-430) on the same number of threads."""
    import time
    from bench import available_cores
    from oracle.pyoracle import Ref
    from msdfgen_amd.shape import autoframe
    if not Ref.available():
        print(json.dumps({"config": "cpu baselines", "skipped": "oracle/_ref not built"}), flush=True)
        return
    cores = available_cores()
    ref = Ref()

    def pool_rate(shapes, xfs, w, h):
        probe = min(len(shapes), 4*cores)
        _, secs = ref.generate_batch_timed(shapes[:probe], 3, w, h, xfs[:probe], threads=cores)
        n = int(min(max(probe, probe/max(secs, 1e-9)*budget_s), 100000))
        idx = [i % len(shapes) for i in range(n)]
        _, secs = ref.generate_batch_timed([shapes[i] for i in idx], 3, w, h, xfs[idx], threads=cores)
        return n, secs

    pick = list(range(0, dj.n_glyphs, 8))
    n, secs = pool_rate([dj.shape(g) for g in pick], xf48[pick], 48, 48)
    print(json.dumps({"config": "cpu: cfg4 real fonts msdf 48x48 default EC, compiled reference, glyph-parallel pool", "threads": cores, "glyphs": n,
                      "seconds": round(secs, 2), "glyphs_per_s": round(n/secs)}), flush=True)
    cx = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in cjk_shapes[:128]])
    n, secs = pool_rate(cjk_shapes[:128], cx, 48, 48)
    print(json.dumps({"config": "cpu: cfg4 CJK-like msdf 48x48 default EC, compiled reference, glyph-parallel pool", "threads": cores, "glyphs": n,
                      "seconds": round(secs, 2), "glyphs_per_s": round(n/secs)}), flush=True)
    t0 = time.perf_counter()
    ref.generate(logo, 3, 1024, 1024, logo_xf)
    single = time.perf_counter()-t0
    line = {"config": "cpu: cfg5 logo msdf 1024x1024 default EC, compiled reference", "single_thread_s": round(single, 2)}
    try:
        os.environ["OMP_NUM_THREADS"] = str(cores)
        omp = Ref(openmp=True)
        omp.generate(logo, 3, 64, 64, autoframe(logo.bounds(), 64, 64, 4))          # spin the OpenMP team up
        t0 = time.perf_counter()
        omp.generate(logo, 3, 1024, 1024, logo_xf)
        line["openmp_rows_s"], line["openmp_threads"] = round(time.perf_counter()-t0, 2), cores
    except (OSError, FileNotFoundError) as e:
        line["openmp_rows_s"] = None
        line["openmp_note"] = str(e)[:120]
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
