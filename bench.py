#!/usr/bin/env python
"""bench.py -- MSDF glyphs/s (64x64, fp32 tiles) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--glyphs G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

`--gpus N` with N > 1 and no torchrun environment re-launches itself under torch.distributed.run with N ranks (one per GPU) and fails
loudly when fewer than N devices are visible, so `python bench.py --gpus 8` is a complete command.

Workload (BASELINE.json metric "MSDF glyphs/sec (64x64, fp32)"): 8 192 DISTINCT glyphs per GPU -- the first 8 192 glyphs with outlines
of DejaVuSans followed by DejaVuSans-Bold (tests/golden/dejavu8192.npz: prepared by the reference's Shape::normalize +
edgeColoringSimple; 23.1 edges and 2.41 contours per glyph, up to 543 edges / 43 contours; SURVEY.md 8d: Roboto / NotoSansCJK are not
on the box) -- mode msdf, 64x64 tiles, 4 px range, library-default config: overlapSupport = true, error correction EDGE_PRIORITY +
CHECK_DISTANCE_AT_EDGE (generateMSDF incl. the msdfErrorCorrection pass). Every tile of this workload is pinned bit-for-bit to the
compiled reference (tests/test_gpu_fullsize.py).
A step = one pass of the hot path over one batch: on-device digestion of the HBM-resident edge buffer -> distance-field kernels ->
error-correction kernels -> G tiles in HBM. Inputs (flattened edge buffer, per-glyph transforms) are resident in HBM before the timed
region; outputs stay in HBM. Scaling is weak: the global list is N rotated copies of the glyph list, cut into N contiguous shards of
equal cost by msdfgen_amd.shard (rank r renders its own shard; no collective in the data path); value = all glyphs / max-over-ranks time.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  roofline      dominant pass = the distance field (HIP-event timed inside this process, algorithmic bytes per SURVEY.md 8d)
  end_to_end    host CSR arrays -> H2D -> kernels incl. error correction -> D2H into pinned caller-owned tiles (the reference's contract,
                core/msdfgen.cpp:52-76), and the same with 8-bit output; measured after the timed steps
  secondary     the Basic-Latin set (94 shapes tiled to G glyphs): round 1's workload, for continuity
  cpu_baseline  the compiled reference (or the oracle port) on the host cores, bounded sample
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_GFLOPS = 78600.0  # MI355X fp64 vector peak (FMA = 2 flop); the path has no dense contraction, so no MFMA


def _batch(z, inverse_y=None):
    from msdfgen_amd.shape import ShapeBatch
    n = len(z["names"])
    return ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                      z["colors"].astype(np.int32), np.zeros(n, bool) if inverse_y is None else inverse_y, [str(v) for v in z["names"]])


def load_latin():
    z = np.load(os.path.join(ROOT, "tests", "golden", "latin.npz"))
    return _batch(z, z["inverse_y"]), z["xf64"]


def load_dejavu():
    z = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
    return _batch(z), z["xf64"], z["bounds"]


def tile_batch(batch, xfs, n, offset=0):
    idx = [(offset+i) % batch.n_glyphs for i in range(n)]
    return batch.select(idx), xfs[idx]


def global_list(batch, xfs, glyphs_per_gpu, world):
    """The weak-scaling workload of `world` GPUs: world*glyphs_per_gpu glyphs = the glyph list repeated with a rotation per copy (so
    that equal ranges are not identical ranges). Returned as index list into `batch`."""
    idx = []
    for r in range(world):
        rot = (r*1237) % batch.n_glyphs
        idx.extend(((rot+i) % batch.n_glyphs) for i in range(glyphs_per_gpu))
    return np.array(idx)


def rank_shard(batch, xfs, glyphs_per_gpu, world, rank, w, h):
    """(sub-batch, xfs, (lo, hi), bounds) of `rank`: contiguous cut of the global list into ranges of equal modelled cost (msdfgen_amd.shard:
    per-class cost model fitted to measured kernel times, profiles/r06_cost_model.json)."""
    from msdfgen_amd.shard import partition_contiguous, glyph_costs
    idx = global_list(batch, xfs, glyphs_per_gpu, world)
    bounds = partition_contiguous(glyph_costs(batch, w, h)[idx], world)
    lo, hi = int(bounds[rank]), int(bounds[rank+1])
    return batch.select(idx[lo:hi]), xfs[idx[lo:hi]], (lo, hi), bounds


def config4_sets():
    """BASELINE.json configs[3] AS STATED: ONE 8 192-glyph atlas, msdf 48x48, glyph-sharded over the GPUs (strong scaling). NotoSansCJK is not
    on the box; two stand-ins: the CJK-like synthetic set (512 distinct many-contour shapes x 16: 82.6 edges / 13.7 contours per glyph, the
    class that runs in the global workspace; pinned by tests/golden/cjk512.npz) and the real-font set (8 192 distinct DejaVu glyphs)."""
    from msdfgen_amd import synth
    from msdfgen_amd.shape import ShapeBatch, autoframe
    base = [synth.cjk_like_shape(20000+i) for i in range(512)]
    cj = ShapeBatch.from_shapes([base[i % 512] for i in range(8192)])
    cx = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])[np.arange(8192) % 512]
    z = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
    return {"cjk_like": (cj, cx), "dejavu": (_batch(z), z["xf48"])}


def step_ms(M, torch, lib, dev, stream, batch, xfs, w, h, cfg, steps, warmup=2):
    """ms per step (digest + distance + correction, HBM resident) of one glyph list alone on the device."""
    gb = M.GlyphBatch(batch, dev)
    out = torch.empty((batch.n_glyphs, h, w, 3), dtype=torch.float32, device=dev)
    elapsed, kd, kc, _ = timed_steps(M, torch, None, lib, gb, gb.descriptors(xfs, w, h, 3), out, cfg, w, h, steps, warmup, 1, dev, stream)
    gb.close()
    return 1e3*elapsed/steps, kd, kc


def strong_scaling_one_gpu(M, torch, lib, dev, stream, cfg, steps=6, parts=(2, 4, 8), only=None):
    """Strong scaling of config 4 REHEARSED ON ONE GPU: the 8 192-glyph set cut into N shards (msdfgen_amd.shard, what `--strong --gpus N` gives
    rank r), every shard timed ALONE on this device.  efficiency(N) = T(whole set) / (N x max_r T(shard r)) -- what N such GPUs would deliver
    relative to N times one GPU, PCIe / host effects aside.  xN: contiguous ranges of equal modelled cost (the default of --strong); dealt_x8: the
    8-way cut with the glyphs dealt out by modelled cost (--strong-cut dealt: every shard the same mix -- measured no better)."""
    from msdfgen_amd.shard import shard_indices
    res = {}
    for name, (batch, xfs) in config4_sets().items():
        if only and name != only:
            continue
        # (the whole set twice, 2x the steps, the faster run counts: a single short run of it right behind another workload read 3.9 / 5.0 / 5.6 ms on three boxes
        # where tools/bench_configs.py measures 3.7-3.8 -- and an inflated T(8192) flatters every efficiency below)
        whole, kd, kc = min((step_ms(M, torch, lib, dev, stream, batch, xfs, 48, 48, cfg, 2*steps, warmup=3) for _ in range(2)), key=lambda r: r[0])
        row = {"ms_whole_set": round(whole, 3), "glyphs_per_s_1gpu": round(batch.n_glyphs/whole*1e3), "kernel_ms": {"distance": round(kd, 3), "error_correction": round(kc, 3)}}
        for cut, n in [("contiguous", n) for n in parts]+[("dealt", 8)]:
            lists = shard_indices(batch, n, 48, 48, cut)
            t = [step_ms(M, torch, lib, dev, stream, batch.select(ix), xfs[ix], 48, 48, cfg, steps)[0] for ix in lists]
            row[("x%d" if cut == "contiguous" else "dealt_x%d") % n] = {
                "efficiency": round(whole/(n*max(t)), 3), "ms_per_shard": [round(v, 3) for v in t], "glyphs_per_shard": [len(ix) for ix in lists],
                "projected_glyphs_per_s": round(batch.n_glyphs/max(t)*1e3)}
        res[name] = row
    res["note"] = ("BASELINE config 4 as stated (ONE 8192-glyph 48x48 msdf atlas over N GPUs), rehearsed on one GPU: every shard timed alone on this device; "
                   "efficiency = T(8192) / (N * max_r T(shard r)); xN = contiguous ranges of equal modelled cost (default of --strong), dealt_x8 = glyphs dealt out by modelled cost; "
                   "measured multi-GPU: bench.py --strong --gpus N")
    return res


def two_batches_in_flight(M, torch, dev, batch, xfs, w, h, cfg, steps):
    """Supplementary, NOT `value`: the same step with TWO batch objects (inputs, outputs and scratch of their own) on two streams, steps alternating -- step k+1's
    digest and distance pass start under step k's tail (the distance checks' launch and the stream joins leave part of the device idle). What an atlas service
    that renders batch after batch would see; every step still does all of its work on its own buffers. tools/r06_overlap.py is the experiment behind it."""
    parts = []
    for _ in range(2):
        gb = M.GlyphBatch(batch, dev)
        parts.append((gb, gb.descriptors(xfs, w, h, 3), torch.empty((batch.n_glyphs, h, w, 3), dtype=torch.float32, device=dev), torch.cuda.Stream(dev)))

    def run(n):
        for k in range(n):
            gb, desc, out, s = parts[k % 2]
            gb.digest(s)
            gb.generate(M.MODE_MSDF, w, h, descriptors=desc, out=out, stream=s, config=cfg)
    run(4)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize(dev)
    ms = 1e3*(time.perf_counter()-t0)/steps
    same = bool((parts[0][2] == parts[1][2]).all())
    for gb, _, _, _ in parts:
        gb.close()
    return {"ms_per_step": ms, "glyphs_per_s": batch.n_glyphs/ms*1e3, "steps": steps, "tiles_of_the_two_batches_identical": same,
            "note": "supplementary (not `value`): two batch objects on two streams, steps alternating, so that a step's low-occupancy tail overlaps the next step's start"}


def algorithmic_bytes(batch, w, h, n):
    """per glyph W*H*N*4 (texels written once) + 72*E (each edge read once) + 48 (transform, mapping, dims)."""
    return batch.n_glyphs*(w*h*n*4+48)+72*batch.n_edges


def algorithmic_flops(batch, w, h):
    """SURVEY.md 8(d) estimate: W*H*(30*E_lin + 300*E_quad + 900*E_cub + 60*C + 40) fp64 flop per glyph."""
    t = batch.types
    return float(w*h)*(30.*(t == 1).sum()+300.*(t == 2).sum()+900.*(t == 3).sum()+60.*batch.n_contours+40.*batch.n_glyphs)


def available_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box reports 256 hardware threads
    but runs the container under cpu.max = 16 CPUs; more threads than that only get throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota)/int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota//period))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(batch, xfs, w, h, budget_s=15.):
    """Reference (or oracle port) on the host cores, glyph-parallel thread pool, on a bounded sample of the same workload."""
    from oracle.pyoracle import Oracle, Ref
    cores = available_cores()
    try:
        impl = Ref() if Ref.available() else Oracle()
    except Exception:  # noqa: BLE001
        impl = Oracle()
    stride = max(1, batch.n_glyphs//1024)
    pick = list(range(0, batch.n_glyphs, stride))                    # an evenly spaced sample of the distinct glyphs (same mix of contours / edges)
    shapes, sx = [batch.shape(g) for g in pick], xfs[pick]
    probe = min(len(shapes), 4*cores)
    _, secs = impl.generate_batch_timed(shapes[:probe], 3, w, h, sx[:probe], threads=cores)
    rate = probe/max(secs, 1e-9)
    n = int(min(max(probe, rate*budget_s), 200000))
    idx = [i % len(shapes) for i in range(n)]
    _, secs = impl.generate_batch_timed([shapes[i] for i in idx], 3, w, h, sx[idx], threads=cores)
    return {"value": n/secs, "unit": "glyphs/s", "cores": cores, "kind": impl.kind,
            "sample": "%d glyphs of the same workload (every %d-th of the distinct glyphs, cycled; msdf %dx%d, default error correction) through %s, "
                      "glyph-parallel thread pool on %d threads (= the CPUs the container may use: %d hardware threads visible, cgroup quota applied), %.1f s" % (
                          n, stride, w, h, "the compiled reference (oracle/_ref)" if impl.kind == "reference" else "the plain-C oracle", cores, os.cpu_count() or 0, secs)}


def profile_counters(args, w, h):
    """Counters of the dominant pass from the separate rocprofv3 --pmc passes of this very command (tools/profile_round.sh; counters cannot be
    collected from inside the timed process): profiles/pmc_traffic.json = HBM bytes (FETCH_SIZE / WRITE_SIZE) and the calibrated VALU busy
    fraction of the distance kernels (tools/pmc_report.py: per-class instruction counts x measured cycles per class). Only used when the
    committed profile is of this exact workload; the entry says at which commit it was measured."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(path))
        if t.get("glyphs_per_gpu") == args.glyphs and t.get("tile") == [w, h] and t.get("workload") == "dejavu8192" and not args.strong:
            from msdfgen_amd.build import source_hash
            if t.get("source_hash") == source_hash():               # counters of OTHER kernels than the ones timed here are not quoted
                return t
    except (OSError, ValueError, KeyError):
        pass
    return None


def timed_steps(M, torch, dist, lib, gb, desc, out, cfg, w, h, steps, warmup, world, dev, stream):
    import ctypes as C

    def step():
        gb.digest(stream)
        gb.generate(M.MODE_MSDF, w, h, descriptors=desc, out=out, stream=stream, config=cfg)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        step()
    fence()
    lib.msdfhip_set_kernel_timing(1)
    lib.msdfhip_kernel_timing(None, None, None, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter()-t0
    lib.msdfhip_set_kernel_timing(0)
    kd, kc, kn = C.c_double(), C.c_double(), C.c_int()
    lib.msdfhip_kernel_timing(C.byref(kd), C.byref(kc), C.byref(kn), 1)
    return elapsed, kd.value, kc.value, kn.value


def end_to_end(M, batch, xfs, w, h, reps=5):
    """The reference's contract (caller-owned host bitmaps): HOST CSR arrays -> msdfhip_batch_create (H2D + digestion) ->
    msdfhip_batch_generate_host (chunked two-stream pipeline: kernels incl. error correction overlapped with the D2H of the previous
    chunk) into PINNED host tiles; and the 8-bit atlas variant (float tiles stay on the device, 1/4 of the D2H bytes)."""
    n = batch.n_glyphs
    tiles = M.host_alloc((n, h, w, 3))
    cols = 128
    rows = (n+cols-1)//cols
    atlas = M.host_alloc((rows*h, cols*w, 3), np.uint8)
    offs = np.array([((g//cols)*h*cols*w+(g % cols)*w)*3 for g in range(n)], np.int64)
    res = {}
    t_create, t_float, t_bytes = [], [], []
    for rep in range(reps+1):
        t0 = time.perf_counter()
        hb = M.HostBatch(batch)
        t1 = time.perf_counter()
        hb.generate_host(M.MODE_MSDF, w, h, xfs, out=tiles)
        t2 = time.perf_counter()
        hb.generate_bytes_host(M.MODE_MSDF, w, h, xfs, atlas, offs, cols*w*3)
        t3 = time.perf_counter()
        hb.close()
        if rep:                                                      # the first round pays one-time allocations
            t_create.append(t1-t0), t_float.append(t2-t1), t_bytes.append(t3-t2)
    c, f, b = float(np.median(t_create)), float(np.median(t_float)), float(np.median(t_bytes))
    h2d = 72*batch.n_edges+4*(batch.n_contours+batch.n_glyphs+2)+64*n
    res["float_tiles"] = {"glyphs_per_s": n/(c+f), "glyphs_per_s_excluding_upload": n/f, "ms_upload_and_digest": 1e3*c, "ms_generate_and_copy_back": 1e3*f,
                          "h2d_bytes": h2d, "d2h_bytes": int(tiles.nbytes), "d2h_gb_per_s_incl_kernels": tiles.nbytes/f/1e9}
    res["uint8_atlas"] = {"glyphs_per_s": n/(c+b), "glyphs_per_s_excluding_upload": n/b, "ms_generate_convert_and_copy_back": 1e3*b,
                          "h2d_bytes": h2d, "d2h_bytes": int(atlas.nbytes), "atlas": [int(atlas.shape[1]), int(atlas.shape[0])]}
    fl = flatten_ms(batch)
    if fl:
        best = min(fl.values())
        res["ms_flatten"] = {"per_pass_over_all_glyphs": fl, "note": "const msdfgen::Shape & -> CSR (the C++ shim's per-call flatten) over the same %d glyphs, real "
                             "msdfgen::Shape objects; NOT inside the two figures above (they start from CSR arrays) -- with it: float tiles %.0f, 8-bit atlas %.0f glyphs/s" % (
                                 n, n/(c+f+best*1e-3), n/(c+b+best*1e-3))}
    res["note"] = ("host CSR arrays in pageable memory, outputs in pinned memory (msdfhip_host_alloc), %d glyphs, median of %d runs; chunks of the glyph "
                   "list rotate through three streams (two chunks' kernels at a time) so that kernels overlap the copies back" % (n, reps))
    M.host_free(tiles)
    M.host_free(atlas)
    return res


def inprocess(args):
    """ONE process driving N GPUs: msdfhip_generate_sharded over devices 0..N-1 (one host thread + two streams per device, every device
    copying its rectangles straight into the caller's pinned buffer) on the same weak-scaling list as the multi-process
    bench. End to end by construction: host CSR arrays in, host tiles out. N = 1 is the single-GPU end_to_end figure."""
    import torch
    import msdfgen_amd as M
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    devices = [0]*args.gpus if args.same_device else list(range(args.gpus))
    if not args.same_device and have < args.gpus:
        raise SystemExit("bench.py --gpus %d --inprocess: only %d GPU(s) visible" % (args.gpus, have))
    M.init(0)
    w = h = args.size
    dejavu, xf64, bounds = load_dejavu()
    idx = global_list(dejavu, xf64, args.glyphs, args.gpus)
    batch, xfs = dejavu.select(idx), xf64[idx]
    n = batch.n_glyphs
    tiles = M.host_alloc((n, h, w, 3))
    cols = 128
    atlas = M.host_alloc((((n+cols-1)//cols)*h, cols*w, 3), np.uint8)
    offs = np.array([((g//cols)*h*cols*w+(g % cols)*w)*3 for g in range(n)], np.int64)
    t_float, t_bytes = [], []
    for rep in range(args.warmup+args.steps):
        t0 = time.perf_counter()
        M.generate_sharded(devices, batch, M.MODE_MSDF, w, h, xfs, out=tiles)
        t1 = time.perf_counter()
        M.generate_sharded(devices, batch, M.MODE_MSDF, w, h, xfs, atlas=atlas, out_offsets=offs, row_stride=cols*w*3)
        t2 = time.perf_counter()
        if rep >= args.warmup:
            t_float.append(t1-t0), t_bytes.append(t2-t1)
    f, b = float(np.median(t_float)), float(np.median(t_bytes))
    print(json.dumps({"metric": "MSDF glyphs/sec (64x64, fp32), end to end, one process driving the GPUs", "value": n/f, "unit": "glyphs/s", "n_gpus": args.gpus,
                      "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3*f, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
                      "data": "synthetic",
                      "config": {"workload": "msdfhip_generate_sharded: %d x %d distinct DejaVu glyphs msdf %dx%d, library-default config; host CSR arrays -> per-device "
                                             "upload + digest + kernels + copy back into ONE pinned host buffer" % (args.gpus, args.glyphs, w, h),
                                 "devices": devices, "parallelism": "one process, one host thread + two streams per device, contiguous ranges of equal modelled cost, no exchange"},
                      "uint8_atlas": {"glyphs_per_s": n/b, "ms_per_step": 1e3*b},
                      "note": "median of %d runs after %d warm-up runs%s" % (args.steps, args.warmup, "; REHEARSAL: every 'device' is GPU 0" if args.same_device else "")}))
    M.host_free(tiles)
    M.host_free(atlas)


def streamed_end_to_end(M, batch, xfs, w, h, reps=7):
    """SURVEY.md 8(d)'s metric as ONE pipelined call. (a) From REAL msdfgen::Shape objects: tests/shim/shim_check (a client of msdfgen's public headers linked
    against the C++ shim; built in the authoring container, it travels to the GPU box) rebuilds the Shape objects of this workload and times
    msdfgen_hip::generateMSDFBatch() -- host threads flatten chunk k+1 / k+2 into pinned staging while chunk k is uploaded, digested and rendered and chunk k-1
    travels back; the clock starts at `const Shape *const *` and stops when the caller's pinned bitmaps are complete. (b) msdfhip_generate_stream_csr from this
    process: the same pipeline fed from host CSR arrays (no Shape walk)."""
    import tempfile
    n = batch.n_glyphs
    res = {}
    exe = os.path.join(ROOT, "tests", "shim", "shim_check")
    if os.path.exists(exe):
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
            path = f.name
        try:
            batch.dump(path, xfs)
            env = dict(os.environ)
            env.setdefault("GPU_MAX_HW_QUEUES", "8")
            r = subprocess.run([exe, "e2e", path, str(w), str(h), str(reps)], capture_output=True, text=True, timeout=300, env=env)
            if r.returncode == 0 and r.stdout.strip():
                d = json.loads(r.stdout.strip().splitlines()[-1])
                res["from_shape_objects"] = {"float_tiles_glyphs_per_s": d["float_tiles_glyphs_per_s"], "uint8_atlas_glyphs_per_s": d["uint8_atlas_glyphs_per_s"],
                                             "float_tiles_ms": d["float_tiles_ms"], "uint8_atlas_ms": d["uint8_atlas_ms"], "reps": d["reps"],
                                             "entry": "msdfgen_hip::generateMSDFBatch(BitmapSection<T,3>[], const Shape *const[], SDFTransformation[], n) -- include/msdfgen_hip_batch.hpp",
                                             "host_threads": available_cores()}
        except (OSError, ValueError, KeyError, subprocess.SubprocessError):
            pass
        finally:
            os.unlink(path)
    tiles = M.host_alloc((n, h, w, 3))
    cols = 128
    atlas = M.host_alloc((((n+cols-1)//cols)*h, cols*w, 3), np.uint8)
    offs = np.array([((g//cols)*h*cols*w+(g % cols)*w)*3 for g in range(n)], np.int64)
    tf, tb = [], []
    for rep in range(reps+1):
        t0 = time.perf_counter()
        M.generate_stream(batch, M.MODE_MSDF, w, h, xfs, out=tiles)
        t1 = time.perf_counter()
        M.generate_stream(batch, M.MODE_MSDF, w, h, xfs, atlas=atlas, out_offsets=offs, row_stride=cols*w*3)
        t2 = time.perf_counter()
        if rep:
            tf.append(t1-t0), tb.append(t2-t1)
    f, b = float(np.median(tf)), float(np.median(tb))
    res["from_csr_arrays"] = {"float_tiles_glyphs_per_s": n/f, "uint8_atlas_glyphs_per_s": n/b, "float_tiles_ms": 1e3*f, "uint8_atlas_ms": 1e3*b, "reps": reps,
                              "entry": "msdfhip_generate_stream_csr (include/msdfgen_hip.h) through ctypes; the figure includes the binding's descriptor preparation in numpy"}
    M.host_free(tiles)
    M.host_free(atlas)
    return res


def flatten_ms(batch):
    """SURVEY 8(d) counts the HOST FLATTEN in the end-to-end metric: const msdfgen::Shape & -> CSR edge buffer, what the C++ shim does per call
    (msdfgen_shim.cpp: flatten). Measured by the shim's own client (tests/shim/shim_check, built against the msdfgen headers in the authoring
    container; it travels to the GPU box): real msdfgen::Shape objects built from this workload, flattened on 1 and on all usable threads."""
    import tempfile
    exe = os.path.join(ROOT, "tests", "shim", "shim_check")
    if not os.path.exists(exe):
        return None
    gco, co = batch.glyph_contour_offsets.astype(np.int32), batch.contour_offsets.astype(np.int32)
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        np.array([len(gco)-1, len(co)-1, int(co[-1])], np.int32).tofile(f)
        gco.tofile(f), co.tofile(f), np.ascontiguousarray(batch.points, np.float64).tofile(f)
        np.ascontiguousarray(batch.types, np.uint8).tofile(f), np.ascontiguousarray(batch.colors, np.uint8).tofile(f)
        path = f.name
    out = {}
    try:
        for threads in (1, available_cores()):
            r = subprocess.run([exe, "flatten", path, str(threads), "7"], capture_output=True, text=True, timeout=120)
            if r.returncode == 0:
                out["threads_%d" % threads] = json.loads(r.stdout.strip().splitlines()[-1])["ms_flatten"]
    except (OSError, ValueError, subprocess.SubprocessError):
        return None
    finally:
        os.unlink(path)
    return out or None


def spawn(args):
    """`python bench.py --gpus N` outside torchrun: re-launch under torch.distributed.run, one rank per GPU."""
    if not args.mock and not args.same_device:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible" % (args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)]+sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def mock_rank(args, rank, world):
    """CPU-only rehearsal of the N > 1 control path (tests): gloo ranks, the real shard computation, barrier + max-reduce, rank 0 prints."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.strong:                                                  # ONE set cut into `world` shards (total work fixed)
        from msdfgen_amd.shard import shard_indices
        whole, wxf = config4_sets()[args.strong_set]
        lists = shard_indices(whole, world, 48, 48, args.strong_cut)
        lo, hi = 0, len(lists[rank])
        bounds = np.concatenate([[0], np.cumsum([len(ix) for ix in lists])])
        sub = whole.select(lists[rank])
    else:
        batch, xfs, _ = load_dejavu()
        sub, sx, (lo, hi), bounds = rank_shard(batch, xfs, args.glyphs, world, rank, args.size, args.size)
    t = torch.tensor([float(sub.n_edges), float(hi-lo)], dtype=torch.float64)
    parts = [torch.zeros_like(t) for _ in range(world)]
    if world > 1:
        dist.barrier()
        dist.all_gather(parts, t)
    else:
        parts = [t]
    if rank == 0:
        print(json.dumps({"mock": True, "n_gpus": world, "scaling": "strong" if args.strong else "weak", "bounds": [int(v) for v in bounds], "glyphs_per_rank": [int(p[1]) for p in parts],
                          "edges_per_rank": [int(p[0]) for p in parts]}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _flush_c_stdio():
    try:
        C.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()


def _silence_stdout():
    """The driver reads ONE JSON line from this command's stdout: ranks other than 0 never write to it (RCCL and gloo print lines of their own per process), and
    rank 0 closes it behind its line."""
    sys.stdout.flush()
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


def _stdout_to_stderr():
    """While the bench runs, whatever the runtimes print on stdout ("[Gloo] Rank 0 is connected ...", "Librccl path : ...") goes to stderr; returns the saved stdout."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return saved


def _restore_stdout(saved):
    _flush_c_stdio()
    os.dup2(saved, 1)
    os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--glyphs", type=int, default=8192, help="glyph tiles per GPU per step")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end / secondary / quality (profiling runs)")
    ap.add_argument("--simple-combiner", action="store_true", help="experiment: overlapSupport=false (NOT the headline config)")
    ap.add_argument("--mock", action="store_true", help="CPU rehearsal of the multi-rank control path (gloo, no kernels); used by the tests")
    ap.add_argument("--same-device", action="store_true", help="rehearsal of the N > 1 path on a ONE-GPU box: every rank uses cuda:0 and the process "
                                                                "group is gloo (RCCL refuses two ranks on one device); the number is not a scaling result")
    ap.add_argument("--strong", action="store_true", help="BASELINE config 4 as stated: ONE 8192-glyph 48x48 msdf atlas (--strong-set) glyph-sharded over "
                                                           "--gpus ranks (strong scaling: total work fixed); with --gpus 1 also the one-GPU rehearsal of the N-way splits")
    ap.add_argument("--strong-set", default="cjk_like", choices=["cjk_like", "dejavu"])
    ap.add_argument("--strong-cut", default="contiguous", choices=["contiguous", "dealt"],
                    help="--strong: contiguous ranges of equal modelled cost, or glyphs in order of modelled cost dealt out to the ranks (every shard the same mix)")
    ap.add_argument("--inprocess", action="store_true", help="one process drives all --gpus devices through msdfhip_generate_sharded (end to end; prints its own JSON line)")
    args = ap.parse_args()

    if args.inprocess:
        return inprocess(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1 and rank != 0 and not args.mock:
        _silence_stdout()
    saved_stdout = _stdout_to_stderr() if rank == 0 and not args.mock else None
    if args.mock:
        return mock_rank(args, rank, world)

    import torch
    import torch.distributed as dist
    import msdfgen_amd as M

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: msdfgen_amd has no CPU compute path")
    if args.same_device:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (only %d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.same_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
    M.init(local_rank)
    lib = M.load()

    w = h = args.size
    if args.strong:
        # config 4 as BASELINE states it: ONE 8 192-glyph set, 48x48, cut into `world` contiguous shards of equal modelled cost (total work fixed)
        from msdfgen_amd.shard import shard_indices
        w = h = 48
        whole, wxf = config4_sets()[args.strong_set]
        mine = shard_indices(whole, world, w, h, args.strong_cut)[rank]
        lo, hi = int(mine[0]) if len(mine) else 0, int(mine[-1])+1 if len(mine) else 0
        batch, xfs = whole.select(mine), wxf[mine]
    else:
        dejavu, xf64, bounds = load_dejavu()
        if w != 64:
            from msdfgen_amd.shape import autoframe
            xf64 = np.stack([autoframe(b, w, h, 4) for b in bounds])
        batch, xfs, (lo, hi), shard_bounds = rank_shard(dejavu, xf64, args.glyphs, world, rank, w, h)
    gb = M.GlyphBatch(batch, dev)
    desc = gb.descriptors(xfs, w, h, 3)
    out = torch.empty((batch.n_glyphs, h, w, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    cfg = M.MSDFGeneratorConfig(overlap_support=not args.simple_combiner)

    elapsed, dist_ms, ec_ms, launches = timed_steps(M, torch, dist, lib, gb, desc, out, cfg, w, h, args.steps, args.warmup, world, dev, stream)
    per_rank = None
    if world > 1:
        cdev = "cpu" if args.same_device else dev
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        # what every rank did, for the line's reader (not part of the timed region): ms per step, glyphs, edges, device index -- one all_gather
        row = torch.tensor([1e3*elapsed/args.steps, batch.n_glyphs, batch.n_edges, local_rank, rank], dtype=torch.float64, device=cdev)
        rows = [torch.zeros_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
        rows = [r.cpu().tolist() for r in rows]
        per_rank = {"backend": dist.get_backend(), "ranks_seen": sorted(int(r[4]) for r in rows), "ms_per_step": [round(r[0], 4) for r in rows],
                    "glyphs": [int(r[1]) for r in rows], "edges": [int(r[2]) for r in rows], "device_index": [int(r[3]) for r in rows]}
        elapsed = float(t.item())

    if rank == 0:
        total_glyphs = (8192 if args.strong else world*args.glyphs)*args.steps
        ab = algorithmic_bytes(batch, w, h, 3)
        achieved = ab/(dist_ms*1e-3)/1e9 if dist_ms > 0 else 0.
        gflops = algorithmic_flops(batch, w, h)/(dist_ms*1e-3)/1e9 if dist_ms > 0 else 0.
        prof = profile_counters(args, w, h)
        if per_rank is not None:
            assert per_rank["ranks_seen"] == list(range(world)), per_rank
        res = {
            "metric": "MSDF glyphs/sec (48x48, fp32), BASELINE config 4: one 8192-glyph atlas glyph-sharded over the GPUs" if args.strong else "MSDF glyphs/sec (64x64, fp32)",
            "value": total_glyphs/elapsed, "unit": "glyphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3*elapsed/args.steps, "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("STRONG scaling, BASELINE config 4: ONE 8192-glyph set (%s) msdf 48x48 cut into %d shards (%s), "
                                    "library-default config; step = digest + distance field + error correction of the rank's shard, HBM resident; value = 8192 glyphs / slowest rank"
                                    % (args.strong_set, world, "glyphs in order of modelled cost dealt out to the ranks" if args.strong_cut == "dealt" else "contiguous ranges of equal modelled cost")) if args.strong else "msdf %dx%d tiles, %d DISTINCT glyphs per GPU per step = the first 8192 glyphs with outlines of DejaVuSans + DejaVuSans-Bold "
                                   "(%.1f edges, %.2f contours per glyph; tests/golden/dejavu8192.npz, every tile pinned to the compiled reference); "
                                   "overlapSupport=true, error correction EDGE_PRIORITY+CHECK_DISTANCE_AT_EDGE (library defaults); "
                                   "step = digest + distance field + error correction, inputs/outputs resident in HBM -- `value` is this RESIDENT step (bench contract); "
                                   "SURVEY 8(d)'s end-to-end rate from Shape objects to caller bitmaps is `end_to_end_metric` of this line" % (
                                       w, h, args.glyphs, batch.n_edges/batch.n_glyphs, batch.n_contours/batch.n_glyphs),
                       "glyphs_per_gpu": batch.n_glyphs if args.strong else args.glyphs, "tile": [w, h], "mode": "msdf",
                       "parallelism": ("glyph-sharded x%d, dealt by modelled cost (msdfgen_amd.shard.partition_dealt), no collective; rank 0 owns %d glyphs of 8192%s" % (
                           world, batch.n_glyphs, " -- REHEARSAL: all ranks on one GPU, gloo" if args.same_device else ""))
                       if args.strong and args.strong_cut == "dealt" else
                       "glyph-sharded x%d into ranges of equal modelled cost (msdfgen_amd.shard), no collective; rank 0 owns glyphs [%d, %d) of %d%s" % (
                           world, lo, hi, 8192 if args.strong else world*args.glyphs, " -- REHEARSAL: all ranks on one GPU, gloo" if args.same_device else "")},
            "roofline": {"bound": "hbm", "kernel": "k_distance<3,...> distance pass (msdf; three launches: 1-contour glyphs / combiner scratch in LDS / in the global workspace)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved/HBM_PEAK_GBS,
                         "traffic": prof["hbm_bytes_per_launch"] if prof else None,
                         "traffic_source": ("profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, measured at commit %s "
                                            "(not collected in this run)" % prof.get("commit", "?")) if prof else None,
                         "algorithmic_bytes_per_launch": ab, "avg_launch_ms": dist_ms, "launches_timed": launches,
                         "note": "arithmetic intensity ~400 fp64 flop/B: the pass is fp64-VALU bound, not HBM bound (SURVEY.md 8d); "
                                 "the HBM fraction is reported as the contract asks, the binding resource is in `valu_fp64`. `traffic`: the overlapping-combiner "
                                 "kernels run at four wavefronts per SIMD (128 VGPRs) and spill OUTSIDE their edge loop; WRITE_SIZE minus the tiles (`non_tile_write_bytes`) is "
                                 "those scratch stores + the global-scratch class's workspace -- two thirds of the spill stores execute in the rare second walks of the "
                                 "combiner (tools/scratch_by_source.py, DESIGN.md 3.1)",
                         "traffic_over_algorithmic": prof.get("traffic_over_algorithmic") if prof else None,
                         "write_over_algorithmic": prof.get("write_over_algorithmic") if prof else None,
                         "non_tile_write_bytes": prof.get("non_tile_write_bytes") if prof else None},
            "valu_fp64": {"achieved": gflops, "peak": FP64_VECTOR_PEAK_GFLOPS, "unit": "GFLOP/s (algorithmic ESTIMATE: SURVEY.md 8d's per-edge flop figures over ALL edges; the "
                                                                                      "kernels cull most of them -- not an achieved rate, see measured_*)", "frac": gflops/FP64_VECTOR_PEAK_GFLOPS,
                          "measured_gflops": prof.get("fp64_gflops_pmc") if prof else None,
                          "measured_frac": (prof["fp64_gflops_pmc"]/FP64_VECTOR_PEAK_GFLOPS if prof and prof.get("fp64_gflops_pmc") else None),
                          "measured_note": prof.get("fp64_note") if prof else None,
                          "valu_other_over_valu_insts": prof.get("valu_other_over_valu_insts") if prof else None,
                          "valu_issue_frac_pmc": prof.get("valu_busy_frac_calibrated") if prof else None,
                          "valu_issue_frac_pmc_per_kernel": prof.get("valu_busy_per_kernel") if prof else None,
                          "valu_issue_frac_pmc_note": prof.get("valu_busy_note") if prof else None},
            "kernel_ms": {"distance": dist_ms, "error_correction": ec_ms},
        }
        if per_rank is not None:
            res["per_rank"] = per_rank
        if not args.no_extras:
            # outside the timed region: the reference-defined quality of what was just rendered (estimateSDFError, core/sdf-error-estimation.h)
            err = gb.estimate_sdf_error(out, xfs)
            res["quality"] = {"metric": "estimateSDFError of the rendered tiles (1 scanline per row, non-zero fill), evaluated on the device",
                              "mean": float(err.mean()), "max": float(err.max())}
    gb.close()
    del out
    if world > 1 and not args.no_extras:
        # every rank runs the host-output pipeline of ITS shard at the same time (host CSR -> H2D -> kernels -> D2H into pinned memory): the
        # contention for PCIe / host memory that the resident number cannot show. Aggregate = all glyphs / slowest rank.
        dist.barrier()
        mine = end_to_end(M, batch, xfs, w, h, reps=2)
        vec = torch.tensor([batch.n_glyphs, mine["float_tiles"]["ms_upload_and_digest"]+mine["float_tiles"]["ms_generate_and_copy_back"],
                            mine["float_tiles"]["ms_upload_and_digest"]+mine["uint8_atlas"]["ms_generate_convert_and_copy_back"]], dtype=torch.float64,
                           device="cpu" if args.same_device else dev)
        parts = [torch.zeros_like(vec) for _ in range(world)]
        dist.all_gather(parts, vec)
        if rank == 0:
            rows = [[float(v) for v in p.cpu()] for p in parts]
            total = sum(r[0] for r in rows)
            res["end_to_end"] = {"float_tiles": {"glyphs_per_s": total/(max(r[1] for r in rows)*1e-3), "ms_per_rank": [round(r[1], 3) for r in rows]},
                                 "uint8_atlas": {"glyphs_per_s": total/(max(r[2] for r in rows)*1e-3), "ms_per_rank": [round(r[2], 3) for r in rows]},
                                 "note": "all %d ranks run msdfhip_batch_create + msdfhip_batch_generate_host / _bytes_host on their shards SIMULTANEOUSLY "
                                         "(barrier, then each rank its own median of 2 runs); aggregate = all glyphs / slowest rank" % world,
                                 "rank0": mine}
    if rank == 0 and world == 1 and not args.no_extras and args.strong:
        res["strong_scaling"] = strong_scaling_one_gpu(M, torch, lib, dev, stream, cfg, steps=max(3, args.steps//4), only=args.strong_set)
    if rank == 0 and world == 1 and not args.no_extras and not args.strong:
        res["end_to_end"] = end_to_end(M, batch, xfs, w, h)
        latin, lxf = load_latin()
        if w != 64:
            from msdfgen_amd.shape import autoframe
            lxf = np.stack([autoframe(b, w, h, 4) for b in np.load(os.path.join(ROOT, "tests", "golden", "latin.npz"))["bounds"]])
        lb, lx = tile_batch(latin, lxf, args.glyphs)
        g2 = M.GlyphBatch(lb, dev)
        o2 = torch.empty((lb.n_glyphs, h, w, 3), dtype=torch.float32, device=dev)
        e2, d2, c2, _ = timed_steps(M, torch, dist, lib, g2, g2.descriptors(lx, w, h, 3), o2, cfg, w, h, max(5, args.steps//3), 2, 1, dev, stream)
        e2e = res["end_to_end"]
        fl = min(e2e["ms_flatten"]["per_pass_over_all_glyphs"].values()) if "ms_flatten" in e2e else None
        ms_u8 = e2e["float_tiles"]["ms_upload_and_digest"]+e2e["uint8_atlas"]["ms_generate_convert_and_copy_back"]
        ms_f32 = e2e["float_tiles"]["ms_upload_and_digest"]+e2e["float_tiles"]["ms_generate_and_copy_back"]
        staged_u8, staged_f32 = batch.n_glyphs/((ms_u8+(fl or 0.))*1e-3), batch.n_glyphs/((ms_f32+(fl or 0.))*1e-3)
        streamed = streamed_end_to_end(M, batch, xfs, w, h)
        res["end_to_end"]["streamed"] = streamed
        src = streamed.get("from_shape_objects")
        res["end_to_end_metric"] = {
            "metric": "MSDF glyphs/sec (64x64) END TO END as SURVEY.md 8(d) defines it: Shape -> CSR flatten + H2D + digest + kernels incl. error correction + D2H into caller bitmaps",
            "uint8_atlas_glyphs_per_s": src["uint8_atlas_glyphs_per_s"] if src else staged_u8, "float_tiles_glyphs_per_s": src["float_tiles_glyphs_per_s"] if src else staged_f32,
            "path": ("ONE pipelined call over real msdfgen::Shape objects, msdfgen_hip::generateMSDFBatch (flatten on the host threads || upload + digest || kernels || copy back, chunk by chunk); "
                     "wall clock of the call, median of %d" % src["reps"]) if src else
                    "staged (flatten, then msdfhip_batch_create, then msdfhip_batch_generate_host): tests/shim/shim_check absent on this box, the streamed C++ entry was not timed",
            "staged": {"uint8_atlas_glyphs_per_s": staged_u8, "float_tiles_glyphs_per_s": staged_f32,
                       "ms": {"flatten": fl, "upload_and_digest": e2e["float_tiles"]["ms_upload_and_digest"], "uint8_generate_convert_copy_back": e2e["uint8_atlas"]["ms_generate_convert_and_copy_back"],
                              "float_generate_copy_back": e2e["float_tiles"]["ms_generate_and_copy_back"]},
                       "note": "round 4's figure: the three stages one after the other (flatten -> msdfhip_batch_create -> msdfhip_batch_generate_host / _bytes_host)"},
            "note": "the 8-bit atlas (pixelFloatToByte on the device, a quarter of the D2H bytes) is what an atlas tool consumes; the float path is PCIe bound "
                    "(%d MB D2H). `value` above is the HBM-resident step as the bench contract asks" % (e2e["float_tiles"]["d2h_bytes"]//1000000)}
        res["strong_scaling"] = strong_scaling_one_gpu(M, torch, lib, dev, stream, cfg, steps=max(3, args.steps//6))
        res["two_batches_in_flight"] = two_batches_in_flight(M, torch, dev, batch, xfs, w, h, cfg, max(6, args.steps))
        res["secondary"] = {"workload": "round 1's bench workload: DejaVuSans Basic-Latin (94 prepared shapes, 15.6 edges / 1.41 contours per glyph) tiled to %d glyphs" % args.glyphs,
                            "glyphs_per_s": args.glyphs*max(5, args.steps//3)/e2, "kernel_ms": {"distance": d2, "error_correction": c2}}
        g2.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(batch, xfs, w, h)
        _restore_stdout(saved_stdout)                                # (everything the runtimes printed meanwhile went to stderr)
        print(json.dumps(res), flush=True)
    if world > 1:
        _silence_stdout()                                            # whatever the runtimes still print at exit must not follow rank 0's line


if __name__ == "__main__":
    main()
