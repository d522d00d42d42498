#!/usr/bin/env python
"""bench.py -- MSDF glyphs/s (64x64, fp32 tiles) on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--glyphs G]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json metric "MSDF glyphs/sec (64x64, fp32)"): the 94 prepared DejaVuSans Basic-Latin glyph shapes
(tests/golden/latin.npz: after Shape::normalize + edgeColoringSimple, 15.6 edges / 1.41 contours per glyph) tiled to G glyphs per
GPU, mode msdf, 64x64 tiles, 4 px range, library-default config: overlapSupport = true, error correction EDGE_PRIORITY +
CHECK_DISTANCE_AT_EDGE (i.e. generateMSDF incl. the msdfErrorCorrection pass).
A step = one pass of the hot path over one batch: on-device digestion of the HBM-resident edge buffer -> distance-field kernel ->
error-correction kernel -> G tiles in HBM.  Inputs (flattened edge buffer, per-glyph transforms) are resident in HBM before the
timed region; outputs stay in HBM.  Scaling is weak: every rank renders its own G glyphs (glyph-sharded, no collective in the data
path); value = N*G*K / max-over-ranks time.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` (dominant kernel = distance field, HIP-event timed
inside this process) and `cpu_baseline` (the compiled reference, or the oracle port, timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_GFLOPS = 78600.0  # MI355X fp64 vector peak (FMA = 2 flop); the path has no dense contraction, so no MFMA


def load_latin():
    from msdfgen_amd.shape import ShapeBatch
    z = np.load(os.path.join(ROOT, "tests", "golden", "latin.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), z["inverse_y"], [str(n) for n in z["names"]])
    return batch, z["xf64"]


def tile_batch(batch, xfs, n, offset=0):
    idx = [(offset+i) % batch.n_glyphs for i in range(n)]
    return batch.select(idx), xfs[idx]


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
    shapes = batch.shapes()[:94*8]                   # the batch is the 94 Basic-Latin shapes tiled: cycle through a slice of it
    probe = min(len(shapes), 4*cores)
    _, secs = impl.generate_batch_timed(shapes[:probe], 3, w, h, xfs[:probe], threads=cores)
    rate = probe/max(secs, 1e-9)
    n = int(min(max(probe, rate*budget_s), 200000))
    idx = [i % len(shapes) for i in range(n)]
    _, secs = impl.generate_batch_timed([shapes[i] for i in idx], 3, w, h, xfs[idx], threads=cores)
    return {"value": n/secs, "unit": "glyphs/s", "cores": cores, "kind": impl.kind,
            "sample": "%d glyphs of the same workload (Basic-Latin shapes cycled, msdf %dx%d, default error correction) through %s, "
                      "glyph-parallel thread pool on %d threads (= the CPUs the container may use: %d hardware threads visible, cgroup quota applied), %.1f s" % (
                          n, w, h, "the compiled reference (oracle/_ref)" if impl.kind == "reference" else "the plain-C oracle", cores, os.cpu_count() or 0, secs)}


def pmc_traffic(args, w, h):
    """HBM bytes per launch of the dominant kernel from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this very
    command (tools/pmc_traffic.py writes profiles/pmc_traffic.json; counters cannot be collected from inside the timed process).
    None when no profile of this exact workload is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(path))
        if t.get("glyphs_per_gpu") == args.glyphs and t.get("tile") == [w, h]:
            return t["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--glyphs", type=int, default=8192, help="glyph tiles per GPU per step")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--simple-combiner", action="store_true", help="experiment: overlapSupport=false (NOT the headline config)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import msdfgen_amd as M

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: msdfgen_amd has no CPU compute path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # nccl == RCCL on ROCm
    M.init(local_rank)

    w = h = args.size
    latin, xf64 = load_latin()
    if w != 64:
        from msdfgen_amd.shape import autoframe
        bounds = np.load(os.path.join(ROOT, "tests", "golden", "latin.npz"))["bounds"]
        xf64 = np.stack([autoframe(b, w, h, 4) for b in bounds])
    # weak scaling: rank r renders its own G glyphs (the global list is rank-major; a static contiguous split gives every rank G)
    batch, xfs = tile_batch(latin, xf64, args.glyphs, offset=rank*args.glyphs)
    gb = M.GlyphBatch(batch, dev)
    desc = gb.descriptors(xfs, w, h, 3)
    out = torch.empty((batch.n_glyphs, h, w, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    cfg = M.MSDFGeneratorConfig(overlap_support=not args.simple_combiner)

    def step():
        gb.digest(stream)
        gb.generate(M.MODE_MSDF, w, h, descriptors=desc, out=out, stream=stream, config=cfg)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    lib = M.load()
    fence()
    lib.msdfhip_set_kernel_timing(1)
    lib.msdfhip_kernel_timing(None, None, None, 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter()-t0
    lib.msdfhip_set_kernel_timing(0)
    import ctypes as C
    kd, kc, kn = C.c_double(), C.c_double(), C.c_int()
    lib.msdfhip_kernel_timing(C.byref(kd), C.byref(kc), C.byref(kn), 1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_glyphs = world*args.glyphs*args.steps
        ab = algorithmic_bytes(batch, w, h, 3)
        dist_ms = kd.value
        achieved = ab/(dist_ms*1e-3)/1e9 if dist_ms > 0 else 0.
        gflops = algorithmic_flops(batch, w, h)/(dist_ms*1e-3)/1e9 if dist_ms > 0 else 0.
        res = {
            "metric": "MSDF glyphs/sec (64x64, fp32)", "value": total_glyphs/elapsed, "unit": "glyphs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3*elapsed/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "msdf %dx%d tiles, %d glyphs per GPU per step = DejaVuSans Basic-Latin (94 prepared shapes, 15.6 edges/glyph) tiled; "
                                   "overlapSupport=true, error correction EDGE_PRIORITY+CHECK_DISTANCE_AT_EDGE (library defaults); "
                                   "step = digest + distance field + error correction, inputs/outputs resident in HBM" % (w, h, args.glyphs),
                       "glyphs_per_gpu": args.glyphs, "tile": [w, h], "mode": "msdf", "parallelism": "glyph-sharded x%d, no collective" % world},
            "roofline": {"bound": "hbm", "kernel": "k_distance<3,true,false> (msdf, overlapping combiner)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved/HBM_PEAK_GBS, "traffic": pmc_traffic(args, w, h),
                         "algorithmic_bytes_per_launch": ab, "avg_launch_ms": dist_ms, "launches_timed": kn.value,
                         "note": "arithmetic intensity ~400 fp64 flop/B: the kernel is fp64-VALU bound, not HBM bound (SURVEY.md 8d); "
                                 "the HBM fraction is reported as the contract asks, the binding resource is in `valu_fp64`"},
            "valu_fp64": {"achieved": gflops, "peak": FP64_VECTOR_PEAK_GFLOPS, "unit": "GFLOP/s (algorithmic estimate, SURVEY.md 8d)", "frac": gflops/FP64_VECTOR_PEAK_GFLOPS},
            "kernel_ms": {"distance": dist_ms, "error_correction": kc.value},
        }
        # outside the timed region: the reference-defined quality of what was just rendered (estimateSDFError, core/sdf-error-estimation.h)
        err = gb.estimate_sdf_error(out, xfs)
        res["quality"] = {"metric": "estimateSDFError of the rendered tiles (1 scanline per row, non-zero fill), evaluated on the device",
                          "mean": float(err.mean()), "max": float(err.max())}
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(batch, xfs, w, h)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
