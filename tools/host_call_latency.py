"""Latency / throughput of the literal drop-in call (single shape, host pointers in and out, PCIe both ways) from one host thread
and from a pool of C++ host threads -- the way msdf-atlas-gen's workers call generateMSDF -- with the library's micro-batcher on
and off.  Reported in DESIGN.md (it is never bench.py's value).

    python tools/host_call_latency.py [--threads 1,4,16,64] [--calls 400]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HERE = os.path.join(ROOT, "tools", "hostbench")
SO = os.path.join(HERE, "libhostbench.so")


def build():
    from msdfgen_amd import build as B
    B.build_lib()
    src = os.path.join(HERE, "hostbench.cpp")
    if not os.path.exists(SO) or os.path.getmtime(src) > os.path.getmtime(SO):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", os.path.join(ROOT, "include"), src, "-o", SO,
                        "-L", B.LIBDIR, "-lmsdfgen_hip", "-Wl,-rpath," + B.LIBDIR], check=True)
    return SO


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", default="1,4,16,64")
    ap.add_argument("--calls", type=int, default=400)
    ap.add_argument("--leaders", default="4,0", help="micro-batcher leader slots per run; 0 = micro-batching off")
    args = ap.parse_args()
    import msdfgen_amd as M
    from msdfgen_amd.shape import distance_mapping
    M.init(0)
    hb = C.CDLL(build())
    hb.hostbench_run.restype = C.c_double
    z = np.load(os.path.join(ROOT, "tests", "golden", "latin.npz"))
    gco = np.ascontiguousarray(z["glyph_contour_offsets"], np.int32)
    co = np.ascontiguousarray(z["contour_offsets"], np.int32)
    pts = np.ascontiguousarray(z["points"], np.float64)
    types = np.ascontiguousarray(z["types"], np.uint8)
    colors = np.ascontiguousarray(z["colors"], np.uint8)
    xfs = np.array([[x[0], x[1], x[2], x[3], *distance_mapping(x[4], x[5])] for x in z["xf64"]], np.float64)
    G = len(gco)-1

    def p(a, t):
        return a.ctypes.data_as(C.POINTER(t))

    def run(nt, calls):
        bad = C.c_int()
        tile = np.zeros((64, 64, 3), np.float32)
        secs = hb.hostbench_run(nt, calls, G, p(gco, C.c_int32), p(co, C.c_int32), p(pts, C.c_double), p(types, C.c_uint8), p(colors, C.c_uint8),
                                p(xfs, C.c_double), 3, 64, 64, C.byref(bad), p(tile, C.c_float))
        assert bad.value == 0, "%d calls failed" % bad.value
        return secs
    for micro in [int(x) for x in args.leaders.split(",")]:
        M.set_microbatch(256 if micro else 1, max(micro, 1))
        run(4, 20)                                                       # warm the arenas
        for nt in [int(t) for t in args.threads.split(",")]:
            M.microbatch_stats(reset=True)
            secs = run(nt, args.calls)
            st = M.microbatch_stats()
            ph = (C.c_double*8)()
            M.load().msdfhip_debug_single_call_phases(ph, 1)
            print(json.dumps({"leaders": micro, "host_threads": nt, "calls": nt*args.calls, "us_per_call_per_thread": round(1e6*secs/args.calls, 1),
                              "glyphs_per_s": round(nt*args.calls/secs), "device_batches": st["batches"], "largest_group": st["largest"],
                              "us_per_batch": {k: round(1e3*st[k+"_ms"]/max(st["batches"], 1), 1) for k in ("stage", "device", "scatter")},
                              "fused_single_calls": int(ph[0]), "fused_shader_mhz": round((ph[0]-int(ph[0]))*1e6), "fused_phase_us": dict(zip(("digest", "distance_wg0", "wait_all_tiles", "sweep_wg0", "candidates_visible", "checks_wg0", "first_to_last"),
                                                                                        [round(v, 2) for v in list(ph)[1:]]))}), flush=True)


if __name__ == "__main__":
    main()
