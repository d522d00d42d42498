"""the LDS budget / class limits of k_distance swept ON THE CJK-LIKE SET (config 4), with the bench workload and the real-font
config 4 beside it. One process: the knobs are re-read between runs (msdfhip_reload_tuning), every run builds a fresh GlyphBatch (class lists are per batch).

    python tools/lds_class_sweep.py [--reps 4]          one JSON line per (set, knob combination)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_configs import timed, kernel_ms  # noqa: E402

KNOBS = ("MSDFHIP_RES_LDS_BUDGET", "MSDFHIP_SMALL_MAX_EDGES", "MSDFHIP_LDS_CLASS_TPW")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--sets", default="cjk,dejavu64,dejavu48")
    args = ap.parse_args()
    import torch
    import msdfgen_amd as M
    from msdfgen_amd import lib as L, synth
    from msdfgen_amd.shape import ShapeBatch, autoframe
    M.init(0)
    lib = L.load()
    sets = {}
    if "cjk" in args.sets:
        base = [synth.cjk_like_shape(20000+i) for i in range(512)]
        cj = ShapeBatch.from_shapes([base[i % 512] for i in range(8192)])
        cx = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])[np.arange(8192) % 512]
        sets["cjk"] = (cj, cx, 48)
    zd = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
    dj = ShapeBatch(zd["glyph_contour_offsets"].astype(np.int32), zd["contour_offsets"].astype(np.int32), zd["points"], zd["types"].astype(np.int32),
                    zd["colors"].astype(np.int32), np.zeros(len(zd["names"]), bool), [str(n) for n in zd["names"]])
    if "dejavu64" in args.sets:
        sets["dejavu64"] = (dj, zd["xf64"], 64)
    if "dejavu48" in args.sets:
        sets["dejavu48"] = (dj, zd["xf48"], 48)
    combos = [{}]
    for budget in (16, 20, 24, 28, 32, 40, 52):
        for tpw in (4, 1):
            for me in (128, 160):
                combos.append({"MSDFHIP_RES_LDS_BUDGET": str(budget*1024), "MSDFHIP_SMALL_MAX_EDGES": str(me), "MSDFHIP_LDS_CLASS_TPW": str(tpw)})
    combos.append({"MSDFHIP_LDS_CLASS_TPW": "1"})
    for name, (batch, xfs, size) in sets.items():
        for combo in combos:
            if name != "cjk" and combo.get("MSDFHIP_SMALL_MAX_EDGES") == "160" and combo.get("MSDFHIP_RES_LDS_BUDGET") not in ("24576", "32768"):
                continue                                             # (the font sets: a thinner sweep)
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(combo)
            lib.msdfhip_reload_tuning()
            gb = M.GlyphBatch(batch)
            out = torch.empty((batch.n_glyphs, size, size, 3), dtype=torch.float32, device="cuda")
            desc = gb.descriptors(xfs, size, size, 3)

            def step():
                gb.digest()
                gb.generate(3, size, size, descriptors=desc, out=out)
            ms = timed(step, args.reps)
            km = kernel_ms(step)
            print(json.dumps({"set": name, "knobs": combo, "ms_per_step": round(ms, 3), "kernel_ms_distance_and_post": km}), flush=True)
            gb.close()
    for k in KNOBS:
        os.environ.pop(k, None)
    lib.msdfhip_reload_tuning()


if __name__ == "__main__":
    main()
