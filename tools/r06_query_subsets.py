"""What k_ec_query's launch waits for: time of the distance checks (generate() with the default correction minus generate() with DO_NOT_CHECK_DISTANCE) on subsets of the
distinct DejaVu set cut by edge count.   [MSDFHIP_QUERY_GRID=n] python tools/r06_query_subsets.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import msdfgen_amd as M
    from msdfgen_amd.shape import ShapeBatch
    M.init(0)
    z = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
    xfs = z["xf64"]
    gco, co = batch.glyph_contour_offsets, batch.contour_offsets
    nE = np.array([co[gco[g+1]]-co[gco[g]] for g in range(batch.n_glyphs)])

    def timed(gb, desc, out, cfg, reps=10):
        for _ in range(2):
            gb.generate(3, 64, 64, descriptors=desc, out=out, config=cfg)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            gb.generate(3, 64, 64, descriptors=desc, out=out, config=cfg)
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b)/reps

    for lo, hi in ((0, 24), (0, 48), (0, 64), (0, 128), (0, 100000), (49, 100000), (49, 128), (129, 100000)):
        idx = [int(g) for g in np.nonzero((nE >= lo) & (nE <= hi))[0]]
        sub = batch.select(idx)
        gb = M.GlyphBatch(sub)
        out = torch.empty((len(idx), 64, 64, 3), dtype=torch.float32, device="cuda")
        desc = gb.descriptors(xfs[idx], 64, 64, 3)
        t_on = timed(gb, desc, out, M.MSDFGeneratorConfig())
        cand = gb.candidate_counts()[1]
        t_nocheck = timed(gb, desc, out, M.MSDFGeneratorConfig(True, M.ErrorCorrectionConfig(M.EC_EDGE_PRIORITY, M.DO_NOT_CHECK_DISTANCE)))
        print(json.dumps({"edges": [lo, hi], "glyphs": len(idx), "candidates": int(np.sum(cand)), "ms_step": round(t_on, 4), "ms_distance_checks": round(t_on-t_nocheck, 4)}))
        gb.close()


if __name__ == "__main__":
    main()
