"""How the correction pass scales with the glyph count (is k_ec_query a throughput problem or a critical path?): time of generate() with the
default correction minus time with the correction disabled, first N glyphs of the distinct DejaVu set.   python tools/ec_scaling.py"""
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
    rng = np.random.RandomState(5)
    order = rng.permutation(8192)

    def timed(gb, desc, out, cfg, reps=8):
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

    for n in (256, 512, 1024, 2048, 4096, 8192):
        idx = sorted(int(i) for i in order[:n])
        sub = batch.select(idx)
        gb = M.GlyphBatch(sub)
        out = torch.empty((n, 64, 64, 3), dtype=torch.float32, device="cuda")
        desc = gb.descriptors(xfs[idx], 64, 64, 3)
        t_on = timed(gb, desc, out, M.MSDFGeneratorConfig())
        t_off = timed(gb, desc, out, M.MSDFGeneratorConfig(True, M.ErrorCorrectionConfig(M.EC_DISABLED)))
        t_nocheck = timed(gb, desc, out, M.MSDFGeneratorConfig(True, M.ErrorCorrectionConfig(M.EC_EDGE_PRIORITY, M.DO_NOT_CHECK_DISTANCE)))
        print(json.dumps({"glyphs": n, "ms_with_correction": round(t_on, 4), "ms_without": round(t_off, 4), "ms_correction_without_distance_checks": round(t_nocheck-t_off, 4),
                          "ms_correction": round(t_on-t_off, 4), "ms_distance_checks": round(t_on-t_nocheck, 4)}))
        gb.close()


if __name__ == "__main__":
    main()
