"""Where a k_distance wavefront's cycles go (measurement build, -DMSDF_PROFILE_WAITS -> variants/profwaits.so): s_memtime stamps around the
hand-placed record loads, the relevance tests and the evaluations, summed per wavefront by the kernel itself.

    MSDFGEN_HIP_LIB=$PWD/variants/profwaits.so python tools/profile_waits.py [bench|latin|cjk]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import msdfgen_amd as M
    from msdfgen_amd import lib as L, synth
    from msdfgen_amd.shape import ShapeBatch, autoframe
    M.init(0)
    lib = L.load()
    which = sys.argv[1] if len(sys.argv) > 1 else "bench"
    if which == "cjk":
        base = [synth.cjk_like_shape(20000+i) for i in range(512)]
        batch = ShapeBatch.from_shapes([base[i % 512] for i in range(8192)])
        xfs, size = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])[np.arange(8192) % 512], 48
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
        batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                           z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
        xfs, size = z["xf64"], 64
        if which in ("lds", "simple"):                                # one glyph class of the bench workload (msdf_capi.hip: dispatchDistance)
            n_c = np.diff(batch.glyph_contour_offsets)
            n_e = batch.contour_offsets[batch.glyph_contour_offsets[1:]]-batch.contour_offsets[batch.glyph_contour_offsets[:-1]]
            pick = np.nonzero((n_c >= 2) & (n_c <= 7) & (n_e <= 128))[0] if which == "lds" else np.nonzero(n_c <= 1)[0]
            batch, xfs = batch.select([int(g) for g in pick]), xfs[pick]
    gb = M.GlyphBatch(batch)
    cfg = M.MSDFGeneratorConfig(True, M.ErrorCorrectionConfig(M.EC_DISABLED))
    for _ in range(2):
        gb.generate(3, size, size, xfs, config=cfg)
    torch.cuda.synchronize()
    out = (C.c_ulonglong*24)()
    lib.msdfhip_debug_wait_profile(out, 1)
    reps = 3
    for _ in range(reps):
        gb.generate(3, size, size, xfs, config=cfg)
    torch.cuda.synchronize()
    lib.msdfhip_debug_wait_profile(out, 0)
    v = [int(x) for x in out]
    waves = max(v[0], 1)
    line = {"workload": which, "glyphs": batch.n_glyphs, "launches": reps, "waves": v[0], "cycles_per_wave": v[1]/waves, "phase1_frac": v[2]/max(v[1], 1), "phase2_frac": v[10]/max(v[1], 1),
            "record_batches_per_wave": v[4]/waves, "cycles_per_record_batch": v[3]/max(v[4], 1), "record_batch_frac_of_wave": v[3]/max(v[1], 1),
            "batches_over_1000_cycles_frac": v[11]/max(v[4], 1), "batches_over_3000_cycles_frac": v[12]/max(v[4], 1),
            "curve_batches_per_wave": v[6]/waves, "cycles_per_curve_batch": v[5]/max(v[6], 1), "curve_batch_frac_of_wave": v[5]/max(v[1], 1),
            "evaluations_per_wave": v[8]/waves, "cycles_per_evaluation": v[7]/max(v[8], 1), "evaluation_frac_of_wave": v[7]/max(v[1], 1),
            "relevance_cycles_per_test": v[9]/max(v[4], 1), "relevance_frac_of_wave": v[9]/max(v[1], 1),
            "contour_walks_frac_of_wave": v[13]/max(v[1], 1), "per_contour_bookkeeping_frac": v[14]/max(v[1], 1), "second_walks_frac": v[15]/max(v[1], 1),
            "combiner_epilogue_frac": v[16]/max(v[1], 1), "tiles_total_frac": v[17]/max(v[1], 1),
            "phase1_header_loads_frac": v[18]/max(v[1], 1), "phase1_pass_a_frac": v[19]/max(v[1], 1), "phase1_pass_b_frac": (v[2]-v[18]-v[19])/max(v[1], 1)}
    print(json.dumps({k: (round(x, 4) if isinstance(x, float) else x) for k, x in line.items()}))


if __name__ == "__main__":
    main()
