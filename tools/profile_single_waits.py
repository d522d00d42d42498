"""Where the distance phase of k_single_call spends a wavefront's cycles (measurement build: -DMSDF_PROFILE_WAITS -> variants/profwaits.so).
    MSDFGEN_HIP_LIB=$PWD/variants/profwaits.so python tools/profile_single_waits.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import msdfgen_amd as M
    from msdfgen_amd import lib as L
    from msdfgen_amd.shape import ShapeBatch
    M.init(0)
    lib = L.load()
    z = np.load(os.path.join(ROOT, "tests", "golden", "latin.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), z["inverse_y"], [str(n) for n in z["names"]])
    shapes = [batch.shape(g) for g in range(batch.n_glyphs)]
    out = np.zeros((64, 64, 3), np.float32)
    for g in range(8):
        M.generate_msdf(out, shapes[g], M.SDFTransformation.from_xf(z["xf64"][g]))
    prof = (C.c_ulonglong*24)()
    lib.msdfhip_debug_wait_profile(prof, 1)
    n = 0
    for rep in range(3):
        for g in range(batch.n_glyphs):
            M.generate_msdf(out, shapes[g], M.SDFTransformation.from_xf(z["xf64"][g]))
            n += 1
    lib.msdfhip_debug_wait_profile(prof, 0)
    v = [int(x) for x in prof]
    waves = max(v[0], 1)
    us = lambda cyc: round(cyc/2400., 2)
    print(json.dumps({"calls": n, "waves_per_call": v[0]/n, "us_per_wave": us(v[1]/waves), "phase1_us": us(v[2]/waves), "phase1_header_us": us(v[18]/waves), "phase1_pass_a_us": us(v[19]/waves),
                      "phase2_us": us(v[10]/waves), "record_batches_per_wave": round(v[4]/waves, 2), "us_per_record_batch": us(v[3]/max(v[4], 1)),
                      "curve_batches_per_wave": round(v[6]/waves, 2), "us_per_curve_batch": us(v[5]/max(v[6], 1)), "evaluations_per_wave": round(v[8]/waves, 2),
                      "us_per_evaluation": us(v[7]/max(v[8], 1)), "us_per_relevance_test": us(v[9]/max(v[4], 1)), "contour_walks_us": us(v[13]/waves),
                      "bookkeeping_us": us(v[14]/waves), "second_walks_us": us(v[15]/waves), "epilogue_us": us(v[16]/waves), "tiles_total_us": us(v[17]/waves)}))


if __name__ == "__main__":
    main()
