"""Is k_ec_query's launch as long as its heaviest glyphs? Correction-pass time (in-library HIP events: everything behind the distance pass) of the bench workload without
its glyphs above an edge bound -- k_ec_fast does not notice a few glyphs less, the distance checks' critical path does.   python tools/r06_query_without.py"""
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
    from msdfgen_amd import lib as L
    from bench import load_dejavu
    M.init(0)
    lib = L.load()
    batch, xfs, _ = load_dejavu()
    gco, co = batch.glyph_contour_offsets, batch.contour_offsets
    nE = np.array([co[gco[g+1]]-co[gco[g]] for g in range(batch.n_glyphs)])
    names = [str(n) for n in batch.names]
    cases = [(100000, None)]+[(100000, n) for n in sys.argv[1:]]+([] if len(sys.argv) > 1 else [(b, None) for b in (400, 256, 128, 64, 48)])
    for bound, without in cases:
        idx = [int(g) for g in np.nonzero(nE <= bound)[0] if names[g] != without]
        gb = M.GlyphBatch(batch.select(idx))
        out = torch.empty((len(idx), 64, 64, 3), dtype=torch.float32, device="cuda")
        desc = gb.descriptors(xfs[idx], 64, 64, 3)
        res = []
        for rep in range(3):
            for _ in range(2):
                gb.generate(3, 64, 64, descriptors=desc, out=out)
            torch.cuda.synchronize()
            lib.msdfhip_set_kernel_timing(1)
            lib.msdfhip_kernel_timing(None, None, None, 1)
            for _ in range(8):
                gb.generate(3, 64, 64, descriptors=desc, out=out)
            torch.cuda.synchronize()
            lib.msdfhip_set_kernel_timing(0)
            kd, kc, kn = C.c_double(), C.c_double(), C.c_int()
            lib.msdfhip_kernel_timing(C.byref(kd), C.byref(kc), C.byref(kn), 1)
            res.append(round(kc.value, 4))
        print(json.dumps({"max_edges": bound, "without": without, "glyphs": len(idx), "ms_correction_pass": res, "per_1000_glyphs": round(min(res)/len(idx)*1000, 4)}))
        gb.close()


if __name__ == "__main__":
    main()
