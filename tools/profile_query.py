"""Where a wavefront of k_ec_query (the deferred distance checks of the error correction) spends its cycles: measurement build
-DMSDF_PROFILE_QUERY -> variants/profquery.so, s_memtime stamps (after a full s_waitcnt) around the ticket, the glyph lookup, the
per-glyph loads, the evaluation and the stores of every work item; one launch per line.

    MSDFGEN_HIP_LIB=$PWD/variants/profquery.so python tools/profile_query.py [bench|cjk|logo]
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
    elif which == "logo":
        z = np.load(os.path.join(ROOT, "tests", "golden", "logo1024.npz"))
        from msdfgen_amd.shape import FlatShape
        shape = FlatShape(z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32), z["colors"].astype(np.int32))
        batch, xfs, size = ShapeBatch.from_shapes([shape]), z["xf"][None], 1024
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
        batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                           z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
        xfs, size = z["xf64"], 64
    gb = M.GlyphBatch(batch)
    for _ in range(2):
        gb.generate(3, size, size, xfs)
    torch.cuda.synchronize()
    out = (C.c_ulonglong*24)()
    lib.msdfhip_debug_wait_profile(out, 1)
    gb.generate(3, size, size, xfs)
    torch.cuda.synchronize()
    lib.msdfhip_debug_wait_profile(out, 0)
    v = [int(x) for x in out]
    waves, busy = max(v[0], 1), max(v[2], 1)
    items = max(v[5]+v[7], 1)
    span = v[4]-v[3]
    acc = max(v[9]+v[10]+v[11]+v[12]+v[13], 1)
    line = {"workload": which, "waves": v[0], "waves_with_work": v[2], "kernel_span_cycles": span, "last_work_ends_at_frac_of_span": (v[16]-v[3])/max(span, 1),
            "wave_cycles_sum_over_span_x_waves_with_work": v[1]/max(span*busy, 1),
            "chunk_items": v[5], "cycles_per_chunk_item": v[6]/max(v[5], 1), "cooperative_items": v[7], "cycles_per_cooperative_item": v[8]/max(v[7], 1),
            "cooperative_rounds_per_item": v[15]/max(v[7], 1), "longest_item_cycles": v[14], "longest_item_over_span": v[14]/max(span, 1),
            "items_per_wave_with_work": items/busy, "longest_wave_cycles": v[18], "mean_wave_cycles": v[1]/waves,
            "frac_ticket": v[9]/acc, "frac_lookup": v[10]/acc, "frac_glyph_state": v[11]/acc, "frac_evaluation": v[12]/acc, "frac_store_barrier": v[13]/acc,
            "cycles_per_item": {"ticket": v[9]/items, "lookup": v[10]/items, "glyph_state": v[11]/items, "evaluation": v[12]/items, "store_barrier": v[13]/items}}
    lib.msdfhip_debug_wait_profile(out, 2)                          # second page: inside the cooperative distance query
    d = [int(x) for x in out]
    q = max(d[0], 1)
    line["item_cycles_histogram_16k_32k_64k_128k_256k_more"] = d[10:16]
    line["longest_item"] = {"cycles": (v[17] >> 24) << 8, "glyph": (v[17] & 0xffffff) >> 1, "chunk": v[17] & 1}
    g = (v[17] & 0xffffff) >> 1
    if g < batch.n_glyphs:
        gco, co = batch.glyph_contour_offsets, batch.contour_offsets
        line["longest_item"].update({"contours": int(gco[g+1]-gco[g]), "edges": int(co[gco[g+1]]-co[gco[g]])})
    line["cooperative_query"] = {"queries": d[0], "with_slots": d[9], "contours_per_query": d[8]/q, "cycles": d[1]/q, "all_edge_evaluation": d[2]/q, "per_contour_slot_merges": d[3]/q,
                                 "contour_walks_pass0": d[4]/q, "bookkeeping": d[5]/q, "second_walks": d[6]/q, "epilogue": d[7]/q}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
