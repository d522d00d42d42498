"""Fits the per-glyph cost model the sharding uses (msdfgen_amd.shard.glyph_costs, msdf_capi.hip: shardRanges) to MEASURED kernel times
on one MI355X (VERDICT r2 next #5b), and checks the split it produces.

The 8 192 distinct DejaVu glyphs are binned by (contours, edges); every bin is rendered on its own (msdf 64x64, library-default config =
distance field + error correction, the bench step) and timed with HIP events. Least squares over the bins:

    microseconds per glyph = (W*H/4096) * (a_k + b_k*E + c_k*C + d_k*E*C)      k = the kernel class of the glyph

(E edges, C contours; classes as msdf_capi.hip: ensureBuckets sorts them: one contour -> simple-combiner kernel; 2..5 contours and <= 128
edges -> per-contour distances in LDS; the rest -> global workspace. The overlapping combiner walks every contour's survivors at every
tile, hence the E*C term). Then: the 2-way and 8-way contiguous splits of the glyph list by the OLD cost W*H*(E+1) and by
the fitted one, each part timed on the GPU -- the imbalance is max/mean-1 of the measured part times.

    python tools/fit_cost_model.py > profiles/r06_cost_model.json        (round 6: refitted for the round-5 / 6 kernels -- four wavefronts per SIMD, LDS class up to 5 contours)
"""
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
    from msdfgen_amd.shard import partition_contiguous, glyph_costs, COST_MODEL
    M.init(0)
    z = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
    xfs = z["xf64"]
    gco, co = batch.glyph_contour_offsets, batch.contour_offsets
    n_c = np.diff(gco).astype(np.int64)
    n_e = (co[gco[1:]]-co[gco[:-1]]).astype(np.int64)

    def time_subset(idx, reps=6, fill=True):
        idx = [int(i) for i in idx]
        if fill and len(idx) < 2048:                                     # fill the device: small BINS are tiled (cost per glyph is what is fitted)
            idx = (idx*(2048//len(idx)+1))[:2048]
        sub = batch.select(idx)
        gb = M.GlyphBatch(sub)
        out = torch.empty((sub.n_glyphs, 64, 64, 3), dtype=torch.float32, device="cuda")
        desc = gb.descriptors(xfs[idx], 64, 64, 3)
        for _ in range(2):
            gb.generate(M.MODE_MSDF, 64, 64, descriptors=desc, out=out)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            gb.generate(M.MODE_MSDF, 64, 64, descriptors=desc, out=out)
        b.record()
        torch.cuda.synchronize()
        gb.close()
        return a.elapsed_time(b)/reps, len(idx)

    bins = []
    c_edges = [(1, 1), (2, 2), (3, 3), (4, 5), (6, 8), (9, 14), (15, 64)]         # (4, 5) ends the LDS class, (6, 8) starts the global one
    e_edges = [(1, 12), (13, 20), (21, 32), (33, 56), (57, 100), (101, 600)]
    for clo, chi in c_edges:
        for elo, ehi in e_edges:
            idx = np.nonzero((n_c >= clo) & (n_c <= chi) & (n_e >= elo) & (n_e <= ehi))[0]
            if len(idx) < 12:
                continue
            ms, n = time_subset(idx)
            bins.append({"contours": [clo, chi], "edges": [elo, ehi], "glyphs": int(len(idx)), "mean_edges": float(n_e[idx].mean()), "mean_contours": float(n_c[idx].mean()),
                         "mean_edges_x_extra_contours": float((n_e[idx]*np.maximum(n_c[idx]-1, 0)).mean()), "us_per_glyph": 1e3*ms/n})
    # piecewise by the kernel class a glyph runs in (msdf_capi.hip: ensureBuckets): one contour -> simple combiner; 2..7 contours and <= 128
    # edges -> per-contour distances in LDS; the rest -> global workspace. Within a class: a + b*E + c*C + d*E*C.
    from msdfgen_amd.shard import LDS_MAX_CONTOURS, LDS_MAX_EDGES     # the class boundaries the launches use (round 5: 5 contours at the 10 KB LDS budget)
    def klass(c, e):
        return 0 if c <= 1 else 1 if (c <= LDS_MAX_CONTOURS and e <= LDS_MAX_EDGES) else 2
    fitted, pred, y = {}, np.zeros(len(bins)), np.array([b["us_per_glyph"] for b in bins])
    for k, name in enumerate(("one_contour", "lds", "global")):
        sel = [i for i, b in enumerate(bins) if klass(b["contours"][0], b["edges"][0]) == k]
        cols = [[1., bins[i]["mean_edges"]]+([bins[i]["mean_contours"], bins[i]["mean_edges"]*bins[i]["mean_contours"]] if k else []) for i in sel]
        A = np.array(cols)
        wgt = np.sqrt(np.array([bins[i]["glyphs"] for i in sel], np.float64))     # bins weigh by their share of the set
        coef, *_ = np.linalg.lstsq(A*wgt[:, None], y[sel]*wgt, rcond=None)
        pred[sel] = A @ coef
        fitted[name] = [float(v) for v in coef]+([0., 0.] if not k else [])

    def cost_fit(m):
        out = np.zeros(len(n_e))
        for g in range(len(n_e)):
            a, b, c, d = m[("one_contour", "lds", "global")[klass(n_c[g], n_e[g])]]
            out[g] = max(a+b*n_e[g]+c*n_c[g]+d*n_e[g]*n_c[g], 1e-3)
        return out

    def split_report(costs, parts):
        bounds = partition_contiguous(costs, parts)
        times = [time_subset(range(int(bounds[r]), int(bounds[r+1])), reps=8, fill=False)[0] for r in range(parts)]   # the parts as they are
        return {"bounds": [int(v) for v in bounds], "ms_per_part": [round(t, 4) for t in times], "imbalance_max_over_mean": round(max(times)/np.mean(times)-1, 4)}

    old = 4096.*(n_e+1.)
    res = {"workload": "8192 distinct DejaVu glyphs, msdf 64x64, default config (distance + error correction)", "bins": bins, "fitted_us_per_glyph_at_64x64": fitted,
           "fit_rms_relative_error": float(np.sqrt(np.mean(((pred-y)/y)**2))),
           "model_in_the_tree": COST_MODEL,
           "splits": {"2-way by W*H*(E+1) (round 2)": split_report(old, 2), "2-way by the fitted model": split_report(cost_fit(fitted), 2),
                      "2-way by the model in the tree": split_report(glyph_costs(batch, 64, 64), 2),
                      "8-way by W*H*(E+1) (round 2)": split_report(old, 8), "8-way by the fitted model": split_report(cost_fit(fitted), 8)}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
