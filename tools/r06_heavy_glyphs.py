"""The largest glyphs of the bench workload and the form their distance checks take (python replica of ecQueryGridSlices).   python tools/r06_heavy_glyphs.py [min_edges]"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def main():
    import torch, msdfgen_amd as M
    from bench import load_dejavu
    M.init(0)
    batch, xfs, _ = load_dejavu()
    gb = M.GlyphBatch(batch)
    gb.generate(3, 64, 64, xfs)
    torch.cuda.synchronize()
    cnt = gb.candidate_counts()[1]
    gco, co = batch.glyph_contour_offsets, batch.contour_offsets
    lim = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(os.environ.get("MSDFHIP_QUERY_GRID", "16"))
    rows = []
    for g in range(batch.n_glyphs):
        C = int(gco[g+1]-gco[g]); nE = int(co[gco[g+1]]-co[gco[g]])
        if nE < lim: continue
        count = int(cnt[g]); J = 1
        while J < count and J < 64: J <<= 1
        S = 64//J; need = 1
        while need*steps < nE and need < 64: need <<= 1
        S = max(S, need)
        form = "coop" if (C > 24 or count == 0 or not (2 <= S <= 32)) else "grid S=%d items=%d steps<=%d" % (S, -(-count//(64//S)), -(-max(int(co[gco[g]+c+1]-co[gco[g]+c]) for c in range(C))//S))
        rows.append((nE, C, count, form, str(batch.names[g])))
    for r in sorted(rows, reverse=True)[:40]: print(r)
main()
