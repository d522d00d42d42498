"""Diagnostics (GPU): distribution of deferred distance checks per glyph (the work of k_ec_query) for the headline and the CJK-like config."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import msdfgen_amd as M  # noqa: E402
from msdfgen_amd import synth  # noqa: E402
from msdfgen_amd.shape import ShapeBatch, autoframe  # noqa: E402
from bench import load_latin, tile_batch  # noqa: E402

M.init(0)
latin, xf64 = load_latin()
b, x = tile_batch(latin, xf64, 940)
gb = M.GlyphBatch(b)
gb.generate(3, 64, 64, x)
ov, c = gb.candidate_counts()
print("latin 64x64: overflow %s, candidates/glyph mean %.1f p50 %d p99 %d max %d" % (ov, c.mean(), np.percentile(c, 50), np.percentile(c, 99), c.max()))
base = [synth.cjk_like_shape(20000+i) for i in range(512)]
cj = ShapeBatch.from_shapes(base)
cx = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])
gb = M.GlyphBatch(cj)
gb.generate(3, 48, 48, cx)
ov, c = gb.candidate_counts()
print("cjk-like 48x48: overflow %s, candidates/glyph mean %.1f p50 %d p99 %d max %d" % (ov, c.mean(), np.percentile(c, 50), np.percentile(c, 99), c.max()))
from bench import load_dejavu  # noqa: E402
dj, djx, _ = load_dejavu()
gb = M.GlyphBatch(dj)
gb.generate(3, 64, 64, djx)
ov, c = gb.candidate_counts()
gco, co = dj.glyph_contour_offsets, dj.contour_offsets
ne = co[gco[1:]]-co[gco[:-1]]
nc = np.diff(gco)
print("dejavu8192 64x64: overflow %s, candidates/glyph mean %.1f p50 %d p90 %d p99 %d max %d; sum(cand*edges) %.3g, sum(cand) %d; glyphs with 0: %d" % (
    ov, c.mean(), np.percentile(c, 50), np.percentile(c, 90), np.percentile(c, 99), c.max(), float((c.astype(np.float64)*ne).sum()), int(c.sum()), int((c == 0).sum())))
for lo, hi in ((1, 1), (2, 3), (4, 7), (8, 99)):
    m = (nc >= lo) & (nc <= hi)
    print("  contours %d..%d: glyphs %d, candidates/glyph %.1f, edges %.1f" % (lo, hi, m.sum(), c[m].mean(), ne[m].mean()))
