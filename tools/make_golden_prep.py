"""tests/golden/prep.npz -- fixtures for SURVEY 8(f3), shape preparation (Shape::normalize + edgeColoringSimple).

Inputs: RAW (un-normalised, white) outlines -- DejaVuSans U+0020..U+007E, a slice of DejaVuSerif / DejaVuSans-Bold glyphs (curvier
outlines, more corner cases) and synthetic contours built to hit the rare branches: single-edge contours (split in thirds), two-edge
contours with one corner ("teardrop" split into six), smooth contours (no corner), cusps (anti-parallel tangents -> deconverge incl.
quadratic -> cubic conversion), degenerate control points.  Outputs: the reference's own results (oracle/_ref, compiled from
/root/reference) for (normalize, edgeColoringSimple(angle 3, seed s)) per glyph.  Run in the authoring container:
    python tools/make_golden_prep.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from fontshapes import font_glyphs  # noqa: E402
from msdfgen_amd.shape import FlatShape, ShapeBatch  # noqa: E402
from oracle.pyoracle import Ref  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def contour(*edges):
    return [(7,)+tuple(tuple(float(v) for v in p) for p in e) for e in edges]            # raw outlines are WHITE (EdgeColor.h:17)


def synthetic():
    """Hand-built raw shapes for the rare branches. Edge = tuple of 2/3/4 points."""
    shapes = []
    k = .5522847498
    circle = contour(((1, 0), (1, k), (k, 1), (0, 1)), ((0, 1), (-k, 1), (-1, k), (-1, 0)), ((-1, 0), (-1, -k), (-k, -1), (0, -1)), ((0, -1), (k, -1), (1, -k), (1, 0)))
    shapes.append([circle])                                                            # smooth: no corner
    shapes.append([contour(((0, 0), (2, 1.5), (-2, 1.5), (0, 0)))])                     # one cubic edge, closed: split in thirds, one corner
    shapes.append([contour(((0, 0), (1, 2), (0, 0)))])                                  # one (degenerate) quadratic edge
    shapes.append([contour(((0, 0), (3, 0)))])                                          # one linear edge (open, degenerate)
    shapes.append([contour(((0, 0), (1, 1), (2, 0)), ((2, 0), (1, -1), (0, 0)))])       # lens: two edges, two corners
    shapes.append([contour(((0, 0), (.5, 1), (1, 1)), ((1, 1), (2, 1), (1, -.5), (0, 0)))])   # two edges, smooth at (1,1): teardrop with 2 edges
    shapes.append([contour(((0, 0), (1, 1), (2, 0)), ((2, 0), (1, 1), (0, 2)), ((0, 2), (-1, 1), (0, 0)))])   # cusp at (2,0): anti-parallel tangents
    shapes.append([contour(((0, 0), (1, 0), (2, 0), (3, 1)), ((3, 1), (2, 0), (1, 0), (0, -1)), ((0, -1), (0, 0)))])  # cubic cusp
    shapes.append([contour(((0, 0), (2, 0)), ((2, 0), (1, 0), (1, 1)), ((1, 1), (0, 0)))])                    # line then quadratic turning straight back
    shapes.append([contour(((0, 0), (0, 0), (1, 1), (2, 0)), ((2, 0), (2, 0), (1, -1), (0, 0)))])             # degenerate first control points
    shapes.append([circle, contour(((3, 0), (4, 1), (5, 0)), ((5, 0), (4, -1), (3, 0))), contour(((0, 3), (1, 4), (0, 3)))])  # colour state runs across contours
    rng = np.random.default_rng(33)
    for i in range(12):                                                                 # random polygons with a few curved edges: many corners
        n = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2*np.pi, n))
        pts = np.stack([np.cos(ang), np.sin(ang)], 1)*rng.uniform(.5, 1.5, (n, 1))
        edges = []
        for j in range(n):
            a, b = pts[j], pts[(j+1) % n]
            if rng.random() < .4:
                edges.append((a, (a+b)/2+rng.normal(0, .2, 2), b))
            else:
                edges.append((a, b))
        shapes.append([contour(*edges)])
    return [FlatShape.from_contours(s) for s in shapes]


def main():
    ref = Ref()
    raws, names = [], []
    for name, raw in font_glyphs("DejaVuSans.ttf", range(0x20, 0x7f)):
        raws.append(raw), names.append("sans-"+name)
    for font, lo, hi in (("DejaVuSerif.ttf", 0x21, 0x7f), ("DejaVuSans-Bold.ttf", 0xa1, 0x100)):
        for name, raw in font_glyphs(font, range(lo, hi)):
            raws.append(raw), names.append(font[6:-4]+"-"+name)
    for i, s in enumerate(synthetic()):
        raws.append(s), names.append("synthetic-%d" % i)
    seeds = np.array([0 if i % 3 else 12345+977*i for i in range(len(raws))], np.uint64)
    prepared, normalized = [], []
    for raw, seed in zip(raws, seeds):
        fa = ref.shape_prepare(raw, True, 1, 3.0, int(seed))
        prepared.append(FlatShape(fa.contour_offsets, fa.points, fa.types, fa.colors))
        fa = ref.shape_prepare(raw, True, 0, 3.0, 0)
        normalized.append(FlatShape(fa.contour_offsets, fa.points, fa.types, fa.colors))
    rb, pb, nb = ShapeBatch.from_shapes(raws, names), ShapeBatch.from_shapes(prepared, names), ShapeBatch.from_shapes(normalized, names)
    out = {"names": np.array(names), "seeds": seeds}
    for tag, b in (("raw", rb), ("prep", pb), ("norm", nb)):
        out[tag+"_gco"] = b.glyph_contour_offsets
        out[tag+"_co"] = b.contour_offsets
        out[tag+"_points"] = b.points
        out[tag+"_types"] = b.types.astype(np.uint8)
        out[tag+"_colors"] = b.colors.astype(np.uint8)
    np.savez_compressed(os.path.join(GOLDEN, "prep.npz"), **out)
    changed = sum(int(r.n_edges != p.n_edges) for r, p in zip(raws, prepared))
    cubics = sum(int((p.types == 3).sum() > (r.types == 3).sum()) for r, p in zip(raws, normalized))
    print("prep.npz: %d glyphs, %d raw edges -> %d prepared; %d glyphs changed edge count, %d gained cubics (deconverge)" % (
        len(raws), rb.n_edges, pb.n_edges, changed, cubics))


if __name__ == "__main__":
    main()
