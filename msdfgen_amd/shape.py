"""Flat (CSR, struct-of-arrays) shape model that crosses the C ABI (include/msdfgen_hip.h).

The reference's `msdfgen::Shape` (core/Shape.h:15-58) is a vector of contours, each a vector of heap-allocated polymorphic
`EdgeSegment*` (core/edge-segments.h:14-55).  The hot path only *reads* it, so the boundary flattens it once into:

    contour_offsets int32[C+1]   CSR offsets into the edge arrays (edges in the Shape's natural order)
    points          float64[E,8] p0x,p0y,p1x,p1y,p2x,p2y,p3x,p3y   (unused slots 0)
    types           int32[E]     1 linear / 2 quadratic / 3 cubic   (EDGE_TYPE, core/edge-segments.h:62,91,122)
    colors          int32[E]     EdgeColor bitmask R=1 G=2 B=4      (core/EdgeColor.h:9-18)
    inverse_y       bool         Shape::getYAxisOrientation() == Y_DOWNWARD (core/Shape.cpp:200)

A batch of glyphs concatenates shapes and adds `glyph_contour_offsets int32[G+1]`.
"""
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

LINEAR, QUADRATIC, CUBIC = 1, 2, 3
BLACK, RED, GREEN, YELLOW, BLUE, MAGENTA, CYAN, WHITE = range(8)


@dataclass
class FlatShape:
    contour_offsets: np.ndarray
    points: np.ndarray
    types: np.ndarray
    colors: np.ndarray
    inverse_y: bool = False

    def __post_init__(self):
        self.contour_offsets = np.ascontiguousarray(self.contour_offsets, np.int32).reshape(-1)
        self.points = np.ascontiguousarray(self.points, np.float64).reshape(-1, 8)
        self.types = np.ascontiguousarray(self.types, np.int32).reshape(-1)
        self.colors = np.ascontiguousarray(self.colors, np.int32).reshape(-1)
        if len(self.contour_offsets) == 0:
            self.contour_offsets = np.zeros(1, np.int32)
        e = int(self.contour_offsets[-1])
        if not (len(self.points) == len(self.types) == len(self.colors) == e):
            raise ValueError("edge arrays do not match contour_offsets[-1] = %d" % e)
        if np.any(np.diff(self.contour_offsets) < 0) or self.contour_offsets[0] != 0:
            raise ValueError("contour_offsets must be a non-decreasing CSR array starting at 0")
        if e and (self.types.min() < 1 or self.types.max() > 3):
            raise ValueError("edge type must be 1, 2 or 3")

    @property
    def n_contours(self) -> int:
        return len(self.contour_offsets)-1

    @property
    def n_edges(self) -> int:
        return int(self.contour_offsets[-1])

    def bounds(self):
        """Control-point bounding box (l, b, r, t). A conservative stand-in for Shape::getBounds (core/Shape.cpp:105)
        that is only used for framing synthetic workloads; framing is the caller's job, not part of the hot path."""
        if self.n_edges == 0:
            return 0., 0., 1., 1.
        xs, ys = [], []
        for k in range(4):
            m = self.types >= max(k, 1)
            xs.append(self.points[m, 2*k])
            ys.append(self.points[m, 2*k+1])
        xs, ys = np.concatenate(xs), np.concatenate(ys)
        return float(xs.min()), float(ys.min()), float(xs.max()), float(ys.max())

    @staticmethod
    def from_contours(contours: Sequence[Sequence[tuple]], inverse_y=False) -> "FlatShape":
        """contours: list of contours, each a list of edges `(color, (x, y), (x, y)[, (x, y)[, (x, y)]])`."""
        offs, pts, types, colors = [0], [], [], []
        for contour in contours:
            for edge in contour:
                color, cps = edge[0], edge[1:]
                row = [0.]*8
                for i, (x, y) in enumerate(cps):
                    row[2*i], row[2*i+1] = float(x), float(y)
                pts.append(row)
                types.append(len(cps)-1)
                colors.append(int(color))
            offs.append(len(pts))
        return FlatShape(np.array(offs, np.int32), np.array(pts, np.float64).reshape(-1, 8), np.array(types, np.int32), np.array(colors, np.int32), inverse_y)


@dataclass
class ShapeBatch:
    """G glyph shapes concatenated into one CSR edge buffer."""
    glyph_contour_offsets: np.ndarray  # int32[G+1]
    contour_offsets: np.ndarray        # int32[C+1], global edge indices
    points: np.ndarray                 # float64[E, 8]
    types: np.ndarray                  # int32[E]
    colors: np.ndarray                 # int32[E]
    inverse_y: np.ndarray              # uint8[G]
    names: List[str] = field(default_factory=list)

    @property
    def n_glyphs(self) -> int:
        return len(self.glyph_contour_offsets)-1

    @property
    def n_contours(self) -> int:
        return len(self.contour_offsets)-1

    @property
    def n_edges(self) -> int:
        return int(self.contour_offsets[-1])

    @staticmethod
    def from_shapes(shapes: Sequence[FlatShape], names=None) -> "ShapeBatch":
        gco, co, pts, types, colors, inv = [0], [0], [], [], [], []
        e = 0
        for s in shapes:
            co.extend((s.contour_offsets[1:]+e).tolist())
            gco.append(len(co)-1)
            pts.append(s.points)
            types.append(s.types)
            colors.append(s.colors)
            inv.append(1 if s.inverse_y else 0)
            e += s.n_edges
        cat = lambda parts, dt, shp: np.concatenate(parts).astype(dt) if parts else np.zeros(shp, dt)
        return ShapeBatch(np.array(gco, np.int32), np.array(co, np.int32), cat(pts, np.float64, (0, 8)).reshape(-1, 8),
                          cat(types, np.int32, (0,)), cat(colors, np.int32, (0,)), np.array(inv, np.uint8), list(names or []))

    def dump(self, path, xfs) -> None:
        """The flat file tests/shim/shim_check reads to rebuild real msdfgen::Shape objects (`batch`, `e2e`, `flatten` modes): int32 nGlyphs, nContours,
        nEdges; gco; co; f64 points[nEdges][8]; u8 types; u8 colors; f64 xfs[nGlyphs][6] (sx, sy, tx, ty, range lower, range upper); u8 inverse_y."""
        gco, co = self.glyph_contour_offsets.astype(np.int32), self.contour_offsets.astype(np.int32)
        with open(path, "wb") as f:
            np.array([len(gco)-1, len(co)-1, int(co[-1])], np.int32).tofile(f)
            gco.tofile(f), co.tofile(f), np.ascontiguousarray(self.points, np.float64).tofile(f)
            np.ascontiguousarray(self.types, np.uint8).tofile(f), np.ascontiguousarray(self.colors, np.uint8).tofile(f)
            np.ascontiguousarray(xfs, np.float64).reshape(len(gco)-1, 6).tofile(f)
            np.ascontiguousarray(self.inverse_y, np.uint8).tofile(f)

    def shape(self, g: int) -> FlatShape:
        c0, c1 = int(self.glyph_contour_offsets[g]), int(self.glyph_contour_offsets[g+1])
        e0, e1 = int(self.contour_offsets[c0]), int(self.contour_offsets[c1])
        return FlatShape(self.contour_offsets[c0:c1+1]-e0, self.points[e0:e1], self.types[e0:e1], self.colors[e0:e1], bool(self.inverse_y[g]))

    def shapes(self) -> List[FlatShape]:
        return [self.shape(g) for g in range(self.n_glyphs)]

    def select(self, idx: Sequence[int]) -> "ShapeBatch":
        names = [self.names[i] for i in idx] if self.names else []
        return ShapeBatch.from_shapes([self.shape(int(i)) for i in idx], names)

    def save(self, path):
        np.savez_compressed(path, glyph_contour_offsets=self.glyph_contour_offsets, contour_offsets=self.contour_offsets, points=self.points,
                            types=self.types.astype(np.uint8), colors=self.colors.astype(np.uint8), inverse_y=self.inverse_y, names=np.array(self.names))

    @staticmethod
    def load(path) -> "ShapeBatch":
        z = np.load(path, allow_pickle=False)
        return ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"].astype(np.float64),
                          z["types"].astype(np.int32), z["colors"].astype(np.int32), z["inverse_y"].astype(np.uint8), [str(n) for n in z["names"]])


def autoframe(bounds, width, height, px_range):
    """Centre `bounds` (l, b, r, t) in a width x height tile leaving px_range/2 texels of margin, exactly as the reference CLI's
    `-autoframe -pxrange` does (main.cpp:1153-1183).  Returns xf = (sx, sy, tx, ty, range_lower, range_upper) in float64.
    Framing is caller-side policy, not part of the hot path; it lives here so that tests, bench and the CPU baseline all feed
    the very same six doubles to the reference and to the HIP path."""
    l, b, r, t = (float(v) for v in bounds)
    lower = -.5*float(px_range)
    fx, fy = float(width)+2*lower, float(height)+2*lower
    if l >= r or b >= t:
        l, b, r, t = 0., 0., 1., 1.
    if fx <= 0 or fy <= 0:
        raise ValueError("cannot fit the specified pixel range")
    dx, dy = r-l, t-b
    if dx*fy < dy*fx:
        tx, ty = .5*(fx/fy*dy-dx)-l, -b
        scale = fy/dy
    else:
        tx, ty = -l, .5*(fy/fx*dx-dy)-b
        scale = fx/dx
    tx -= lower/scale
    ty -= lower/scale
    return np.array([scale, scale, tx, ty, lower/scale, -lower/scale], np.float64)


def distance_mapping(range_lower, range_upper):
    """DistanceMapping(Range) (core/DistanceMapping.cpp:13): scale = 1/(upper-lower), translate = -lower."""
    lo, up = np.float64(range_lower), np.float64(range_upper)
    return float(np.float64(1)/(up-lo)), float(-lo)
