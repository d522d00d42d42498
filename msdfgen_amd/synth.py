"""Seeded synthetic shapes for tests and benchmarks (no fonts, no reference needed).  They are valid hot-path inputs: closed
contours, edges already coloured.  The colouring is a simple corner-alternating scheme (not edgeColoringSimple); parity only
needs both sides to be fed the same Shape."""
import math

import numpy as np

from .shape import FlatShape, CYAN, MAGENTA, YELLOW, WHITE

_CYCLE = (CYAN, MAGENTA, YELLOW)


def _blob(rng, cx, cy, radius, n_edges, kinds, wobble, clockwise, color0):
    ang0 = rng.uniform(0, 2*math.pi)
    angs = ang0+np.sort(rng.uniform(0, 2*math.pi, n_edges))
    if clockwise:
        angs = angs[::-1]
    rad = radius*(1+wobble*rng.uniform(-1, 1, n_edges))
    pts = [(cx+r*math.cos(a), cy+r*math.sin(a)) for a, r in zip(angs, rad)]
    edges = []
    for i in range(n_edges):
        p0, p3 = pts[i], pts[(i+1) % n_edges]
        kind = int(rng.choice(kinds))
        color = _CYCLE[(color0+i) % 3] if n_edges > 1 else WHITE
        dx, dy = p3[0]-p0[0], p3[1]-p0[1]
        nx, ny = -dy, dx
        if kind == 1:
            edges.append((color, p0, p3))
        elif kind == 2:
            b = rng.uniform(-.6, .6)
            edges.append((color, p0, (p0[0]+.5*dx+b*nx, p0[1]+.5*dy+b*ny), p3))
        else:
            b1, b2 = rng.uniform(-.9, .9, 2)
            edges.append((color, p0, (p0[0]+.3*dx+b1*nx, p0[1]+.3*dy+b1*ny), (p0[0]+.7*dx+b2*nx, p0[1]+.7*dy+b2*ny), p3))
    return edges


def random_shape(seed, n_contours=2, edges_per_contour=(3, 9), kinds=(1, 2, 3), spread=.6, wobble=.35, holes=True):
    """Overlapping blobs (positive winding) plus optional reversed 'hole' contours; linear/quadratic/cubic mix."""
    rng = np.random.default_rng(seed)
    contours = []
    for c in range(n_contours):
        n = int(rng.integers(edges_per_contour[0], edges_per_contour[1]+1))
        cx, cy = rng.uniform(-spread, spread, 2)
        r = rng.uniform(.15, .6)
        cw = holes and c > 0 and rng.random() < .4
        contours.append(_blob(rng, cx, cy, r, n, kinds, wobble, cw, int(rng.integers(0, 3))))
    return FlatShape.from_contours(contours)


def cjk_like_shape(seed):
    """8-20 stroke-like contours (rectangles and curved strokes), 60-150 edges: stands in for CJK edge density."""
    rng = np.random.default_rng(seed)
    contours = []
    for c in range(int(rng.integers(8, 21))):
        cx, cy = rng.uniform(.1, .9, 2)
        w, h = (rng.uniform(.2, .7), rng.uniform(.03, .07)) if rng.random() < .5 else (rng.uniform(.03, .07), rng.uniform(.2, .7))
        x0, x1, y0, y1 = cx-w/2, cx+w/2, cy-h/2, cy+h/2
        corners = [(x0, y0), (x1, y0), (x1, y1), (x0, y1)]
        edges = []
        k = int(rng.integers(0, 3))
        for i in range(4):
            p0, p1 = corners[i], corners[(i+1) % 4]
            if rng.random() < .5:
                edges.append((_CYCLE[(k+i) % 3], p0, p1))
            else:  # split a side into two segments, one of them curved
                mx, my = .5*(p0[0]+p1[0]), .5*(p0[1]+p1[1])
                bx, by = rng.uniform(-.01, .01, 2)
                edges.append((_CYCLE[(k+i) % 3], p0, (.5*(p0[0]+mx)+bx, .5*(p0[1]+my)+by), (mx, my)))
                edges.append((_CYCLE[(k+i) % 3], (mx, my), p1))
        contours.append(edges)
    return FlatShape.from_contours(contours)


def logo_shape(seed=5, n_blobs=40, edges=(10, 38)):
    """Union of overlapping closed cubic blobs incl. self-intersecting loops: 400-1500 cubic edges."""
    rng = np.random.default_rng(seed)
    contours = []
    for c in range(n_blobs):
        n = int(rng.integers(edges[0], edges[1]+1))
        cx, cy = rng.uniform(.15, .85, 2)
        contours.append(_blob(rng, cx, cy, rng.uniform(.05, .22), n, (3,), .5, rng.random() < .25, c))
    return FlatShape.from_contours(contours)
