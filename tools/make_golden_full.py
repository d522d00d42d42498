"""Full-size parity pins for BASELINE configs 4 and 5, generated ONCE in the authoring container by the UNMODIFIED reference
(oracle/_ref/libmsdfgen_ref.so, compiled in place from /root/reference).  The fixtures travel to the GPU box; the `-m gpu` tests
compare EVERY tile / texel of the device output against them (tests/test_gpu_fullsize.py).

    python tools/make_golden_full.py [--threads N]

  dejavu8192.npz   config 4: the first 8 192 glyphs with outlines of DejaVuSans followed by DejaVuSans-Bold (glyph
                   order), each after Shape::normalize + edgeColoringSimple(3.0, seed 0) by the reference; flattened CSR arrays,
                   control-point bounds, and per glyph the sha256 of the reference's msdf 48x48 tile (autoframe, 4 px range,
                   library-default config = overlapping combiner + EDGE_PRIORITY / CHECK_DISTANCE_AT_EDGE error correction) and of
                   its msdf 64x64 tile (the bench workload re-frames the same glyphs at 64x64).
  dejavu8192_modes.npz  the same 8 192 glyphs in the other three field types (round 3): per-glyph sha256[:16] of the reference's mtsdf 64x64
                   (config 3's mode on the tail of a real font), sdf 48x48 and psdf 48x48 tiles, library-default config.
  cjk512.npz       config 4's CJK-like stand-in: the 512 distinct shapes msdfgen_amd.synth.cjk_like_shape(20000..20511) (8-20 contours each:
                   the only workload that takes the persistent global-scratch form of k_distance), msdf 48x48: per-shape sha256 of the
                   reference's tile + a checksum of the generated outlines (the test regenerates them from the seeds).
  logo1024.npz     config 5: the 926-edge cubic logo (msdfgen_amd.synth.logo_shape(5)), msdf 1024x1024, 8 px range, default error
                   correction: sha256 of the texels and of the final stencil, per-row sha256 (to localise a mismatch), the
                   pre-correction field's sha256, a 64x64 crop of the texels and of the stencil, and the shape itself.
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from fontshapes import font_glyphs  # noqa: E402
from msdfgen_amd.shape import FlatShape, ShapeBatch, autoframe  # noqa: E402
from msdfgen_amd import synth  # noqa: E402
from oracle.pyoracle import Ref  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha_bytes(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def dejavu(ref, threads):
    names, shapes, bounds = [], [], []
    for font in ("DejaVuSans.ttf", "DejaVuSans-Bold.ttf"):
        for name, raw in font_glyphs(font):
            if len(shapes) >= 8192:
                break
            h = ref.shape_from_flat(raw)
            ref.prepare(h, 3.0, 0)
            fa = ref.flatten(h)
            b = ref.bounds(h)
            ref.free(h)
            names.append(font[:-4]+":"+name), shapes.append(FlatShape(fa.contour_offsets, fa.points, fa.types, fa.colors)), bounds.append(b)
    assert len(shapes) == 8192, len(shapes)
    batch = ShapeBatch.from_shapes(shapes, names)
    out = {"glyph_contour_offsets": batch.glyph_contour_offsets, "contour_offsets": batch.contour_offsets, "points": batch.points,
           "types": batch.types.astype(np.uint8), "colors": batch.colors.astype(np.uint8), "names": np.array(names), "bounds": np.stack(bounds)}
    for size in (48, 64):
        xfs = np.stack([autoframe(b, size, size, 4) for b in bounds])
        t0 = time.time()
        tiles, _ = ref.generate_batch_timed(shapes, 3, size, size, xfs, threads=threads)
        print("dejavu msdf %dx%d: %.1f s" % (size, size, time.time()-t0), flush=True)
        out["xf%d" % size] = xfs
        out["sha%d" % size] = np.stack([sha_bytes(t) for t in tiles])
        out["sha_all%d" % size] = sha_bytes(tiles)
        if size == 48:
            out["sample48"] = tiles[::1024].copy()                      # 8 whole tiles for a readable diff when a hash disagrees
    np.savez_compressed(os.path.join(GOLDEN, "dejavu8192.npz"), **out)
    print("dejavu8192: %d glyphs, %d contours, %d edges (%.1f per glyph, max %d), max contours %d" % (
        batch.n_glyphs, batch.n_contours, batch.n_edges, batch.n_edges/batch.n_glyphs,
        max(s.n_edges for s in shapes), max(s.n_contours for s in shapes)))


def dejavu_modes(ref, threads):
    """mtsdf / sdf / psdf on the 8 192 distinct glyphs of dejavu8192.npz (which must exist: same shapes, same framing)."""
    z = np.load(os.path.join(GOLDEN, "dejavu8192.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
    shapes = [batch.shape(g) for g in range(batch.n_glyphs)]
    out = {}
    for tag, mode, size in (("mtsdf64", 4, 64), ("sdf48", 1, 48), ("psdf48", 2, 48)):
        t0 = time.time()
        tiles, _ = ref.generate_batch_timed(shapes, mode, size, size, z["xf%d" % size], threads=threads)
        print("dejavu %s: %.1f s" % (tag, time.time()-t0), flush=True)
        out["sha_"+tag] = np.stack([sha_bytes(t)[:16] for t in tiles])
        out["sha_all_"+tag] = sha_bytes(tiles)
        out["sample_"+tag] = tiles[::2048].copy()
    np.savez_compressed(os.path.join(GOLDEN, "dejavu8192_modes.npz"), **out)


def cjk(ref, threads):
    base = [synth.cjk_like_shape(20000+i) for i in range(512)]
    batch = ShapeBatch.from_shapes(base)
    xfs = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])
    t0 = time.time()
    tiles, _ = ref.generate_batch_timed(base, 3, 48, 48, xfs, threads=threads)
    print("cjk-like msdf 48x48, 512 shapes: %.1f s" % (time.time()-t0), flush=True)
    np.savez_compressed(os.path.join(GOLDEN, "cjk512.npz"), sha48=np.stack([sha_bytes(t) for t in tiles]), sha_all48=sha_bytes(tiles),
                        sha_points=sha_bytes(batch.points), sha_xf=sha_bytes(xfs), n_edges=np.array(batch.n_edges), n_contours=np.array(batch.n_contours),
                        sample48=tiles[::128].copy())
    print("cjk512: %d contours, %d edges (%.1f / %.1f per glyph)" % (batch.n_contours, batch.n_edges, batch.n_contours/512, batch.n_edges/512))


def logo(ref):
    """The reference's generateMSDF evaluates distances through ShapeDistanceFinder::distance(), whose per-edge cache
    (edge-selectors.cpp:64-79, DISTANCE_DELTA_FACTOR) prunes edges relative to the PREVIOUS texel of its serpentine walk; on this shape
    the pruning is not exact: 1 of 1 048 576 texels differs (by 3e-5) from the reference's own exact evaluation
    ShapeDistanceFinder::oneShotDistance (ShapeDistanceFinder.hpp:36-58) -- and which texels do depends on the walk (the OpenMP build
    restarts the cache per row chunk). The device evaluates every texel exactly, so the pin is: the reference's oneShotDistance at
    every texel centre, mapped as DistanceMapping does, then the reference's msdfErrorCorrection on that field; the texels where the
    cached walk disagrees are listed in the fixture."""
    s = synth.logo_shape(5)
    w = h = 1024
    xf = autoframe(s.bounds(), w, h, 8)
    t0 = time.time()
    cached = ref.generate(s, 3, w, h, xf, ec_mode=0)
    ys, xs = np.mgrid[0:h, 0:w]
    pts = np.stack([(xs.ravel()+.5)/xf[0]-xf[2], (ys.ravel()+.5)/xf[1]-xf[3]], 1)       # Projection::unproject, Projection.cpp:14-16
    d = ref.shape_distance(s, 3, True, pts)[:, :3]
    pre = (np.float64(1)/(xf[5]-xf[4])*(d+(-xf[4]))).astype(np.float32).reshape(h, w, 3)   # DistanceMapping.cpp:13-17
    diff = np.argwhere((pre.view(np.uint32) != cached.view(np.uint32)).any(axis=2))
    stencil = np.zeros((h, w), np.uint8)
    out = ref.error_correction(s, pre, xf, stencil=stencil)
    st_cached = np.zeros((h, w), np.uint8)
    out_cached = ref.generate(s, 3, w, h, xf, stencil=st_cached)
    print("logo msdf 1024x1024: %.1f s (%d edges, %d contours); cached walk differs from oneShotDistance at %d texel(s): %s" % (
        time.time()-t0, s.n_edges, s.n_contours, len(diff), diff[:4].tolist()), flush=True)
    y0, x0 = 480, 480
    np.savez_compressed(os.path.join(GOLDEN, "logo1024.npz"), contour_offsets=s.contour_offsets, points=s.points, types=s.types.astype(np.uint8),
                        colors=s.colors.astype(np.uint8), xf=xf, sha_out=sha_bytes(out), sha_stencil=sha_bytes(stencil), sha_pre=sha_bytes(pre),
                        sha_rows=np.stack([sha_bytes(out[y]) for y in range(h)]), sha_stencil_rows=np.stack([sha_bytes(stencil[y]) for y in range(h)]),
                        sha_pre_rows=np.stack([sha_bytes(pre[y]) for y in range(h)]),
                        crop_origin=np.array([y0, x0]), crop_out=out[y0:y0+64, x0:x0+64].copy(), crop_stencil=stencil[y0:y0+64, x0:x0+64].copy(),
                        n_error=np.array(int((stencil & 1).sum())),
                        cached_diff_yx=diff, cached_diff_values=np.stack([cached[y, x] for y, x in diff]) if len(diff) else np.zeros((0, 3), np.float32),
                        exact_values_there=np.stack([pre[y, x] for y, x in diff]) if len(diff) else np.zeros((0, 3), np.float32),
                        sha_pre_cached_walk=sha_bytes(cached), sha_stencil_cached_walk=sha_bytes(st_cached),
                        out_cached_diff_yx=np.argwhere((out.view(np.uint32) != out_cached.view(np.uint32)).any(axis=2)),
                        out_cached_diff_values=out_cached[(out.view(np.uint32) != out_cached.view(np.uint32)).any(axis=2)],
                        sha_out_cached_walk=sha_bytes(out_cached), n_out_differs_from_cached_walk=np.array(int((out.view(np.uint32) != out_cached.view(np.uint32)).any(axis=2).sum())))
    print("logo1024: %d ERROR texels; final bitmap differs from the cached walk's at %d texel(s)" % (
        int((stencil & 1).sum()), int((out.view(np.uint32) != out_cached.view(np.uint32)).any(axis=2).sum())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    ref = Ref()
    if args.only in ("", "dejavu"):
        dejavu(ref, args.threads)
    if args.only in ("", "modes"):
        dejavu_modes(ref, args.threads)
    if args.only in ("", "cjk"):
        cjk(ref, args.threads)
    if args.only in ("", "logo"):
        logo(ref)
    for f in ("dejavu8192.npz", "dejavu8192_modes.npz", "cjk512.npz", "logo1024.npz"):
        p = os.path.join(GOLDEN, f)
        if os.path.exists(p):
            print("%-16s %8d bytes" % (f, os.path.getsize(p)))


if __name__ == "__main__":
    main()
