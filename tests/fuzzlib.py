"""Randomized parity sweep of the HIP path against the oracle (test infrastructure: used by tests/test_gpu_parity.py with a bounded
shape count and by tools/fuzz_parity.py for long runs): random shapes (lines / quadratics / cubics, holes, many contours, degenerate
pieces), random tile sizes and ranges, both combiners, every error-correction mode, sdf / psdf / msdf / mtsdf."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def run(n_shapes, seed, deadline_s=None, single=False):
    """Returns a dict: shapes, groups, values_compared, values_differing_bitwise, max_abs_delta, worst_case, seed.
    single: every shape through its own generate*() call (the literal drop-in: one fused launch per call, msdf_single.hpp) instead of one batch per group."""
    import time
    import msdfgen_amd as M
    from msdfgen_amd import synth
    from msdfgen_amd.shape import ShapeBatch, autoframe
    from oracle.pyoracle import Oracle
    M.init(0)
    orc = Oracle()
    rng = np.random.default_rng(seed)
    total = differing = 0
    worst = 0.
    worst_case = None
    groups = 0
    pool = ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 8))
    done = 0
    t0 = time.time()
    seen = set()
    while done < n_shapes and (deadline_s is None or time.time()-t0 < deadline_s):
        n = int(min(n_shapes-done, rng.integers(20, 80)))
        mode = int(rng.choice([1, 2, 3, 3, 4]))
        w, h = int(rng.integers(8, 72)), int(rng.integers(8, 72))
        overlap = bool(rng.integers(0, 2))
        ec_mode, ec_dist = int(rng.integers(0, 4)), int(rng.integers(0, 3))
        px_range = min(float(rng.choice([2, 4, 8, 1.5])), .45*min(w, h))      # autoframe needs room for the range inside the tile
        kind = int(rng.integers(0, 5))
        shapes = []
        for i in range(n):
            sd = int(rng.integers(0, 2**31))
            if kind == 0:
                s = synth.random_shape(sd, n_contours=int(rng.integers(1, 4)), kinds=(1, 2, 3), holes=bool(sd & 1))
            elif kind == 1:
                s = synth.random_shape(sd, n_contours=int(rng.integers(1, 3)), edges_per_contour=(3, 14), kinds=(3,), wobble=.6)
            elif kind == 2:
                s = synth.cjk_like_shape(sd)
            elif kind == 4:                                                   # heavily overlapping / nested blobs: texels inside several contours at once
                s = synth.random_shape(sd, n_contours=int(rng.integers(3, 8)), kinds=(1, 2, 3), spread=.25, holes=bool(sd & 1))
            else:
                s = synth.random_shape(sd, n_contours=int(rng.integers(4, 12)), edges_per_contour=(3, 6), kinds=(1, 2), spread=.9)
            s.inverse_y = bool(rng.integers(0, 2))
            shapes.append(s)
        xfs = np.stack([autoframe(s.bounds(), w, h, px_range) for s in shapes])
        if rng.random() < .3:                                             # asymmetric range (CLI -arange)
            xfs[:, 4] *= .5
        batch = ShapeBatch.from_shapes(shapes)
        cfg = M.MSDFGeneratorConfig(overlap, M.ErrorCorrectionConfig(ec_mode, ec_dist)) if mode >= 3 else M.GeneratorConfig(overlap)
        if single:
            fn = {1: M.generate_sdf, 2: M.generate_psdf, 3: M.generate_msdf, 4: M.generate_mtsdf}[mode]
            got = np.stack([fn(np.zeros((h, w, M.CHANNELS[mode]), np.float32), shapes[g], M.SDFTransformation.from_xf(xfs[g]), cfg, M.Y_UPWARD) for g in range(n)])
        else:
            gb = M.GlyphBatch(batch)
            got = gb.generate(mode, w, h, xfs, config=cfg).cpu().numpy()
            gb.close()
        want = list(pool.map(lambda g: orc.generate(shapes[g], mode, w, h, xfs[g], overlap=overlap, ec_mode=ec_mode, ec_dist=ec_dist), range(n)))
        want = np.stack(want)
        bad = got.view(np.uint32) != want.view(np.uint32)
        bad &= ~(np.isnan(got) & np.isnan(want))
        total += got.size
        differing += int(bad.sum())
        if bad.any():
            d = np.abs(got.astype(np.float64)-want.astype(np.float64))
            d[~bad] = 0
            m = float(np.nanmax(d))
            if m > worst:
                worst = m
                g = int(np.argwhere(bad)[0][0])
                worst_case = {"mode": mode, "size": [w, h], "overlap": overlap, "ec": [ec_mode, ec_dist], "kind": kind, "glyph_edges": int(shapes[g].n_edges)}
        seen.add((mode, overlap, ec_mode if mode >= 3 else -1, ec_dist if mode >= 3 else -1, kind))
        done += n
        groups += 1
    pool.shutdown()
    return {"shapes": done, "groups": groups, "values_compared": total, "values_differing_bitwise": differing, "max_abs_delta": worst,
            "worst_case": worst_case, "seed": seed, "single_calls": bool(single), "distinct_mode_combiner_ec_kind": len(seen), "seconds": round(time.time()-t0, 1)}
