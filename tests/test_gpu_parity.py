"""Parity of the HIP path (through the C ABI, on a real MI355X) against the oracle and the committed reference fixtures.

Tolerance (BASELINE.json north_star): |delta| <= 1e-5 per fp32 texel.  The geometry is evaluated in fp64 with the reference's
operation order and no FMA contraction, so in practice outputs are bit-identical except where gfx950's OCML acos/cos/pow differ
from glibc's in the last ulp; the tests report the number of texels whose bits differ and assert the 1e-5 bound on all of them.
Stencils (integer flags) must match exactly.
"""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import msdfgen_amd as M
from conftest import load_npz, bits
from msdfgen_amd import synth
from msdfgen_amd.shape import FlatShape, ShapeBatch, autoframe, distance_mapping

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _device():
    M.init(0)
    info = M.device_info()
    assert info["arch"].startswith("gfx950"), info
    return info


def close(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    same_nan = np.isnan(got) & np.isnan(want)
    d = np.abs(got.astype(np.float64)-want.astype(np.float64))
    d[same_nan] = 0
    d[(got == want)] = 0  # also covers equal infinities
    nbits = int((bits(got) != bits(want)).sum()) if got.dtype == np.float32 else int((got != want).sum())
    worst = float(np.nanmax(d)) if d.size else 0.
    assert worst <= TOL, "%s: max |delta| %.3g > %g (%d of %d values differ bitwise)" % (what, worst, TOL, nbits, got.size)
    return nbits


def gen(mode, s, w, h, xf, config=None, y_down=False):
    out = np.zeros((h, w, M.CHANNELS[mode]), np.float32)
    fn = {1: M.generate_sdf, 2: M.generate_psdf, 3: M.generate_msdf, 4: M.generate_mtsdf}[mode]
    return fn(out, s, M.SDFTransformation.from_xf(xf), config, M.Y_DOWNWARD if y_down else M.Y_UPWARD)


def cfg(overlap=True, ec_mode=M.EC_EDGE_PRIORITY, ec_dist=M.CHECK_DISTANCE_AT_EDGE, min_dev=1.11111111111111111, min_imp=1.11111111111111111, buffer=None, stage=0):
    c = M.MSDFGeneratorConfig(overlap, M.ErrorCorrectionConfig(ec_mode, ec_dist, min_dev, min_imp, buffer))
    c._stage_limit = stage
    return c


def test_config1_shape_a():
    z = load_npz("shape_a.npz")
    s = FlatShape(z["contour_offsets"], z["points"], z["types"], z["colors"])
    close(gen(1, s, 32, 32, z["xf"]), z["sdf32"], "shape A sdf 32x32")
    close(gen(3, s, 32, 32, z["xf"]), z["msdf32"], "shape A msdf 32x32")


def test_config2_glyphs_m_and_s_msdf64(latin):
    batch, xf64, _ = latin
    z = load_npz("outputs.npz")
    sub = [int(g) for g in z["subset"]]
    for ch in "MS":
        g = batch.names.index("U+%04X" % ord(ch))
        close(gen(3, batch.shape(g), 64, 64, xf64[g]), z["msdf64"][sub.index(g)], "msdf64 "+ch)


def test_latin_subset_golden_all_modes(latin):
    batch, xf64, _ = latin
    z = load_npz("outputs.npz")
    diff = 0
    for k, g in enumerate(z["subset"]):
        s = batch.shape(int(g))
        diff += close(gen(1, s, 32, 32, z["xf32"][k]), z["sdf32"][k], "sdf32 %s" % batch.names[g])
        diff += close(gen(2, s, 32, 32, z["xf32"][k]), z["psdf32"][k], "psdf32 %s" % batch.names[g])
        diff += close(gen(3, s, 64, 64, xf64[g]), z["msdf64"][k], "msdf64 %s" % batch.names[g])
        diff += close(gen(4, s, 64, 64, xf64[g]), z["mtsdf64"][k], "mtsdf64 %s" % batch.names[g])
        diff += close(gen(3, s, 64, 64, xf64[g], cfg(ec_mode=M.EC_DISABLED)), z["msdf64_noec"][k], "msdf64 noec %s" % batch.names[g])
        if k < 4:
            diff += close(gen(3, s, 64, 64, xf64[g], cfg(overlap=False)), z["msdf64_simple"][k], "msdf64 simple %s" % batch.names[g])
    print("texels differing bitwise from the reference fixtures:", diff)


def test_error_correction_modes_golden(latin):
    batch, _, _ = latin
    z = load_npz("outputs.npz")
    for mode in (1, 2, 3):
        for dist in (0, 1, 2):
            for k, g in enumerate(z["subset"][2:5]):
                got = gen(3, batch.shape(int(g)), 32, 32, z["xf32"][2+k], cfg(ec_mode=mode, ec_dist=dist))
                close(got, z["msdf32_ec%d%d" % (mode, dist)][k], "ec mode %d dist %d %s" % (mode, dist, batch.names[g]))


def test_stencil_stages_and_standalone_correction(latin):
    batch, xf64, _ = latin
    z = load_npz("outputs.npz")
    for k, g in enumerate(z["subset"]):
        s = batch.shape(int(g))
        t = M.SDFTransformation.from_xf(xf64[g])
        for stage in range(4):
            st = np.zeros((64, 64), np.uint8)
            px = z["msdf64_noec"][k].copy()
            M.msdf_error_correction(px, s, t, cfg(buffer=st, stage=stage+1))
            assert (st == z["stages64"][k, stage]).all(), (batch.names[g], stage, int((st != z["stages64"][k, stage]).sum()))
            assert (bits(px) == bits(z["msdf64_noec"][k])).all()  # stage snapshots do not apply
        px = z["msdf64_noec"][k].copy()
        M.msdf_error_correction(px, s, t)
        close(px, z["msdf64"][k], "standalone msdfErrorCorrection %s" % batch.names[g])


def test_synthetic_golden_incl_flips_and_cubics():
    z = load_npz("synth.npz")
    for name in z["cases"]:
        name = str(name)
        w, h, mode, inv, ydown = (int(v) for v in z[name+"_meta"])
        s = FlatShape(z[name+"_co"], z[name+"_pts"], z[name+"_types"], z[name+"_colors"], bool(inv))
        close(gen(mode, s, w, h, z[name+"_xf"], None, bool(ydown)), z[name+"_out"], name)
        close(gen(mode, s, w, h, z[name+"_xf"], cfg(overlap=False), bool(ydown)), z[name+"_out_simple"], name+" simple")


def test_windings_and_distance_queries(latin, oracle):
    batch, _, _ = latin
    z = load_npz("kats.npz")
    s = batch.shape(int(z["oneshot_glyph"]))
    assert (M.contour_windings(s) == oracle.windings(s)).all()
    for sel in (1, 2, 3, 4):
        for ov in (0, 1):
            got = M.shape_distance(s, sel, ov, z["oneshot_pts"])
            want = z["oneshot_%d_%d" % (sel, ov)]
            assert np.allclose(got, want, rtol=1e-12, atol=1e-13), (sel, ov, np.abs(got-want).max())


def test_batched_full_latin_vs_oracle(latin, oracle):
    """BASELINE config 3: the whole Basic-Latin set, mtsdf 64x64, one batched launch; plus msdf."""
    import torch
    batch, xf64, _ = latin
    gb = M.GlyphBatch(batch)
    assert (gb.windings() == np.concatenate([oracle.windings(batch.shape(g)) for g in range(batch.n_glyphs)])).all()
    for mode in (4, 3):
        tiles = gb.generate(mode, 64, 64, xf64)
        torch.cuda.synchronize()
        got = tiles.cpu().numpy()
        want = np.stack([oracle.generate(batch.shape(g), mode, 64, 64, xf64[g]) for g in range(batch.n_glyphs)])
        n = close(got, want, "batched latin mode %d" % mode)
        print("mode %d: %d of %d texels differ bitwise from the oracle" % (mode, n, got.size))
    gb.close()


def test_batch_output_placement_row_stride_and_stencil(latin, oracle):
    """Tiles written into sections of a larger atlas bitmap (out_offset / row_stride), Y-down atlas, stencil returned."""
    import torch
    batch, xf64, _ = latin
    sub = batch.select(range(0, 12))
    gb = M.GlyphBatch(sub)
    w = h = 40
    xfs = np.stack([autoframe(b, w, h, 4) for b in latin[2][:12]])
    atlas_w = 4*w
    atlas = torch.full((3*h, atlas_w, 3), -7., dtype=torch.float32, device="cuda")
    offs = [((g//4)*h*atlas_w+(g % 4)*w)*3 for g in range(12)]
    desc = gb.descriptors(xfs, w, h, 3, y_orientation=M.Y_DOWNWARD, out_offsets=offs, row_stride=atlas_w*3)
    st = torch.zeros((12, h, w), dtype=torch.uint8, device="cuda")
    gb.generate(3, w, h, descriptors=desc, out=atlas, stencil=st)
    torch.cuda.synchronize()
    a = atlas.cpu().numpy()
    for g in range(12):
        sb = np.zeros((h, w), np.uint8)
        want = oracle.generate(sub.shape(g), 3, w, h, xfs[g], y_down=True, stencil=sb)
        y0, x0 = (g//4)*h, (g % 4)*w
        close(a[y0:y0+h, x0:x0+w], want, "atlas tile %d" % g)
        assert (st[g].cpu().numpy() == sb[::-1]).all()  # our stencil rows follow the bitmap's memory rows; the reference keeps it Y-up
    gb.close()


@pytest.mark.parametrize("ec_mode,ec_dist", [(1, 0), (2, 1), (2, 2), (3, 1), (3, 0)])
def test_random_shapes_vs_oracle(oracle, ec_mode, ec_dist):
    rng = np.random.default_rng(77)
    for seed in range(6):
        s = synth.random_shape(5000+seed, n_contours=1+seed % 5, kinds=(1, 2, 3), holes=bool(seed & 1))
        s.inverse_y = bool(seed & 2)
        w, h = int(rng.integers(9, 50)), int(rng.integers(9, 50))  # ragged: not multiples of the 8x8 tile
        xf = autoframe(s.bounds(), w, h, 3)
        xf[1] *= 1.1
        xf[4] *= .7
        for mode in (1, 2, 3, 4):
            for ov in (True, False):
                c = cfg(ov, ec_mode, ec_dist, 1.2, 1.05)
                want = oracle.generate(s, mode, w, h, xf, overlap=ov, ec_mode=ec_mode, ec_dist=ec_dist, min_dev=1.2, min_imp=1.05, y_down=bool(seed & 4))
                close(gen(mode, s, w, h, xf, c, bool(seed & 4)), want, "seed %d mode %d ov %d" % (seed, mode, ov))


def test_fuzz_sweep_vs_oracle():
    """The randomized sweep of tools/fuzz_parity.py, bounded (parity evidence the driver can see): ~2 000 random shapes in ~40
    groups, each group a random (field type, combiner, error-correction mode x distance check, tile size, range, shape family) -- lines /
    quadratics / cubics, holes, nested and overlapping blobs, CJK-like many-contour shapes.  Every texel value against the oracle; 1e-5 is
    the bound, the bitwise count is reported (and has been 0 on every run so far)."""
    import fuzzlib
    r = fuzzlib.run(2000, 401, deadline_s=60)
    print(r)
    assert r["shapes"] >= 500, r                                           # the deadline only trims a slow box's sweep, it must not empty it
    assert r["max_abs_delta"] <= TOL, r
    assert r["values_differing_bitwise"] <= r["values_compared"]*1e-6, r   # bit-identical in practice; a last-ulp libm difference may flip a handful


def test_fuzz_sweep_single_calls_vs_oracle():
    """The same generator, every shape through its own generate*() call: the fused one-launch path (msdf_single.hpp) incl. its fall-backs."""
    import fuzzlib
    r = fuzzlib.run(1500, 411, deadline_s=60, single=True)
    print(r)
    assert r["shapes"] >= 300 and r["max_abs_delta"] <= TOL and r["values_differing_bitwise"] <= r["values_compared"]*1e-6, r


def test_degenerate_and_empty_inputs(oracle):
    empty = FlatShape(np.zeros(1, np.int32), np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros(0, np.int32))
    xf = np.array([10., 10., .1, .1, -.2, .2])
    for mode in (1, 2, 3, 4):
        close(gen(mode, empty, 5, 4, xf), oracle.generate(empty, mode, 5, 4, xf), "empty shape mode %d" % mode)
    assert gen(3, empty, 0, 0, xf).shape == (0, 0, 3)  # zero-size bitmap: no-op
    s = FlatShape.from_contours([
        [(7, (0, 0), (1, 1.5), (2, 0))],
        [],
        [(3, (0, 0), (1, 0)), (5, (1, 0), (.5, 1), (0, 0))],
        [(0, (.2, .2), (.8, .2)), (6, (.8, .2), (.8, .2)), (3, (.8, .2), (.5, .9)), (5, (.5, .9), (.2, .2))],
    ])
    xf = autoframe((0, 0, 2, 1.5), 20, 16, 2)
    for mode in (1, 2, 3, 4):
        for ov in (True, False):
            close(gen(mode, s, 20, 16, xf, cfg(ov)), oracle.generate(s, mode, 20, 16, xf, overlap=ov), "degenerate mode %d ov %d" % (mode, ov))


def test_empty_contours_in_batched_launches(oracle):
    """Empty contours before, between and after the real ones, in launches large enough for the batched forms of k_distance (four tiles
    per wavefront with the scratch in LDS; global scratch): phase 1 walks the glyph's edges across contour boundaries and has to give
    every contour -- also an empty one -- the right slice of the survivor list."""
    def tri(cx, cy, r, colours=(6, 5, 3), flip=False):
        pts = [(cx+r*np.cos(a), cy+r*np.sin(a)) for a in (0.3, 2.4, 4.5)]
        if flip:
            pts = pts[::-1]
        return [(colours[k], pts[k], pts[(k+1) % 3]) for k in range(3)]
    few = FlatShape.from_contours([[], tri(.3, .3, .22), [], [], tri(.65, .6, .3, flip=True), []])
    many = FlatShape.from_contours([[]]+[c for k in range(9) for c in (tri(.15+.08*k, .2+.07*k, .12, flip=bool(k % 2)), [])]+[[], []])
    assert few.n_contours == 6 and many.n_contours == 21
    # 16 or more contour INDICES inside one 16-edge row of phase 1 (a run of empty contours between two small ones) -- the rank
    # key's contour segment must count contour changes, not index differences -- and more contours than edges in k_ec_query's slot path.
    runs = FlatShape.from_contours([tri(.25, .3, .2)]+[[]]*19+[tri(.6, .55, .3, flip=True)]+[[]]*17+[tri(.75, .25, .15)]+[[]]*3)
    assert runs.n_contours == 42 and runs.n_edges == 9
    shapes = [few, many, runs]*60                                             # 180 glyphs x 64 tiles: not a small launch
    xfs = np.stack([autoframe((0, 0, 1, 1), 64, 64, 4)]*len(shapes))
    for ov in (True, False):
        gb = M.GlyphBatch(ShapeBatch.from_shapes(shapes))
        got = gb.generate(3, 64, 64, xfs, config=cfg(ov)).cpu().numpy()
        gb.close()
        for g, s in enumerate((few, many, runs)):
            close(got[g], oracle.generate(s, 3, 64, 64, xfs[g], overlap=ov), "empty contours, shape %d, overlap %d" % (g, ov))
        assert (bits(got[0::3]) == bits(got[0])).all() and (bits(got[1::3]) == bits(got[1])).all() and (bits(got[2::3]) == bits(got[2])).all()
    one = np.zeros((64, 64, 3), np.float32)                                   # the single-shape call (one tile per wavefront, small-launch forms)
    M.generate_msdf(one, runs, M.SDFTransformation.from_xf(xfs[2]))
    close(one, oracle.generate(runs, 3, 64, 64, xfs[2]), "empty contour runs, single-shape call")


def test_many_contours_cjk_like_48(oracle):
    """BASELINE config 4 stand-in at parity size: CJK-like glyphs (8-20 contours, 60-150 edges), msdf 48x48."""
    shapes = [synth.cjk_like_shape(8192+i) for i in range(6)]
    xfs = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in shapes])
    gb = M.GlyphBatch(ShapeBatch.from_shapes(shapes))
    got = gb.generate(3, 48, 48, xfs).cpu().numpy()
    for g, s in enumerate(shapes):
        close(got[g], oracle.generate(s, 3, 48, 48, xfs[g]), "cjk-like %d (%d contours, %d edges)" % (g, s.n_contours, s.n_edges))
    gb.close()


def test_config4_full_size_atlas_8192_glyphs_48(oracle):
    """BASELINE config 4 at FULL size: an 8 192-glyph atlas, msdf 48x48, rendered in one batched launch (CJK-like synthetic glyphs
    stand in for NotoSansCJK, which is not on the box: 8-20 contours, 60-150 edges each). The oracle cannot render 8 192 such glyphs
    in test time, so: a random sample of tiles is compared exactly with the oracle, the batch is re-rendered as 8 glyph-sharded
    parts (what 8 GPUs would do) and must be byte-identical, and duplicated glyphs must produce identical tiles."""
    import torch
    from msdfgen_amd.shard import shard
    base = [synth.cjk_like_shape(20000+i) for i in range(512)]
    shapes = [base[i % 512] for i in range(8192)]
    xfs512 = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])
    xfs = xfs512[np.arange(8192) % 512]
    batch = ShapeBatch.from_shapes(shapes)
    gb = M.GlyphBatch(batch)
    tiles = gb.generate(3, 48, 48, xfs)
    torch.cuda.synchronize()
    got = tiles.cpu().numpy()
    gb.close()
    assert got.shape == (8192, 48, 48, 3) and np.isfinite(got).all()
    rng = np.random.default_rng(4)
    worst = 0
    for g in rng.choice(8192, 24, replace=False):
        worst += close(got[g], oracle.generate(shapes[g], 3, 48, 48, xfs[g]), "atlas glyph %d" % g)
    assert (bits(got[:512]) == bits(got[512:1024])).all() and (bits(got[:512]) == bits(got[7680:])).all()
    parts = []
    for r in range(8):
        sub, sx, (lo, hi) = shard(batch, xfs, r, 8, 48, 48)
        parts.append(M.GlyphBatch(sub).generate(3, 48, 48, sx).cpu().numpy())
    assert (bits(np.concatenate(parts)) == bits(got)).all()
    # 295 k tiles of the global-scratch class = a PERSISTENT launch (work queue, one workspace slice per resident wavefront); the direct
    # mapping (one slice per tile, chunked launches) must give the same bytes
    import os
    os.environ["MSDFHIP_PERSISTENT_ROUNDS"] = "0"
    M.load().msdfhip_reload_tuning()
    try:
        direct = M.GlyphBatch(batch).generate(3, 48, 48, xfs).cpu().numpy()
    finally:
        del os.environ["MSDFHIP_PERSISTENT_ROUNDS"]
        M.load().msdfhip_reload_tuning()
    assert (bits(direct) == bits(got)).all()
    print("config 4: 24 sampled tiles, %d texels differing bitwise" % worst)


def test_scheduling_knobs_do_not_change_a_byte():
    """Round 3 changed WHEN work runs, never what it computes: glyph classes heaviest first (MSDFHIP_NO_CLASS_SORT), side classes at low queue
    priority (MSDFHIP_SIDE_PRIORITY), distance checks per ticket
    (MSDFHIP_QUERY_BATCH), the form and the ticket schedule of the distance checks (round 6: MSDFHIP_QUERY_GRID, MSDFHIP_QUERY_STATIC), which glyphs take the lane-per-candidate chunks / how many edges get LDS slots (MSDFHIP_QUERY_POLICY,
    MSDFHIP_QUERY_LDS -- i.e. k_ec_query's cooperative path with register records vs its chunk walk with batched scalar loads on the SAME
    candidates). 1 024 distinct DejaVu glyphs incl. the 543-edge symbol, msdf with the default correction and mtsdf with ALWAYS_CHECK."""
    import os
    z = load_npz("dejavu8192.npz")
    full = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                      z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
    n_e = full.contour_offsets[full.glyph_contour_offsets[1:]]-full.contour_offsets[full.glyph_contour_offsets[:-1]]
    idx = sorted(set(int(i) for i in np.argsort(-n_e)[:64]) | set(range(0, 8192, 8)))[:1024]
    batch, xfs = full.select(idx), z["xf64"][idx]
    always = M.MSDFGeneratorConfig(True, M.ErrorCorrectionConfig(M.EC_EDGE_PRIORITY, M.ALWAYS_CHECK_DISTANCE))

    def render():
        gb = M.GlyphBatch(batch)
        a = gb.generate(3, 64, 64, xfs).cpu().numpy()
        b = gb.generate(4, 40, 40, xfs, config=always).cpu().numpy()
        gb.close()
        return a, b
    want = render()
    knobs = [{"MSDFHIP_NO_CLASS_SORT": "1"}, {"MSDFHIP_QUERY_BATCH": "5"}, {"MSDFHIP_QUERY_POLICY": "150,0,2147483647,0,4e8"},
             {"MSDFHIP_QUERY_POLICY": "1,128,0,128,0"}, {"MSDFHIP_QUERY_LDS": "700,30"}, {"MSDFHIP_QUERY_LDS": "16,2"}, {"MSDFHIP_SERIAL_CLASSES": "1"}, {"MSDFHIP_SIDE_PRIORITY": "none"},
             {"MSDFHIP_SIDE_PRIORITY": "high"}, {"MSDFHIP_SHORT_ROUNDS": "0"}, {"MSDFHIP_SHORT_ROUNDS": "100000"}, {"MSDFHIP_SHARE_GRID": "0"}, {"MSDFHIP_SHARE_GRID": "3"},
             {"MSDFHIP_PERSISTENT_ROUNDS": "1", "MSDFHIP_PERSISTENT_GRID": "300"},   # (last two: four tiles / one tile per wavefront in every class launch)
             # round 6: the grid form of the distance checks (lanes = candidates x edge slices, partial selectors merged by shuffles) off / with other
             # slice counts, and the three ways a wavefront of k_ec_query comes by its tickets (one counter / dealt statically / first dealt + eight counters)
             {"MSDFHIP_QUERY_GRID": "0"}, {"MSDFHIP_QUERY_GRID": "2"}, {"MSDFHIP_QUERY_GRID": "5"}, {"MSDFHIP_QUERY_GRID": "64"},
             {"MSDFHIP_QUERY_STATIC": "0"}, {"MSDFHIP_QUERY_STATIC": "1"}, {"MSDFHIP_QUERY_STATIC": "1", "MSDFHIP_QUERY_GRID": "0", "MSDFHIP_QUERY_BATCH": "3"},
             {"MSDFHIP_QUERY_STATIC": "2", "MSDFHIP_QUERY_BATCH": "4"}]
    for env in knobs:
        os.environ.update(env)
        M.load().msdfhip_reload_tuning()
        try:
            got = render()
        finally:
            for k in env:
                del os.environ[k]
            M.load().msdfhip_reload_tuning()
        assert (bits(got[0]) == bits(want[0])).all() and (bits(got[1]) == bits(want[1])).all(), env


def test_single_shape_call_paths_are_byte_identical(latin, oracle):
    """One generate*() call of one shape takes ONE launch (k_single_call, msdf_single.hpp): digest, distance field, correction sweep and distance
    checks as phases of one grid, the inputs inside the kernel arguments or read from the pinned staging area, the tile written straight to host
    memory. The same call through every other route -- inputs staged in pinned memory, uploaded inputs + device outputs, and the batched
    launch sequence of round 3 -- must give the same bytes; and all of them the oracle's (1e-5; in practice bit-identical). Covers all four field
    types, both combiners, an error-correction config with distance checks at every texel, a stencil buffer (which keeps the device-output
    route), ragged sizes, a many-contour shape (combiner scratch in LDS / in the global workspace), a shape too large for the kernel arguments
    and one beyond the fused path's limits (falls back by itself)."""
    import os
    batch, xf64, bounds = latin
    cases = []
    for g in (1, 12, 33, 45, 77):
        cases.append((batch.shape(g), 64, 64, xf64[g]))
    s = synth.cjk_like_shape(8601)
    cases.append((s, 48, 40, autoframe(s.bounds(), 48, 40, 4)))
    s = synth.random_shape(8700, n_contours=30, edges_per_contour=(3, 5), kinds=(1, 2))
    cases.append((s, 33, 47, autoframe(s.bounds(), 33, 47, 3)))
    s = synth.logo_shape(5)                                                       # 926 edges: neither argument payload nor (at 136 x 136 = 289 tiles) the fused path
    cases.append((s, 72, 72, autoframe(s.bounds(), 72, 72, 4)))
    cases.append((s, 136, 136, autoframe(s.bounds(), 136, 136, 4)))
    always = cfg(True, M.EC_EDGE_PRIORITY, M.ALWAYS_CHECK_DISTANCE)
    crowded = cfg(False, M.EC_EDGE_PRIORITY, M.ALWAYS_CHECK_DISTANCE)              # overlapping strokes without overlap support: candidate overflow -> the fused call hands over

    def render():
        out = []
        for s, w, h, xf in cases:
            out.append(gen(1, s, w, h, xf))
            out.append(gen(2, s, w, h, xf, M.GeneratorConfig(False)))
            out.append(gen(3, s, w, h, xf))
            out.append(gen(3, s, w, h, xf, cfg(False, M.EC_DISABLED)))
            out.append(gen(4, s, w, h, xf, always))
            st = np.zeros((h, w), np.uint8)
            out.append(gen(3, s, w, h, xf, cfg(True, buffer=st)))
            out.append(st)
        s = synth.cjk_like_shape(8801)
        out.append(gen(3, s, 48, 48, autoframe(s.bounds(), 48, 48, 4), crowded))
        return out
    want = render()
    lib = M.load()
    ph = (C.c_double*8)()
    lib.msdfhip_debug_single_call_phases(ph, 1)
    render()
    lib.msdfhip_debug_single_call_phases(ph, 1)
    assert int(ph[0]) >= 10, "the fused launch did not run (%d calls counted)" % int(ph[0])
    for env in ({"MSDFHIP_NO_ARG_PAYLOAD_SINGLE": "1"}, {"MSDFHIP_NO_ZERO_COPY_SINGLE": "1"}, {"MSDFHIP_NO_FUSED_SINGLE": "1"}):
        os.environ.update(env)
        lib.msdfhip_reload_tuning()
        try:
            got = render()
        finally:
            for k in env:
                del os.environ[k]
            lib.msdfhip_reload_tuning()
        for i, (a, b) in enumerate(zip(got, want)):
            assert a.shape == b.shape and (a.view(np.uint8) == b.view(np.uint8)).all(), (env, i)
    k = 0
    for s, w, h, xf in cases[:7]:                                                 # (the 926-edge logo costs the oracle seconds per render: the byte comparison above covers it)
        close(want[k], oracle.generate(s, 1, w, h, xf), "sdf")
        close(want[k+1], oracle.generate(s, 2, w, h, xf, overlap=False), "psdf")
        close(want[k+2], oracle.generate(s, 3, w, h, xf), "msdf")
        close(want[k+3], oracle.generate(s, 3, w, h, xf, overlap=False, ec_mode=0), "msdf, simple combiner, no correction")
        close(want[k+4], oracle.generate(s, 4, w, h, xf, ec_mode=2, ec_dist=2), "mtsdf, distance checks everywhere")
        sb = np.zeros((h, w), np.uint8)
        close(want[k+5], oracle.generate(s, 3, w, h, xf, stencil=sb), "msdf with stencil")
        assert (want[k+6] == sb).all()
        k += 7
    s = synth.cjk_like_shape(8801)
    close(want[-1], oracle.generate(s, 3, 48, 48, autoframe(s.bounds(), 48, 48, 4), overlap=False, ec_mode=2, ec_dist=2), "candidate overflow")


def _fallbacks(lib, reset=0):
    t, l, r = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
    lib.msdfhip_single_call_fallbacks(C.byref(t), C.byref(l), C.byref(r), reset)
    return int(t.value), int(l.value), int(r.value)


def test_single_call_whose_grid_barrier_gives_up_is_rerun_not_failed(latin):
    """core/msdfgen.cpp:92-98 cannot fail. k_single_call's grid barrier needs the whole launch resident; when it is not (another thread's persistent
    kernel holds the slots) the barrier gives up after a bounded wait and the call must be RERUN through the batched sequence, not failed (ADVICE r4,
    VERDICT r4 next #6). Forced here: a spin limit of one look makes (nearly) every launch give up; the bytes must be those of the undisturbed call, the
    counters must show that the slow road was taken, and the arena must be usable afterwards (its barrier counters restart from zero)."""
    import os
    batch, xf64, _ = latin
    lib = M.load()
    cases = [(batch.shape(g), 64, 64, xf64[g]) for g in (1, 12, 33, 45, 77)]
    s = synth.cjk_like_shape(8601)
    cases.append((s, 48, 40, autoframe(s.bounds(), 48, 40, 4)))

    def render():
        return [gen(m, s, w, h, xf) for s, w, h, xf in cases for m in (3, 4)]
    want = render()
    _fallbacks(lib, 1)
    os.environ["MSDFHIP_SINGLE_SPIN_LIMIT"] = "1"
    lib.msdfhip_reload_tuning()
    try:
        got = render()
    finally:
        del os.environ["MSDFHIP_SINGLE_SPIN_LIMIT"]
        lib.msdfhip_reload_tuning()
    timeouts, lost, refused = _fallbacks(lib, 1)
    assert timeouts >= len(cases), "the forced barrier time-outs did not happen (%d)" % timeouts
    for i, (a, b) in enumerate(zip(got, want)):
        assert (a.view(np.uint8) == b.view(np.uint8)).all(), i
    ph = (C.c_double*8)()
    lib.msdfhip_debug_single_call_phases(ph, 1)
    again = render()                                                              # the same arenas, counters restarted: the fused launch works again
    lib.msdfhip_debug_single_call_phases(ph, 1)
    assert int(ph[0]) >= len(cases) and _fallbacks(lib, 1)[0] == 0
    for i, (a, b) in enumerate(zip(again, want)):
        assert (a.view(np.uint8) == b.view(np.uint8)).all(), i


def test_single_calls_next_to_a_large_batch_never_fail(latin):
    """Eight threads hammer generateMSDF() single calls while another thread loops an 8 192-glyph CJK-like batch, whose global-scratch class runs as a
    PERSISTENT launch that holds every wavefront slot it gets. No call may fail, every tile must be byte-identical to the undisturbed call
    (VERDICT r4 next #6). Whether launches had to give up depends on timing; the counters are printed, not asserted."""
    import torch
    batch, xf64, _ = latin
    lib = M.load()
    picks = [1, 5, 12, 20, 33, 45, 60, 77]
    want = [gen(3, batch.shape(g), 64, 64, xf64[g]) for g in picks]
    base = [synth.cjk_like_shape(20000+i) for i in range(128)]
    cj = ShapeBatch.from_shapes([base[i % 128] for i in range(8192)])
    cx = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])[np.arange(8192) % 128]
    gb = M.GlyphBatch(cj)
    out = torch.empty((8192, 48, 48, 3), dtype=torch.float32, device="cuda")
    desc = gb.descriptors(cx, 48, 48, 3)
    stop = threading.Event()
    errors = []

    def big():
        try:
            side = torch.cuda.Stream()
            while not stop.is_set():
                gb.generate(3, 48, 48, descriptors=desc, out=out, stream=side)
                side.synchronize()
        except Exception as e:                                                    # noqa: BLE001
            errors.append(("batch", repr(e)))

    def small(k):
        try:
            for it in range(40):
                g = picks[(k+it) % len(picks)]
                got = gen(3, batch.shape(g), 64, 64, xf64[g])
                if not (got.view(np.uint8) == want[(k+it) % len(picks)].view(np.uint8)).all():
                    errors.append(("bytes", k, it))
        except Exception as e:                                                    # noqa: BLE001
            errors.append(("call", k, repr(e)))
    _fallbacks(lib, 1)
    tb = threading.Thread(target=big)
    tb.start()
    ts = [threading.Thread(target=small, args=(k,)) for k in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop.set()
    tb.join()
    gb.close()
    print("fused launches next to a persistent batch: barrier time-outs %d, lost flags %d, refused %d" % _fallbacks(lib, 1))
    assert not errors, errors[:5]


def test_very_many_contours_use_the_global_combiner_scratch(oracle):
    """Maximum-size edge case: 260 overlapping contours. The overlapping combiner's per-contour scratch (260*3*512 B) exceeds the CU's
    LDS, so k_distance / k_ec_query keep it in a global workspace; results must not change."""
    rng = np.random.default_rng(9)
    contours = []
    for i in range(260):
        cx, cy, r = rng.uniform(.05, .95), rng.uniform(.05, .95), rng.uniform(.02, .08)
        a = rng.uniform(0, 6.28)
        pts = [(cx+r*np.cos(a+k*2.0944), cy+r*np.sin(a+k*2.0944)) for k in range(3)]
        if i % 5 == 0:
            pts = pts[::-1]
        contours.append([((3, 5, 6)[k], pts[k], pts[(k+1) % 3]) if (i+k) % 2 else ((3, 5, 6)[k], pts[k], (cx, cy), pts[(k+1) % 3]) for k in range(3)])
    s = FlatShape.from_contours(contours)
    xf = autoframe((0, 0, 1, 1), 64, 56, 3)
    for mode in (3, 4, 2):
        close(gen(mode, s, 64, 56, xf), oracle.generate(s, mode, 64, 56, xf), "260 contours mode %d" % mode)
    close(gen(3, s, 64, 56, xf, cfg(ec_dist=M.ALWAYS_CHECK_DISTANCE)), oracle.generate(s, 3, 64, 56, xf, ec_dist=2), "260 contours, always check distance")


def test_sharding_is_byte_invariant(latin):
    """Glyph-sharded execution: the bytes of every tile are identical whether the batch is rendered whole or in parts."""
    from msdfgen_amd.shard import shard
    batch, xf64, _ = latin
    whole = M.GlyphBatch(batch).generate(3, 64, 64, xf64).cpu().numpy()
    for world in (2, 8):
        parts = []
        for r in range(world):
            sub, xfs, (lo, hi) = shard(batch, xf64, r, world, 64, 64)
            parts.append(M.GlyphBatch(sub).generate(3, 64, 64, xfs).cpu().numpy() if hi > lo else np.zeros((0, 64, 64, 3), np.float32))
        assert (bits(np.concatenate(parts)) == bits(whole)).all()


def test_config5_logo_1024_sampled_against_oracle(oracle):
    """BASELINE config 5 at FULL size (1024x1024, ~900 cubic edges, 40 overlapping / self-intersecting contours): the oracle cannot
    render 1M texels in test time, so (a) the full pipeline is compared exactly on a 96x96 render of the same shape, and (b) at
    1024x1024 a random sample of texels of the pre-correction field is compared with oracle distance queries at the texel centres."""
    s = synth.logo_shape(5)
    xf = autoframe(s.bounds(), 96, 96, 4)
    close(gen(3, s, 96, 96, xf), oracle.generate(s, 3, 96, 96, xf), "logo 96x96 full pipeline")
    xf = autoframe(s.bounds(), 1024, 1024, 8)
    big = gen(3, s, 1024, 1024, xf, cfg(ec_mode=M.EC_DISABLED))
    rng = np.random.default_rng(5)
    xy = rng.integers(0, 1024, (1500, 2))
    pts = np.stack([(xy[:, 0]+.5)/xf[0]-xf[2], (xy[:, 1]+.5)/xf[1]-xf[3]], 1)
    d = oracle.shape_distance(s, 3, True, pts)[:, :3]
    want = (np.float64(1)/(xf[5]-xf[4])*(d+(-xf[4]))).astype(np.float32)
    close(big[xy[:, 1], xy[:, 0]], want, "logo 1024x1024 sampled texels")
    full = gen(3, s, 1024, 1024, xf)  # with error correction: corrected texels are exactly the median of the uncorrected ones
    changed = (bits(full) != bits(big)).any(axis=2)
    med = np.median(big, axis=2)
    assert changed.any() and (full[changed] == med[changed][:, None]).all()


def test_concurrent_host_threads(latin, oracle):
    """The reference's generate* are re-entrant; so are ours."""
    batch, xf64, _ = latin
    want = {g: oracle.generate(batch.shape(g), 3, 64, 64, xf64[g]) for g in range(8)}
    errors = []

    def work(g):
        try:
            for _ in range(3):
                close(gen(3, batch.shape(g), 64, 64, xf64[g]), want[g], "thread %d" % g)
        except Exception as e:  # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work, args=(g,)) for g in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


def test_micro_batched_concurrent_calls_mixed_parameters(latin, oracle):
    """concurrent single-shape calls are combined into device batches per (mode, size, config). 24 host threads
    issue different kinds of calls at once (msdf 64x64, mtsdf 40x40 Y-down with stencil, sdf 32x32, standalone sign correction,
    one thread with an invalid shape); every result must equal the oracle's, the bad call must fail alone, and the outcome must
    not depend on whether batching is enabled."""
    batch, xf64, bounds = latin
    xf40 = [autoframe(b, 40, 40, 4) for b in bounds]
    xf32 = [autoframe(b, 32, 32, 4) for b in bounds]
    errors, failures = [], []

    def work(t):
        try:
            for i in range(6):
                g = (5*t+i) % batch.n_glyphs
                s = batch.shape(g)
                kind = t % 4
                if t == 23:
                    bad = FlatShape(s.contour_offsets, s.points, s.types.copy(), s.colors)
                    bad.types[:] = 9                                            # past the Python-side validation: the C ABI must reject it
                    try:
                        gen(3, bad, 64, 64, xf64[g])
                        errors.append("invalid edge type accepted")
                    except M.MsdfHipError as e:
                        failures.append(e.code)
                elif kind == 0:
                    close(gen(3, s, 64, 64, xf64[g]), oracle.generate(s, 3, 64, 64, xf64[g]), "msdf %d" % g)
                elif kind == 1:
                    st, want_st = np.zeros((40, 40), np.uint8), np.zeros((40, 40), np.uint8)
                    got = gen(4, s, 40, 40, xf40[g], cfg(buffer=st), y_down=True)
                    close(got, oracle.generate(s, 4, 40, 40, xf40[g], y_down=True, stencil=want_st), "mtsdf %d" % g)
                    assert (st == want_st).all()                                 # rows in the reference's (upward) order, also for a Y-down bitmap
                elif kind == 2:
                    close(gen(1, s, 32, 32, xf32[g]), oracle.generate(s, 1, 32, 32, xf32[g]), "sdf %d" % g)
                else:
                    f = oracle.generate(s, 3, 32, 32, xf32[g], ec_mode=0)
                    got = M.distance_sign_correction((1-f).copy(), s, M.SDFTransformation.from_xf(xf32[g]), .5, M.FILL_ODD)
                    assert (bits(got) == bits(oracle.sign_correction(s, 1-f, xf32[g], .5, 1))).all()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))
    for enabled in (True, False):
        M.set_microbatch(256 if enabled else 1, 2)
        M.microbatch_stats(reset=True)
        del failures[:]
        threads = [threading.Thread(target=work, args=(t,)) for t in range(24)]
        [t.start() for t in threads]
        [t.join() for t in threads]
        st = M.microbatch_stats()
        print("micro-batching %s: %s" % ("on" if enabled else "off", st))
        assert not errors, errors[:3]
        assert failures == [M.lib.ERR_INVALID]*6
        assert st["calls"] == 23*6 and (enabled or st["batches"] == st["calls"])
    M.set_microbatch(256, 4)


# ---- distanceSignCorrection (core/rasterization.cpp:19-92) on the device

@pytest.mark.parametrize("seed", range(6))
def test_distance_sign_correction_standalone_vs_oracle(oracle, seed):
    """msdfhip_distance_sign_correction in place on 1-, 3- and 4-channel fields: all fill rules, inverted fields (every texel
    flips), exact zero-value medians (the neighbour vote), Y-down bitmaps and inverse-Y shapes."""
    rng = np.random.default_rng(1300+seed)
    s = synth.random_shape(7100+seed, n_contours=1+seed % 4, kinds=(1, 2, 3), holes=bool(seed & 1))
    s.inverse_y = bool(seed & 2)
    w, h = int(rng.integers(9, 70)), int(rng.integers(9, 70))
    xf = autoframe(s.bounds(), w, h, 3)
    yd = bool(seed & 4)
    for mode in (1, 3, 4):
        field = oracle.generate(s, mode, w, h, xf, ec_mode=0, y_down=yd)
        field = 1-field if seed % 3 == 0 else field
        if mode >= 3:
            field[rng.integers(0, h, 10), rng.integers(0, w, 10)] = .5
            field[0, 0] = field[h-1, w-1] = field[0, w-1] = .5
        for rule in range(4):
            want = oracle.sign_correction(s, field, xf, .5, rule, y_down=yd)
            got = M.distance_sign_correction(field.copy(), s, M.SDFTransformation.from_xf(xf), .5, rule, M.Y_DOWNWARD if yd else M.Y_UPWARD)
            assert (bits(got) == bits(want)).all(), "seed %d mode %d rule %d: %d texels differ" % (seed, mode, rule, int((bits(got) != bits(want)).sum()))
        want = oracle.sign_correction(s, field, xf, .25, 0, y_down=yd)
        got = M.distance_sign_correction(field.copy(), s, M.Projection((xf[0], xf[1]), (xf[2], xf[3])), .25, M.FILL_NONZERO, M.Y_DOWNWARD if yd else M.Y_UPWARD)
        assert (bits(got) == bits(want)).all()


def test_distance_sign_correction_many_edges_chunked_lists(oracle):
    """Shapes with more than 128 edges exceed the per-row list capacity kept in LDS: the kernel then walks the edges in chunks
    (k_sign_correction). CJK-like glyphs (~100 edges, unchunked), 150-300 edge blobs, the 926-edge cubic logo on a wide bitmap."""
    rng = np.random.default_rng(77)
    cases = [(synth.cjk_like_shape(8300+i), 48, 48) for i in range(2)]
    cases += [(synth.random_shape(8400+i, n_contours=18+4*i, edges_per_contour=(7, 12), kinds=(1, 2, 3)), 40+9*i, 56-7*i) for i in range(3)]
    cases += [(synth.logo_shape(5), 136, 72)]
    assert max(s.n_edges for s, _, _ in cases) > 900 and sum(s.n_edges > 128 for s, _, _ in cases) >= 4
    for s, w, h in cases:
        xf = autoframe(s.bounds(), w, h, 4)
        for mode in (1, 3, 4):
            field = oracle.generate(s, mode, w, h, xf, ec_mode=0)
            if mode >= 3:
                field[rng.integers(0, h, 12), rng.integers(0, w, 12)] = .5
            for rule in (0, 1, 3):
                want = oracle.sign_correction(s, field, xf, .5, rule)
                got = M.distance_sign_correction(field.copy(), s, M.SDFTransformation.from_xf(xf), .5, rule)
                assert (bits(got) == bits(want)).all(), "%d edges mode %d rule %d: %d texels differ" % (s.n_edges, mode, rule, int((bits(got) != bits(want)).sum()))
        assert (M.rasterize(np.zeros((h, w, 1), np.float32), s, M.SDFTransformation.from_xf(xf), 1) == oracle.rasterize(s, w, h, xf, 1)).all()


def test_scanline_pass_pipeline_stays_on_device(latin, oracle):
    """The reference's -scanline flow (main.cpp:1233-1298): generate with the simple combiner and no correction, sign-correct
    against the scanline fill, then msdfErrorCorrection with DO_NOT_CHECK_DISTANCE -- as one batched call, vs the oracle's
    composition of the three steps. Also the 1-channel flow (no correction step) and the sign pass with correction disabled."""
    import torch
    batch, xf64, _ = latin
    gb = M.GlyphBatch(batch)
    G = batch.n_glyphs
    for mode in (3, 4):
        for rule in (M.FILL_NONZERO, M.FILL_ODD):
            c = cfg(overlap=False, ec_mode=M.EC_EDGE_PRIORITY, ec_dist=M.DO_NOT_CHECK_DISTANCE)
            st = torch.zeros((G, 64, 64), dtype=torch.uint8, device="cuda")
            got = gb.generate(mode, 64, 64, xf64, config=c, stencil=st, scanline_pass=True, fill_rule=rule).cpu().numpy()
            want = np.zeros_like(got)
            for g in range(G):
                s = batch.shape(g)
                f = oracle.generate(s, mode, 64, 64, xf64[g], overlap=False, ec_mode=0)
                f = oracle.sign_correction(s, f, xf64[g], .5, rule)
                want[g] = oracle.error_correction(s, f, xf64[g], overlap=False, ec_mode=2, ec_dist=0)
            n = close(got, want, "scanline flow mode %d rule %d" % (mode, rule))
            print("scanline flow mode %d rule %d: %d of %d texels differ bitwise" % (mode, rule, n, got.size))
    # correction disabled: distance -> sign pass -> caller's tiles
    c = cfg(overlap=True, ec_mode=M.EC_DISABLED)
    got = gb.generate(3, 64, 64, xf64, config=c, scanline_pass=True).cpu().numpy()
    for g in range(0, G, 7):
        s = batch.shape(g)
        want = oracle.sign_correction(s, oracle.generate(s, 3, 64, 64, xf64[g], ec_mode=0), xf64[g], .5, 0)
        close(got[g], want, "sign pass only, glyph %d" % g)
    got = gb.generate(1, 64, 64, xf64, config=M.GeneratorConfig(False), scanline_pass=True, sdf_zero_value=.5).cpu().numpy()
    for g in range(0, G, 7):
        s = batch.shape(g)
        want = oracle.sign_correction(s, oracle.generate(s, 1, 64, 64, xf64[g], overlap=False), xf64[g], .5, 0)
        close(got[g], want, "sdf + sign pass, glyph %d" % g)
    gb.close()


def test_scanline_pass_makes_the_sign_follow_the_fill(oracle):
    """What the pass is for: whatever the winding of the contours (here: as generated, and with the field inverted as a reversed
    outline would produce), afterwards `distance > zero value` agrees with the scanline fill at every texel centre."""
    s = synth.random_shape(7300, n_contours=2, kinds=(1, 2), holes=True)
    w = h = 48
    xf = autoframe(s.bounds(), w, h, 4)
    plain = gen(1, s, w, h, xf, M.GeneratorConfig(False))
    fill = M.rasterize(np.zeros((h, w, 1), np.float32), s, M.SDFTransformation.from_xf(xf)) > 0
    assert (fill == (oracle.rasterize(s, w, h, xf) > 0)).all() and 0 < fill.mean() < 1
    for field in (plain, 1-plain):
        fixed = M.distance_sign_correction(field.copy(), s, M.SDFTransformation.from_xf(xf))
        assert (bits(fixed) == bits(oracle.sign_correction(s, field, xf, .5, 0))).all()
        assert ((fixed > .5) == fill)[fixed != .5].all()


def test_rasterize_vs_oracle(oracle):
    """msdfhip_rasterize (core/rasterization.cpp:8-16): the fill bit of every texel centre, all fill rules."""
    for seed in range(5):
        s = synth.random_shape(7400+seed, n_contours=1+seed % 4, kinds=(1, 2, 3), holes=bool(seed & 1))
        s.inverse_y = bool(seed & 2)
        w, h = 30+7*seed, 61-5*seed
        xf = autoframe(s.bounds(), w, h, 2)
        for rule in range(4):
            got = M.rasterize(np.full((h, w, 1), -3, np.float32), s, M.Projection((xf[0], xf[1]), (xf[2], xf[3])), rule, M.Y_DOWNWARD if seed & 4 else M.Y_UPWARD)
            assert (got == oracle.rasterize(s, w, h, xf, rule, y_down=bool(seed & 4))).all(), (seed, rule)


def test_tiles_to_bytes_atlas_blit(latin, oracle):
    """pixelFloatToByte + atlas-rectangle blit on the device (msdfhip_tiles_to_bytes): generated mtsdf tiles and a
    tile of boundary values go into a uint8 atlas with padding; untouched atlas bytes must stay untouched."""
    import torch
    batch, xf64, _ = latin
    sub = batch.select(range(20, 32))
    gb = M.GlyphBatch(sub)
    w = h = 40
    xfs = np.stack([autoframe(b, w, h, 4) for b in latin[2][20:32]])
    tiles = gb.generate(4, w, h, xfs)
    k = np.arange(0, 256, dtype=np.float64)
    edge = np.concatenate([(k+d)/255. for d in (0, .5, -.5)]).astype(np.float32)
    special = np.concatenate([edge, np.nextafter(edge, np.float32(2)), np.nextafter(edge, np.float32(-2)), np.array([np.nan, np.inf, -np.inf, -3, 7], np.float32)])
    tiles[0].view(-1)[:special.size] = torch.from_numpy(special).to(tiles.device)
    cols, pad = 4, 3
    aw, ah = cols*(w+pad)+5, 3*(h+pad)+2
    atlas = torch.full((ah, aw, 4), 77, dtype=torch.uint8, device="cuda")
    offs = [(((g//cols)*(h+pad)+1)*aw+(g % cols)*(w+pad)+2)*4 for g in range(12)]
    gb.to_bytes(tiles, atlas, offs, aw*4)
    torch.cuda.synchronize()
    got, src = atlas.cpu().numpy(), tiles.cpu().numpy()
    want = np.full((ah, aw, 4), 77, np.uint8)
    for g in range(12):
        y0, x0 = (g//cols)*(h+pad)+1, (g % cols)*(w+pad)+2
        want[y0:y0+h, x0:x0+w] = oracle.pixel_float_to_byte(src[g])
    assert (got == want).all()
    with pytest.raises(ValueError):
        gb.to_bytes(tiles, atlas, [o+aw*ah*4 for o in offs], aw*4)
    gb.close()


# ---- shape preparation on the device (Shape::normalize + edgeColoringSimple)

def _prep_fixture():
    z = load_npz("prep.npz")
    mk = lambda t: ShapeBatch(z[t+"_gco"].astype(np.int32), z[t+"_co"].astype(np.int32), z[t+"_points"], z[t+"_types"].astype(np.int32),  # noqa: E731
                              z[t+"_colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
    return mk("raw"), mk("prep"), mk("norm"), z["seeds"]


def _same_batch(a, b, what):
    assert (a.glyph_contour_offsets == b.glyph_contour_offsets).all() and (a.contour_offsets == b.contour_offsets).all(), what+": offsets"
    assert (a.types == b.types).all(), what+": types"
    assert (a.colors == b.colors).all(), what+": colours"
    pa, pb = np.ascontiguousarray(a.points, np.float64), np.ascontiguousarray(b.points, np.float64)
    assert (pa.view(np.uint64) == pb.view(np.uint64)).all(), what+": %d control-point values differ bitwise" % int((pa.view(np.uint64) != pb.view(np.uint64)).sum())


def test_shape_preparation_on_device_matches_reference_fixture(oracle):
    """Raw outlines (306 font glyphs + hand-built rare cases: single-edge contours, two-edge teardrops, cusps that deconverge into
    cubics) prepared by msdfhip_batch_create_prepared: bit-identical to the compiled reference's normalize + edgeColoringSimple
    (tests/golden/prep.npz); the prepared batch then renders exactly what the oracle renders from the reference-prepared shapes."""
    raw, prep, norm, seeds = _prep_fixture()
    gb = M.GlyphBatch.from_raw(raw, True, 1, 3.0, seeds=seeds)
    _same_batch(gb.shapes, prep, "normalize + edgeColoringSimple")
    pick = list(range(0, raw.n_glyphs, 13))+list(range(raw.n_glyphs-23, raw.n_glyphs))
    extent = [gb.shapes.shape(g).bounds() for g in range(raw.n_glyphs)]
    pick = [g for g in pick if extent[g][2]-extent[g][0] > 1e-3 and extent[g][3]-extent[g][1] > 1e-3]   # a flat outline has no frame
    # synthetic-2 / -10 contain a quadratic that runs out and straight back over itself (p0 == p2, collinear control points): the sign of
    # its distance is nonZeroSign(cross) of two PARALLEL vectors, i.e. rounding noise of the cubic root -- the reference's own result
    # there changes with the last ulp of acos/cos. Their preparation is compared above; their rendering is not.
    pick = [g for g in pick if raw.names[g] not in ("synthetic-2", "synthetic-10")]
    xfs = np.stack([autoframe(b if b[2]-b[0] > 1e-3 and b[3]-b[1] > 1e-3 else (0, 0, 1, 1), 32, 32, 4) for b in extent])
    tiles = gb.generate(3, 32, 32, xfs).cpu().numpy()
    for g in pick:
        close(tiles[g], oracle.generate(prep.shape(g), 3, 32, 32, xfs[g]), "prepared glyph %s" % raw.names[g])
    gb.close()
    gb = M.GlyphBatch.from_raw(raw, True, 0)
    _same_batch(gb.shapes, norm, "normalize only")
    gb.close()
    gb = M.GlyphBatch.from_raw(raw, True, 2, 3.0, seeds=seeds)          # edgeColoringInkTrap
    want = [oracle.shape_prepare(raw.shape(g), True, 2, 3.0, int(seeds[g])) for g in range(raw.n_glyphs)]
    _same_batch(gb.shapes, ShapeBatch.from_shapes([FlatShape(f.contour_offsets, f.points, f.types, f.colors) for f in want]), "normalize + edgeColoringInkTrap")
    gb.close()
    gb = M.GlyphBatch.from_raw(prep, False, 0)                          # nothing to do: a plain upload
    _same_batch(gb.shapes, prep, "identity")
    gb.close()


def test_contour_windings_of_the_digest_kernels_vs_oracle(oracle):
    """Contour::winding as the digest computes it since round 4 (lanes = edges, ordered sums; msdf_prep.hpp: contourWindingsWave): a batch of shapes
    with long contours, > 64 contours, empty / short contours and contours of almost no area through k_prep_records (a lane per contour, the long
    ones by their wavefront together), and each of the first 40 alone (its own upload)."""
    from test_device_logic_host import winding_stress_shapes
    shapes = winding_stress_shapes(12)
    want = [oracle.windings(s) for s in shapes]
    gb = M.GlyphBatch(ShapeBatch.from_shapes(shapes))
    assert (gb.windings() == np.concatenate(want)).all()
    gb.close()
    for s, w in zip(shapes[:40], want[:40]):
        assert (M.contour_windings(s) == w).all()


def test_backtracking_curves_follow_the_host_build_with_the_devices_own_transcendentals(oracle):
    """synthetic-2 / -10 (excluded above): a quadratic that runs out and straight back over itself. The sign of its distance is last-ulp noise
    of acos / cos in solveCubicNormed, where the kernels use their own < 1 ulp implementations (msdf_device.hpp) and the reference glibc's.
    DESIGN.md 4 claims the kernels' LOGIC is not involved: the product's device headers compiled for the host with the same transcendentals
    (tests/hostemu, -DMSDF_LEAN_MATH) must then reproduce the GPU's tiles bit for bit -- asserted here -- and every value that differs from
    the oracle must be one of those sign flips (|got| == |want| about the mapped zero level), not a different magnitude."""
    from emu import Emu
    raw, prep, norm, seeds = _prep_fixture()
    emu = Emu(lean=True)
    names = list(raw.names)
    for name in ("synthetic-2", "synthetic-10"):
        g = names.index(name)
        s = prep.shape(g)
        xf = autoframe(s.bounds(), 32, 32, 4)
        for mode in (1, 3):
            for ov in (True, False):
                got = gen(mode, s, 32, 32, xf, cfg(ov, M.EC_DISABLED) if mode >= 3 else M.GeneratorConfig(ov))
                want = emu.generate(s, mode, 32, 32, xf, overlap=ov, ec_mode=0)
                assert (bits(got) == bits(want)).all(), "%s mode %d overlap %d: %d values differ from the lean host build" % (name, mode, ov, int((bits(got) != bits(want)).sum()))
                ref = oracle.generate(s, mode, 32, 32, xf, overlap=ov, ec_mode=0)
                bad = bits(got) != bits(ref)
                # mapped value = scale*(d+translate): a sign flip of d mirrors the value about scale*translate
                if bad.any():
                    ms, mt = distance_mapping(xf[4], xf[5])
                    zero = ms*mt
                    assert np.allclose(got[bad].astype(np.float64)-zero, -(ref[bad].astype(np.float64)-zero), atol=1e-5), (name, mode, ov)
                assert bad.sum() <= 16, (name, mode, ov, int(bad.sum()))


def test_shape_preparation_random_vs_oracle(oracle):
    rng = np.random.default_rng(8)
    shapes, want, seeds = [], [], []
    for i in range(200):
        s = synth.random_shape(9500+i, n_contours=int(rng.integers(1, 6)), edges_per_contour=(1, 8), kinds=(1, 2, 3), holes=bool(i & 1))
        s.colors[:] = 7
        sd = int(rng.integers(0, 2**50))
        shapes.append(s), seeds.append(sd)
        fa = oracle.shape_prepare(s, True, 1, 2.5, sd)
        want.append(FlatShape(fa.contour_offsets, fa.points, fa.types, fa.colors))
    gb = M.GlyphBatch.from_raw(ShapeBatch.from_shapes(shapes), True, 1, 2.5, seeds=np.array(seeds, np.uint64))
    _same_batch(gb.shapes, ShapeBatch.from_shapes(want), "random shapes, angle 2.5")
    gb.close()
    # long contours: several 64-edge rounds of the lanes-=-edges colouring kernel (corner ballots, prefix counts across words), smooth / one-corner /
    # many-corner alike, and one beyond its LDS tables (serial form on lane 0)
    big, bwant, bseeds = [], [], []
    for i, (n_edges, wobble, kinds) in enumerate([(70, .05, (2,)), (130, .6, (1, 2)), (200, .9, (1,)), (257, .3, (1, 2, 3)), (64, .5, (1, 2)), (2100, .7, (1, 2))]):
        s = synth.random_shape(9900+i, n_contours=1+i % 2, edges_per_contour=(n_edges, n_edges), kinds=kinds, wobble=wobble)
        s.colors[:] = 7
        big.append(s), bseeds.append(1000+i)
        fa = oracle.shape_prepare(s, False, 1, 3.0, 1000+i)
        bwant.append(FlatShape(fa.contour_offsets, fa.points, fa.types, fa.colors))
    def polygon(pts):                                                          # closed contour of line edges through pts
        pts = np.asarray(pts, np.float64)
        e = np.zeros((len(pts), 8))
        e[:, 0:2], e[:, 2:4] = pts, np.roll(pts, -1, axis=0)
        return FlatShape(np.array([0, len(pts)], np.int32), e, np.ones(len(pts), np.int32), np.full(len(pts), 7, np.int32))
    ang = np.linspace(0, 2*np.pi, 200, endpoint=False)
    ring = np.stack([np.cos(ang), np.sin(ang)], 1)
    td = np.linspace(0, 2*np.pi, 301, endpoint=False)
    drop = np.stack([np.cos(td), np.sin(td)*np.sin(td/2)], 1)                  # smooth but for the cusp at t = 0
    for k, s in enumerate([polygon(ring),                                      # 200-gon, 1.8 degrees per vertex: smooth, no corner
                           polygon(drop),                                      # ONE corner, 301 edges: a teardrop, coloured in thirds by position (WHITE in the middle)
                           polygon(np.concatenate([ring[:70], [[2.5, .5]], ring[80:160], [[-.2, -2.5]]]))]):   # two apexes + the two open ends: a few long splines
        big.append(s), bseeds.append(2000+k)
        fa = oracle.shape_prepare(s, False, 1, 3.0, 2000+k)
        bwant.append(FlatShape(fa.contour_offsets, fa.points, fa.types, fa.colors))
    gb = M.GlyphBatch.from_raw(ShapeBatch.from_shapes(big), False, 1, 3.0, seeds=np.array(bseeds, np.uint64))
    _same_batch(gb.shapes, ShapeBatch.from_shapes(bwant), "long contours")
    nchange = [int((np.diff(np.asarray(w.colors).astype(int)) != 0).sum()) for w in bwant]
    assert max(nchange) >= 20 and nchange[-3] == 0 and nchange[-2] == 2 and 7 in np.asarray(bwant[-2].colors) and 2 <= nchange[-1] <= 8, nchange   # many-corner, smooth, teardrop and few-spline contours were in it
    gb.close()
    # edgeColoringInkTrap on the same long contours (lanes = corners: spline lengths, minor corners; the 2 100-edge one keeps its tables in global memory)
    iwant = [oracle.shape_prepare(s, False, 2, 3.0, int(sd)) for s, sd in zip(big, bseeds)]
    gb = M.GlyphBatch.from_raw(ShapeBatch.from_shapes(big), False, 2, 3.0, seeds=np.array(bseeds, np.uint64))
    _same_batch(gb.shapes, ShapeBatch.from_shapes([FlatShape(f.contour_offsets, f.points, f.types, f.colors) for f in iwant]), "long contours, ink trap")
    gb.close()
    assert sum(int((np.asarray(a.colors) != np.asarray(b.colors)).any()) for a, b in zip(iwant, bwant)) >= 3    # minor corners were in it
    # the colouring kernel's SMALL LDS tier (no contour beyond 256 edges: what font batches are) on the contours of 64..256 edges of the same list, and the
    # large tier forced on the same batch -- both against the oracle
    mid = [k for k, s in enumerate(big) if int(np.diff(s.contour_offsets).max()) <= 256]
    assert len(mid) >= 5 and max(int(np.diff(big[k].contour_offsets).max()) for k in mid) >= 200
    for forced in (False, True):
        if forced:
            os.environ["MSDFHIP_PREP_LARGE_TIER"] = "1"
        M.load().msdfhip_reload_tuning()
        try:
            for coloring, wants in ((1, bwant), (2, iwant)):
                gb = M.GlyphBatch.from_raw(ShapeBatch.from_shapes([big[k] for k in mid]), False, coloring, 3.0, seeds=np.array([bseeds[k] for k in mid], np.uint64))
                _same_batch(gb.shapes, ShapeBatch.from_shapes([FlatShape(wants[k].contour_offsets, wants[k].points, wants[k].types, wants[k].colors) for k in mid]),
                            "contours up to 256 edges, colouring %d, %s tier" % (coloring, "large" if forced else "small"))
                gb.close()
        finally:
            os.environ.pop("MSDFHIP_PREP_LARGE_TIER", None)
            M.load().msdfhip_reload_tuning()
    # shapes built for the preparation passes (tests/test_shape_prep_oracle.py): contours of hundreds of edges, > 64 corners, CUSPS (normalize's
    # serial repair next to the lanes-=-edges pass), one- and two-edge contours (split in thirds) -- normalize x {keep, simple, ink trap}
    from test_shape_prep_oracle import prep_stress_shapes
    stress = prep_stress_shapes(41, 120)
    sseeds = np.arange(5000, 5000+len(stress), dtype=np.uint64)
    for normalize in (True, False):
        for coloring, angle in ((0, 3.0), (1, 3.0), (2, 3.0), (1, .05), (2, 1.0)):
            want = [oracle.shape_prepare(s, normalize, coloring, angle, int(sd)) for s, sd in zip(stress, sseeds)]
            gb = M.GlyphBatch.from_raw(ShapeBatch.from_shapes(stress), normalize, coloring, angle, seeds=sseeds)
            _same_batch(gb.shapes, ShapeBatch.from_shapes([FlatShape(f.contour_offsets, f.points, f.types, f.colors) for f in want]),
                        "stress shapes, normalize %s colouring %d angle %g" % (normalize, coloring, angle))
            gb.close()
    gb = M.GlyphBatch.from_raw(ShapeBatch.from_shapes(shapes[:50]), False, 1, 3.0, seed=77)    # one seed for all, colouring only
    want = [oracle.shape_prepare(s, False, 1, 3.0, 77) for s in shapes[:50]]
    _same_batch(gb.shapes, ShapeBatch.from_shapes([FlatShape(f.contour_offsets, f.points, f.types, f.colors) for f in want]), "colouring only")
    gb.close()


# ---- renderSDF / simulate8bit on the device

@pytest.mark.parametrize("n_out,mode", [(1, 1), (3, 1), (1, 3), (3, 3), (1, 4), (4, 4)])
def test_render_sdf_and_simulate_8bit(latin, oracle, n_out, mode):
    import torch
    batch, xf64, _ = latin
    sub = batch.select(range(40, 52))
    gb = M.GlyphBatch(sub)
    xfs = np.stack([autoframe(b, 40, 32, 4) for b in latin[2][40:52]])
    tiles = gb.generate(mode, 40, 32, xfs)
    src = tiles.cpu().numpy()
    for ow, oh in ((40, 32), (97, 64), (17, 23)):
        for lo, hi, thr in ((0, 0, .5), (-2, 2, .5), (-1, 3, .4)):
            got = M.render_sdf(tiles, ow, oh, n_out, (lo, hi), thr).cpu().numpy()
            for g in range(12):
                want = oracle.render_sdf(src[g], ow, oh, n_out, lo, hi, thr)
                assert (bits(got[g]) == bits(want)).all(), "renderSDF %d<-%d %dx%d glyph %d" % (n_out, src.shape[3], ow, oh, g)
    noisy = tiles+torch.randn_like(tiles)*.3
    want = oracle.simulate_8bit(noisy.cpu().numpy().reshape(-1, 40, noisy.shape[3]))
    assert (bits(M.simulate_8bit(noisy).cpu().numpy().reshape(want.shape)) == bits(want)).all()
    with pytest.raises(M.MsdfHipError):
        M.render_sdf(tiles, 8, 8, 2)
    gb.close()


def test_estimate_sdf_error_on_device(latin, oracle):
    """estimateSDFError of every tile of a batch without leaving the device, exact doubles vs the oracle:
    msdf / mtsdf / sdf, 1 and 3 scanlines per row, two fill rules, clean and perturbed fields, an inverse-Y glyph set."""
    import torch
    batch, xf64, _ = latin
    for inv in (False, True):
        sub = batch.select(range(30, 60))
        sub.inverse_y[:] = inv
        gb = M.GlyphBatch(sub)
        w, h = 40, 36
        xfs = np.stack([autoframe(b, w, h, 4) for b in latin[2][30:60]])
        for mode in (3, 4, 1):
            tiles = gb.generate(mode, w, h, xfs)
            noisy = (tiles+torch.randn_like(tiles)*.12).contiguous()
            for field in (tiles, noisy):
                src = field.cpu().numpy()
                for spr, rule in ((1, 0), (3, 1)):
                    got = gb.estimate_sdf_error(field, xfs, spr, rule).cpu().numpy()
                    want = np.array([oracle.estimate_sdf_error(sub.shape(g), src[g], xfs[g], spr, rule) for g in range(sub.n_glyphs)])
                    assert (got == want).all(), "mode %d spr %d rule %d inverse_y %s: %d of %d glyphs differ" % (mode, spr, rule, inv, int((got != want).sum()), len(got))
            if mode == 3 and not inv:
                print("mean fill error of clean msdf tiles: %.3g, of perturbed: %.3g" % (
                    float(gb.estimate_sdf_error(tiles, xfs).mean()), float(gb.estimate_sdf_error(noisy, xfs).mean())))
        gb.close()


def test_mixed_batch_is_bucketed_by_contour_count(latin, oracle):
    """A batch of mostly few-contour glyphs plus some many-contour ones: the kernels split it (LDS scratch for the former, global
    workspace for the latter); the result must not depend on that -- compared per glyph with the oracle, msdf and mtsdf, and through
    the single-shape entry point (which owns and frees its workspaces per call)."""
    batch, xf64, bounds = latin
    many = [synth.cjk_like_shape(8600+i) for i in range(5)]+[synth.random_shape(8700, n_contours=30, edges_per_contour=(3, 5), kinds=(1, 2))]
    shapes = [batch.shape(g) for g in range(0, 40)]
    for i, s in enumerate(many):
        shapes.insert(3+6*i, s)
    w = h = 32
    xfs = np.stack([autoframe(s.bounds(), w, h, 4) for s in shapes])
    gb = M.GlyphBatch(ShapeBatch.from_shapes(shapes))
    assert gb.max_contours >= 20
    for mode in (3, 4):
        got = gb.generate(mode, w, h, xfs).cpu().numpy()
        for g, s in enumerate(shapes):
            close(got[g], oracle.generate(s, mode, w, h, xfs[g]), "mixed batch glyph %d (%d contours) mode %d" % (g, s.n_contours, mode))
    gb.close()
    for s, xf in zip(many[:3], xfs[3::6]):
        for _ in range(3):
            close(gen(3, s, w, h, xf), oracle.generate(s, 3, w, h, xf), "single call, %d contours" % s.n_contours)


def test_candidate_segment_overflow_is_handled_per_glyph(latin, oracle):
    """The deferred distance checks of a glyph live in a fixed-size segment; a glyph that overflows it is redone by the full
    per-texel pipeline, the other glyphs of the batch are not affected. Overlapping strokes rendered WITHOUT overlap support produce
    hundreds of artifacts per tile; a heavily perturbed field does the same through the standalone correction."""
    batch, xf64, bounds = latin
    shapes = [batch.shape(g) for g in range(10, 26)]
    shapes.insert(5, synth.cjk_like_shape(8801))
    shapes.insert(11, synth.cjk_like_shape(8802))
    w = h = 48
    xfs = np.stack([autoframe(s.bounds(), w, h, 4) for s in shapes])
    gb = M.GlyphBatch(ShapeBatch.from_shapes(shapes))
    c = cfg(overlap=False, ec_mode=M.EC_EDGE_PRIORITY, ec_dist=M.ALWAYS_CHECK_DISTANCE)
    got = gb.generate(3, w, h, xfs, config=c).cpu().numpy()
    changed = 0
    for g, s in enumerate(shapes):
        want = oracle.generate(s, 3, w, h, xfs[g], overlap=False, ec_mode=2, ec_dist=2)
        close(got[g], want, "glyph %d (%d contours)" % (g, s.n_contours))
        changed += int((bits(want) != bits(oracle.generate(s, 3, w, h, xfs[g], overlap=False, ec_mode=0))).any(axis=2).sum()) if s.n_contours > 6 else 0
    assert changed > 200, "the stroke glyphs were meant to need many corrections (%d)" % changed
    gb.close()
    # the single-shape host call mirrors the overflow count next to its results instead of launching the overflow pass every time; when the
    # count is non-zero it runs the sequence again with that pass (round 3)
    for g in (5, 11):
        close(gen(3, shapes[g], w, h, xfs[g], c), oracle.generate(shapes[g], 3, w, h, xfs[g], overlap=False, ec_mode=2, ec_dist=2), "single call with an overflowing segment")
    rng = np.random.default_rng(9)
    s = batch.shape(33)
    xf = autoframe(bounds[33], 32, 32, 4)
    noisy = oracle.generate(s, 3, 32, 32, xf, ec_mode=0)+rng.normal(0, .2, (32, 32, 3)).astype(np.float32)
    for dist in (M.CHECK_DISTANCE_AT_EDGE, M.ALWAYS_CHECK_DISTANCE):
        want = oracle.error_correction(s, noisy, xf, ec_mode=2, ec_dist=dist)
        got = M.msdf_error_correction(noisy.copy(), s, M.SDFTransformation.from_xf(xf), cfg(ec_dist=dist))
        close(got, want, "standalone correction of a noisy field, distance check %d" % dist)


def test_stencil_rows_follow_the_reference_for_y_downward_bitmaps(ref):
    """ErrorCorrectionConfig::buffer keeps its rows in upward order whatever the bitmap's orientation (core/msdf-error-correction.cpp:19,
    core/MSDFErrorCorrection.cpp:122,192,415): compared with the compiled reference for all four shape / bitmap orientations,
    and left untouched when no correction pass runs."""
    z = load_npz("latin.npz")
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), z["inverse_y"], [str(n) for n in z["names"]])
    g = batch.names.index("U+0040")
    for inv in (False, True):
        s = batch.shape(g)
        s.inverse_y = inv
        for y_down in (False, True):
            want_st = np.zeros((64, 64), np.uint8)
            want = ref.generate(s, 3, 64, 64, z["xf64"][g], y_down=y_down, stencil=want_st)
            got_st = np.full((64, 64), 77, np.uint8)
            got = gen(3, s, 64, 64, z["xf64"][g], cfg(buffer=got_st), y_down=y_down)
            close(got, want, "tile, shape inverse_y=%s bitmap y_down=%s" % (inv, y_down))
            assert (got_st == want_st).all(), (inv, y_down, int((got_st != want_st).sum()))
    untouched = np.full((64, 64), 77, np.uint8)
    gen(3, batch.shape(g), 64, 64, z["xf64"][g], cfg(ec_mode=M.EC_DISABLED, buffer=untouched))
    assert (untouched == 77).all()


def test_stencil_rows_of_the_batched_paths_for_y_downward_bitmaps(ref):
    """The same contract on the BATCHED entry points: GlyphBatch.generate (device tensors) and HostBatch.generate_host take
    y_orientation and a stencil, and must lay the stencil out like the single-shape path and the compiled reference do."""
    import torch
    z = load_npz("latin.npz")
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), z["inverse_y"], [str(n) for n in z["names"]])
    pick = [batch.names.index(n) for n in ("U+0040", "U+0067", "U+0025")]
    sub, xfs = batch.select(pick), z["xf64"][pick]
    for y in (M.Y_UPWARD, M.Y_DOWNWARD):
        want_st = np.zeros((3, 64, 64), np.uint8)
        want = np.stack([ref.generate(sub.shape(g), 3, 64, 64, xfs[g], y_down=(y == M.Y_DOWNWARD), stencil=want_st[g]) for g in range(3)])
        gb = M.GlyphBatch(sub)
        st = torch.full((3, 64, 64), 77, dtype=torch.uint8, device="cuda")
        got = gb.generate(3, 64, 64, xfs, stencil=st, y_orientation=y).cpu().numpy()
        gb.close()
        close(got, want, "GlyphBatch.generate y=%d" % y)
        assert (st.cpu().numpy() == want_st).all(), "GlyphBatch.generate: stencil rows, y_orientation %d" % y
        hb = M.HostBatch(sub)
        hst = np.full((3, 64, 64), 77, np.uint8)
        hgot = hb.generate_host(3, 64, 64, xfs, stencil=hst, y_orientation=y)
        hb.close()
        close(hgot, want, "HostBatch.generate_host y=%d" % y)
        assert (hst == want_st).all(), "HostBatch.generate_host: stencil rows, y_orientation %d" % y


def test_shapeless_error_correction_entry_points(ref, oracle):
    """msdfFastDistanceErrorCorrection / msdfFastEdgeErrorCorrection (core/msdf-error-correction.cpp:50-59, 87-113; VERDICT r2 missing #5):
    findErrors(sdf) + apply without a shape, on pre-correction fields of real glyphs (msdf and mtsdf), several deviation ratios -- bit for
    bit against the COMPILED REFERENCE's own functions, and against the oracle's general pass on an empty shape (the identity the device
    path relies on: mode INDISCRIMINATE resp. EDGE_ONLY with DO_NOT_CHECK_DISTANCE)."""
    z = load_npz("latin.npz")
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), z["inverse_y"], [str(n) for n in z["names"]])
    empty = FlatShape.from_contours([])
    changed = 0
    for name in ("U+0040", "U+0067", "U+0026", "U+0057"):
        g = batch.names.index(name)
        xf = z["xf64"][g]
        t = M.SDFTransformation.from_xf(xf)
        for mode, n in ((3, 3), (4, 4)):
            pre = gen(mode, batch.shape(g), 64, 60, xf, cfg(ec_mode=M.EC_DISABLED))
            for ratio in (1.11111111111111111, 1.0, 2.5):
                for protect_all, fn in ((False, M.msdf_fast_distance_error_correction), (True, M.msdf_fast_edge_error_correction)):
                    want = ref.fast_error_correction(pre, xf, ratio, protect_all)
                    got = fn(pre.copy(), t, ratio)
                    assert (bits(got) == bits(want)).all(), (name, mode, ratio, protect_all, int((bits(got) != bits(want)).sum()))
                    same = oracle.error_correction(empty, pre, xf, overlap=False, ec_mode=M.EC_EDGE_ONLY if protect_all else M.EC_INDISCRIMINATE,
                                                   ec_dist=M.DO_NOT_CHECK_DISTANCE, min_dev=ratio)
                    assert (bits(same) == bits(want)).all()
                    changed += int((bits(got) != bits(pre)).any())
    assert changed >= 8                                                        # the passes do change these fields


def _ring(cx, cy, r, n, phase=0., flip=False, quad_every=3):
    """Closed polygon ring of n edges around (cx, cy); every quad_every-th edge a quadratic bulge. Colours cycle (corners everywhere)."""
    ang = phase+np.arange(n+1)*(2*np.pi/n)
    if flip:
        ang = ang[::-1]
    pts = [(cx+r*np.cos(a)*(1+.03*np.sin(7*a)), cy+r*np.sin(a)*(1+.03*np.cos(5*a))) for a in ang[:-1]]
    edges = []
    for k in range(n):
        p0, p1 = pts[k], pts[(k+1) % n]
        col = (6, 5, 3)[k % 3]
        if k % quad_every == 0:
            mx, my = .5*(p0[0]+p1[0]), .5*(p0[1]+p1[1])
            edges.append((col, p0, (mx+(mx-cx)*.01, my+(my-cy)*.01), p1))
        else:
            edges.append((col, p0, p1))
    return edges


def test_shapes_beyond_the_lds_lists_are_rendered_not_refused(oracle):
    """The reference's generators return void for ANY Shape (core/msdfgen.cpp:78-106); round 2 refused glyphs whose survivor lists exceed a
    CU's LDS (MSDFHIP_ERR_TOO_COMPLEX beyond ~40 000 edges). They now take the list-free kernel (k_distance_unculled), and beyond ~19 000
    contours the error-correction pass takes the full per-texel pipeline with its scratch in the global workspace. Against the oracle:
      * 61 440 edges in 24 nested / overlapping rings, msdf with the default correction and mtsdf (single-shape call);
      * 12 000 contours (36 000 edges), msdf default correction, inside a BATCH next to an ordinary glyph;
      * 21 000 contours (63 000 edges): also past the error-correction kernel's per-contour LDS scratch; plus a distance query
        (msdfhip_shape_distance) on the same shape, whose combiner scratch does not fit LDS either."""
    big = FlatShape.from_contours([_ring(.5+.01*(c % 5), .5-.008*(c % 7), .46-.017*c, 2560, phase=.1*c, flip=bool(c % 3 == 1)) for c in range(24)])
    assert big.n_edges == 61440
    xf = autoframe((0, 0, 1, 1), 40, 36, 4)
    close(gen(3, big, 40, 36, xf), oracle.generate(big, 3, 40, 36, xf), "61 440 edges msdf")
    close(gen(4, big, 40, 36, xf, cfg(ec_mode=M.EC_DISABLED)), oracle.generate(big, 4, 40, 36, xf, ec_mode=0), "61 440 edges mtsdf")

    def grid_of_triangles(n, cols):
        out = []
        for i in range(n):
            cx, cy, r = (i % cols+.5)/cols, (i//cols+.5)/cols, .45/cols
            pts = [(cx+r*np.cos(a+i), cy+r*np.sin(a+i)) for a in (0., 2.1, 4.2)]
            if i % 4 == 0:
                pts = pts[::-1]
            out.append([((6, 5, 3)[k], pts[k], pts[(k+1) % 3]) for k in range(3)])
        return FlatShape.from_contours(out)
    many = grid_of_triangles(12000, 110)
    small = FlatShape.from_contours([_ring(.5, .5, .4, 12), _ring(.5, .5, .2, 9, flip=True)])
    xf2 = autoframe((0, 0, 1, 1), 24, 24, 2)
    gb = M.GlyphBatch(ShapeBatch.from_shapes([small, many, small]))
    got = gb.generate(3, 24, 24, np.stack([xf2]*3)).cpu().numpy()
    gb.close()
    close(got[1], oracle.generate(many, 3, 24, 24, xf2), "12 000 contours in a batch")
    close(got[0], oracle.generate(small, 3, 24, 24, xf2), "ordinary glyph next to it")
    assert (bits(got[2]) == bits(got[0])).all()
    # Only the oversized glyphs take the list-free kernel; the others of the batch keep their classes (one contour / LDS / global workspace):
    # one glyph of each class next to TWO oversized ones, with the overlapping and with the simple combiner
    one = FlatShape.from_contours([_ring(.5, .5, .4, 17)])
    ten = FlatShape.from_contours([_ring(.5+.02*c, .5-.01*c, .45-.04*c, 14+c, phase=.3*c, flip=bool(c % 2)) for c in range(10)])
    mixed = [ten, many, one, big, small]
    gb = M.GlyphBatch(ShapeBatch.from_shapes(mixed))
    for overlap in (True, False):
        got = gb.generate(3, 24, 24, np.stack([xf2]*len(mixed)), cfg(overlap=overlap)).cpu().numpy()
        for k, s_ in enumerate(mixed):
            close(got[k], oracle.generate(s_, 3, 24, 24, xf2, overlap=overlap), "mixed batch with oversized glyphs, glyph %d, overlap %s" % (k, overlap))
    gb.close()
    huge = grid_of_triangles(21000, 145)
    xf3 = autoframe((0, 0, 1, 1), 16, 16, 2)
    close(gen(3, huge, 16, 16, xf3), oracle.generate(huge, 3, 16, 16, xf3), "21 000 contours msdf + correction")
    pts = np.stack([np.linspace(.1, .9, 64), np.linspace(.9, .2, 64)], 1)
    got_d = M.shape_distance(huge, 3, True, pts)
    want_d = oracle.shape_distance(huge, 3, True, pts)
    assert np.abs(got_d[:, :3]-want_d[:, :3]).max() <= 1e-9
