"""Full-size parity of BASELINE configs 4 and 5 against fixtures rendered ONCE by the unmodified, compiled reference
(tools/make_golden_full.py -> tests/golden/dejavu8192.npz, logo1024.npz).  EVERY tile / texel is compared:

  * config 4: 8 192 distinct glyphs (DejaVuSans + DejaVuSans-Bold, glyph order), msdf 48x48, library-default config -- per-glyph sha256
    of the reference's tile; the same glyphs re-framed at 64x64 (the bench workload) likewise.
  * config 5: the 926-edge / 40-contour cubic logo, msdf 1024x1024 incl. the error-correction stages
    (core/MSDFErrorCorrection.cpp:412-457) -- sha256 of all texels and of the final stencil.

The contract is |delta| <= 1e-5 per texel (BASELINE.json); in practice the device is bit-identical, and a hash only proves the
latter.  A tile whose hash differs is therefore re-rendered by the plain-C oracle (bit-equal to the reference,
tests/test_oracle_vs_reference.py) and held to the 1e-5 bound, and the number of such tiles is bounded -- see `resolve_mismatches`.
"""
import hashlib

import numpy as np
import pytest

import msdfgen_amd as M
from conftest import load_npz, bits
from msdfgen_amd.shape import FlatShape, ShapeBatch

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _device():
    M.init(0)


@pytest.fixture(scope="module")
def dejavu():
    z = load_npz("dejavu8192.npz")
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
    return batch, z


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def resolve_mismatches(batch, tiles, want_sha, xfs, size, oracle, what, allowed=4):
    """Tiles whose sha256 differs from the reference's: each must be within 1e-5 of the oracle's render; at most `allowed` of them
    (a last-ulp difference of acos/cos/pow surviving the fp32 rounding is a ~2^-29 event per value)."""
    bad = [g for g in range(batch.n_glyphs) if not (sha(tiles[g]) == want_sha[g]).all()]
    for g in bad[:allowed]:
        want = oracle.generate(batch.shape(g), 3, size, size, xfs[g])
        assert (sha(want) == want_sha[g]).all(), "%s: the oracle itself disagrees with the reference fixture on glyph %d" % (what, g)
        d = np.abs(tiles[g].astype(np.float64)-want)
        assert float(d.max()) <= TOL, "%s: glyph %d (%s) max |delta| %.3g" % (what, g, batch.names[g], float(d.max()))
    assert len(bad) <= allowed, "%s: %d of %d tiles differ from the reference bitwise (first: %s)" % (what, len(bad), batch.n_glyphs, bad[:8])
    return len(bad)


@pytest.mark.parametrize("size", [48, 64])
def test_config4_dejavu_8192_every_tile(dejavu, oracle, size):
    batch, z = dejavu
    xfs = z["xf%d" % size]
    gb = M.GlyphBatch(batch)
    tiles = gb.generate(M.MODE_MSDF, size, size, xfs).cpu().numpy()
    gb.close()
    assert tiles.shape == (8192, size, size, 3)
    nbad = resolve_mismatches(batch, tiles, z["sha%d" % size], xfs, size, oracle, "config 4 (%dx%d)" % (size, size))
    if nbad == 0:
        assert (sha(tiles) == z["sha_all%d" % size]).all()
    if size == 48:
        assert (bits(tiles[::1024]) == bits(z["sample48"])).all() or nbad > 0
    print("config 4 %dx%d: %d of 8192 tiles differ bitwise from the compiled reference" % (size, size, nbad))


def test_batch_of_more_than_8192_glyphs_every_tile(dejavu):
    """One device batch of 10 000 glyphs (the 8 192 distinct ones in reverse order + the first 1 808 again) at 48x48: k_ec_scan's work list takes a second round of its
    1 024 x 8 positions, the distance checks' statically dealt first tickets and eight ticket counters (round 6) cover a list longer than one round -- every tile against
    the compiled reference's hash of its glyph."""
    batch, z = dejavu
    idx = list(range(8191, -1, -1))+list(range(1808))
    sub, xfs = batch.select(idx), z["xf48"][idx]
    gb = M.GlyphBatch(sub)
    tiles = gb.generate(M.MODE_MSDF, 48, 48, xfs).cpu().numpy()
    gb.close()
    bad = [k for k, g in enumerate(idx) if not (sha(tiles[k]) == z["sha48"][g]).all()]
    assert not bad, "%d of %d tiles differ from the reference bitwise (first: %s)" % (len(bad), len(idx), bad[:8])


@pytest.mark.parametrize("tag,mode,size", [("mtsdf64", 4, 64), ("sdf48", 1, 48), ("psdf48", 2, 48)])
def test_dejavu_8192_every_tile_other_field_types(dejavu, oracle, tag, mode, size):
    """The other three generators on the 8 192 distinct glyphs (config 3's mtsdf on the tail of a real font; sdf; psdf): every tile against
    the compiled reference's sha256 (tools/make_golden_full.py: dejavu_modes). core/msdfgen.cpp:78-106, core/edge-selectors.cpp:229-260."""
    batch, z = dejavu
    zm = load_npz("dejavu8192_modes.npz")
    xfs = z["xf%d" % size]
    gb = M.GlyphBatch(batch)
    tiles = gb.generate(mode, size, size, xfs).cpu().numpy()
    gb.close()
    want = zm["sha_"+tag]
    bad = [g for g in range(8192) if not (sha(tiles[g])[:16] == want[g]).all()]
    for g in bad[:4]:
        ref = oracle.generate(batch.shape(g), mode, size, size, xfs[g])
        assert (sha(ref)[:16] == want[g]).all(), "%s: the oracle itself disagrees with the reference fixture on glyph %d" % (tag, g)
        assert float(np.abs(tiles[g].astype(np.float64)-ref).max()) <= TOL, "%s glyph %d (%s)" % (tag, g, batch.names[g])
    assert len(bad) <= 4, "%s: %d of 8192 tiles differ from the reference bitwise (first: %s)" % (tag, len(bad), bad[:8])
    if not bad:
        assert (sha(tiles) == zm["sha_all_"+tag]).all()
        assert (bits(tiles[::2048]) == bits(zm["sample_"+tag])).all()
    print("%s on 8192 distinct glyphs: %d tiles differ bitwise from the compiled reference" % (tag, len(bad)))


def test_config4_cjk_like_512_distinct_shapes_every_tile_both_mappings(oracle):
    """Config 4's CJK-like stand-in, EVERY tile: 8 192 glyphs = 16 copies of 512 distinct shapes (13.7 contours / 82.6 edges each), msdf
    48x48 -- the workload that takes the PERSISTENT global-scratch form of k_distance (per-XCD work queues, one workspace slice per
    resident wavefront). Each of the 8 192 tiles must hash to the compiled reference's tile of its shape, in the persistent and in the
    direct mapping. core/contour-combiners.cpp:77-134."""
    import os
    from msdfgen_amd import synth
    from msdfgen_amd.shape import autoframe
    zc = load_npz("cjk512.npz")
    base = [synth.cjk_like_shape(20000+i) for i in range(512)]
    xfs512 = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])
    assert (sha(ShapeBatch.from_shapes(base).points) == zc["sha_points"]).all() and (sha(xfs512) == zc["sha_xf"]).all(), \
        "the seeded generator no longer reproduces the shapes the fixture was rendered from"
    batch = ShapeBatch.from_shapes([base[i % 512] for i in range(8192)])
    xfs = xfs512[np.arange(8192) % 512]
    want = zc["sha48"]
    for knob, what in ((None, "persistent"), ("0", "direct")):
        if knob is not None:
            os.environ["MSDFHIP_PERSISTENT_ROUNDS"] = knob
            M.load().msdfhip_reload_tuning()
        try:
            gb = M.GlyphBatch(batch)
            tiles = gb.generate(M.MODE_MSDF, 48, 48, xfs).cpu().numpy()
            gb.close()
        finally:
            os.environ.pop("MSDFHIP_PERSISTENT_ROUNDS", None)
            M.load().msdfhip_reload_tuning()
        bad = [g for g in range(8192) if not (sha(tiles[g]) == want[g % 512]).all()]
        for g in bad[:4]:
            ref = oracle.generate(base[g % 512], 3, 48, 48, xfs[g])
            assert (sha(ref) == want[g % 512]).all()
            assert float(np.abs(tiles[g].astype(np.float64)-ref).max()) <= TOL, "%s mapping, glyph %d" % (what, g)
        assert len(bad) <= 4, "%s mapping: %d of 8192 tiles differ from the reference bitwise (first: %s)" % (what, len(bad), bad[:8])
        print("config 4 (CJK-like), %s mapping: %d of 8192 tiles differ bitwise from the compiled reference" % (what, len(bad)))


def test_config4_single_shape_calls_match_the_batch(dejavu):
    """The drop-in entry point (one generateMSDF call per glyph, host pointers) on the glyphs with the most contours / edges."""
    batch, z = dejavu
    gco, co = batch.glyph_contour_offsets, batch.contour_offsets
    n_c = np.diff(gco)
    n_e = co[gco[1:]]-co[gco[:-1]]
    pick = sorted(set(np.argsort(n_c)[-6:].tolist()+np.argsort(n_e)[-6:].tolist()+[0, 4095, 8191]))
    for g in pick:
        out = np.zeros((48, 48, 3), np.float32)
        M.generate_msdf(out, batch.shape(g), M.SDFTransformation.from_xf(z["xf48"][g]))
        assert (sha(out) == z["sha48"][g]).all(), "glyph %d (%s, %d contours, %d edges)" % (g, batch.names[g], n_c[g], n_e[g])


def test_config5_logo_1024_every_texel_and_stencil(oracle):
    """Pin (tools/make_golden_full.py:logo): the reference's exact per-texel evaluation (ShapeDistanceFinder::oneShotDistance, mapped as
    DistanceMapping does) at all 1 048 576 texel centres, then the reference's msdfErrorCorrection on that field. The reference's
    generateMSDF itself walks the texels with a per-edge cache (edge-selectors.cpp:64-79) whose pruning is not exact on this shape: the
    fixture lists the texel(s) where it departs from its own exact evaluation (1 texel, by 3e-5); there the device must equal the exact
    value, everywhere else all three agree."""
    z = load_npz("logo1024.npz")
    s = FlatShape(z["contour_offsets"], z["points"], z["types"].astype(np.int32), z["colors"].astype(np.int32))
    xf = z["xf"]
    t = M.SDFTransformation.from_xf(xf)
    pre = M.generate_msdf(np.zeros((1024, 1024, 3), np.float32), s, t, M.MSDFGeneratorConfig(True, M.ErrorCorrectionConfig(M.EC_DISABLED)))
    stencil = np.zeros((1024, 1024), np.uint8)
    out = M.generate_msdf(np.zeros((1024, 1024, 3), np.float32), s, t, M.MSDFGeneratorConfig(True, M.ErrorCorrectionConfig(buffer=stencil)))
    y0, x0 = (int(v) for v in z["crop_origin"])
    pre_ok = (sha(pre) == z["sha_pre"]).all()
    if not pre_ok:                                                            # hold the differing rows to the 1e-5 contract against the oracle
        rows = [y for y in range(1024) if not (sha(pre[y]) == z["sha_pre_rows"][y]).all()]
        assert len(rows) <= 8, "pre-correction field: %d rows differ from the reference" % len(rows)
        for y in rows:
            pts = np.stack([(np.arange(1024)+.5)/xf[0]-xf[2], np.full(1024, (y+.5)/xf[1]-xf[3])], 1)
            want = (np.float64(1)/(xf[5]-xf[4])*(oracle.shape_distance(s, 3, True, pts)[:, :3]+(-xf[4]))).astype(np.float32)
            assert (sha(want) == z["sha_pre_rows"][y]).all(), "the oracle disagrees with the reference fixture in row %d" % y
            assert float(np.abs(pre[y].astype(np.float64)-want).max()) <= TOL, "row %d" % y
    for (yy, xx), exact, cached in zip(z["cached_diff_yx"], z["exact_values_there"], z["cached_diff_values"]):
        assert (bits(pre[yy, xx]) == bits(exact)).all() or not pre_ok
        assert float(np.abs(exact.astype(np.float64)-cached).max()) < 1e-4     # what the reference's cached walk writes there instead
    ok = pre_ok and (sha(out) == z["sha_out"]).all() and (sha(stencil) == z["sha_stencil"]).all()
    if not ok:
        want_st = np.zeros((1024, 1024), np.uint8)
        want = oracle.error_correction(s, pre, xf, stencil=want_st)           # the oracle's correction pass == the reference's (bit-exact, CPU tests)
        nbits = int((bits(out) != bits(want)).sum())
        assert nbits == 0 and (stencil == want_st).all(), "error correction of the 1024x1024 field: %d values / %d stencil bytes differ" % (
            nbits, int((stencil != want_st).sum()))
    else:
        assert (bits(out[y0:y0+64, x0:x0+64]) == bits(z["crop_out"])).all()
        assert (stencil[y0:y0+64, x0:x0+64] == z["crop_stencil"]).all()
        assert int((stencil & 1).sum()) == int(z["n_error"])
    # The literal generateMSDF (cached serpentine walk + its own error correction): the device output must turn into it -- bitwise, sha256
    # over all texels -- by replacing exactly the texels the fixture lists (1 of 1 048 576; VERDICT r2 "parity footnote").
    if ok:
        as_cached = pre.copy()
        for (yy, xx), v in zip(z["cached_diff_yx"], z["cached_diff_values"]):
            as_cached[yy, xx] = v
        assert (sha(as_cached) == z["sha_pre_cached_walk"]).all()
        as_cached = out.copy()
        for (yy, xx), v in zip(z["out_cached_diff_yx"], z["out_cached_diff_values"]):
            as_cached[yy, xx] = v
        assert (sha(as_cached) == z["sha_out_cached_walk"]).all()
        assert len(z["out_cached_diff_yx"]) == int(z["n_out_differs_from_cached_walk"]) == 1
    print("config 5: 1024x1024 msdf + error correction %s the compiled reference's exact evaluation (%d ERROR texels; the reference's cached walk "
          "differs from it at %d texel(s))" % ("bit-identical to" if ok else "within 1e-5 of", int((stencil & 1).sum()), len(z["cached_diff_yx"])))
