"""distanceSignCorrection (core/rasterization.cpp:19-88) + Shape::scanline + EdgeSegment::scanlineIntersections
(core/edge-segments.cpp:279-403).  The plain-C oracle against the compiled reference, bit-exact."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from emu import Emu
from msdfgen_amd import synth
from msdfgen_amd.shape import autoframe


@pytest.fixture(scope="module")
def emu():
    return Emu()


def test_scanline_intersections_kat(oracle, ref):
    rng = np.random.default_rng(5)
    for t in (1, 2, 3):
        for i in range(400):
            pts = np.zeros(8)
            pts[:2*(t+1)] = np.round(rng.uniform(-1, 1, 2*(t+1)), 2) if i % 3 == 0 else rng.uniform(-1, 1, 2*(t+1))
            ys = [rng.uniform(-1.2, 1.2), pts[1], pts[2*t+1], pts[3]]          # incl. rows through control points (the == branches)
            for y in ys:
                na, xa, da = ref.scanline_intersections(t, pts, y)
                nb, xb, db = oracle.scanline_intersections(t, pts, y)
                assert na == nb and (da == db).all(), (t, i, y)
                assert_bit_equal(xb, xa, "scanlineIntersections type %d" % t)


@pytest.mark.parametrize("seed", range(10))
def test_sign_correction_vs_reference(oracle, ref, seed):
    rng = np.random.default_rng(300+seed)
    s = synth.random_shape(6000+seed, n_contours=1+seed % 4, kinds=(1, 2, 3), holes=bool(seed & 1))
    s.inverse_y = bool(seed & 2)
    w, h = int(rng.integers(12, 40)), int(rng.integers(12, 40))
    xf = autoframe(s.bounds(), w, h, 3)
    for mode, n in ((1, 1), (3, 3), (4, 4)):
        field = ref.generate(s, mode, w, h, xf, ec_mode=0, y_down=bool(seed & 4))
        field = 1-field if seed % 3 == 0 else field                         # a wholly inverted field exercises every texel
        if n >= 3:
            field[rng.integers(0, h, 6), rng.integers(0, w, 6)] = .5           # exact zero-value medians: the ambiguity pass (:69-88)
        for rule in range(4):
            a = ref.sign_correction(s, field, xf, .5, rule, y_down=bool(seed & 4))
            b = oracle.sign_correction(s, field, xf, .5, rule, y_down=bool(seed & 4))
            assert_bit_equal(b, a, "seed %d mode %d rule %d" % (seed, mode, rule))
        a = ref.sign_correction(s, field, xf, .25, 0)
        assert_bit_equal(oracle.sign_correction(s, field, xf, .25, 0), a, "zero value .25")


# ---- the product's device code (msdf_scanline.hpp + the k_sign_correction walk) compiled for the host, against the oracle

def test_device_scanline_intersections_host(oracle, emu):
    rng = np.random.default_rng(6)
    for t in (1, 2, 3):
        for i in range(600):
            pts = np.zeros(8)
            pts[:2*(t+1)] = np.round(rng.uniform(-1, 1, 2*(t+1)), 2) if i % 3 == 0 else rng.uniform(-1, 1, 2*(t+1))
            if i % 7 == 0:
                pts[2*t:2*t+2] = pts[0:2]+[rng.uniform(-1, 1), 0]             # end point level with the start
            for y in (rng.uniform(-1.2, 1.2), pts[1], pts[2*t+1], pts[3]):
                na, xa, da = oracle.scanline_intersections(t, pts, y)
                nb, xb, db = emu.scanline_intersections(t, pts, y)
                assert na == nb and (da == db).all(), (t, i, y)
                assert_bit_equal(xb, xa, "device scanlineIntersections type %d" % t)


@pytest.mark.parametrize("seed", range(8))
def test_device_sign_correction_host(oracle, emu, seed):
    rng = np.random.default_rng(900+seed)
    s = synth.random_shape(6100+seed, n_contours=1+seed % 4, kinds=(1, 2, 3), holes=bool(seed & 1))
    s.inverse_y = bool(seed & 2)
    w, h = int(rng.integers(9, 40)), int(rng.integers(9, 40))
    xf = autoframe(s.bounds(), w, h, 3)
    for mode in (1, 3, 4):
        field = oracle.generate(s, mode, w, h, xf, ec_mode=0, y_down=bool(seed & 4))
        field = 1-field if seed % 3 == 0 else field
        if mode >= 3:
            field[rng.integers(0, h, 8), rng.integers(0, w, 8)] = .5
            field[0, 0] = field[h-1, w-1] = .5                                  # ambiguous texels on the border: fewer neighbours
        for rule in range(4):
            a = oracle.sign_correction(s, field, xf, .5, rule, y_down=bool(seed & 4))
            b = emu.sign_correction(s, field, xf, .5, rule, y_down=bool(seed & 4))
            assert_bit_equal(b, a, "seed %d mode %d rule %d" % (seed, mode, rule))


@pytest.mark.parametrize("seed", range(8))
def test_rasterize_oracle_reference_and_device_host(oracle, ref, emu, seed):
    """rasterize (core/rasterization.cpp:8-16): oracle == reference == the product's scanline code on the host."""
    s = synth.random_shape(6200+seed, n_contours=1+seed % 4, kinds=(1, 2, 3), holes=bool(seed & 1))
    s.inverse_y = bool(seed & 2)
    w, h = 21+seed, 37-seed
    xf = autoframe(s.bounds(), w, h, 2)
    for rule in range(4):
        a = ref.rasterize(s, w, h, xf, rule, y_down=bool(seed & 4))
        assert (oracle.rasterize(s, w, h, xf, rule, y_down=bool(seed & 4)) == a).all()
        assert (emu.rasterize(s, w, h, xf, rule, y_down=bool(seed & 4)) == a).all()
    assert 0 < ref.rasterize(s, w, h, xf, 1).mean() < 1


def test_pixel_float_to_byte_oracle_vs_reference(oracle, ref):
    """pixelFloatToByte (core/pixel-conversion.hpp:8-10): every rounding boundary k/255 +- a few ulps, out-of-range, inf, NaN."""
    k = np.arange(0, 256, dtype=np.float64)
    edges = np.concatenate([(k+d)/255. for d in (0, .5, -.5, 1/3.)]).astype(np.float32)
    vals = np.concatenate([edges, np.nextafter(edges, np.float32(2)), np.nextafter(edges, np.float32(-2)),
                           np.array([-1, -0., 0, 1, 1.0000001, 2, 1e30, -1e30, np.inf, -np.inf, np.nan], np.float32),
                           np.random.default_rng(3).uniform(-.2, 1.2, 20000).astype(np.float32)])
    a, b = ref.pixel_float_to_byte(vals), oracle.pixel_float_to_byte(vals)
    assert (a == b).all() and a.min() == 0 and a.max() == 255


@pytest.mark.parametrize("n_out,n_sdf", [(1, 1), (3, 1), (1, 3), (3, 3), (1, 4), (4, 4)])
def test_render_sdf_oracle_vs_reference(oracle, ref, latin, n_out, n_sdf):
    """renderSDF (core/render-sdf.cpp): every overload, up- and down-scaling, thresholded (range 0) and ranged, plus simulate8bit."""
    batch, xf64, _ = latin
    rng = np.random.default_rng(n_out*10+n_sdf)
    mode = {1: 1, 3: 3, 4: 4}[n_sdf]
    for g in (3, 33, 51):
        sdf = ref.generate(batch.shape(g), mode, 40, 32, autoframe(latin[2][g], 40, 32, 4))
        for ow, oh in ((40, 32), (97, 64), (17, 23)):
            for lo, hi, thr in ((0, 0, .5), (-2, 2, .5), (-1, 3, .4), (0, 0, .6)):
                a = ref.render_sdf(sdf, ow, oh, n_out, lo, hi, thr)
                b = oracle.render_sdf(sdf, ow, oh, n_out, lo, hi, thr)
                assert_bit_equal(b, a, "renderSDF %d<-%d %dx%d range (%g,%g)" % (n_out, n_sdf, ow, oh, lo, hi))
        noisy = sdf+rng.normal(0, .3, sdf.shape).astype(np.float32)
        assert_bit_equal(oracle.simulate_8bit(noisy), ref.simulate_8bit(noisy), "simulate8bit")


@pytest.mark.parametrize("seed", range(8))
def test_estimate_sdf_error_oracle_vs_reference(oracle, ref, seed):
    """estimateSDFError (core/sdf-error-estimation.cpp): sdf / msdf / mtsdf fields, several scanlines per row, fill rules, inverse-Y
    shapes, noisy fields (many spurious crossings: the consistency check of scanlineMSDF), flat fields (no crossing at all)."""
    rng = np.random.default_rng(400+seed)
    s = synth.random_shape(6400+seed, n_contours=1+seed % 4, kinds=(1, 2, 3), holes=bool(seed & 1))
    s.inverse_y = bool(seed & 2)
    w, h = int(rng.integers(12, 48)), int(rng.integers(12, 48))
    xf = autoframe(s.bounds(), w, h, 3)
    for mode in (1, 3, 4):
        clean = ref.generate(s, mode, w, h, xf, y_down=bool(seed & 2))
        fields = [clean, clean+rng.normal(0, .15, clean.shape).astype(np.float32), np.full_like(clean, .25), 1-clean]
        for f in fields:
            for spr, rule in ((1, 0), (3, 1), (2, 2)):
                a = ref.estimate_sdf_error(s, f, xf, spr, rule)
                b, lines = oracle.estimate_sdf_error(s, f, xf, spr, rule, per_line=True)
                assert a == b, "seed %d mode %d spr %d rule %d: %r vs %r" % (seed, mode, spr, rule, a, b)
                assert len(lines) == (h-1)*spr
    assert oracle.estimate_sdf_error(s, clean[:1], xf) == 0 == ref.estimate_sdf_error(s, clean[:1], xf)


@pytest.mark.parametrize("seed", range(6))
def test_device_estimate_sdf_error_host(oracle, emu, seed):
    """The product's per-scanline error code (msdf_scanline.hpp: sdfErrorOfLine) on the host against the oracle, exact doubles."""
    rng = np.random.default_rng(500+seed)
    s = synth.random_shape(6500+seed, n_contours=1+seed % 4, kinds=(1, 2, 3), holes=bool(seed & 1))
    s.inverse_y = bool(seed & 2)
    w, h = int(rng.integers(12, 48)), int(rng.integers(12, 48))
    xf = autoframe(s.bounds(), w, h, 3)
    for mode in (1, 3, 4):
        clean = oracle.generate(s, mode, w, h, xf, y_down=bool(seed & 2))
        for f in (clean, clean+rng.normal(0, .15, clean.shape).astype(np.float32), np.full_like(clean, .75), 1-clean):
            for spr, rule in ((1, 0), (3, 1), (2, 3)):
                assert emu.estimate_sdf_error(s, f, xf, spr, rule) == oracle.estimate_sdf_error(s, f, xf, spr, rule), (seed, mode, spr, rule)
