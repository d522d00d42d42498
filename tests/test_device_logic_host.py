"""The PRODUCT's device headers (msdfgen_amd/csrc/msdf_{device,prep,ec}.hpp) compiled for the host (tests/hostemu) and walked
texel by texel like the gfx950 kernels do per lane -- checked bit-for-bit against the oracle.  This validates the restructured
algorithm (pre-digested edge records in visit order, LDS-style combiner scratch, gather-form error correction) without a GPU.
The GPU build of the same headers is checked by tests/test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import load_npz, assert_bit_equal
from emu import Emu
from msdfgen_amd import synth
from msdfgen_amd.shape import FlatShape, autoframe


@pytest.fixture(scope="module")
def emu():
    return Emu()


def test_latin_all_modes(emu, oracle, latin):
    batch, xf64, _ = latin
    for g in range(0, batch.n_glyphs, 3):
        s = batch.shape(g)
        for mode in (1, 2, 3, 4):
            for ov in (True, False):
                assert_bit_equal(emu.generate(s, mode, 64, 64, xf64[g], overlap=ov), oracle.generate(s, mode, 64, 64, xf64[g], overlap=ov),
                                 "%s mode %d overlap %d" % (batch.names[g], mode, ov))


def test_golden_synthetic(emu):
    z = load_npz("synth.npz")
    for name in z["cases"]:
        name = str(name)
        w, h, mode, inv, ydown = (int(v) for v in z[name+"_meta"])
        s = FlatShape(z[name+"_co"], z[name+"_pts"], z[name+"_types"], z[name+"_colors"], bool(inv))
        assert_bit_equal(emu.generate(s, mode, w, h, z[name+"_xf"], y_down=bool(ydown)), z[name+"_out"], name)
        assert_bit_equal(emu.generate(s, mode, w, h, z[name+"_xf"], y_down=bool(ydown), overlap=False), z[name+"_out_simple"], name+" simple")


@pytest.mark.parametrize("ec_mode", [1, 2, 3])
@pytest.mark.parametrize("ec_dist", [0, 1, 2])
def test_error_correction_matrix(emu, oracle, ec_mode, ec_dist):
    for seed in range(3):
        s = synth.random_shape(2000+seed, n_contours=2+seed % 2, kinds=(1, 2, 3))
        s.inverse_y = bool(seed & 1)
        xf = autoframe(s.bounds(), 28, 26, 3)
        st_a, st_b = np.zeros((26, 28), np.uint8), np.zeros((26, 28), np.uint8)
        a = oracle.generate(s, 3+seed % 2, 28, 26, xf, ec_mode=ec_mode, ec_dist=ec_dist, min_dev=1.2, min_imp=1.05, stencil=st_a)
        b = emu.generate(s, 3+seed % 2, 28, 26, xf, ec_mode=ec_mode, ec_dist=ec_dist, min_dev=1.2, min_imp=1.05, stencil=st_b)
        assert_bit_equal(b, a, "ec %d/%d seed %d" % (ec_mode, ec_dist, seed))
        assert (st_a == st_b).all()


def test_stencil_stages_and_standalone_correction(emu, oracle, latin):
    batch, xf64, _ = latin
    z = load_npz("outputs.npz")
    for k, g in enumerate(z["subset"]):
        s = batch.shape(int(g))
        pre = z["msdf64_noec"][k]
        for stage in range(4):
            st = np.zeros((64, 64), np.uint8)
            emu.generate(s, 3, 64, 64, xf64[g], stage=stage+1, correct_only=pre, stencil=st)
            assert (st == z["stages64"][k, stage]).all(), (batch.names[g], stage)
        assert_bit_equal(emu.generate(s, 3, 64, 64, xf64[g], correct_only=pre), z["msdf64"][k], "standalone EC %s" % batch.names[g])


def test_windings_and_queries(emu, oracle, latin):
    batch, _, _ = latin
    z = load_npz("kats.npz")
    s = batch.shape(int(z["oneshot_glyph"]))
    assert (emu.windings(s) == oracle.windings(s)).all()
    for sel in (1, 2, 3, 4):
        for ov in (0, 1):
            assert_bit_equal(emu.shape_distance(s, sel, ov, z["oneshot_pts"]), z["oneshot_%d_%d" % (sel, ov)], "sel %d ov %d" % (sel, ov))


def winding_stress_shapes(seed):
    """Shapes for Contour::winding: contours of hundreds of edges, > 64 contours per shape with empty / one-edge / two-edge ones in between, and
    contours of (almost) no area, where the order of the shoelace sum decides the sign."""
    from test_shape_prep_oracle import prep_stress_shapes
    shapes = prep_stress_shapes(77, 60)
    rng = np.random.default_rng(seed)
    for k in range(30):                                                   # many contours of random lengths 0 .. 70, random orientation
        contours = []
        for c in range(int(rng.integers(60, 200))):
            n = int(rng.choice([0, 1, 2, 3, 4, 5, 9, 33, 63, 64, 65, 70]))
            cx, cy, r = rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(.01, .5)
            ang = np.sort(rng.uniform(0, 2*np.pi, max(n, 1)))[::(1 if rng.random() < .5 else -1)]
            pts = [(cx+r*np.cos(a), cy+r*np.sin(a)) for a in ang]
            if n == 1:
                contours.append([(7, pts[0], (cx+r, cy+r), (cx-r, cy+r), pts[0])])
            elif n == 2:
                contours.append([(7, pts[0], (cx, cy+r), pts[1]), (7, pts[1], (cx, cy-r), pts[0])])
            else:
                contours.append([(7, pts[i], pts[(i+1) % n]) if rng.random() < .7 else (7, pts[i], (cx, cy), pts[(i+1) % n]) for i in range(n)])
        shapes.append(FlatShape.from_contours(contours))
    for k in range(40):                                                   # out and back along (almost) the same path: the terms cancel to rounding
        n = int(rng.integers(3, 90))
        xs = np.sort(rng.uniform(0, 1, n))
        eps = float(rng.choice([0., 1e-17, 1e-13, -1e-15]))
        path = [(x, .3*x+eps*np.sin(40*x)) for x in xs]+[(x, .3*x) for x in xs[::-1][1:-1]]
        shapes.append(FlatShape.from_contours([[(7, path[i], path[(i+1) % len(path)]) for i in range(len(path))]]))
    return shapes


def test_wave_form_of_the_contour_windings_matches_oracle(emu, oracle, latin):
    """Round 4: the digest kernels compute Contour::winding with lanes = edges and a wave-uniform pass that adds the shoelace terms in edge order
    (msdf_prep.hpp: contourWindingsWave): all of a shape's contours in k_single_call, the long ones in k_prep_records (a lane per contour otherwise). The same source with a
    64-lane context on the host against the oracle: fonts, contours of hundreds of edges, > 64 contours per shape (several wavefronts, contours
    ending at and across the 64-edge rounds), empty / one-edge / two-edge contours in between, and contours of (almost) no area, where the
    ORDER of the sum decides the sign."""
    batch, _, _ = latin
    shapes = [batch.shape(g) for g in range(batch.n_glyphs)]+winding_stress_shapes(12)
    zero = both = 0
    for s in shapes:
        want = oracle.windings(s)
        for form in ("single", "batch"):
            assert (emu.windings(s, wave=form) == want).all(), (form, s.n_contours, s.n_edges)
        zero += int((want == 0).sum())
        both += int((want > 0).any() and (want < 0).any())
    assert zero >= 10 and both >= 30


def test_degenerate_inputs(emu, oracle):
    empty = FlatShape(np.zeros(1, np.int32), np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros(0, np.int32))
    xf = np.array([10., 10., .1, .1, -.2, .2])
    for mode in (1, 2, 3, 4):
        assert_bit_equal(emu.generate(empty, mode, 5, 4, xf), oracle.generate(empty, mode, 5, 4, xf), "empty shape mode %d" % mode)
    s = FlatShape.from_contours([
        [(7, (0, 0), (1, 1.5), (2, 0))],
        [],
        [(3, (0, 0), (1, 0)), (5, (1, 0), (.5, 1), (0, 0))],
        [(0, (.2, .2), (.8, .2)), (6, (.8, .2), (.8, .2)), (3, (.8, .2), (.5, .9)), (5, (.5, .9), (.2, .2))],
    ])
    xf = autoframe((0, 0, 2, 1.5), 20, 16, 2)
    assert (emu.windings(s) == oracle.windings(s)).all() and (emu.windings(s, wave="single") == oracle.windings(s)).all() and (emu.windings(s, wave="batch") == oracle.windings(s)).all()
    for mode in (1, 2, 3, 4):
        for ov in (True, False):
            assert_bit_equal(emu.generate(s, mode, 20, 16, xf, overlap=ov), oracle.generate(s, mode, 20, 16, xf, overlap=ov), "degenerate mode %d" % mode)


def test_quadratic_prefilter_never_skips_a_real_candidate(emu):
    """msdf_ec_fast.hpp skips solveQuadratic when the quadratic provably has no root in [0.005, 0.995]; fuzz that claim."""
    import ctypes as C
    rng = np.random.default_rng(11)
    n = 2_000_000
    abc = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    abc[:n//4] *= rng.choice([1e-3, 1e-6, 1, 10], (n//4, 1)).astype(np.float32)           # mixed magnitudes
    abc[n//4:n//2, 2] = abc[n//4:n//2, 0]*rng.uniform(.9, 1.1, n//4).astype(np.float32)   # near-degenerate: dD ~ dA
    abc[n//2:3*n//4, 1] = 2*abc[n//2:3*n//4, 0]                                           # b ~ 0 region, tiny a
    abc[-1000:] = 0
    emu.lib.emu_quadratic_prefilter_violations.restype = C.c_long
    skipped = C.c_long()
    bad = emu.lib.emu_quadratic_prefilter_violations(abc.ctypes.data_as(C.POINTER(C.c_float)), C.c_long(n), C.byref(skipped))
    assert bad == 0
    assert skipped.value > n//4  # the prefilter is actually doing something


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_fast_error_correction_all_configs(emu, oracle, seed):
    rng = np.random.default_rng(900+seed)
    s = synth.random_shape(7000+seed, n_contours=1+seed % 4, kinds=(1, 2, 3), holes=bool(seed & 1))
    s.inverse_y = bool(seed & 2)
    w, h = int(rng.integers(12, 44)), int(rng.integers(12, 44))
    xf = autoframe(s.bounds(), w, h, float(rng.uniform(1.5, 6)))
    for ec_mode in (1, 2, 3):
        for ec_dist in (0, 1, 2):
            for mode in (3, 4):
                a = oracle.generate(s, mode, w, h, xf, ec_mode=ec_mode, ec_dist=ec_dist, y_down=bool(seed & 4))
                b = emu.generate(s, mode, w, h, xf, ec_mode=ec_mode, ec_dist=ec_dist, y_down=bool(seed & 4))
                assert_bit_equal(b, a, "seed %d ec %d/%d mode %d" % (seed, ec_mode, ec_dist, mode))


def test_full_latin_default_pipeline_checksum(emu, latin):
    import hashlib
    batch, xf64, _ = latin
    z = load_npz("outputs.npz")
    for mode, key in ((3, "msdf"), (4, "mtsdf")):
        full = np.stack([emu.generate(batch.shape(g), mode, 64, 64, xf64[g]) for g in range(batch.n_glyphs)])
        assert hashlib.sha256(full.tobytes()).hexdigest() == str(z["sha_full_%s64" % key]), key


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_tile_culling_is_exact(emu, oracle, seed):
    """k_distance drops edges per 8x8 tile (msdf_cull.hpp); the result must stay bit-identical for every selector and combiner,
    also when tiles are small relative to the shape (large bitmaps, tight ranges) and when many contours overlap."""
    import ctypes as C
    rng = np.random.default_rng(4000+seed)
    if seed % 4 == 0:
        s = synth.cjk_like_shape(100+seed)
    elif seed % 4 == 1:
        s = synth.logo_shape(seed, n_blobs=5+seed % 5, edges=(4, 9))
    else:
        s = synth.random_shape(9000+seed, n_contours=1+seed % 6, kinds=(1, 2, 3), holes=bool(seed & 1), spread=.9)
    s.inverse_y = bool(seed & 2)
    w, h = int(rng.integers(20, 100)), int(rng.integers(20, 100))
    xf = autoframe(s.bounds(), w, h, float(rng.uniform(.5, 8)))
    xf[1] *= rng.uniform(.7, 1.4)
    kept, total = C.c_long(), C.c_long()
    emu.lib.emu_cull_stats(C.byref(kept), C.byref(total), 1)
    for mode in (1, 2, 3, 4):
        for ov in (True, False):
            a = oracle.generate(s, mode, w, h, xf, overlap=ov, ec_mode=0)
            b = emu.generate(s, mode, w, h, xf, overlap=ov, ec_mode=0)
            assert_bit_equal(b, a, "seed %d mode %d overlap %d" % (seed, mode, ov))
    emu.lib.emu_cull_stats(C.byref(kept), C.byref(total), 1)
    assert 0 < kept.value <= total.value


def test_reciprocal_fma_division_is_correctly_rounded(emu):
    """divExact (msdf_device.hpp) must equal IEEE division for every divisor that divSafe() admits."""
    import ctypes as C
    rng = np.random.default_rng(3)
    n = 3_000_000
    a = rng.standard_normal(n)*np.exp(rng.uniform(-30, 30, n))
    b = rng.standard_normal(n)*np.exp(rng.uniform(-30, 30, n))
    a[:1000] = 0
    a[1000:2000] = b[1000:2000]*rng.integers(1, 9, 1000)                 # exact quotients
    b[2000:3000] = np.ldexp(2-2.0**-52, rng.integers(-20, 20, 1000))     # all-ones significands must be rejected by divSafe
    a[3000:4000] = np.nextafter(b[3000:4000]*3, np.inf)                  # quotients next to ties
    emu.lib.emu_div_exact_violations.restype = C.c_long
    safe = C.c_long()
    bad = emu.lib.emu_div_exact_violations(a.ctypes.data_as(C.POINTER(C.c_double)), b.ctypes.data_as(C.POINTER(C.c_double)), C.c_long(n), C.byref(safe))
    assert bad == 0
    assert n-1500 < safe.value < n


def test_lean_transcendentals_are_within_one_ulp(emu):
    """The device build replaces OCML's generic cos/pow inside solveCubicNormed by range-specific kernels (msdf_device.hpp).
    Check them against 80-bit long double evaluations of exactly the reference's expressions: < 1 ulp, like glibc/OCML."""
    import ctypes as C
    rng = np.random.default_rng(17)
    n = 400_000
    t = np.concatenate([rng.uniform(0, np.pi, n-4), [0., np.pi, np.pi/2, 1e-9]])
    out = np.zeros((len(t), 3))
    emu.lib.emu_lean_cos_thirds(t.ctypes.data_as(C.POINTER(C.c_double)), C.c_long(len(t)), out.ctypes.data_as(C.POINTER(C.c_double)))
    third = np.float64(1)/np.float64(3)
    twopi = np.float64(2)*np.float64(np.pi)
    args = np.stack([third*t, third*(t+twopi), third*(t-twopi)], 1)            # the reference's fp64 argument arithmetic
    ref = np.cos(args.astype(np.longdouble))
    ulp = np.spacing(np.abs(ref.astype(np.float64)))
    err = np.abs(out.astype(np.longdouble)-ref)/ulp
    assert err.max() < 1.0, err.max(axis=0)
    glibc = np.abs(np.cos(args).astype(np.longdouble)-ref)/ulp
    print("cos thirds max ulp error: lean %.3f, libm %.3f" % (err.max(), glibc.max()))
    x = np.exp(rng.uniform(-40, 40, n))
    outp = np.zeros(n)
    emu.lib.emu_lean_pow_third(x.ctypes.data_as(C.POINTER(C.c_double)), C.c_long(n), outp.ctypes.data_as(C.POINTER(C.c_double)))
    refp = np.power(x.astype(np.longdouble), np.longdouble(third))              # x^(1/3.) with the fp64-rounded exponent
    errp = np.abs(outp.astype(np.longdouble)-refp)/np.spacing(refp.astype(np.float64))
    assert errp.max() < 1.0, errp.max()
    print("pow third max ulp error: lean %.3f" % errp.max())


def test_cooperative_psdf_query_equals_sequential(emu, oracle, latin):
    """k_ec_query evaluates a glyph's edges with lanes = edges and merges the single-edge selector states with a shuffle tree
    (EdgesCooperative). The merge must reproduce the sequential visit-order result bit for bit, ties included: points ON vertices
    and on the bisectors of corners make adjacent edges tie in |distance|."""
    batch, xf64, bounds = latin
    rng = np.random.default_rng(12)
    shapes = [batch.shape(g) for g in (1, 4, 32, 38, 51, 77)]+[synth.random_shape(900+i, n_contours=3, edges_per_contour=(30, 45), kinds=(1, 2, 3)) for i in range(3)]
    shapes.append(synth.cjk_like_shape(8200))
    for s in shapes:
        b = s.bounds()
        pts = np.column_stack([rng.uniform(b[0]-.1, b[2]+.1, 300), rng.uniform(b[1]-.1, b[3]+.1, 300)])
        verts = s.points[:, 0:2]
        pts = np.vstack([pts, verts[:40], verts[:40]+rng.normal(0, 1e-3, (min(40, len(verts)), 2))])
        for overlap in (True, False):
            want = oracle.shape_distance(s, 2, overlap, pts)[:, 0]
            got = emu.psdf_cooperative(s, overlap, pts)
            assert_bit_equal(got, want, "cooperative PSDF, overlap=%s, %d edges" % (overlap, s.n_edges))
            assert_bit_equal(emu.psdf_cooperative(s, overlap, pts, slotted=True), want, "slotted cooperative PSDF, overlap=%s" % overlap)


def test_combiner_restructuring_on_nested_and_overlapping_contours(emu, oracle):
    """The overlapping combiner as the device walks it (only the shape selector merged eagerly, inner / outer selectors by member count
    with a second walk for 2+ members; nearest-first survivor order with the visit-index tie-break; msdf_device.hpp) against the oracle
    on the inputs that exercise it: the font glyphs with the most contours and the hashes of the reference fixture, heavily overlapping
    blobs (texels inside several contours at once), the 40-contour logo, all selectors."""
    import hashlib
    from msdfgen_amd.shape import ShapeBatch
    z = load_npz("dejavu8192.npz")
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), np.zeros(len(z["names"]), bool), [str(n) for n in z["names"]])
    n_c = np.diff(batch.glyph_contour_offsets)
    for g in sorted(set(np.argsort(n_c)[-6:].tolist()+list(range(5, 8192, 683)))):
        got = emu.generate(batch.shape(g), 3, 48, 48, z["xf48"][g])
        assert (np.frombuffer(hashlib.sha256(got.tobytes()).digest(), np.uint8) == z["sha48"][g]).all(), (g, batch.names[g], int(n_c[g]))
    for seed in range(4):
        s = synth.random_shape(100+seed, n_contours=6, spread=.25)
        xf = autoframe(s.bounds(), 36, 36, 4)
        for mode in (1, 2, 3, 4):
            assert_bit_equal(emu.generate(s, mode, 36, 36, xf), oracle.generate(s, mode, 36, 36, xf), "overlapping blobs %d mode %d" % (seed, mode))
        assert_bit_equal(emu.generate(s, 3, 36, 36, xf, ec_dist=2), oracle.generate(s, 3, 36, 36, xf, ec_dist=2), "overlapping blobs %d, always check" % seed)
    logo = synth.logo_shape(5)
    xf = autoframe(logo.bounds(), 48, 48, 4)
    assert_bit_equal(emu.generate(logo, 3, 48, 48, xf), oracle.generate(logo, 3, 48, 48, xf), "logo 48x48")


def test_split_form_of_the_combiner_matches_oracle(emu, oracle):
    """The form of the overlapping combiner the kernels actually instantiate since round 6 -- two instances of the contour loop (msdf_device.hpp:
    shapeDistanceOverlapSplit; contour-combiners.cpp:77-134) -- walked on the host against the oracle: CJK-like shapes (13.7 contours), heavily overlapping
    blobs (second walks, texels inside several contours), the 40-contour logo, every selector."""
    form = 1
    emu.set_combiner_form(form)
    try:
        for i in range(6):
            s = synth.cjk_like_shape(20000+37*i)
            xf = autoframe(s.bounds(), 32, 32, 3)
            for mode in (3, 4):
                assert_bit_equal(emu.generate(s, mode, 32, 32, xf, ec_mode=0), oracle.generate(s, mode, 32, 32, xf, ec_mode=0), "cjk-like %d mode %d form %d" % (i, mode, form))
        for seed in range(6):
            s = synth.random_shape(300+seed, n_contours=3+seed, spread=.2)
            xf = autoframe(s.bounds(), 28, 28, 3)
            for mode in (1, 2, 3, 4):
                assert_bit_equal(emu.generate(s, mode, 28, 28, xf), oracle.generate(s, mode, 28, 28, xf), "blobs %d mode %d form %d" % (seed, mode, form))
        logo = synth.logo_shape(5)
        xf = autoframe(logo.bounds(), 40, 40, 4)
        for mode in (3, 4):
            assert_bit_equal(emu.generate(logo, mode, 40, 40, xf), oracle.generate(logo, mode, 40, 40, xf), "logo mode %d form %d" % (mode, form))
    finally:
        emu.set_combiner_form(0)
