"""Randomised differential test: the plain-C oracle against the compiled reference (oracle/_ref), where that library exists.
Bit-exact on all outputs, stencils and per-function hooks."""
import numpy as np
import pytest

from conftest import assert_bit_equal
from msdfgen_amd import synth
from msdfgen_amd.shape import autoframe


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_all_modes(oracle, ref, seed):
    rng = np.random.default_rng(100+seed)
    s = synth.random_shape(1000+seed, n_contours=1+seed % 5, kinds=(1, 2, 3) if seed % 3 else (2,), holes=bool(seed & 1))
    s.inverse_y = bool(seed & 2)
    w, h = int(rng.integers(9, 40)), int(rng.integers(9, 40))
    xf = autoframe(s.bounds(), w, h, float(rng.uniform(1, 6)))
    xf[0] *= rng.uniform(.8, 1.25)
    xf[4] *= rng.uniform(.5, 1.5)
    ydown = bool(seed & 4)
    for mode in (1, 2, 3, 4):
        for ov in (True, False):
            a = ref.generate(s, mode, w, h, xf, overlap=ov, y_down=ydown)
            b = oracle.generate(s, mode, w, h, xf, overlap=ov, y_down=ydown)
            assert_bit_equal(b, a, "seed %d mode %d overlap %d" % (seed, mode, ov))


@pytest.mark.parametrize("ec_mode", [1, 2, 3])
@pytest.mark.parametrize("ec_dist", [0, 1, 2])
def test_error_correction_matrix(oracle, ref, ec_mode, ec_dist):
    for seed in range(4):
        s = synth.random_shape(2000+seed, n_contours=2+seed % 2, kinds=(1, 2, 3))
        xf = autoframe(s.bounds(), 28, 26, 3)
        st_a, st_b = np.zeros((26, 28), np.uint8), np.zeros((26, 28), np.uint8)
        a = ref.generate(s, 3+seed % 2, 28, 26, xf, ec_mode=ec_mode, ec_dist=ec_dist, min_dev=1.2, min_imp=1.05, stencil=st_a, y_down=bool(seed & 1))
        b = oracle.generate(s, 3+seed % 2, 28, 26, xf, ec_mode=ec_mode, ec_dist=ec_dist, min_dev=1.2, min_imp=1.05, stencil=st_b, y_down=bool(seed & 1))
        assert_bit_equal(b, a, "ec %d/%d seed %d" % (ec_mode, ec_dist, seed))
        assert (st_a == st_b).all()


def test_standalone_error_correction_and_stages(oracle, ref):
    for seed in range(4):
        s = synth.random_shape(3000+seed, n_contours=3, kinds=(1, 2, 3))
        xf = autoframe(s.bounds(), 32, 32, 4)
        pre = ref.generate(s, 3, 32, 32, xf, ec_mode=0)
        assert_bit_equal(oracle.error_correction(s, pre, xf), ref.error_correction(s, pre, xf), "standalone EC")
        assert (oracle.ec_stages(s, pre, xf) == ref.ec_stages(s, pre, xf)).all()


def test_degenerate_inputs(oracle, ref):
    from msdfgen_amd.shape import FlatShape
    empty = FlatShape(np.zeros(1, np.int32), np.zeros((0, 8)), np.zeros(0, np.int32), np.zeros(0, np.int32))
    xf = np.array([10., 10., .1, .1, -.2, .2])
    for mode in (1, 2, 3, 4):
        assert_bit_equal(oracle.generate(empty, mode, 5, 4, xf), ref.generate(empty, mode, 5, 4, xf), "empty shape mode %d" % mode)
    # contours with one and two edges, an empty contour in between, a BLACK edge, a zero-length line
    s = FlatShape.from_contours([
        [(7, (0, 0), (1, 1.5), (2, 0))],
        [],
        [(3, (0, 0), (1, 0)), (5, (1, 0), (.5, 1), (0, 0))],
        [(0, (.2, .2), (.8, .2)), (6, (.8, .2), (.8, .2)), (3, (.8, .2), (.5, .9)), (5, (.5, .9), (.2, .2))],
    ])
    xf = autoframe((0, 0, 2, 1.5), 20, 16, 2)
    for mode in (1, 2, 3, 4):
        for ov in (True, False):
            assert_bit_equal(oracle.generate(s, mode, 20, 16, xf, overlap=ov), ref.generate(s, mode, 20, 16, xf, overlap=ov), "degenerate mode %d" % mode)
    assert (oracle.windings(s) == ref.flatten(ref.shape_from_flat(s)).windings).all()
