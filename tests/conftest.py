import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    """The compiled reference; only available where oracle/_ref/libmsdfgen_ref.so exists (built from /root/reference)."""
    from oracle.pyoracle import Ref
    if not Ref.available() and not os.path.isdir("/root/reference/core"):
        pytest.skip("oracle/_ref/libmsdfgen_ref.so not present")
    return Ref()


@pytest.fixture(scope="session")
def latin():
    """(ShapeBatch of the 94 prepared DejaVuSans Basic-Latin glyphs, xf rows for 64x64 / 4 px range, bounds)."""
    from msdfgen_amd.shape import ShapeBatch
    z = np.load(os.path.join(GOLDEN, "latin.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), z["inverse_y"], [str(n) for n in z["names"]])
    return batch, z["xf64"], z["bounds"]


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype == np.float32:
        n = int((bits(a) != bits(b)).sum())
    else:
        n = int((a.view(np.uint64) != b.view(np.uint64)).sum()) if a.dtype == np.float64 else int((a != b).sum())
    assert n == 0, "%s: %d of %d values differ (max |d| %.3g)" % (what, n, a.size, float(np.nanmax(np.abs(a.astype(np.float64)-b.astype(np.float64)))))
