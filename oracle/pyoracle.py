"""TEST INFRASTRUCTURE ONLY -- ctypes bindings for the two CPU checkers.

    Oracle   oracle/libmsdf_oracle.so        our plain-C restatement (oracle/msdf_oracle.c)
    Ref      oracle/_ref/libmsdfgen_ref.so   the unmodified reference compiled from /root/reference (oracle/ref_driver.cpp)

Only tests/, tools/make_golden.py, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (msdfgen_amd/) never does.

A shape is passed as plain numpy arrays (the flat CSR edge buffer described in oracle/msdf_oracle.h):
    contour_offsets int32[C+1], points float64[E, 8], types int32[E], colors int32[E], inverse_y bool
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libmsdf_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libmsdfgen_ref.so")
REF_OMP_SO = os.path.join(HERE, "_ref", "libmsdfgen_ref_omp.so")   # same sources, MSDFGEN_USE_OPENMP (CPU baseline of one large bitmap)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)
_bp = C.POINTER(C.c_uint8)

DEFAULT_RATIO = 1.11111111111111111  # ErrorCorrectionConfig::defaultMinDeviationRatio, MSDFErrorCorrection.cpp:22-23
EC_DISABLED, EC_INDISCRIMINATE, EC_EDGE_PRIORITY, EC_EDGE_ONLY = 0, 1, 2, 3
DC_DO_NOT_CHECK, DC_CHECK_AT_EDGE, DC_ALWAYS_CHECK = 0, 1, 2
CHANNELS = {1: 1, 2: 1, 3: 3, 4: 4}


def build(target="all"):
    subprocess.run(["make", "-s", "-C", HERE, target], check=True)


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a, t):
    return a.ctypes.data_as(t)


class _OrcShape(C.Structure):
    _fields_ = [("n_contours", C.c_int32), ("contour_offsets", _ip), ("points", _dp), ("types", _ip), ("colors", _ip), ("inverse_y", C.c_int32)]


class FlatArrays:
    """Keeps the contiguous arrays of one shape alive while C code points into them."""

    def __init__(self, contour_offsets, points, types, colors, inverse_y=False):
        self.contour_offsets = _arr(contour_offsets, np.int32)
        self.points = _arr(points, np.float64).reshape(-1, 8)
        self.types = _arr(types, np.int32)
        self.colors = _arr(colors, np.int32)
        self.inverse_y = bool(inverse_y)
        self.n_contours = len(self.contour_offsets)-1
        self.n_edges = int(self.contour_offsets[-1]) if self.n_contours >= 0 and len(self.contour_offsets) else 0

    def orc(self):
        return _OrcShape(self.n_contours, _p(self.contour_offsets, _ip), _p(self.points, _dp), _p(self.types, _ip), _p(self.colors, _ip), int(self.inverse_y))


def _flat(shape):
    if isinstance(shape, FlatArrays):
        return shape
    return FlatArrays(shape.contour_offsets, shape.points, shape.types, shape.colors, getattr(shape, "inverse_y", False))


def _xf(xf):
    a = _arr(xf, np.float64).reshape(-1)
    assert a.size == 6, "xf = (sx, sy, tx, ty, range_lower, range_upper)"
    return a


class Oracle:
    _PFX = "orc_"
    """The plain-C restatement."""

    kind = "port"

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build("libmsdf_oracle.so")
        self.lib = L = C.CDLL(ORACLE_SO)
        L.orc_signed_distance.argtypes = [C.c_int, _dp, C.c_double, C.c_double, _dp]
        L.orc_solve_cubic.argtypes = [_dp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_solve_quadratic.argtypes = [_dp, C.c_double, C.c_double, C.c_double]
        L.orc_contour_windings.argtypes = [C.POINTER(_OrcShape), _ip]
        L.orc_shape_distance.argtypes = [C.POINTER(_OrcShape), C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.orc_generate.argtypes = [C.POINTER(_OrcShape), C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _bp]
        L.orc_error_correction.argtypes = [C.POINTER(_OrcShape), C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _bp]
        L.orc_ec_stages.argtypes = [C.POINTER(_OrcShape), C.c_int, _fp, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_double, _bp]
        L.orc_generate_batch_timed.argtypes = [C.POINTER(_OrcShape), C.c_int, C.c_int, _fp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_generate_batch_timed.restype = C.c_double

    def signed_distance(self, type_, pts8, ox, oy):
        p = _arr(pts8, np.float64)
        out = np.zeros(3)
        self.lib.orc_signed_distance(type_, _p(p, _dp), ox, oy, _p(out, _dp))
        return out

    def solve_cubic(self, a, b, c, d):
        x = np.zeros(3)
        n = self.lib.orc_solve_cubic(_p(x, _dp), a, b, c, d)
        return n, x

    def solve_quadratic(self, a, b, c):
        x = np.zeros(2)
        n = self.lib.orc_solve_quadratic(_p(x, _dp), a, b, c)
        return n, x

    def windings(self, shape):
        f = _flat(shape)
        w = np.zeros(max(f.n_contours, 1), np.int32)
        s = f.orc()
        self.lib.orc_contour_windings(C.byref(s), _p(w, _ip))
        return w[:f.n_contours]

    def shape_distance(self, shape, selector, overlap, pts):
        f = _flat(shape)
        pts = _arr(pts, np.float64).reshape(-1, 2)
        out = np.zeros((len(pts), 4))
        s = f.orc()
        self.lib.orc_shape_distance(C.byref(s), selector, int(overlap), len(pts), _p(pts, _dp), _p(out, _dp))
        return out

    def generate(self, shape, mode, w, h, xf, overlap=True, ec_mode=EC_EDGE_PRIORITY, ec_dist=DC_CHECK_AT_EDGE,
                 min_dev=DEFAULT_RATIO, min_imp=DEFAULT_RATIO, y_down=False, stencil=None, out=None, row_stride=None):
        f = _flat(shape)
        n = CHANNELS[mode]
        if out is None:
            out = np.zeros((h, w, n), np.float32)
        if row_stride is None:
            row_stride = w*n
        xf = _xf(xf)
        s = f.orc()
        self.lib.orc_generate(C.byref(s), mode, _p(out, _fp), w, h, row_stride, int(y_down), _p(xf, _dp), int(overlap), ec_mode, ec_dist,
                              min_dev, min_imp, _p(stencil, _bp) if stencil is not None else None)
        return out

    def error_correction(self, shape, pixels, xf, overlap=True, ec_mode=EC_EDGE_PRIORITY, ec_dist=DC_CHECK_AT_EDGE,
                         min_dev=DEFAULT_RATIO, min_imp=DEFAULT_RATIO, y_down=False, stencil=None):
        f = _flat(shape)
        px = np.array(pixels, np.float32, order="C")
        h, w, n = px.shape
        xf = _xf(xf)
        s = f.orc()
        self.lib.orc_error_correction(C.byref(s), n, _p(px, _fp), w, h, w*n, int(y_down), _p(xf, _dp), int(overlap), ec_mode, ec_dist,
                                      min_dev, min_imp, _p(stencil, _bp) if stencil is not None else None)
        return px

    def ec_stages(self, shape, pixels, xf, overlap=True, min_dev=DEFAULT_RATIO, min_imp=DEFAULT_RATIO):
        f = _flat(shape)
        px = _arr(pixels, np.float32)
        h, w, n = px.shape
        xf = _xf(xf)
        stages = np.zeros((4, h, w), np.uint8)
        s = f.orc()
        self.lib.orc_ec_stages(C.byref(s), n, _p(px, _fp), w, h, _p(xf, _dp), int(overlap), min_dev, min_imp, _p(stages, _bp))
        return stages

    def sign_correction(self, shape, pixels, xf, zero=.5, fill_rule=0, y_down=False):
        """distanceSignCorrection (core/rasterization.h:17-19) on a copy of `pixels` (h, w, N)."""
        f = _flat(shape)
        px = np.array(pixels, np.float32, order="C")
        h, w, n = px.shape
        x4 = _arr(np.asarray(xf, np.float64).reshape(-1)[:4], np.float64)
        s = f.orc()
        self.lib.orc_sign_correction.argtypes = [C.POINTER(_OrcShape), C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_float, C.c_int]
        self.lib.orc_sign_correction(C.byref(s), n, _p(px, _fp), w, h, w*n, int(y_down), _p(x4, _dp), C.c_float(zero), fill_rule)
        return px

    def shape_prepare(self, shape, normalize=True, coloring=1, angle=3.0, seed=0):
        """Shape::normalize + edgeColoringSimple (core/Shape.cpp:65-92, core/edge-coloring.cpp:68-142) -> FlatArrays."""
        f = _flat(shape)
        s = f.orc()
        ne = int(f.contour_offsets[-1])
        offs = np.zeros(f.n_contours+1, np.int32)
        pts = np.zeros((max(3*ne, 1), 8))
        types = np.zeros(max(3*ne, 1), np.int32)
        colors = np.zeros(max(3*ne, 1), np.int32)
        self.lib.orc_shape_prepare.argtypes = [C.POINTER(_OrcShape), C.c_int, C.c_int, C.c_double, C.c_ulonglong, _ip, _dp, _ip, _ip]
        n = self.lib.orc_shape_prepare(C.byref(s), int(normalize), int(coloring), float(angle), int(seed), _p(offs, _ip), _p(pts, _dp), _p(types, _ip), _p(colors, _ip))
        return FlatArrays(offs, pts[:n], types[:n], colors[:n], bool(f.inverse_y))

    def estimate_sdf_error(self, shape, sdf, xf, scanlines_per_row=1, fill_rule=0, per_line=False):
        """estimateSDFError (core/sdf-error-estimation.h:18-20) of sdf (h, w, N), memory rows."""
        f = _flat(shape)
        px = _arr(sdf, np.float32)
        h, w, n = px.shape
        x4 = _arr(np.asarray(xf, np.float64).reshape(-1)[:4], np.float64)
        s = f.orc()
        lines = np.zeros(max((h-1)*scanlines_per_row, 1))
        self.lib.orc_estimate_sdf_error.restype = C.c_double
        self.lib.orc_estimate_sdf_error.argtypes = [C.POINTER(_OrcShape), _fp, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, _dp]
        v = self.lib.orc_estimate_sdf_error(C.byref(s), _p(px, _fp), w, h, n, _p(x4, _dp), scanlines_per_row, fill_rule, _p(lines, _dp))
        return (v, lines[:max(h-1, 0)*scanlines_per_row]) if per_line else v

    def render_sdf(self, sdf, ow, oh, n_out, range_lower=0., range_upper=0., threshold=.5):
        """renderSDF (core/render-sdf.h:12-17): (oh, ow, n_out) float32 from sdf (sh, sw, Ns)."""
        sdf = _arr(sdf, np.float32)
        sh, sw, ns = sdf.shape
        out = np.zeros((oh, ow, n_out), np.float32)
        fn = getattr(self.lib, self._PFX+"render_sdf")
        fn.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_float]
        if fn(_p(out, _fp), ow, oh, n_out, _p(sdf, _fp), sw, sh, ns, float(range_lower), float(range_upper), C.c_float(threshold)) != 0:
            raise ValueError("renderSDF has no overload for %d <- %d channels" % (n_out, ns))
        return out

    def simulate_8bit(self, a):
        a = np.array(a, np.float32, order="C")
        if self._PFX == "orc_":
            self.lib.orc_simulate_8bit.argtypes = [_fp, C.c_long]
            self.lib.orc_simulate_8bit(_p(a, _fp), a.size)
        else:
            h, w, n = a.shape
            self.lib.ref_simulate_8bit.argtypes = [_fp, C.c_int, C.c_int, C.c_int]
            self.lib.ref_simulate_8bit(_p(a, _fp), w, h, n)
        return a

    def pixel_float_to_byte(self, a):
        """pixelFloatToByte (core/pixel-conversion.hpp:8-10), elementwise."""
        a = _arr(a, np.float32)
        out = np.zeros(a.shape, np.uint8)
        fn = getattr(self.lib, self._PFX+"pixel_float_to_byte")
        fn.argtypes = [_fp, _bp, C.c_long]
        fn(_p(a, _fp), _p(out, _bp), a.size)
        return out

    def rasterize(self, shape, w, h, xf, fill_rule=0, y_down=False):
        """rasterize (core/rasterization.h:13): (h, w, 1) coverage."""
        f = _flat(shape)
        px = np.zeros((h, w, 1), np.float32)
        x4 = _arr(np.asarray(xf, np.float64).reshape(-1)[:4], np.float64)
        s = f.orc()
        self.lib.orc_rasterize.argtypes = [C.POINTER(_OrcShape), _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int]
        self.lib.orc_rasterize(C.byref(s), _p(px, _fp), w, h, w, int(y_down), _p(x4, _dp), fill_rule)
        return px

    def scanline_intersections(self, type_, pts8, y):
        p = _arr(pts8, np.float64)
        x = np.zeros(3)
        dy = np.zeros(3, np.int32)
        self.lib.orc_scanline_intersections.argtypes = [C.c_int, _dp, C.c_double, _dp, _ip]
        n = self.lib.orc_scanline_intersections(type_, _p(p, _dp), y, _p(x, _dp), _p(dy, _ip))
        return n, x[:n], dy[:n]

    def generate_batch_timed(self, shapes, mode, w, h, xfs, overlap=True, ec_mode=EC_EDGE_PRIORITY, ec_dist=DC_CHECK_AT_EDGE,
                             min_dev=DEFAULT_RATIO, min_imp=DEFAULT_RATIO, threads=1):
        flats = [_flat(s) for s in shapes]
        arr = (_OrcShape*len(flats))(*[f.orc() for f in flats])
        xfs = _arr(xfs, np.float64).reshape(len(flats), 6)
        out = np.zeros((len(flats), h, w, CHANNELS[mode]), np.float32)
        secs = self.lib.orc_generate_batch_timed(arr, len(flats), mode, _p(out, _fp), w, h, _p(xfs, _dp), int(overlap), ec_mode, ec_dist,
                                                 min_dev, min_imp, threads)
        return out, secs


class Ref:
    _PFX = "ref_"
    """The compiled reference (msdfgen v1.13.0 core)."""

    kind = "reference"

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, openmp=False):
        so = REF_OMP_SO if openmp else REF_SO
        if not os.path.exists(so):
            if os.path.isdir("/root/reference/core"):
                build("ref")
            else:
                raise FileNotFoundError(so+" (build it in the authoring container: make -C oracle ref)")
        self.lib = L = C.CDLL(so)
        vp = C.c_void_p
        L.ref_version.restype = C.c_char_p
        L.ref_shape_from_desc.argtypes = [C.c_char_p]
        L.ref_shape_from_desc.restype = vp
        L.ref_shape_from_flat.argtypes = [C.c_int, _ip, _dp, _ip, _ip, C.c_int]
        L.ref_shape_from_flat.restype = vp
        for name in ("ref_shape_free", "ref_shape_normalize", "ref_shape_orient_contours"):
            getattr(L, name).argtypes = [vp]
            getattr(L, name).restype = None
        L.ref_shape_validate.argtypes = [vp]
        L.ref_shape_inverse_y.argtypes = [vp]
        L.ref_shape_set_inverse_y.argtypes = [vp, C.c_int]
        L.ref_shape_color_simple.argtypes = [vp, C.c_double, C.c_ulonglong]
        L.ref_shape_color_inktrap.argtypes = [vp, C.c_double, C.c_ulonglong]
        L.ref_shape_counts.argtypes = [vp, _ip, _ip]
        L.ref_shape_bounds.argtypes = [vp, _dp]
        L.ref_shape_flatten.argtypes = [vp, _ip, _dp, _ip, _ip, _ip]
        L.ref_generate.argtypes = [vp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _bp]
        L.ref_error_correction.argtypes = [vp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, _bp]
        L.ref_ec_stages.argtypes = [vp, C.c_int, _fp, C.c_int, C.c_int, _dp, C.c_int, C.c_double, C.c_double, _bp]
        if hasattr(L, "ref_fast_error_correction"):                 # (a prebuilt _ref from before round 3 lacks it)
            L.ref_fast_error_correction.argtypes = [C.c_int, _fp, C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.c_int]
        L.ref_signed_distance.argtypes = [C.c_int, _dp, C.c_double, C.c_double, _dp]
        L.ref_solve_cubic.argtypes = [_dp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.ref_solve_quadratic.argtypes = [_dp, C.c_double, C.c_double, C.c_double]
        L.ref_oneshot_distance.argtypes = [vp, C.c_int, C.c_int, C.c_int, _dp, _dp]
        L.ref_generate_batch_timed.argtypes = [C.POINTER(vp), C.c_int, C.c_int, _fp, C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.ref_generate_batch_timed.restype = C.c_double

    def version(self):
        return self.lib.ref_version().decode()

    # -- shape handles -------------------------------------------------------------------------------------------
    def shape_from_desc(self, text):
        h = self.lib.ref_shape_from_desc(text.encode())
        if not h:
            raise ValueError("readShapeDescription failed")
        return h

    def shape_from_flat(self, shape):
        f = _flat(shape)
        h = self.lib.ref_shape_from_flat(f.n_contours, _p(f.contour_offsets, _ip), _p(f.points, _dp), _p(f.types, _ip), _p(f.colors, _ip), int(f.inverse_y))
        if not h:
            raise ValueError("bad edge type")
        return h

    def free(self, h):
        self.lib.ref_shape_free(h)

    def flatten(self, h):
        nc, ne = C.c_int32(), C.c_int32()
        self.lib.ref_shape_counts(h, C.byref(nc), C.byref(ne))
        nc, ne = nc.value, ne.value
        offs = np.zeros(nc+1, np.int32)
        pts = np.zeros((max(ne, 1), 8))
        types = np.zeros(max(ne, 1), np.int32)
        colors = np.zeros(max(ne, 1), np.int32)
        wind = np.zeros(max(nc, 1), np.int32)
        self.lib.ref_shape_flatten(h, _p(offs, _ip), _p(pts, _dp), _p(types, _ip), _p(colors, _ip), _p(wind, _ip))
        fa = FlatArrays(offs, pts[:ne], types[:ne], colors[:ne], bool(self.lib.ref_shape_inverse_y(h)))
        fa.windings = wind[:nc]
        return fa

    def prepare(self, h, angle=3.0, seed=0, normalize=True, color=True):
        """Caller-side prep that precedes the hot path: Shape::normalize + edgeColoringSimple (main.cpp:1125, 1255)."""
        if normalize:
            self.lib.ref_shape_normalize(h)
        if color:
            self.lib.ref_shape_color_simple(h, angle, seed)

    def bounds(self, h):
        b = np.zeros(4)
        self.lib.ref_shape_bounds(h, _p(b, _dp))
        return b

    # -- hot path ------------------------------------------------------------------------------------------------
    def _handle(self, shape):
        if isinstance(shape, int):
            return shape, False
        return self.shape_from_flat(shape), True

    def generate(self, shape, mode, w, h, xf, overlap=True, ec_mode=EC_EDGE_PRIORITY, ec_dist=DC_CHECK_AT_EDGE,
                 min_dev=DEFAULT_RATIO, min_imp=DEFAULT_RATIO, y_down=False, stencil=None, out=None, row_stride=None):
        hd, own = self._handle(shape)
        n = CHANNELS[mode]
        if out is None:
            out = np.zeros((h, w, n), np.float32)
        if row_stride is None:
            row_stride = w*n
        xf = _xf(xf)
        self.lib.ref_generate(hd, mode, _p(out, _fp), w, h, row_stride, int(y_down), _p(xf, _dp), int(overlap), ec_mode, ec_dist, min_dev, min_imp,
                              _p(stencil, _bp) if stencil is not None else None)
        if own:
            self.free(hd)
        return out

    def error_correction(self, shape, pixels, xf, overlap=True, ec_mode=EC_EDGE_PRIORITY, ec_dist=DC_CHECK_AT_EDGE,
                         min_dev=DEFAULT_RATIO, min_imp=DEFAULT_RATIO, y_down=False, stencil=None):
        hd, own = self._handle(shape)
        px = np.array(pixels, np.float32, order="C")
        h, w, n = px.shape
        xf = _xf(xf)
        self.lib.ref_error_correction(hd, n, _p(px, _fp), w, h, w*n, int(y_down), _p(xf, _dp), int(overlap), ec_mode, ec_dist, min_dev, min_imp,
                                      _p(stencil, _bp) if stencil is not None else None)
        if own:
            self.free(hd)
        return px

    def fast_error_correction(self, pixels, xf, min_dev=DEFAULT_RATIO, protect_all=False):
        """msdfFastDistanceErrorCorrection (protect_all False) / msdfFastEdgeErrorCorrection (True), core/msdf-error-correction.h:21-34."""
        px = np.array(pixels, np.float32, order="C")
        h, w, n = px.shape
        xf = _xf(xf)
        self.lib.ref_fast_error_correction(n, _p(px, _fp), w, h, w*n, _p(xf, _dp), C.c_double(min_dev), int(protect_all))
        return px

    def ec_stages(self, shape, pixels, xf, overlap=True, min_dev=DEFAULT_RATIO, min_imp=DEFAULT_RATIO):
        hd, own = self._handle(shape)
        px = _arr(pixels, np.float32)
        h, w, n = px.shape
        xf = _xf(xf)
        stages = np.zeros((4, h, w), np.uint8)
        self.lib.ref_ec_stages(hd, n, _p(px, _fp), w, h, _p(xf, _dp), int(overlap), min_dev, min_imp, _p(stages, _bp))
        if own:
            self.free(hd)
        return stages

    def signed_distance(self, type_, pts8, ox, oy):
        p = _arr(pts8, np.float64)
        out = np.zeros(3)
        self.lib.ref_signed_distance(type_, _p(p, _dp), ox, oy, _p(out, _dp))
        return out

    def solve_cubic(self, a, b, c, d):
        x = np.zeros(3)
        n = self.lib.ref_solve_cubic(_p(x, _dp), a, b, c, d)
        return n, x

    def solve_quadratic(self, a, b, c):
        x = np.zeros(2)
        n = self.lib.ref_solve_quadratic(_p(x, _dp), a, b, c)
        return n, x

    def shape_distance(self, shape, selector, overlap, pts):
        hd, own = self._handle(shape)
        pts = _arr(pts, np.float64).reshape(-1, 2)
        out = np.zeros((len(pts), 4))
        self.lib.ref_oneshot_distance(hd, selector, int(overlap), len(pts), _p(pts, _dp), _p(out, _dp))
        if own:
            self.free(hd)
        return out

    def sign_correction(self, shape, pixels, xf, zero=.5, fill_rule=0, y_down=False):
        """distanceSignCorrection (core/rasterization.h:17-19) on a copy of `pixels` (h, w, N)."""
        hd, own = self._handle(shape)
        px = np.array(pixels, np.float32, order="C")
        h, w, n = px.shape
        x4 = _arr(np.asarray(xf, np.float64).reshape(-1)[:4], np.float64)
        self.lib.ref_sign_correction.argtypes = [C.c_void_p, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_float, C.c_int]
        self.lib.ref_sign_correction(hd, n, _p(px, _fp), w, h, w*n, int(y_down), _p(x4, _dp), C.c_float(zero), fill_rule)
        if own:
            self.free(hd)
        return px

    def shape_prepare(self, shape, normalize=True, coloring=1, angle=3.0, seed=0):
        """The reference's own Shape::normalize + edgeColoringSimple on a copy of `shape` -> FlatArrays."""
        h = self.shape_from_flat(shape)
        self.prepare(h, angle, seed, normalize=normalize, color=coloring == 1)
        if coloring == 2:
            self.lib.ref_shape_color_inktrap(h, angle, seed)
        fa = self.flatten(h)
        self.free(h)
        return fa

    def estimate_sdf_error(self, shape, sdf, xf, scanlines_per_row=1, fill_rule=0):
        """The reference's estimateSDFError (core/sdf-error-estimation.h:18-20) of sdf (h, w, N), memory rows."""
        hd, own = self._handle(shape)
        px = _arr(sdf, np.float32)
        h, w, n = px.shape
        x4 = _arr(np.asarray(xf, np.float64).reshape(-1)[:4], np.float64)
        self.lib.ref_estimate_sdf_error.restype = C.c_double
        self.lib.ref_estimate_sdf_error.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_int]
        v = self.lib.ref_estimate_sdf_error(hd, _p(px, _fp), w, h, n, _p(x4, _dp), scanlines_per_row, fill_rule)
        if own:
            self.free(hd)
        return v

    def render_sdf(self, sdf, ow, oh, n_out, range_lower=0., range_upper=0., threshold=.5):
        """renderSDF (core/render-sdf.h:12-17): (oh, ow, n_out) float32 from sdf (sh, sw, Ns)."""
        sdf = _arr(sdf, np.float32)
        sh, sw, ns = sdf.shape
        out = np.zeros((oh, ow, n_out), np.float32)
        fn = getattr(self.lib, self._PFX+"render_sdf")
        fn.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_float]
        if fn(_p(out, _fp), ow, oh, n_out, _p(sdf, _fp), sw, sh, ns, float(range_lower), float(range_upper), C.c_float(threshold)) != 0:
            raise ValueError("renderSDF has no overload for %d <- %d channels" % (n_out, ns))
        return out

    def simulate_8bit(self, a):
        a = np.array(a, np.float32, order="C")
        if self._PFX == "orc_":
            self.lib.orc_simulate_8bit.argtypes = [_fp, C.c_long]
            self.lib.orc_simulate_8bit(_p(a, _fp), a.size)
        else:
            h, w, n = a.shape
            self.lib.ref_simulate_8bit.argtypes = [_fp, C.c_int, C.c_int, C.c_int]
            self.lib.ref_simulate_8bit(_p(a, _fp), w, h, n)
        return a

    def pixel_float_to_byte(self, a):
        """pixelFloatToByte (core/pixel-conversion.hpp:8-10), elementwise."""
        a = _arr(a, np.float32)
        out = np.zeros(a.shape, np.uint8)
        fn = getattr(self.lib, self._PFX+"pixel_float_to_byte")
        fn.argtypes = [_fp, _bp, C.c_long]
        fn(_p(a, _fp), _p(out, _bp), a.size)
        return out

    def rasterize(self, shape, w, h, xf, fill_rule=0, y_down=False):
        """rasterize (core/rasterization.h:13): (h, w, 1) coverage."""
        hd, own = self._handle(shape)
        px = np.zeros((h, w, 1), np.float32)
        x4 = _arr(np.asarray(xf, np.float64).reshape(-1)[:4], np.float64)
        self.lib.ref_rasterize.argtypes = [C.c_void_p, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, C.c_int]
        self.lib.ref_rasterize(hd, _p(px, _fp), w, h, w, int(y_down), _p(x4, _dp), fill_rule)
        if own:
            self.free(hd)
        return px

    def scanline_intersections(self, type_, pts8, y):
        p = _arr(pts8, np.float64)
        x = np.zeros(3)
        dy = np.zeros(3, np.int32)
        self.lib.ref_scanline_intersections.argtypes = [C.c_int, _dp, C.c_double, _dp, _ip]
        n = self.lib.ref_scanline_intersections(type_, _p(p, _dp), y, _p(x, _dp), _p(dy, _ip))
        return n, x[:n], dy[:n]

    def generate_batch_timed(self, shapes, mode, w, h, xfs, overlap=True, ec_mode=EC_EDGE_PRIORITY, ec_dist=DC_CHECK_AT_EDGE,
                             min_dev=DEFAULT_RATIO, min_imp=DEFAULT_RATIO, threads=1):
        handles = [self.shape_from_flat(s) for s in shapes]
        arr = (C.c_void_p*len(handles))(*handles)
        xfs = _arr(xfs, np.float64).reshape(len(handles), 6)
        out = np.zeros((len(handles), h, w, CHANNELS[mode]), np.float32)
        secs = self.lib.ref_generate_batch_timed(arr, len(handles), mode, _p(out, _fp), w, h, _p(xfs, _dp), int(overlap), ec_mode, ec_dist,
                                                 min_dev, min_imp, threads)
        for hd in handles:
            self.free(hd)
        return out, secs
