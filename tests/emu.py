"""Binding for tests/hostemu (the product's device headers compiled for the host; logic check only, see hostemu.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from msdfgen_amd.shape import distance_mapping

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostemu")
SO = os.path.join(HERE, "libhostemu.so")
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "msdfgen_amd", "csrc")


def build(lean=False):
    """lean: -DMSDF_LEAN_MATH, i.e. the device's own < 1 ulp cos / cbrt (msdf_device.hpp) instead of libm's in solveCubicNormed -- the host
    rendition then follows the KERNELS bit for bit also where the last ulp of a transcendental decides."""
    so = SO.replace(".so", "_lean.so") if lean else SO
    srcs = [os.path.join(HERE, "hostemu.cpp")]+[os.path.join(CSRC, f) for f in ("msdf_device.hpp", "msdf_prep.hpp", "msdf_ec.hpp", "msdf_ec_fast.hpp", "msdf_cull.hpp", "msdf_scanline.hpp", "msdf_shapeprep.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]+(["-DMSDF_LEAN_MATH"] if lean else [])+["-o", so, srcs[0]], check=True)
    return so


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


class Emu:
    def __init__(self, lean=False):
        self.lib = C.CDLL(build(lean))

    def _shape(self, s):
        co = np.ascontiguousarray(s.contour_offsets, np.int32)
        pts = np.ascontiguousarray(s.points, np.float64)
        t8 = np.ascontiguousarray(s.types, np.uint8)
        c8 = np.ascontiguousarray(s.colors, np.uint8)
        return (co, pts, t8, c8), (s.n_contours, _p(co, C.c_int32), _p(pts, C.c_double), _p(t8, C.c_uint8), _p(c8, C.c_uint8))

    def set_combiner_form(self, form):
        """0: the rolled pass loop (shapeDistanceOverlap); 1: the kernels' two instances of the contour loop (shapeDistanceOverlapSplit)."""
        self.lib.emu_set_combiner_form(int(form))

    def windings(self, s, wave=None):
        """wave: the forms the digest kernels run since round 4 (contourWindingsWave: lanes = edges, ordered sums) -- "single" = k_single_call's (all
        contours in one cooperative walk), "batch" = k_prep_records' (a lane per contour, long contours by the wavefront together); None: a contour per lane."""
        keep, args = self._shape(s)
        out = np.zeros(max(s.n_contours, 1), np.int32)
        if wave:
            self.lib.emu_windings_wave(*args, _p(out, C.c_int32), {"single": 0, "batch": 1}[wave])
        else:
            self.lib.emu_windings(*args, _p(out, C.c_int32))
        return out[:s.n_contours]

    def shape_distance(self, s, sel, overlap, pts):
        keep, args = self._shape(s)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
        out = np.zeros((len(pts), 4))
        self.lib.emu_shape_distance(sel, int(overlap), *args, len(pts), _p(pts, C.c_double), _p(out, C.c_double))
        return out

    def generate(self, s, mode, w, h, xf, overlap=True, ec_mode=2, ec_dist=1, stage=0, y_down=False, correct_only=None, stencil=None,
                 min_dev=1.11111111111111111, min_imp=1.11111111111111111):
        ms, mt = distance_mapping(xf[4], xf[5])
        x6 = np.array([xf[0], xf[1], xf[2], xf[3], ms, mt], np.float64)
        flip = int(bool(s.inverse_y) != bool(y_down))
        if correct_only is None:
            n = {1: 1, 2: 1, 3: 3, 4: 4}[mode]
            out = np.zeros((h, w, n), np.float32)
        else:
            out = np.array(correct_only, np.float32, order="C")
            n = mode = out.shape[2]
        keep, args = self._shape(s)
        self.lib.emu_generate(mode, int(correct_only is not None), _p(out, C.c_float), w, h, w*n, flip, *args, _p(x6, C.c_double), int(overlap),
                              ec_mode, ec_dist, stage, C.c_double(min_dev), C.c_double(min_imp), _p(stencil, C.c_uint8) if stencil is not None else None)
        return out

    def sign_correction(self, s, field, xf, zero=.5, rule=0, y_down=False):
        out = np.array(field, np.float32, order="C")
        h, w, n = out.shape
        x6 = np.array([xf[0], xf[1], xf[2], xf[3], 1, 0], np.float64)
        flip = int(bool(s.inverse_y) != bool(y_down))
        keep, args = self._shape(s)
        self.lib.emu_sign_correction(n, _p(out, C.c_float), w, h, w*n, flip, *args, _p(x6, C.c_double), C.c_float(zero), int(rule))
        return out

    def scanline_intersections(self, t, pts, y):
        pts = np.ascontiguousarray(pts, np.float64)
        x = np.zeros(3)
        dy = np.zeros(3, np.int32)
        self.lib.emu_scanline_intersections.restype = C.c_int
        n = self.lib.emu_scanline_intersections(int(t), _p(pts, C.c_double), C.c_double(y), _p(x, C.c_double), _p(dy, C.c_int32))
        return n, x[:n], dy[:n]

    def rasterize(self, s, w, h, xf, rule=0, y_down=False):
        out = np.zeros((h, w, 1), np.float32)
        x6 = np.array([xf[0], xf[1], xf[2], xf[3], 1, 0], np.float64)
        keep, args = self._shape(s)
        self.lib.emu_rasterize(_p(out, C.c_float), w, h, w, int(bool(s.inverse_y) != bool(y_down)), *args, _p(x6, C.c_double), int(rule))
        return out

    def psdf_cooperative(self, s, overlap, pts, slotted=False):
        keep, args = self._shape(s)
        pts = np.ascontiguousarray(pts, np.float64).reshape(-1, 2)
        out = np.zeros(len(pts))
        self.lib.emu_psdf_cooperative(int(overlap)+2*int(slotted), *args, len(pts), _p(pts, C.c_double), _p(out, C.c_double))
        return out

    def shape_prepare(self, s, normalize=True, coloring=1, angle=3.0, seed=0, wave=False):
        """wave: the lanes = edges / corners forms the kernels run since round 4 (emu_shape_prepare_wave), else the serial ones;
        with wave, self.cusp_contours = the contours the flat normalize pass handed to the serial cusp repair."""
        from msdfgen_amd.shape import FlatShape
        keep, args = self._shape(s)
        ne = int(np.asarray(s.contour_offsets)[-1])
        offs = np.zeros(s.n_contours+1, np.int32)
        pts = np.zeros((max(3*ne, 1), 8))
        types = np.zeros(max(3*ne, 1), np.uint8)
        colors = np.zeros(max(3*ne, 1), np.uint8)
        if wave:
            flagged = C.c_int(0)
            self.lib.emu_shape_prepare_wave.restype = C.c_int
            n = self.lib.emu_shape_prepare_wave(*args, int(normalize), int(coloring), C.c_double(angle), C.c_ulonglong(int(seed)), _p(offs, C.c_int32),
                                                _p(pts, C.c_double), _p(types, C.c_uint8), _p(colors, C.c_uint8), C.byref(flagged))
            self.cusp_contours = flagged.value
            return FlatShape(offs, pts[:n], types[:n].astype(np.int32), colors[:n].astype(np.int32))
        self.lib.emu_shape_prepare.restype = C.c_int
        n = self.lib.emu_shape_prepare(*args, int(normalize), int(coloring), C.c_double(angle), C.c_ulonglong(int(seed)), _p(offs, C.c_int32), _p(pts, C.c_double),
                                       _p(types, C.c_uint8), _p(colors, C.c_uint8))
        return FlatShape(offs, pts[:n], types[:n].astype(np.int32), colors[:n].astype(np.int32))

    def estimate_sdf_error(self, s, sdf, xf, scanlines_per_row=1, fill_rule=0):
        px = np.ascontiguousarray(sdf, np.float32)
        h, w, n = px.shape
        x4 = np.ascontiguousarray(np.asarray(xf, np.float64).reshape(-1)[:4])
        keep, args = self._shape(s)
        self.lib.emu_estimate_sdf_error.restype = C.c_double
        return self.lib.emu_estimate_sdf_error(n, _p(px, C.c_float), w, h, int(bool(s.inverse_y)), *args, _p(x4, C.c_double), int(scanlines_per_row), int(fill_rule))
