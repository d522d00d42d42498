"""Host-side mirror of the reference's generator interface for the hot path (msdfgen.h:46-56, core/msdf-error-correction.h:15-18),
on top of the C ABI.  Same names, argument meaning and defaults as the reference; outputs are numpy float32 bitmaps
`(height, width, N)` standing in for `BitmapSection<float, N>`.

    generate_sdf / generate_psdf / generate_msdf / generate_mtsdf (output, shape, transformation, config)
    msdf_error_correction(sdf, shape, transformation, config)
    distance_sign_correction(sdf, shape, projection, sdf_zero_value, fill_rule)
    rasterize(output, shape, projection, fill_rule)
    render_sdf(sdf_tiles, out_width, out_height, out_channels, sdf_px_range, sd_threshold) / simulate_8bit(tiles)   -- device tensors
    shape_distance(shape, selector, overlap_support, points)      -- ShapeDistanceFinder::oneShotDistance

`GlyphBatch` is the batched, device-resident front door (one launch for thousands of glyph tiles); it uses torch only to own
device memory and streams.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import lib as _lib
from .lib import (MODE_SDF, MODE_PSDF, MODE_MSDF, MODE_MTSDF, CHANNELS, EC_DISABLED, EC_INDISCRIMINATE, EC_EDGE_PRIORITY, EC_EDGE_ONLY,
                  DO_NOT_CHECK_DISTANCE, CHECK_DISTANCE_AT_EDGE, ALWAYS_CHECK_DISTANCE, FILL_NONZERO, FILL_ODD, FILL_POSITIVE, FILL_NEGATIVE,
                  MsdfHipError)
from .shape import FlatShape, ShapeBatch, distance_mapping

DEFAULT_MIN_DEVIATION_RATIO = 1.11111111111111111  # ErrorCorrectionConfig::defaultMinDeviationRatio (MSDFErrorCorrection.cpp:22)
DEFAULT_MIN_IMPROVE_RATIO = 1.11111111111111111    # ErrorCorrectionConfig::defaultMinImproveRatio   (MSDFErrorCorrection.cpp:23)
Y_UPWARD, Y_DOWNWARD = 0, 1                        # YAxisOrientation (core/YAxisOrientation.h:9-16)


@dataclass
class Projection:
    """core/Projection.h: shape -> pixel affine map, project(c) = scale*(c+translate)."""
    scale: Sequence[float] = (1., 1.)
    translate: Sequence[float] = (0., 0.)


@dataclass
class Range:
    """core/Range.hpp:12-18. Range(w) is the symmetric range (-w/2, +w/2)."""
    lower: float = 0.
    upper: Optional[float] = None

    def __post_init__(self):
        if self.upper is None:
            w = float(self.lower)
            self.lower, self.upper = -.5*w, .5*w


@dataclass
class DistanceMapping:
    """core/DistanceMapping.h: d -> scale*(d+translate). Built from a Range as in DistanceMapping.cpp:13."""
    scale: float = 1.
    translate: float = 0.

    @staticmethod
    def from_range(r: Range) -> "DistanceMapping":
        s, t = distance_mapping(r.lower, r.upper)
        return DistanceMapping(s, t)


@dataclass
class SDFTransformation:
    """core/SDFTransformation.h:13-24."""
    projection: Projection = field(default_factory=Projection)
    distance_mapping: DistanceMapping = field(default_factory=DistanceMapping)

    @staticmethod
    def from_xf(xf) -> "SDFTransformation":
        """xf = (sx, sy, tx, ty, range_lower, range_upper) as produced by shape.autoframe()."""
        xf = [float(v) for v in xf]
        return SDFTransformation(Projection((xf[0], xf[1]), (xf[2], xf[3])), DistanceMapping.from_range(Range(xf[4], xf[5])))

    def xf6(self) -> np.ndarray:
        return np.array([self.projection.scale[0], self.projection.scale[1], self.projection.translate[0], self.projection.translate[1],
                         self.distance_mapping.scale, self.distance_mapping.translate], np.float64)


@dataclass
class ErrorCorrectionConfig:
    """core/generator-config.h:13-47."""
    mode: int = EC_EDGE_PRIORITY
    distance_check_mode: int = CHECK_DISTANCE_AT_EDGE
    min_deviation_ratio: float = DEFAULT_MIN_DEVIATION_RATIO
    min_improve_ratio: float = DEFAULT_MIN_IMPROVE_RATIO
    buffer: Optional[np.ndarray] = None  # optional uint8 (h, w): receives the final stencil (rows in the bitmap's memory order)


@dataclass
class GeneratorConfig:
    """core/generator-config.h:50-56."""
    overlap_support: bool = True


@dataclass
class MSDFGeneratorConfig(GeneratorConfig):
    """core/generator-config.h:58-64."""
    error_correction: ErrorCorrectionConfig = field(default_factory=ErrorCorrectionConfig)
    _stage_limit: int = 0  # test hook: stop the stencil pipeline after stage k (see MsdfHipConfig.ec_stage_limit)


def _c_config(config, y_orientation=None) -> _lib.Config:
    """y_orientation: the OUTPUT bitmap's orientation; the reference keeps the stencil's rows upward whatever it is
    (core/MSDFErrorCorrection.cpp:122,192,415), which is what MsdfHipConfig.stencil_y_down tells the device -- set on every path that
    takes a stencil, single-shape and batched alike."""
    cfg = _lib.default_config()
    if y_orientation is not None:
        cfg.stencil_y_down = int(y_orientation == Y_DOWNWARD)
    if config is None:
        return cfg
    cfg.overlap_support = 1 if config.overlap_support else 0
    ec = getattr(config, "error_correction", None)
    if ec is not None:
        cfg.ec_mode = int(ec.mode)
        cfg.ec_distance_check = int(ec.distance_check_mode)
        cfg.min_deviation_ratio = float(ec.min_deviation_ratio)
        cfg.min_improve_ratio = float(ec.min_improve_ratio)
    cfg.ec_stage_limit = int(getattr(config, "_stage_limit", 0))
    return cfg


def _with_scanline_pass(cfg, scanline_pass, fill_rule, sdf_zero_value):
    cfg.sign_correction = 1 if scanline_pass else 0
    cfg.fill_rule = int(fill_rule)
    cfg.sdf_zero_value = float(sdf_zero_value)
    return cfg


def _shape_args(shape: FlatShape):
    co = np.ascontiguousarray(shape.contour_offsets, np.int32)
    pts = np.ascontiguousarray(shape.points, np.float64)
    types = np.ascontiguousarray(shape.types, np.uint8)
    colors = np.ascontiguousarray(shape.colors, np.uint8)
    keep = (co, pts, types, colors)
    return keep, (_lib.ptr(co, _lib._ip), shape.n_contours, _lib.ptr(pts, _lib._dp), _lib.ptr(types, _lib._bp), _lib.ptr(colors, _lib._bp))


def _bitmap_args(output: np.ndarray, n: int):
    if output.dtype != np.float32 or output.ndim != 3 or output.shape[2] != n:
        raise ValueError("output must be float32 of shape (height, width, %d)" % n)
    if output.size and (output.strides[2] != 4 or output.strides[1] != 4*n or output.strides[0] % 4):
        raise ValueError("output texels must be channel-interleaved and rows float-aligned")
    h, w = output.shape[:2]
    return _lib.ptr(output, _lib._fp), w, h, output.strides[0]//4  # row stride in floats; may be negative for a flipped view


def _generate(mode, output, shape, transformation, config, y_orientation):
    lib = _lib.load()
    px, w, h, stride = _bitmap_args(output, CHANNELS[mode])
    keep, sargs = _shape_args(shape)
    xf = transformation.xf6()
    cfg = _c_config(config)
    flip = int(bool(shape.inverse_y) != (y_orientation == Y_DOWNWARD))  # shape.getYAxisOrientation() != output.yOrientation
    cfg.stencil_y_down = int(y_orientation == Y_DOWNWARD)
    stencil = None
    ec = getattr(config, "error_correction", None)
    if ec is not None and ec.buffer is not None:
        stencil = ec.buffer
        if stencil.dtype != np.uint8 or stencil.size < w*h or not stencil.flags.c_contiguous:
            raise ValueError("error_correction.buffer must be C-contiguous uint8 with at least width*height bytes")
    _lib.check(lib.msdfhip_generate(mode, px, w, h, stride, flip, *sargs, _lib.ptr(xf, _lib._dp), C.byref(cfg),
                                    _lib.ptr(stencil, _lib._bp) if stencil is not None else None))
    del keep
    return output


def generate_sdf(output, shape, transformation, config: Optional[GeneratorConfig] = None, y_orientation=Y_UPWARD):
    """generateSDF (msdfgen.h:47, core/msdfgen.cpp:78-83)."""
    return _generate(MODE_SDF, output, shape, transformation, config or GeneratorConfig(), y_orientation)


def generate_psdf(output, shape, transformation, config: Optional[GeneratorConfig] = None, y_orientation=Y_UPWARD):
    """generatePSDF (msdfgen.h:49, core/msdfgen.cpp:85-90)."""
    return _generate(MODE_PSDF, output, shape, transformation, config or GeneratorConfig(), y_orientation)


def generate_msdf(output, shape, transformation, config: Optional[MSDFGeneratorConfig] = None, y_orientation=Y_UPWARD):
    """generateMSDF (msdfgen.h:51, core/msdfgen.cpp:92-98), including the error-correction pass."""
    return _generate(MODE_MSDF, output, shape, transformation, config or MSDFGeneratorConfig(), y_orientation)


def generate_mtsdf(output, shape, transformation, config: Optional[MSDFGeneratorConfig] = None, y_orientation=Y_UPWARD):
    """generateMTSDF (msdfgen.h:53, core/msdfgen.cpp:100-106), including the error-correction pass."""
    return _generate(MODE_MTSDF, output, shape, transformation, config or MSDFGeneratorConfig(), y_orientation)


def msdf_error_correction(sdf, shape, transformation, config: Optional[MSDFGeneratorConfig] = None, y_orientation=Y_UPWARD):
    """msdfErrorCorrection (core/msdf-error-correction.h:15-16), in place on a 3- or 4-channel bitmap."""
    lib = _lib.load()
    config = config or MSDFGeneratorConfig()
    n = sdf.shape[2]
    px, w, h, stride = _bitmap_args(sdf, n)
    keep, sargs = _shape_args(shape)
    xf = transformation.xf6()
    cfg = _c_config(config)
    flip = int(bool(shape.inverse_y) != (y_orientation == Y_DOWNWARD))
    cfg.stencil_y_down = int(y_orientation == Y_DOWNWARD)
    stencil = config.error_correction.buffer
    if stencil is not None and (stencil.dtype != np.uint8 or stencil.size < w*h or not stencil.flags.c_contiguous):
        raise ValueError("error_correction.buffer must be C-contiguous uint8 with at least width*height bytes")
    _lib.check(lib.msdfhip_error_correction(n, px, w, h, stride, flip, *sargs, _lib.ptr(xf, _lib._dp), C.byref(cfg),
                                            _lib.ptr(stencil, _lib._bp) if stencil is not None else None))
    del keep
    return sdf


def _fast_error_correction(sdf, transformation, min_deviation_ratio, protect_all):
    assert sdf.dtype == np.float32 and sdf.ndim == 3 and sdf.shape[2] in (3, 4) and sdf.flags.c_contiguous
    h, w, n = sdf.shape
    xf = np.ascontiguousarray(transformation.xf6(), np.float64)
    _lib.check(_lib.load().msdfhip_error_correction_shapeless(n, _lib.ptr(sdf, _lib._fp), w, h, w*n, _lib.ptr(xf, _lib._dp), float(min_deviation_ratio), int(protect_all)))
    return sdf


def msdf_fast_distance_error_correction(sdf, transformation, min_deviation_ratio=DEFAULT_MIN_DEVIATION_RATIO):
    """msdfFastDistanceErrorCorrection(sdf, transformation, minDeviationRatio), core/msdf-error-correction.h:21-26: findErrors(sdf) + apply, no shape."""
    return _fast_error_correction(sdf, transformation, min_deviation_ratio, False)


def msdf_fast_edge_error_correction(sdf, transformation, min_deviation_ratio=DEFAULT_MIN_DEVIATION_RATIO):
    """msdfFastEdgeErrorCorrection(sdf, transformation, minDeviationRatio), core/msdf-error-correction.h:29-34: protectAll + findErrors(sdf) + apply."""
    return _fast_error_correction(sdf, transformation, min_deviation_ratio, True)


def distance_sign_correction(sdf, shape, projection, sdf_zero_value=.5, fill_rule=FILL_NONZERO, y_orientation=Y_UPWARD):
    """distanceSignCorrection (core/rasterization.h:15-19), in place on a 1-, 3- or 4-channel bitmap. `projection` is a Projection
    (or anything with one: SDFTransformation)."""
    lib = _lib.load()
    n = sdf.shape[2]
    px, w, h, stride = _bitmap_args(sdf, n)
    keep, sargs = _shape_args(shape)
    proj = getattr(projection, "projection", projection)
    xf = np.array([proj.scale[0], proj.scale[1], proj.translate[0], proj.translate[1], 1, 0], np.float64)
    flip = int(bool(shape.inverse_y) != (y_orientation == Y_DOWNWARD))
    _lib.check(lib.msdfhip_distance_sign_correction(n, px, w, h, stride, flip, *sargs, _lib.ptr(xf, _lib._dp), float(sdf_zero_value), int(fill_rule)))
    del keep
    return sdf


def rasterize(output, shape, projection, fill_rule=FILL_NONZERO, y_orientation=Y_UPWARD):
    """rasterize (core/rasterization.h:13): 1-channel coverage, 1.0 where the texel centre is inside under `fill_rule`."""
    lib = _lib.load()
    px, w, h, stride = _bitmap_args(output, 1)
    keep, sargs = _shape_args(shape)
    proj = getattr(projection, "projection", projection)
    xf = np.array([proj.scale[0], proj.scale[1], proj.translate[0], proj.translate[1], 1, 0], np.float64)
    flip = int(bool(shape.inverse_y) != (y_orientation == Y_DOWNWARD))
    _lib.check(lib.msdfhip_rasterize(px, w, h, stride, flip, *sargs, _lib.ptr(xf, _lib._dp), int(fill_rule)))
    del keep
    return output


def render_sdf(sdf_tiles, out_width, out_height, out_channels, sdf_px_range=(0., 0.), sd_threshold=.5, out=None, stream=None):
    """renderSDF (core/render-sdf.h:12-17) of a device tensor of tiles (G, H, W, N) -> (G, out_height, out_width, out_channels).
    sdf_px_range: Range as (lower, upper), or a width w meaning (-w/2, +w/2); (0, 0) = thresholded rendering."""
    import torch
    lo, hi = (-.5*sdf_px_range, .5*sdf_px_range) if np.isscalar(sdf_px_range) else sdf_px_range
    g, h, w, n = sdf_tiles.shape
    assert sdf_tiles.dtype == torch.float32 and sdf_tiles.is_contiguous()
    if out is None:
        out = torch.empty((g, out_height, out_width, out_channels), dtype=torch.float32, device=sdf_tiles.device)
    s = (stream or torch.cuda.current_stream(sdf_tiles.device)).cuda_stream
    _lib.check(_lib.load().msdfhip_render_sdf(sdf_tiles.data_ptr(), g, w, h, n, out.data_ptr(), out_width, out_height, out_channels, float(lo), float(hi),
                                              float(sd_threshold), s))
    return out


def simulate_8bit(tiles, stream=None):
    """simulate8bit (core/render-sdf.h:20-22), in place on a contiguous float32 device tensor."""
    import torch
    assert tiles.dtype == torch.float32 and tiles.is_contiguous()
    s = (stream or torch.cuda.current_stream(tiles.device)).cuda_stream
    _lib.check(_lib.load().msdfhip_simulate_8bit(tiles.data_ptr(), tiles.numel(), s))
    return tiles


def shape_distance(shape, selector, overlap_support, points):
    """ShapeDistanceFinder<CC<Selector>>::oneShotDistance (core/ShapeDistanceFinder.hpp:36-58) at shape-space points; (n, 4) float64."""
    lib = _lib.load()
    pts = np.ascontiguousarray(points, np.float64).reshape(-1, 2)
    out = np.zeros((len(pts), 4), np.float64)
    keep, sargs = _shape_args(shape)
    _lib.check(lib.msdfhip_shape_distance(int(selector), int(bool(overlap_support)), *sargs, len(pts), _lib.ptr(pts, _lib._dp), _lib.ptr(out, _lib._dp)))
    del keep
    return out


def contour_windings(shape: FlatShape):
    """Contour::winding per contour (core/Contour.cpp:57-81) as computed by the device digestion kernel."""
    lib = _lib.load()
    keep, sargs = _shape_args(shape)
    gco = np.array([0, shape.n_contours], np.int32)
    handle = C.c_void_p()
    _lib.check(lib.msdfhip_batch_create(C.byref(handle), 1, _lib.ptr(gco, _lib._ip), sargs[0], sargs[2], sargs[3], sargs[4]))
    try:
        out = np.zeros(max(shape.n_contours, 1), np.int32)
        _lib.check(lib.msdfhip_batch_windings(handle, _lib.ptr(out, _lib._ip)))
    finally:
        lib.msdfhip_batch_destroy(handle)
    del keep
    return out[:shape.n_contours]


class GlyphBatch:
    """G glyph shapes resident in HBM ("uploaded once"), digested on the device, ready to be rendered into G tiles per launch.

    torch owns the device buffers; the kernels run on torch's current stream unless one is given.
    """

    def __init__(self, shapes: ShapeBatch, device=None):
        import torch
        if not torch.cuda.is_available():
            raise MsdfHipError(_lib.ERR_NO_DEVICE, "no GPU visible to torch; msdfgen_amd has no CPU path")
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device) if not isinstance(device, torch.device) else device
        _lib.init(self.device.index or 0)
        self.shapes = shapes
        self.n_glyphs = shapes.n_glyphs
        dev = self.device
        self._gco = torch.from_numpy(np.ascontiguousarray(shapes.glyph_contour_offsets, np.int32)).to(dev)
        self._co = torch.from_numpy(np.ascontiguousarray(shapes.contour_offsets, np.int32)).to(dev)
        n_e = max(shapes.n_edges, 1)
        pts = np.zeros((n_e, 8), np.float64)
        pts[:shapes.n_edges] = shapes.points
        types = np.ones(n_e, np.uint8)
        types[:shapes.n_edges] = shapes.types
        colors = np.zeros(n_e, np.uint8)
        colors[:shapes.n_edges] = shapes.colors
        self._pts = torch.from_numpy(pts).to(dev)
        self._types = torch.from_numpy(types).to(dev)
        self._colors = torch.from_numpy(colors).to(dev)
        gco, co = shapes.glyph_contour_offsets, shapes.contour_offsets
        self.max_contours = int(np.diff(gco).max()) if self.n_glyphs else 0
        self.max_edges = int((co[gco[1:]]-co[gco[:-1]]).max()) if self.n_glyphs else 0
        self._handle = C.c_void_p()
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.load().msdfhip_batch_create_device(C.byref(self._handle), self.n_glyphs, shapes.n_contours, shapes.n_edges, self.max_contours,
                                                           self.max_edges, self._gco.data_ptr(), self._co.data_ptr(), self._pts.data_ptr(),
                                                           self._types.data_ptr(), self._colors.data_ptr(), stream))
        self._scratch = None
        self._glyph_cache = {}

    @classmethod
    def from_raw(cls, raw: ShapeBatch, normalize=True, coloring=1, angle_threshold=3.0, seeds=None, seed=0, device=None):
        """Uploads RAW outlines and prepares them on the device: Shape::normalize (core/Shape.cpp:65-92) and edgeColoringSimple
        (coloring=1, core/edge-coloring.cpp:68-142) or edgeColoringInkTrap (coloring=2, :151-258) with (angle_threshold, seed; `seeds`: one per glyph). The prepared
        shapes are read back once into `.shapes` (they are small); the batch is digested and ready for generate()."""
        import torch
        if not torch.cuda.is_available():
            raise MsdfHipError(_lib.ERR_NO_DEVICE, "no GPU visible to torch; msdfgen_amd has no CPU path")
        self = object.__new__(cls)
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device) if not isinstance(device, torch.device) else device
        _lib.init(self.device.index or 0)
        lib = _lib.load()
        gco = np.ascontiguousarray(raw.glyph_contour_offsets, np.int32)
        co = np.ascontiguousarray(raw.contour_offsets, np.int32)
        pts = np.ascontiguousarray(raw.points, np.float64).reshape(-1, 8)
        types = np.ascontiguousarray(raw.types, np.uint8)
        colors = np.ascontiguousarray(raw.colors, np.uint8)
        sd = None if seeds is None else np.ascontiguousarray(seeds, np.uint64)
        cfg = _lib.PrepConfig(int(bool(normalize)), int(coloring), float(angle_threshold), int(seed))
        self._handle = C.c_void_p()
        _lib.check(lib.msdfhip_batch_create_prepared(C.byref(self._handle), raw.n_glyphs, _lib.ptr(gco, _lib._ip), _lib.ptr(co, _lib._ip), _lib.ptr(pts, _lib._dp),
                                                     _lib.ptr(types, _lib._bp), _lib.ptr(colors, _lib._bp),
                                                     sd.ctypes.data_as(C.POINTER(C.c_uint64)) if sd is not None else None, C.byref(cfg)))
        ng, nc, ne, mc, me = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.msdfhip_batch_info(self._handle, C.byref(ng), C.byref(nc), C.byref(ne), C.byref(mc), C.byref(me)))
        co2 = np.zeros(nc.value+1, np.int32)
        p2 = np.zeros((max(ne.value, 1), 8), np.float64)
        t2, c2 = np.ones(max(ne.value, 1), np.uint8), np.zeros(max(ne.value, 1), np.uint8)
        _lib.check(lib.msdfhip_batch_download(self._handle, _lib.ptr(co2, _lib._ip), _lib.ptr(p2, _lib._dp), _lib.ptr(t2, _lib._bp), _lib.ptr(c2, _lib._bp)))
        self.shapes = ShapeBatch(gco.copy(), co2, p2[:ne.value], t2[:ne.value].astype(np.int32), c2[:ne.value].astype(np.int32), np.asarray(raw.inverse_y).copy(),
                                 list(raw.names) if raw.names is not None else None)
        self.n_glyphs = raw.n_glyphs
        self.max_contours, self.max_edges = mc.value, me.value
        self._scratch = None
        self._glyph_cache = {}
        return self

    def close(self):
        if getattr(self, "_handle", None) is not None and self._handle:
            self.torch.cuda.synchronize(self.device)
            _lib.load().msdfhip_batch_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def digest(self, stream=None):
        """Re-runs the on-device digestion of the edge buffer (records + windings); part of every fresh upload."""
        s = (stream or self.torch.cuda.current_stream(self.device)).cuda_stream
        _lib.check(_lib.load().msdfhip_batch_digest(self._handle, s))

    def windings(self):
        out = np.zeros(max(self.shapes.n_contours, 1), np.int32)
        _lib.check(_lib.load().msdfhip_batch_windings(self._handle, _lib.ptr(out, _lib._ip)))
        return out[:self.shapes.n_contours]

    def descriptors(self, xfs, width, height, channels, y_orientation=Y_UPWARD, out_offsets=None, row_stride=None):
        """Device array of MsdfHipGlyph for contiguous tiles [g][h][w][N] (or custom atlas placement via out_offsets/row_stride).
        xfs: (G, 6) rows (sx, sy, tx, ty, range_lower, range_upper)."""
        xfs = np.ascontiguousarray(xfs, np.float64).reshape(self.n_glyphs, 6)
        d = np.zeros(self.n_glyphs, _lib.GLYPH_DTYPE)
        d["xf"][:, :4] = xfs[:, :4]
        d["xf"][:, 4] = np.float64(1)/(xfs[:, 5]-xfs[:, 4])  # DistanceMapping.cpp:13
        d["xf"][:, 5] = -xfs[:, 4]
        tile = width*height*channels
        d["out_offset"] = np.arange(self.n_glyphs, dtype=np.int64)*tile if out_offsets is None else np.asarray(out_offsets, np.int64)
        d["row_stride"] = width*channels if row_stride is None else row_stride
        d["flip"] = (self.shapes.inverse_y.astype(bool) != (y_orientation == Y_DOWNWARD)).astype(np.int32)
        return self.torch.from_numpy(d.view(np.uint8).reshape(self.n_glyphs, 64)).to(self.device)

    def estimate_sdf_error(self, tiles, xfs, scanlines_per_row=1, fill_rule=FILL_NONZERO, stream=None):
        """estimateSDFError (core/sdf-error-estimation.h:18-20) of every tile (G, H, W, N) of this batch's shapes; returns a float64
        device tensor (G,). xfs: the (G, 6) rows the tiles were generated with (only the projection part is read)."""
        torch = self.torch
        g, h, w, n = tiles.shape
        assert g == self.n_glyphs and tiles.dtype == torch.float32 and tiles.is_contiguous()
        d = np.zeros(g, _lib.GLYPH_DTYPE)
        d["xf"][:, :4] = np.ascontiguousarray(xfs, np.float64).reshape(g, 6)[:, :4]
        d["flip"] = self.shapes.inverse_y.astype(bool).astype(np.int32)      # shape.getYAxisOrientation() == Y_DOWNWARD
        desc = torch.from_numpy(d.view(np.uint8).reshape(g, 64)).to(self.device)
        out = torch.empty(g, dtype=torch.float64, device=self.device)
        s = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        _lib.check(_lib.load().msdfhip_batch_estimate_sdf_error(self._handle, n, w, h, desc.data_ptr(), tiles.data_ptr(), int(scanlines_per_row), int(fill_rule),
                                                                out.data_ptr(), s))
        return out

    def candidate_counts(self):
        """Diagnostics of the last error-correction pass: (overflow flag, per-glyph number of deferred distance checks)."""
        out = np.zeros(self.n_glyphs+1, np.uint32)
        _lib.check(_lib.load().msdfhip_batch_candidate_counts(self._handle, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return bool(out[0]), out[1:]

    def to_bytes(self, tiles, atlas, out_offsets, row_stride, stream=None):
        """pixelFloatToByte (core/pixel-conversion.hpp:8-10) of packed float tiles (G, H, W, N) + blit of glyph g's rectangle into the
        uint8 device tensor `atlas` at byte offset out_offsets[g] with `row_stride` bytes per atlas row. Returns `atlas`."""
        torch = self.torch
        g, h, w, n = tiles.shape
        assert g == self.n_glyphs and tiles.dtype == torch.float32 and tiles.is_contiguous() and atlas.dtype == torch.uint8
        d = np.zeros(g, _lib.GLYPH_DTYPE)
        d["out_offset"] = np.asarray(out_offsets, np.int64)
        d["row_stride"] = row_stride
        hi = int(d["out_offset"].max())+int(row_stride)*(h-1)+w*n if g else 0
        if g and (int(d["out_offset"].min()) < 0 or hi > atlas.numel()):
            raise ValueError("a glyph rectangle lies outside the atlas")
        desc = torch.from_numpy(d.view(np.uint8).reshape(g, 64)).to(self.device)
        s = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        _lib.check(_lib.load().msdfhip_tiles_to_bytes(tiles.data_ptr(), g, w, h, n, desc.data_ptr(), atlas.data_ptr(), s))
        return atlas

    def generate(self, mode, width, height, xfs=None, config=None, out=None, stencil=None, descriptors=None, stream=None, y_orientation=Y_UPWARD,
                 scanline_pass=False, fill_rule=FILL_NONZERO, sdf_zero_value=.5):
        """Renders every glyph of the batch into its tile; returns the float32 device tensor (G, height, width, N).
        Asynchronous on `stream` (torch.cuda.Stream) or torch's current stream.
        scanline_pass: run distanceSignCorrection(field, shape, projection, sdf_zero_value, fill_rule) between distance generation
        and error correction, the order of the reference's -scanline flow (main.cpp:1281-1298) -- the field stays on the device.
        The caller picks the configs that flow uses (overlap_support False, distance_check_mode DO_NOT_CHECK_DISTANCE)."""
        torch = self.torch
        n = CHANNELS[mode]
        if descriptors is None:
            descriptors = self.descriptors(xfs, width, height, n, y_orientation)
        if out is None:
            out = torch.empty((self.n_glyphs, height, width, n), dtype=torch.float32, device=self.device)
        cfg = _c_config(config if config is not None else (MSDFGeneratorConfig() if mode >= 3 else GeneratorConfig()), y_orientation)
        _with_scanline_pass(cfg, scanline_pass, fill_rule, sdf_zero_value)
        scratch_ptr = None
        stages = int(mode >= 3 and cfg.ec_mode != EC_DISABLED)+int(bool(scanline_pass))   # intermediate fields, msdfgen_hip.h
        if stages:
            need = stages*self.n_glyphs*height*width*n
            if self._scratch is None or self._scratch.numel() < need:
                self._scratch = torch.empty(need, dtype=torch.float32, device=self.device)
            scratch_ptr = self._scratch.data_ptr()
        s = (stream or torch.cuda.current_stream(self.device)).cuda_stream
        _lib.check(_lib.load().msdfhip_batch_generate(self._handle, mode, width, height, descriptors.data_ptr(), out.data_ptr(),
                                                      stencil.data_ptr() if stencil is not None else None, scratch_ptr, C.byref(cfg), s))
        return out


# ------------------------------------------------------------------------------------------------- end to end, host memory


def host_alloc(shape, dtype=np.float32):
    """Pinned (page-locked) host array for the outputs of generate_host / generate_bytes_host / generate_sharded
    (msdfhip_host_alloc): device-to-host copies into it run at full PCIe speed and truly asynchronously."""
    n = int(np.prod(shape))*np.dtype(dtype).itemsize
    p = C.c_void_p()
    _lib.check(_lib.load().msdfhip_host_alloc(C.byref(p), n))
    buf = (C.c_char*max(n, 1)).from_address(p.value)
    a = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    _PINNED[a.__array_interface__["data"][0]] = p.value
    return a


_PINNED = {}


def host_free(a):
    p = _PINNED.pop(a.__array_interface__["data"][0], None)
    if p is not None:
        _lib.check(_lib.load().msdfhip_host_free(p))


def _descriptors_host(shapes: ShapeBatch, xfs, out_offsets, row_stride, y_orientation=Y_UPWARD):
    xfs = np.ascontiguousarray(xfs, np.float64).reshape(shapes.n_glyphs, 6)
    d = np.zeros(shapes.n_glyphs, _lib.GLYPH_DTYPE)
    d["xf"][:, :4] = xfs[:, :4]
    d["xf"][:, 4] = np.float64(1)/(xfs[:, 5]-xfs[:, 4])  # DistanceMapping.cpp:13
    d["xf"][:, 5] = -xfs[:, 4]
    d["out_offset"] = np.asarray(out_offsets, np.int64)
    d["row_stride"] = row_stride
    d["flip"] = (np.asarray(shapes.inverse_y).astype(bool) != (y_orientation == Y_DOWNWARD)).astype(np.int32)
    return d


class HostBatch:
    """The batch API without torch: shapes uploaded from host arrays (msdfhip_batch_create_on), tiles delivered into host memory
    by the chunked two-stream pipeline (msdfhip_batch_generate_host / _bytes_host). This is the end-to-end path of SURVEY.md 8(d):
    host flatten -> H2D -> kernels incl. error correction -> D2H into caller-owned bitmaps."""

    def __init__(self, shapes: ShapeBatch, device=-1):
        self.shapes = shapes
        self.n_glyphs = shapes.n_glyphs
        gco = np.ascontiguousarray(shapes.glyph_contour_offsets, np.int32)
        co = np.ascontiguousarray(shapes.contour_offsets, np.int32)
        pts = np.ascontiguousarray(shapes.points, np.float64).reshape(-1, 8)
        types = np.ascontiguousarray(shapes.types, np.uint8)
        colors = np.ascontiguousarray(shapes.colors, np.uint8)
        self._handle = C.c_void_p()
        _lib.check(_lib.load().msdfhip_batch_create_on(C.byref(self._handle), int(device), shapes.n_glyphs, _lib.ptr(gco, _lib._ip), _lib.ptr(co, _lib._ip),
                                                       _lib.ptr(pts, _lib._dp), _lib.ptr(types, _lib._bp), _lib.ptr(colors, _lib._bp)))

    def close(self):
        if getattr(self, "_handle", None):
            _lib.load().msdfhip_batch_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device(self):
        d = C.c_int()
        _lib.check(_lib.load().msdfhip_batch_device(self._handle, C.byref(d)))
        return d.value

    def generate_host(self, mode, width, height, xfs, out=None, config=None, stencil=None, out_offsets=None, row_stride=None, y_orientation=Y_UPWARD):
        """Float tiles into `out` (host float32; default: a fresh packed (G, height, width, N) array). out_offsets / row_stride (floats)
        place the tiles elsewhere, e.g. as rectangles of one atlas bitmap (BitmapSection semantics)."""
        n = CHANNELS[mode]
        tile = width*height*n
        if out is None:
            out = np.zeros((self.n_glyphs, height, width, n), np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous
        d = _descriptors_host(self.shapes, xfs, np.arange(self.n_glyphs, dtype=np.int64)*tile if out_offsets is None else out_offsets,
                              width*n if row_stride is None else row_stride, y_orientation)
        cfg = _c_config(config if config is not None else (MSDFGeneratorConfig() if mode >= 3 else GeneratorConfig()), y_orientation)
        if stencil is not None:
            assert stencil.dtype == np.uint8 and stencil.flags.c_contiguous and stencil.size >= self.n_glyphs*width*height
        _lib.check(_lib.load().msdfhip_batch_generate_host(self._handle, mode, width, height, d.ctypes.data, out.ctypes.data, out.size,
                                                           stencil.ctypes.data if stencil is not None else None, C.byref(cfg)))
        return out

    def generate_bytes_host(self, mode, width, height, xfs, atlas, out_offsets, row_stride, config=None, y_orientation=Y_UPWARD):
        """8-bit output (pixelFloatToByte, core/pixel-conversion.hpp:8-10) blitted into the host uint8 `atlas`; offsets / stride in bytes."""
        assert atlas.dtype == np.uint8 and atlas.flags.c_contiguous
        d = _descriptors_host(self.shapes, xfs, out_offsets, row_stride, y_orientation)
        cfg = _c_config(config if config is not None else (MSDFGeneratorConfig() if mode >= 3 else GeneratorConfig()), y_orientation)
        _lib.check(_lib.load().msdfhip_batch_generate_bytes_host(self._handle, mode, width, height, d.ctypes.data, atlas.ctypes.data, atlas.size, C.byref(cfg)))
        return atlas


def generate_sharded(devices, shapes: ShapeBatch, mode, width, height, xfs, out=None, atlas=None, out_offsets=None, row_stride=None, config=None,
                     y_orientation=Y_UPWARD):
    """msdfhip_generate_sharded: the glyph list split over `devices` (one host thread + streams per entry, no exchange between devices),
    every device writing its tiles into the caller's `out` (float32) or `atlas` (uint8)."""
    n = CHANNELS[mode]
    tile = width*height*n
    if out is None and atlas is None:
        out = np.zeros((shapes.n_glyphs, height, width, n), np.float32)
    d = _descriptors_host(shapes, xfs, np.arange(shapes.n_glyphs, dtype=np.int64)*tile if out_offsets is None else out_offsets,
                          width*n if row_stride is None else row_stride, y_orientation)
    cfg = _c_config(config if config is not None else (MSDFGeneratorConfig() if mode >= 3 else GeneratorConfig()), y_orientation)
    gco = np.ascontiguousarray(shapes.glyph_contour_offsets, np.int32)
    co = np.ascontiguousarray(shapes.contour_offsets, np.int32)
    pts = np.ascontiguousarray(shapes.points, np.float64).reshape(-1, 8)
    types = np.ascontiguousarray(shapes.types, np.uint8)
    colors = np.ascontiguousarray(shapes.colors, np.uint8)
    devs = (C.c_int*len(devices))(*[int(v) for v in devices])
    _lib.check(_lib.load().msdfhip_generate_sharded(devs, len(devices), mode, width, height, shapes.n_glyphs, _lib.ptr(gco, _lib._ip), _lib.ptr(co, _lib._ip),
                                                    _lib.ptr(pts, _lib._dp), _lib.ptr(types, _lib._bp), _lib.ptr(colors, _lib._bp), d.ctypes.data,
                                                    out.ctypes.data if out is not None else None, out.size if out is not None else 0,
                                                    atlas.ctypes.data if atlas is not None else None, atlas.size if atlas is not None else 0, C.byref(cfg)))
    return out if out is not None else atlas


def generate_stream(shapes: ShapeBatch, mode, width, height, xfs, out=None, atlas=None, out_offsets=None, row_stride=None, config=None, stencil=None,
                    y_orientation=Y_UPWARD, device=-1):
    """msdfhip_generate_stream_csr: host CSR arrays in, host tiles (float32 `out`) or an 8-bit `atlas` out, as ONE pipelined call -- the glyph list is cut
    into chunks and chunk k+1's staging + upload + digest run under chunk k's kernels and chunk k-1's copy back (SURVEY.md 8d's end-to-end metric;
    msdfgen_hip::generate*Batch() of the C++ shim is the same pipeline fed from msdfgen::Shape objects)."""
    n = CHANNELS[mode]
    tile = width*height*n
    if out is None and atlas is None:
        out = np.zeros((shapes.n_glyphs, height, width, n), np.float32)
    d = _descriptors_host(shapes, xfs, np.arange(shapes.n_glyphs, dtype=np.int64)*tile if out_offsets is None else out_offsets,
                          width*n if row_stride is None else row_stride, y_orientation)
    cfg = _c_config(config if config is not None else (MSDFGeneratorConfig() if mode >= 3 else GeneratorConfig()), y_orientation)
    gco = np.ascontiguousarray(shapes.glyph_contour_offsets, np.int32)
    co = np.ascontiguousarray(shapes.contour_offsets, np.int32)
    pts = np.ascontiguousarray(shapes.points, np.float64).reshape(-1, 8)
    types = np.ascontiguousarray(shapes.types, np.uint8)
    colors = np.ascontiguousarray(shapes.colors, np.uint8)
    if stencil is not None:
        assert stencil.dtype == np.uint8 and stencil.flags.c_contiguous and stencil.size >= shapes.n_glyphs*width*height
    _lib.check(_lib.load().msdfhip_generate_stream_csr(int(device), mode, width, height, shapes.n_glyphs, _lib.ptr(gco, _lib._ip), _lib.ptr(co, _lib._ip),
                                                       _lib.ptr(pts, _lib._dp), _lib.ptr(types, _lib._bp), _lib.ptr(colors, _lib._bp), d.ctypes.data,
                                                       out.ctypes.data if out is not None else None, out.size if out is not None else 0,
                                                       atlas.ctypes.data if atlas is not None else None, atlas.size if atlas is not None else 0,
                                                       stencil.ctypes.data if stencil is not None else None, C.byref(cfg)))
    return out if out is not None else atlas
