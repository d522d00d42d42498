"""C-ABI surface and host-side logic (no GPU compute): the library builds/loads, exports every symbol include/msdfgen_hip.h
declares, has the reference's defaults, and FAILS LOUDLY (no CPU fallback) when no gfx950 device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import msdfgen_amd as M
from msdfgen_amd import lib as L
from msdfgen_amd.shape import FlatShape, ShapeBatch, autoframe, distance_mapping
from msdfgen_amd.shard import partition_contiguous, glyph_costs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "msdfgen_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(msdfhip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libmsdfgen_hip.so does not export "+s
    assert set(syms) == set(L.EXPORTED_SYMBOLS), set(syms) ^ set(L.EXPORTED_SYMBOLS)
    assert lib.msdfhip_abi_version() == 5


def test_default_config_matches_reference_defaults():
    cfg = L.default_config()
    assert (cfg.overlap_support, cfg.ec_mode, cfg.ec_distance_check, cfg.ec_stage_limit) == (1, L.EC_EDGE_PRIORITY, L.CHECK_DISTANCE_AT_EDGE, 0)
    assert cfg.min_deviation_ratio == 1.11111111111111111 and cfg.min_improve_ratio == 1.11111111111111111
    assert C.sizeof(L.Config) == 48 and C.sizeof(L.Glyph) == 64


def test_no_device_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    s = FlatShape.from_contours([[(3, (0, 0), (1, 0)), (5, (1, 0), (0, 1)), (6, (0, 1), (0, 0))]])
    out = np.zeros((8, 8, 3), np.float32)
    with pytest.raises(M.MsdfHipError) as e:
        M.generate_msdf(out, s, M.SDFTransformation.from_xf(autoframe(s.bounds(), 8, 8, 2)))
    assert e.value.code == L.ERR_NO_DEVICE
    assert not out.any()
    with pytest.raises(M.MsdfHipError):
        M.shape_distance(s, 1, True, [[.5, .5]])


def test_argument_validation_without_device():
    lib = L.load()
    cfg = L.default_config()
    assert lib.msdfhip_generate(7, None, 4, 4, 4, 0, None, 0, None, None, None, None, C.byref(cfg), None) == L.ERR_INVALID
    assert b"mode" in lib.msdfhip_last_error()
    assert lib.msdfhip_error_correction(2, None, 4, 4, 4, 0, None, 0, None, None, None, None, C.byref(cfg), None) == L.ERR_INVALID


def test_flat_shape_and_batch_roundtrip(tmp_path, latin):
    batch, xf64, bounds = latin
    assert batch.n_glyphs == 94 and batch.n_edges == 1463
    sub = batch.select([3, 10, 50])
    assert sub.n_glyphs == 3 and sub.shape(1).n_edges == batch.shape(10).n_edges
    assert np.array_equal(sub.shape(2).points, batch.shape(50).points)
    p = str(tmp_path/"b.npz")
    sub.save(p)
    again = ShapeBatch.load(p)
    assert np.array_equal(again.points, sub.points) and np.array_equal(again.contour_offsets, sub.contour_offsets) and again.names == sub.names
    with pytest.raises(ValueError):
        FlatShape([0, 2], np.zeros((1, 8)), [1], [7])
    with pytest.raises(ValueError):
        FlatShape([0, 1], np.zeros((1, 8)), [4], [7])


def test_autoframe_follows_cli_rule():
    xf = autoframe((0., 0., 8., 10.), 32, 32, 4)  # the reference CLI prints scale = 2.8 for shape 'A' at -autoframe -pxrange 4 (28/10)
    assert xf[0] == xf[1] == 2.8
    assert xf[4] == -2/2.8 and xf[5] == 2/2.8
    s, t = distance_mapping(xf[4], xf[5])
    assert s == 1/(xf[5]-xf[4]) and t == -xf[4]
    with pytest.raises(ValueError):
        autoframe((0, 0, 1, 1), 4, 4, 4)


def test_partition_is_contiguous_balanced_and_deterministic(latin):
    batch, _, _ = latin
    costs = glyph_costs(batch, 64, 64)
    for parts in (1, 2, 4, 8):
        b = partition_contiguous(costs, parts)
        assert b[0] == 0 and b[-1] == batch.n_glyphs and np.all(np.diff(b) >= 0) and len(b) == parts+1
        sums = np.array([costs[b[i]:b[i+1]].sum() for i in range(parts)])
        assert sums.max() <= costs.sum()/parts+costs.max()
    assert np.array_equal(partition_contiguous([1, 1, 1, 1], 2), [0, 2, 4])
    assert np.array_equal(partition_contiguous([], 3), [0, 0, 0, 0])


def test_shim_status_mode_without_a_device():
    """The C++ shim's non-throwing mode: without a device every generator fails; with msdfgen_hip_shim_set_nothrow(1)
    the failure is a per-thread status instead of a std::runtime_error thrown out of the caller's thread. Runs in a child process
    (the default mode would terminate a ctypes caller)."""
    import subprocess
    import sys
    import textwrap
    from msdfgen_amd import build as B
    shim = B.build_shim()
    if shim is None:
        pytest.skip("msdfgen headers not available: shim not built")
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    from oracle.pyoracle import REF_SO
    if not os.path.exists(REF_SO):
        pytest.skip("the shim links against msdfgen's other objects (here: oracle/_ref), not present")
    code = textwrap.dedent("""
        import ctypes as C, sys
        C.CDLL(%r, mode=C.RTLD_GLOBAL)                       # msdfgen's untouched parts (Projection, DistanceMapping, ...)
        C.CDLL(%r, mode=C.RTLD_GLOBAL)
        shim = C.CDLL(%r)
        shim.msdfgen_hip_shim_last_error.restype = C.c_char_p
        assert shim.msdfgen_hip_shim_last_status() == 0
        shim.msdfgen_hip_shim_set_nothrow(1)
        # renderSDF(BitmapSection<float,1>, BitmapConstSection<float,1>, Range, float): plain structs of pointer + 4 ints, passed by reference
        class Section(C.Structure):
            _fields_ = [("pixels", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("rowStride", C.c_int), ("yOrientation", C.c_int)]
        buf = (C.c_float*16)()
        a, b = Section(C.addressof(buf), 4, 4, 4, 0), Section(C.addressof(buf), 4, 4, 4, 0)
        class Range(C.Structure):
            _fields_ = [("lower", C.c_double), ("upper", C.c_double)]
        fn = getattr(shim, "_ZN7msdfgen9renderSDFERKNS_13BitmapSectionIfLi1EEERKNS_18BitmapConstSectionIfLi1EEENS_5RangeEf")
        fn.argtypes = [C.POINTER(Section), C.POINTER(Section), Range, C.c_float]
        fn(C.byref(a), C.byref(b), Range(-1, 1), C.c_float(.5))
        assert shim.msdfgen_hip_shim_last_status() == -1, shim.msdfgen_hip_shim_last_status()      # MSDFHIP_ERR_NO_DEVICE
        assert b"no HIP device" in shim.msdfgen_hip_shim_last_error()
        print("ok")
    """) % (REF_SO, B.LIB, shim)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "ok" in p.stdout, p.stderr[-1500:]
