"""ctypes binding of the C ABI (include/msdfgen_hip.h).  Loading never falls back to anything: if libmsdfgen_hip.so is
missing it is built with hipcc, and if no gfx950 device is usable every compute entry point raises MsdfHipError."""
import ctypes as C
import os
import threading

import numpy as np

from . import build as _build

MODE_SDF, MODE_PSDF, MODE_MSDF, MODE_MTSDF = 1, 2, 3, 4
CHANNELS = {1: 1, 2: 1, 3: 3, 4: 4}
EC_DISABLED, EC_INDISCRIMINATE, EC_EDGE_PRIORITY, EC_EDGE_ONLY = 0, 1, 2, 3
DO_NOT_CHECK_DISTANCE, CHECK_DISTANCE_AT_EDGE, ALWAYS_CHECK_DISTANCE = 0, 1, 2
FILL_NONZERO, FILL_ODD, FILL_POSITIVE, FILL_NEGATIVE = 0, 1, 2, 3
ERR_NO_DEVICE, ERR_INVALID, ERR_HIP, ERR_TOO_COMPLEX, ERR_NOMEM = -1, -2, -3, -4, -5

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_fp = C.POINTER(C.c_float)
_bp = C.POINTER(C.c_uint8)
_vp = C.c_void_p


class MsdfHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("msdfgen_hip error %d: %s" % (code, message))
        self.code = code


class Config(C.Structure):
    """MsdfHipConfig == MSDFGeneratorConfig + ErrorCorrectionConfig (core/generator-config.h:13-64)."""
    _fields_ = [("overlap_support", C.c_int32), ("ec_mode", C.c_int32), ("ec_distance_check", C.c_int32), ("ec_stage_limit", C.c_int32),
                ("min_deviation_ratio", C.c_double), ("min_improve_ratio", C.c_double),
                ("sign_correction", C.c_int32), ("fill_rule", C.c_int32), ("sdf_zero_value", C.c_float), ("stencil_y_down", C.c_int32)]


class PrepConfig(C.Structure):
    """MsdfHipPrepConfig."""
    _fields_ = [("normalize", C.c_int32), ("coloring", C.c_int32), ("angle_threshold", C.c_double), ("seed", C.c_uint64)]


class Glyph(C.Structure):
    """MsdfHipGlyph."""
    _fields_ = [("xf", C.c_double*6), ("out_offset", C.c_int64), ("row_stride", C.c_int32), ("flip", C.c_int32)]


GLYPH_DTYPE = np.dtype([("xf", np.float64, 6), ("out_offset", np.int64), ("row_stride", np.int32), ("flip", np.int32)])
assert GLYPH_DTYPE.itemsize == C.sizeof(Glyph) == 64

_SHAPE_ARGS = [_ip, C.c_int, _dp, _bp, _bp]

_PROTOS = {
    "msdfhip_default_config": (None, [C.POINTER(Config)]),
    "msdfhip_abi_version": (C.c_int, []),
    "msdfhip_init": (C.c_int, [C.c_int]),
    "msdfhip_last_error": (C.c_char_p, []),
    "msdfhip_device_info": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "msdfhip_generate": (C.c_int, [C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int] + _SHAPE_ARGS + [_dp, C.POINTER(Config), _bp]),
    "msdfhip_generate_sdf": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int] + _SHAPE_ARGS + [_dp, C.POINTER(Config)]),
    "msdfhip_generate_psdf": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int] + _SHAPE_ARGS + [_dp, C.POINTER(Config)]),
    "msdfhip_generate_msdf": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int] + _SHAPE_ARGS + [_dp, C.POINTER(Config), _bp]),
    "msdfhip_generate_mtsdf": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int] + _SHAPE_ARGS + [_dp, C.POINTER(Config), _bp]),
    "msdfhip_error_correction": (C.c_int, [C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int] + _SHAPE_ARGS + [_dp, C.POINTER(Config), _bp]),
    "msdfhip_distance_sign_correction": (C.c_int, [C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int] + _SHAPE_ARGS + [_dp, C.c_float, C.c_int]),
    "msdfhip_rasterize": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int] + _SHAPE_ARGS + [_dp, C.c_int]),
    "msdfhip_set_microbatch": (C.c_int, [C.c_int, C.c_int]),
    "msdfhip_microbatch_stats": (C.c_int, [C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int]),
    "msdfhip_microbatch_times": (C.c_int, [_dp, _dp, _dp, C.c_int]),
    "msdfhip_shape_distance": (C.c_int, [C.c_int, C.c_int] + _SHAPE_ARGS + [C.c_int, _dp, _dp]),
    "msdfhip_batch_create": (C.c_int, [C.POINTER(_vp), C.c_int, _ip, _ip, _dp, _bp, _bp]),
    "msdfhip_batch_create_device": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "msdfhip_batch_create_prepared": (C.c_int, [C.POINTER(_vp), C.c_int, _ip, _ip, _dp, _bp, _bp, C.POINTER(C.c_uint64), C.POINTER(PrepConfig)]),
    "msdfhip_batch_candidate_counts": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "msdfhip_batch_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "msdfhip_batch_download": (C.c_int, [_vp, _ip, _dp, _bp, _bp]),
    "msdfhip_batch_digest": (C.c_int, [_vp, _vp]),
    "msdfhip_batch_destroy": (None, [_vp]),
    "msdfhip_batch_windings": (C.c_int, [_vp, _ip]),
    "msdfhip_batch_generate": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(Config), _vp]),
    "msdfhip_batch_generate_host": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp, C.POINTER(Config)]),
    "msdfhip_batch_generate_bytes_host": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, C.POINTER(Config)]),
    "msdfhip_hw_queues_env": (C.c_int, []),
    "msdfhip_pipeline_overflow_reruns": (C.c_ulonglong, [C.c_int]),
    "msdfhip_set_pipeline_chunk": (C.c_int, [C.c_int]),
    "msdfhip_host_alloc": (C.c_int, [C.POINTER(_vp), C.c_size_t]),
    "msdfhip_host_free": (C.c_int, [_vp]),
    "msdfhip_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "msdfhip_batch_create_on": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, _ip, _ip, _dp, _bp, _bp]),
    "msdfhip_batch_device": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "msdfhip_generate_sharded": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp, _bp, _bp, _vp, _vp, C.c_size_t, _vp, C.c_size_t,
                                           C.POINTER(Config)]),
    "msdfhip_tiles_to_bytes": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "msdfhip_batch_estimate_sdf_error": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "msdfhip_render_sdf": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_float, _vp]),
    "msdfhip_simulate_8bit": (C.c_int, [_vp, C.c_size_t, _vp]),
    "msdfhip_render_sdf_host": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_float]),
    "msdfhip_simulate_8bit_host": (C.c_int, [_fp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "msdfhip_set_kernel_timing": (C.c_int, [C.c_int]),
    "msdfhip_kernel_timing": (C.c_int, [_dp, _dp, C.POINTER(C.c_int), C.c_int]),
    "msdfhip_error_correction_shapeless": (C.c_int, [C.c_int, _fp, C.c_int, C.c_int, C.c_int, _dp, C.c_double, C.c_int]),
    "msdfhip_reload_tuning": (C.c_int, []),
    "msdfhip_trim": (C.c_int, []),
    "msdfhip_front_door_devices": (C.c_int, [C.POINTER(C.c_int), C.c_int]),
    "msdfhip_debug_wait_profile": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int]),
    "msdfhip_debug_single_call_phases": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "msdfhip_debug_bbcount": (C.c_int, [C.POINTER(C.c_uint32), C.c_int, C.c_int]),
    "msdfhip_generate_stream": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.POINTER(Config)]),
    "msdfhip_generate_stream_csr": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, _ip, _dp, _bp, _bp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp,
                                              C.POINTER(Config)]),
    "msdfhip_set_host_threads": (C.c_int, [C.c_int]),
    "msdfhip_single_call_fallbacks": (C.c_int, [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.c_int]),
}

EXPORTED_SYMBOLS = tuple(sorted(_PROTOS))

_lib = None
_lock = threading.Lock()


def library_path():
    return _build.LIB


def load(build_if_missing=True):
    """Returns the loaded C ABI library (ctypes.CDLL). Builds it with hipcc first if it does not exist."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        try:
            # torch wheels bundle their own libamdhip64; if our library pulls in /opt/rocm's first, a later `import torch` no longer
            # sees the GPU (two HIP runtimes in one process). Importing torch first makes both share torch's runtime.
            import torch  # noqa: F401
        except ImportError:
            pass
        # The host-output pipeline wants its three chunk streams on distinct hardware queues ("environment"). The variable is the
        # HOST's to set -- this binding is the host of the tests and the bench -- and is read when the HIP runtime starts (first HIP call).
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        path = os.environ.get("MSDFGEN_HIP_LIB") or _build.LIB   # override: A/B-ing kernel build variants from bench.py
        if path == _build.LIB and not os.path.exists(path):
            if not build_if_missing:
                raise MsdfHipError(ERR_NO_DEVICE, "libmsdfgen_hip.so is not built (python -m msdfgen_amd.build)")
            _build.build_lib()
        lib = C.CDLL(path)
        for name, (restype, argtypes) in _PROTOS.items():
            fn = getattr(lib, name)  # AttributeError here == a symbol of include/msdfgen_hip.h is missing from the library
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.msdfhip_abi_version() != 5:
            raise MsdfHipError(ERR_INVALID, "ABI version mismatch")
        _lib = lib
        return lib


def check(rc):
    if rc != 0:
        raise MsdfHipError(rc, load().msdfhip_last_error().decode(errors="replace"))


def default_config(**overrides):
    cfg = Config()
    load().msdfhip_default_config(C.byref(cfg))
    for k, v in overrides.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


def init(device=0):
    check(load().msdfhip_init(int(device)))


def device_info():
    name = C.create_string_buffer(128)
    cus, lds = C.c_int(), C.c_int()
    check(load().msdfhip_device_info(name, 128, C.byref(cus), C.byref(lds)))
    return {"arch": name.value.decode(), "cus": cus.value, "lds_bytes": lds.value}


def ptr(a, t):
    return a.ctypes.data_as(t)


def set_microbatch(max_group=256, max_leaders=2):
    """Micro-batching of concurrent single-shape calls (msdfgen_hip.h). max_group <= 1 disables it."""
    check(load().msdfhip_set_microbatch(int(max_group), int(max_leaders)))


def microbatch_stats(reset=False):
    a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
    x, y, z = C.c_double(), C.c_double(), C.c_double()
    check(load().msdfhip_microbatch_times(C.byref(x), C.byref(y), C.byref(z), int(reset)))
    check(load().msdfhip_microbatch_stats(C.byref(a), C.byref(b), C.byref(c), int(reset)))
    return {"calls": a.value, "batches": b.value, "largest": c.value, "stage_ms": round(x.value, 3), "device_ms": round(y.value, 3), "scatter_ms": round(z.value, 3)}
