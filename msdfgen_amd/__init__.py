"""msdfgen_amd -- MI355X-native (gfx950, hand-written HIP) implementation of msdfgen's per-texel signed-distance hot path:
generateSDF / generatePSDF / generateMSDF / generateMTSDF and the MSDF error-correction pass, behind a C ABI
(include/msdfgen_hip.h) and this thin host-side mirror of the reference's interface.  There is no CPU compute path."""
from .shape import FlatShape, ShapeBatch, autoframe, distance_mapping  # noqa: F401
from .lib import MsdfHipError, load, init, device_info, default_config, set_microbatch, microbatch_stats  # noqa: F401
from .api import (  # noqa: F401
    Projection, Range, DistanceMapping, SDFTransformation, ErrorCorrectionConfig, GeneratorConfig, MSDFGeneratorConfig,
    generate_sdf, generate_psdf, generate_msdf, generate_mtsdf, msdf_error_correction, msdf_fast_distance_error_correction, msdf_fast_edge_error_correction, distance_sign_correction, rasterize, render_sdf, simulate_8bit, shape_distance, contour_windings, GlyphBatch,
    HostBatch, generate_sharded, generate_stream, host_alloc, host_free,
    MODE_SDF, MODE_PSDF, MODE_MSDF, MODE_MTSDF, CHANNELS, EC_DISABLED, EC_INDISCRIMINATE, EC_EDGE_PRIORITY, EC_EDGE_ONLY,
    DO_NOT_CHECK_DISTANCE, CHECK_DISTANCE_AT_EDGE, ALWAYS_CHECK_DISTANCE, Y_UPWARD, Y_DOWNWARD,
    FILL_NONZERO, FILL_ODD, FILL_POSITIVE, FILL_NEGATIVE)

__version__ = "0.3.0"
