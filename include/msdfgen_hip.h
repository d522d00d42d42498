/*
 * msdfgen_hip.h -- C ABI of the MI355X-native MSDF hot path (libmsdfgen_hip.so).
 *
 * Drop-in boundary for the per-texel signed-distance path of Chlumsky/msdfgen v1.13.0.  Each entry point names the
 * reference interface it replaces (file:line relative to the reference tree).  Plain pointers and sizes only.
 *
 * Shape encoding (the reference's `const Shape &`, core/Shape.h:15-58, flattened once by the caller/binding):
 *   contour_offsets int32[n_contours+1]  CSR offsets into the edge arrays; edges of a contour in Shape order
 *   points          double[n_edges*8]    p0x,p0y,p1x,p1y,p2x,p2y,p3x,p3y (unused slots ignored)
 *   types           uint8[n_edges]       1 linear, 2 quadratic, 3 cubic       (EDGE_TYPE, core/edge-segments.h:62,91,122)
 *   colors          uint8[n_edges]       EdgeColor bitmask R=1 G=2 B=4        (core/EdgeColor.h:9-18)
 *
 * Transformation (the reference's `const SDFTransformation &`, core/SDFTransformation.h:13-24):
 *   xf[6] = { Projection.scale.x, .scale.y, .translate.x, .translate.y,      (core/Projection.h:31-33)
 *             DistanceMapping.scale, DistanceMapping.translate }             (core/DistanceMapping.h:28-30)
 *   i.e. texel centre (x+.5, y+.5) -> shape point  coord/scale - translate  (core/Projection.cpp:14-16)
 *        distance d -> float(mapScale*(d+mapTranslate))                      (core/DistanceMapping.cpp:15-17)
 *
 * Output bitmap (the reference's `const BitmapSection<float, N> &`, core/BitmapRef.hpp:74-111):
 *   pixels, width, height, row_stride (in floats, may be negative), channel-interleaved.
 *   `flip` != 0 when shape.getYAxisOrientation() != bitmap.yOrientation: rows are then written in reverse
 *   (BitmapSection::reorient, core/BitmapRef.hpp:103-109).
 *
 * All functions return MSDFHIP_OK (0) or a negative MSDFHIP_ERR_* code; msdfhip_last_error() gives the text.
 * There is NO CPU fallback: without a usable gfx950 device every compute entry point fails with MSDFHIP_ERR_NO_DEVICE.
 * Thread safety: all entry points may be called concurrently from many host threads (the reference's generate* functions
 * are re-entrant); host-pointer calls use a per-thread stream and staging buffers.
 */
#ifndef MSDFGEN_HIP_H
#define MSDFGEN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSDFHIP_ABI_VERSION 5

/* mode: which generator (msdfgen.h:46-56) */
#define MSDFHIP_MODE_SDF   1 /* generateSDF   msdfgen.h:47  (TrueDistanceSelector)          1 channel  */
#define MSDFHIP_MODE_PSDF  2 /* generatePSDF  msdfgen.h:49  (PerpendicularDistanceSelector) 1 channel  */
#define MSDFHIP_MODE_MSDF  3 /* generateMSDF  msdfgen.h:51  (MultiDistanceSelector)         3 channels */
#define MSDFHIP_MODE_MTSDF 4 /* generateMTSDF msdfgen.h:53  (MultiAndTrueDistanceSelector)  4 channels */

/* ErrorCorrectionConfig::Mode, core/generator-config.h:20-29 */
#define MSDFHIP_EC_DISABLED       0
#define MSDFHIP_EC_INDISCRIMINATE 1
#define MSDFHIP_EC_EDGE_PRIORITY  2
#define MSDFHIP_EC_EDGE_ONLY      3
/* ErrorCorrectionConfig::DistanceCheckMode, core/generator-config.h:31-38 */
#define MSDFHIP_DO_NOT_CHECK_DISTANCE  0
#define MSDFHIP_CHECK_DISTANCE_AT_EDGE 1
#define MSDFHIP_ALWAYS_CHECK_DISTANCE  2

#define MSDFHIP_OK                  0
#define MSDFHIP_ERR_NO_DEVICE      -1 /* no HIP device / not gfx950 / runtime error at init */
#define MSDFHIP_ERR_INVALID        -2 /* bad argument */
#define MSDFHIP_ERR_HIP            -3 /* a HIP call failed; see msdfhip_last_error() */
#define MSDFHIP_ERR_TOO_COMPLEX    -4 /* (round 3: no longer returned for large shapes -- they take list-free kernels; kept for the error-correction
                                         kernel's fixed LDS exceeding a device's limit, which no gfx950 input reaches) */
#define MSDFHIP_ERR_NOMEM          -5

/* MSDFGeneratorConfig + ErrorCorrectionConfig (core/generator-config.h:13-64) without the buffer pointer. */
typedef struct MsdfHipConfig {
    int32_t overlap_support;     /* GeneratorConfig::overlapSupport, default 1 */
    int32_t ec_mode;             /* default MSDFHIP_EC_EDGE_PRIORITY */
    int32_t ec_distance_check;   /* default MSDFHIP_CHECK_DISTANCE_AT_EDGE */
    int32_t ec_stage_limit;      /* 0 = full pipeline. 1..4: debug -- stop the stencil pipeline after stage k of
                                    core/msdf-error-correction.cpp:27-46 (1 protectCorners, 2 +protectEdges, 3 +findErrors(sdf),
                                    4 +protectAll+findErrors(sdf,shape)) and do not apply; used by the parity tests */
    double min_deviation_ratio;  /* default 1.11111111111111111 (core/MSDFErrorCorrection.cpp:22) */
    double min_improve_ratio;    /* default 1.11111111111111111 (core/MSDFErrorCorrection.cpp:23) */
    /* Optional scanline pass between the distance field and the error correction, as the reference CLI does in no-Skia builds
     * (main.cpp:1281-1298): distanceSignCorrection (core/rasterization.h:17-19). Default off (the library functions do not do it). */
    int32_t sign_correction;     /* 0 / 1 */
    int32_t fill_rule;           /* FillRule, core/Scanline.h:10-15: 0 NONZERO (default), 1 ODD, 2 POSITIVE, 3 NEGATIVE */
    float sdf_zero_value;        /* default 0.5 */
    int32_t stencil_y_down;      /* 1: the output bitmap's yOrientation is Y_DOWNWARD. Only the row order of the optional stencil
                                    (ErrorCorrectionConfig::buffer) depends on it: the reference creates the stencil section with the
                                    default orientation (core/msdf-error-correction.cpp:19) and re-orients it against the bitmap
                                    (core/MSDFErrorCorrection.cpp:122,192,415), so its memory row r is always the r-th row counted
                                    upwards -- the bitmap's memory row height-1-r when the bitmap is Y_DOWNWARD. Default 0. */
} MsdfHipConfig;

/* Per-glyph descriptor of a batch (one output tile per glyph). Lives in device memory for the *_device entry points. */
typedef struct MsdfHipGlyph {
    double xf[6];                /* sx, sy, tx, ty, mapScale, mapTranslate */
    int64_t out_offset;          /* index (in floats) of texel (0,0) of memory row 0 of this tile in the output buffer */
    int32_t row_stride;          /* floats between consecutive memory rows of the tile (BitmapSection::rowStride) */
    int32_t flip;                /* 1: shape Y orientation != bitmap orientation */
} MsdfHipGlyph;

/* Fills *cfg with the reference's defaults (MSDFGeneratorConfig(), core/generator-config.h:46,53,62). */
void msdfhip_default_config(MsdfHipConfig *cfg);

int msdfhip_abi_version(void);
/* Makes HIP device `device` the process default (checks that it is gfx950): the device of the single-shape calls and of batches
 * created without an explicit device. Batches remember their device; every call on a batch binds the calling thread to it, so one
 * process can drive all GPUs of a node (msdfhip_batch_create_on, msdfhip_generate_sharded below). */
int msdfhip_init(int device);
int msdfhip_device_count(int *count);
/* Text of the last error on the calling thread ("" if none). */
const char *msdfhip_last_error(void);
/* name buffer receives e.g. "gfx950:sramecc+:xnack-"; cus/lds_bytes may be NULL. */
int msdfhip_device_info(char *name, size_t name_len, int *cus, int *lds_bytes);

/* ------------------------------------------------------------------------------------------------------------------
 * Single-shape, host-pointer calls: the literal replacements of the reference's functions.
 * ------------------------------------------------------------------------------------------------------------------ */

/* generateSDF / generatePSDF / generateMSDF / generateMTSDF(output, shape, transformation, config)
 *   replaces core/msdfgen.cpp:78-106 (declared msdfgen.h:46-53); the Projection+Range overloads (msdfgen.h:59-63) and the legacy
 *   Range/scale/translate overloads (msdfgen.h:65-69) forward here after building xf.
 * For modes 3, 4 the error-correction pass (core/msdfgen.cpp:97,105) runs as configured. `stencil` is the optional
 * ErrorCorrectionConfig::buffer (core/generator-config.h:44): NULL, or width*height bytes that receive the final stencil. */
int msdfhip_generate(int mode, float *pixels, int width, int height, int row_stride, int flip,
                     const int32_t *contour_offsets, int n_contours, const double *points, const uint8_t *types, const uint8_t *colors,
                     const double *xf, const MsdfHipConfig *cfg, uint8_t *stencil);

int msdfhip_generate_sdf(float *pixels, int width, int height, int row_stride, int flip,
                         const int32_t *contour_offsets, int n_contours, const double *points, const uint8_t *types, const uint8_t *colors,
                         const double *xf, const MsdfHipConfig *cfg);                      /* msdfgen.h:47 */
int msdfhip_generate_psdf(float *pixels, int width, int height, int row_stride, int flip,
                          const int32_t *contour_offsets, int n_contours, const double *points, const uint8_t *types, const uint8_t *colors,
                          const double *xf, const MsdfHipConfig *cfg);                     /* msdfgen.h:49 */
int msdfhip_generate_msdf(float *pixels, int width, int height, int row_stride, int flip,
                          const int32_t *contour_offsets, int n_contours, const double *points, const uint8_t *types, const uint8_t *colors,
                          const double *xf, const MsdfHipConfig *cfg, uint8_t *stencil);   /* msdfgen.h:51 */
int msdfhip_generate_mtsdf(float *pixels, int width, int height, int row_stride, int flip,
                           const int32_t *contour_offsets, int n_contours, const double *points, const uint8_t *types, const uint8_t *colors,
                           const double *xf, const MsdfHipConfig *cfg, uint8_t *stencil);  /* msdfgen.h:53 */

/* msdfErrorCorrection(sdf, shape, transformation, config) on an existing 3- or 4-channel bitmap, in place.
 *   replaces core/msdf-error-correction.cpp:61-72 (declared core/msdf-error-correction.h:15-18). */
int msdfhip_error_correction(int channels, float *pixels, int width, int height, int row_stride, int flip,
                             const int32_t *contour_offsets, int n_contours, const double *points, const uint8_t *types, const uint8_t *colors,
                             const double *xf, const MsdfHipConfig *cfg, uint8_t *stencil);
/* msdfFastDistanceErrorCorrection (protect_all = 0) / msdfFastEdgeErrorCorrection (protect_all = 1): findErrors(sdf) + apply with no shape.
 *   replaces core/msdf-error-correction.cpp:50-59, 87-113 (declared core/msdf-error-correction.h:21-34). */
int msdfhip_error_correction_shapeless(int channels, float *pixels, int width, int height, int row_stride, const double *xf,
                                       double min_deviation_ratio, int protect_all);

/* distanceSignCorrection(sdf, shape, projection, sdfZeroValue, fillRule) on an existing 1-, 3- or 4-channel bitmap, in place.
 *   replaces core/rasterization.cpp:19-92 (declared core/rasterization.h:17-19; the legacy overloads :21-27 forward).
 * xf: only the Projection part {sx, sy, tx, ty} is read. fill_rule: FillRule (core/Scanline.h:10-15). */
int msdfhip_distance_sign_correction(int channels, float *pixels, int width, int height, int row_stride, int flip,
                                     const int32_t *contour_offsets, int n_contours, const double *points, const uint8_t *types, const uint8_t *colors,
                                     const double *xf, float sdf_zero_value, int fill_rule);

/* rasterize(output, shape, projection, fillRule): 1-channel coverage bitmap, 1.f where the texel centre is filled.
 *   replaces core/rasterization.cpp:8-16 (declared core/rasterization.h:13, legacy overload :22) -- the same scanline fill
 *   test as the sign correction; provided so that a build can drop the reference's rasterization.cpp as a whole. */
int msdfhip_rasterize(float *pixels, int width, int height, int row_stride, int flip,
                      const int32_t *contour_offsets, int n_contours, const double *points, const uint8_t *types, const uint8_t *colors,
                      const double *xf, int fill_rule);

/* Transparent micro-batching of the single-shape entry points above. Host threads that call them concurrently
 * (msdf-atlas-gen's workers) are combined group-commit style into one device batch per set of calls with equal launch parameters
 * (mode, size, config); a lone caller never waits. On by default (groups of up to 256 calls, 4 concurrent leaders); the
 * environment variable MSDFHIP_MICROBATCH=0 or max_group <= 1 turns it off (every call then runs on its own stream).
 * msdfhip_microbatch_stats: calls served / device batches run / largest group since the last reset. */
int msdfhip_set_microbatch(int max_group, int max_leaders);
int msdfhip_microbatch_stats(long long *calls, long long *batches, long long *largest, int reset);
/* Where the host-pointer calls spend their wall time, summed over device batches since the last reset (milliseconds):
 * staging the inputs, H2D + kernels + D2H + stream sync, scattering the tiles into the callers' bitmaps. */
int msdfhip_microbatch_times(double *stage_ms, double *device_ms, double *scatter_ms, int reset);

/* ShapeDistanceFinder<CC<Selector>>::oneShotDistance at n shape-space points (core/ShapeDistanceFinder.hpp:36-58; the
 * per-pixel engine under generate*).  selector = mode 1..4; out receives n*4 doubles (unused channels 0). */
int msdfhip_shape_distance(int selector, int overlap_support, const int32_t *contour_offsets, int n_contours, const double *points,
                           const uint8_t *types, const uint8_t *colors, int n_points, const double *pts, double *out);

/* ------------------------------------------------------------------------------------------------------------------
 * Batched, device-resident path (what an atlas generator uses): G glyph shapes -> G tiles of width x height.
 * "Upload once": the edge buffer is flattened by the caller, lives in HBM, and is pre-digested on the device into
 * per-edge records (end tangents, corner bisectors, polynomial coefficients, contour windings).
 * ------------------------------------------------------------------------------------------------------------------ */

typedef struct MsdfHipBatch MsdfHipBatch; /* opaque */

/* Creates a batch from HOST arrays (copied to the device; contour_offsets are global edge indices):
 *   glyph_contour_offsets int32[n_glyphs+1], contour_offsets int32[n_contours+1], points double[n_edges*8], types/colors uint8[n_edges]. */
int msdfhip_batch_create(MsdfHipBatch **batch, int n_glyphs, const int32_t *glyph_contour_offsets, const int32_t *contour_offsets,
                         const double *points, const uint8_t *types, const uint8_t *colors);
/* Same, on an explicit device (-1 = the process default). One host thread per device may create and run its own batch concurrently. */
int msdfhip_batch_create_on(MsdfHipBatch **batch, int device, int n_glyphs, const int32_t *glyph_contour_offsets, const int32_t *contour_offsets,
                            const double *points, const uint8_t *types, const uint8_t *colors);
int msdfhip_batch_device(const MsdfHipBatch *batch, int *device);
/* Same, from DEVICE arrays that already live in HBM (e.g. torch tensors); they must stay valid until the batch is destroyed.
 * max_contours_per_glyph / max_edges_per_glyph are host-known upper bounds used to size LDS. `stream` is a hipStream_t (or NULL). */
int msdfhip_batch_create_device(MsdfHipBatch **batch, int n_glyphs, int n_contours, int n_edges, int max_contours_per_glyph, int max_edges_per_glyph,
                                const int32_t *d_glyph_contour_offsets, const int32_t *d_contour_offsets,
                                const double *d_points, const uint8_t *d_types, const uint8_t *d_colors, void *stream);
/* Re-runs the on-device digestion (records + windings) on `stream`, e.g. after the caller rewrote the edge arrays in place
 * (device-array batches) -- this is the per-upload part of the pipeline that a benchmark step should include. */
int msdfhip_batch_digest(MsdfHipBatch *batch, void *stream);
void msdfhip_batch_destroy(MsdfHipBatch *batch);
/* Contour::winding (core/Contour.cpp:57-81) of every contour of the batch, as computed on the device (host int32[n_contours]). */
int msdfhip_batch_windings(const MsdfHipBatch *batch, int32_t *windings);

/* Generates all tiles of the batch: glyph g's tile is written at d_out + d_glyphs[g].out_offset with d_glyphs[g].row_stride.
 * d_glyphs (device, MsdfHipGlyph[n_glyphs]), d_out (device floats), d_stencil (device, n_glyphs*width*height bytes, or NULL),
 * d_scratch: device floats for the intermediate fields: one tile extent (n_glyphs*width*height*N) if error correction OR sign
 * correction runs, two if both do; may be NULL, then internal buffers are used.
 * Asynchronous on `stream` (hipStream_t or NULL = default stream). Replaces a loop of msdfgen.h:46-53 calls.
 * Thread safety: a batch keeps per-batch work buffers (candidate lists, bucket lists, the internal scratch); issue its calls from one
 * stream at a time. Different batches, and the single-shape entry points above, are independent of each other. */
int msdfhip_batch_generate(const MsdfHipBatch *batch, int mode, int width, int height, const MsdfHipGlyph *d_glyphs,
                           float *d_out, uint8_t *d_stencil, float *d_scratch, const MsdfHipConfig *cfg, void *stream);
/* End to end into HOST memory (what replaces an atlas generator's loop of generate*() calls into caller-owned bitmaps,
 * core/msdfgen.cpp:52-76, README.md:133): HOST descriptors, HOST output. The glyph list is processed in chunks on two streams, so the
 * kernels of one chunk overlap the device-to-host copy of the previous one; synchronous for the caller.
 *   glyphs[g].out_offset / row_stride place tile g in `out` (floats). A chunk whose rectangles exactly tile one contiguous range of
 *   `out` (tiles packed in glyph order; glyphs in row-major order filling whole row bands of an atlas) is written by the device in that
 *   layout and copied back as ONE contiguous copy; any other placement (gaps, negative strides) is copied into pinned staging and
 *   scattered row by row on the host while the next chunk runs -- texels outside the rectangles are never touched.
 *   stencil: NULL or n_glyphs*width*height bytes; written only when an error-correction pass runs (like the reference's buffer).
 * For full copy speed `out` should be pinned (msdfhip_host_alloc); pageable memory works, the runtime then stages the copies. */
int msdfhip_batch_generate_host(const MsdfHipBatch *batch, int mode, int width, int height, const MsdfHipGlyph *glyphs,
                                float *out, size_t out_floats, uint8_t *stencil, const MsdfHipConfig *cfg);
/* Same pipeline with 8-bit output: every chunk's float tiles stay on the device, are converted with pixelFloatToByte
 * (core/pixel-conversion.hpp:8-10) and blitted into the caller's uint8 atlas; glyphs[g].out_offset / row_stride are in BYTES of
 * `atlas`. The copy back is a quarter of the float tiles'. */
int msdfhip_batch_generate_bytes_host(const MsdfHipBatch *batch, int mode, int width, int height, const MsdfHipGlyph *glyphs,
                                      uint8_t *atlas, size_t atlas_bytes, const MsdfHipConfig *cfg);
/* STREAMED generation: shapes in, host bitmaps out, with everything in between overlapped -- SURVEY.md 8(d)'s end-to-end metric (host flatten + H2D +
 * kernels incl. error correction + D2H into caller bitmaps) as ONE pipelined call. The two-step form above (msdfhip_batch_create, then
 * msdfhip_batch_generate_host) flattens everything, uploads and digests everything, and only then starts the chunk pipeline; here the glyph list is cut
 * into the pipeline's chunks first and the library's host threads flatten chunk k+1 / k+2 straight into pinned staging WHILE chunk k is uploaded, digested and
 * rendered and chunk k-1 is copied back. Replaces a caller loop of generate*() calls over a list of shapes (core/msdfgen.cpp:52-76, README.md:133,
 * main.cpp:1243-1275); the C++ shim's msdfgen_hip::generate*Batch() are built on it.
 *
 * MsdfHipShapeSource: how the library reads the caller's shape objects (for msdfgen: `const Shape &`, core/Shape.h:15-58 -- contours[i].edges[j]->type(),
 * controlPoints(), color; core/edge-segments.h:28-31). Both callbacks are called from several library threads at once, each glyph exactly once per pass;
 * they must only READ the shapes.
 *   count(user, glyph, &n_contours, &n_edges)   cheap first pass over the whole list (sizes the chunks and the staging)
 *   fill(user, glyph, edge_base, contour_end, points, types, colors)
 *        contour_end int32[n_contours]: edge_base + the index one past contour c's last edge (edges of the glyph numbered from 0 in Shape order)
 *        points double[n_edges*8], types uint8[n_edges] (1 linear, 2 quadratic, 3 cubic), colors uint8[n_edges]: as for msdfhip_generate above
 *   The arrays handed to fill ARE the pinned staging: what fill writes is what the device reads, there is no intermediate copy.
 * device: -1 = the process default. glyphs / out / atlas / stencil / cfg: exactly as msdfhip_batch_generate_host / _bytes_host (exactly one of out, atlas). */
typedef struct MsdfHipShapeSource {
    void *user;
    void (*count)(void *user, int glyph, int32_t *n_contours, int32_t *n_edges);
    void (*fill)(void *user, int glyph, int32_t edge_base, int32_t *contour_end, double *points, uint8_t *types, uint8_t *colors);
} MsdfHipShapeSource;
int msdfhip_generate_stream(int device, int mode, int width, int height, int n_glyphs, const MsdfHipShapeSource *source, const MsdfHipGlyph *glyphs,
                            float *out, size_t out_floats, uint8_t *atlas, size_t atlas_bytes, uint8_t *stencil, const MsdfHipConfig *cfg);
/* The same pipeline over HOST CSR arrays (the arguments of msdfhip_batch_create): upload and digest go chunk by chunk under the kernels instead of in front. */
int msdfhip_generate_stream_csr(int device, int mode, int width, int height, int n_glyphs, const int32_t *glyph_contour_offsets, const int32_t *contour_offsets,
                                const double *points, const uint8_t *types, const uint8_t *colors, const MsdfHipGlyph *glyphs,
                                float *out, size_t out_floats, uint8_t *atlas, size_t atlas_bytes, uint8_t *stencil, const MsdfHipConfig *cfg);
/* Host threads of the streamed generator's flatten pool, incl. the calling thread (0 = the usable cores -- affinity mask and cgroup quota --, at most 32; also
 * MSDFHIP_HOST_THREADS). The pool is created on first use: returns 0 when the value was taken, else the size of the pool that already exists. */
int msdfhip_set_host_threads(int threads);
/* GPU_MAX_HW_QUEUES as the host process exported it (0: not set -- the HIP runtime then multiplexes all streams onto 4 hardware queues and the chunks of the
 * generators above wait behind one another's copies: 12.0 instead of 10.2 ms per 8 192 glyphs). The library does not touch the environment; a host that
 * runs the pipeline with fewer than 8 gets ONE note on stderr if MSDFHIP_VERBOSE is set (a drop-in library stays silent by default). Figures in README.md / DESIGN.md assume GPU_MAX_HW_QUEUES=8. */
int msdfhip_hw_queues_env(void);
/* How often a host-output / streamed call was run a second time because a glyph's distance-check candidates overflowed their segment (more than 1/16 of its
 * texels needed a check: overlapping strokes rendered without overlap support, ALWAYS_CHECK_DISTANCE on noise). The chunks of those calls do not launch the
 * per-texel overflow pass; the count is mirrored to the host with the chunk's results and such a call is repeated with the pass -- same bytes either way. */
unsigned long long msdfhip_pipeline_overflow_reruns(int reset);
/* Glyphs per pipeline chunk (0 = automatic: about 96 MB of float tiles). */
int msdfhip_set_pipeline_chunk(int glyphs_per_chunk);
/* Pinned (page-locked, portable across devices) host memory for outputs of the two functions above. */
int msdfhip_host_alloc(void **p, size_t bytes);
int msdfhip_host_free(void *p);

/* Glyph-sharded generation on several devices of one node: the glyph list (HOST CSR arrays as for
 * msdfhip_batch_create) is cut into n_devices contiguous ranges balanced by edge count; one host thread per entry of `devices`
 * uploads its range, runs the pipeline above on that device and copies its tiles straight into the caller's buffer. No exchange
 * between devices; the bytes do not depend on the split (a device may be listed more than once), and any tile placement is fine: a
 * device only ever writes its own glyphs' rectangles. Exactly one of `out` (float tiles, offsets in floats) and `atlas` (8-bit,
 * offsets in bytes) is non-NULL. */
int msdfhip_generate_sharded(const int *devices, int n_devices, int mode, int width, int height, int n_glyphs,
                             const int32_t *glyph_contour_offsets, const int32_t *contour_offsets, const double *points, const uint8_t *types,
                             const uint8_t *colors, const MsdfHipGlyph *glyphs, float *out, size_t out_floats, uint8_t *atlas, size_t atlas_bytes,
                             const MsdfHipConfig *cfg);

/* Shape preparation on the device: what callers run on every glyph before the generators.
 *   normalize  Shape::normalize (core/Shape.cpp:65-92): single-edge contours split in thirds, convergent edges pushed apart
 *   coloring   0 keep `colors`, 1 edgeColoringSimple(shape, angle_threshold, seed) (core/edge-coloring.cpp:68-142),
 *              2 edgeColoringInkTrap(shape, angle_threshold, seed) (core/edge-coloring.cpp:151-258)
 * msdfhip_batch_create_prepared uploads RAW outlines (colors may be NULL = all WHITE), prepares them on the device and returns a
 * digested batch of the prepared shapes; seeds: one per glyph, or NULL to use cfg->seed for every glyph. The prepared shapes can be
 * read back with msdfhip_batch_info (sizes) + msdfhip_batch_download (arrays sized from those; any pointer may be NULL). */
typedef struct MsdfHipPrepConfig {
    int32_t normalize;
    int32_t coloring;
    double angle_threshold;       /* main.cpp:32: 3.0 */
    uint64_t seed;
} MsdfHipPrepConfig;
int msdfhip_batch_create_prepared(MsdfHipBatch **batch, int n_glyphs, const int32_t *glyph_contour_offsets, const int32_t *contour_offsets,
                                  const double *points, const uint8_t *types, const uint8_t *colors, const uint64_t *seeds,
                                  const MsdfHipPrepConfig *cfg);
/* Diagnostics: after an error-correction pass, counts[0] = 1 if some glyph's candidate segment overflowed (those glyphs were redone by
 * the full per-texel pipeline), counts[1+g] = deferred distance checks pushed for glyph g. counts holds n_glyphs+1 entries. */
int msdfhip_batch_candidate_counts(const MsdfHipBatch *batch, uint32_t *counts);
int msdfhip_batch_info(const MsdfHipBatch *batch, int *n_glyphs, int *n_contours, int *n_edges, int *max_contours, int *max_edges);
int msdfhip_batch_download(const MsdfHipBatch *batch, int32_t *contour_offsets, double *points, uint8_t *types, uint8_t *colors);

/* 8-bit atlas output: converts packed fp32 tiles [g][h][w][channels] (msdfhip_batch_generate's output with out_offset =
 * g*h*w*channels, row_stride = w*channels) with pixelFloatToByte (core/pixel-conversion.hpp:8-10, what the reference's savers
 * apply, core/save-bmp.cpp:174-224, ext/save-png.cpp:83-187) and stores glyph g's rectangle at
 * d_atlas + d_glyphs[g].out_offset + d_glyphs[g].row_stride*y + channels*x, offsets and strides in BYTES (xf and flip are not read).
 * Asynchronous on `stream`. The device-to-host copy of an 8-bit atlas is a quarter of the float tiles'. */
int msdfhip_tiles_to_bytes(const float *d_tiles, int n_glyphs, int width, int height, int channels, const MsdfHipGlyph *d_glyphs,
                           uint8_t *d_atlas, void *stream);

/* estimateSDFError(sdf, shape, projection, scanlinesPerRow, fillRule) for every glyph of a batch: d_errors[g] from the
 * packed tiles d_tiles[g][h][w][channels] (memory rows; what msdfhip_batch_generate wrote), without leaving the device.
 *   replaces core/sdf-error-estimation.cpp:134-164 (declared core/sdf-error-estimation.h:18-20).
 * d_glyphs[g].xf[0..3] = the Projection; d_glyphs[g].flip is read as "shape.getYAxisOrientation() == Y_DOWNWARD"
 * (sdf-error-estimation.cpp:146; for Y-upward bitmaps that is the flip flag the tiles were generated with). Asynchronous on `stream`. */
int msdfhip_batch_estimate_sdf_error(const MsdfHipBatch *batch, int channels, int width, int height, const MsdfHipGlyph *d_glyphs, const float *d_tiles,
                                     int scanlines_per_row, int fill_rule, double *d_errors, void *stream);

/* renderSDF(output, sdf, sdfPxRange, sdThreshold) for n_glyphs packed tiles: d_out[g][oh][ow][out_channels] from
 * d_sdf[g][sh][sw][sdf_channels], memory rows as they are.
 *   replaces core/render-sdf.cpp:14-170 (declared core/render-sdf.h:12-17); channel pairs as the reference's overloads:
 *   1<-1, 3<-1, 1<-3, 3<-3, 1<-4, 4<-4. range_lower == range_upper selects the thresholded rendering (core/render-sdf.cpp:16-23).
 * msdfhip_simulate_8bit: simulate8bit (core/render-sdf.cpp:172-188) on n floats in place. Both asynchronous on `stream`. */
int msdfhip_render_sdf(const float *d_sdf, int n_glyphs, int sdf_width, int sdf_height, int sdf_channels, float *d_out, int out_width, int out_height,
                       int out_channels, double range_lower, double range_upper, float sd_threshold, void *stream);
int msdfhip_simulate_8bit(float *d_pixels, size_t n, void *stream);
/* The same two on HOST bitmaps (what the C++ shim's renderSDF / simulate8bit overloads call; synchronous). Strides in floats. */
int msdfhip_render_sdf_host(float *out, int out_width, int out_height, int out_row_stride, int out_channels,
                            const float *sdf, int sdf_width, int sdf_height, int sdf_row_stride, int sdf_channels,
                            double range_lower, double range_upper, float sd_threshold);
int msdfhip_simulate_8bit_host(float *pixels, int width, int height, int row_stride, int channels);

/* Timing hook for bench.py: average device time in milliseconds of the dominant kernel (the distance-field kernel) and of
 * the passes after it (sign correction + error correction, per error-correction launch) over the launches recorded since the
 * last call with reset != 0, measured with hipEvents on the launching stream.
 * Enable with msdfhip_set_kernel_timing(1) (adds two event records per launch). */
int msdfhip_set_kernel_timing(int enable);
int msdfhip_kernel_timing(double *avg_ms_distance, double *avg_ms_correction, int *launches, int reset);

/* Frees the pooled resources of the host-pointer entry points (arenas of the single-shape calls, pipelines of the host-output calls) that are
 * not in use right now. The pools grow to the peak number of concurrent calls and are otherwise kept for the life of the process. */
int msdfhip_trim(void);

/* MSDFHIP_DEVICES = "all" | "0,1,...": the devices over which the host-pointer single-shape entry points (msdfhip_generate* etc., and with them
 * the C++ shim) spread their micro-batched groups, round robin; unset = the default device (msdfhip_init). Returns how many are configured
 * and writes up to `cap` of them to `out`. */
int msdfhip_front_door_devices(int *out, int cap);

/* The library reads its MSDFHIP_* environment knobs ("environment") once, at first use. Tests and A/B scripts that change
 * the environment of a running process call this to have them read again. Not meant for production code. */
int msdfhip_reload_tuning(void);
/* Measurement builds only: the per-wavefront cycle table of the distance kernel (-DMSDF_PROFILE_WAITS, tools/profile_waits.py) or of the
 * distance checks of the error correction (-DMSDF_PROFILE_QUERY, tools/profile_query.py; reset = 2 reads its second page); a regular build
 * reports zeros. out24: 24 counters. reset = 1 clears the table after reading. */
int msdfhip_debug_wait_profile(unsigned long long *out24, int reset);
/* Measurement builds only (tools/isa_bbcount.py: -DMSDF_BBCOUNT, counter bumps inserted into the kernels' gfx950 assembly): execution counts per basic block of
 * the instrumented kernels since the last reset. Returns the number of counters written to `out` (at most `cap`); a regular build returns 0. */
int msdfhip_debug_bbcount(unsigned *out, int cap, int reset);
/* Fused single-shape launches (k_single_call): out8[0] = calls since the last reset (+ 1e-6 x the shader clock in MHz the launches ran at), out8[1..6] = microseconds per call, as seen by workgroup 0, of:
 * digest | its own distance tile | waiting for all tiles (grid barrier) | its own correction sweep | waiting for all sweeps | distance checks,
 * out8[7] = start of workgroup 0 to the last workgroup's end. Diagnostics (tools/host_call_latency.py). */
int msdfhip_debug_single_call_phases(double *out8, int reset);

/* The one-launch form of a single-shape call (k_single_call) needs all its workgroups resident at once. Calls that would not fit next to the fused
 * launches already in flight take the batched launch sequence instead (*refused); a launch that still finds the device occupied -- a persistent
 * kernel of another thread or process holds the slots -- gives up at its grid barrier after a bounded wait (*barrier_timeouts) and is rerun through the
 * batched sequence, as is one that ended without its completion flag (*lost_flags). The caller's generate*() succeeds either way
 * (core/msdfgen.cpp:78-106 cannot fail); these counters only say how often the slow road was taken since the last reset. Any pointer may be NULL. */
int msdfhip_single_call_fallbacks(unsigned long long *barrier_timeouts, unsigned long long *lost_flags, unsigned long long *refused, int reset);

#ifdef __cplusplus
}
#endif
#endif /* MSDFGEN_HIP_H */
