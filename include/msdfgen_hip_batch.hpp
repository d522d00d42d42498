// msdfgen_hip_batch.hpp -- the batch entry of the C++ drop-in (libmsdfgen_hip_shim.so): MANY shapes in one call.
//
// msdfgen's generators take one Shape and one bitmap per call (msdfgen.h:46-69); an atlas tool calls them in a loop, one glyph at a time
// (main.cpp:1243-1275 for a single shape; msdf-atlas-gen's glyph generators over a list). The per-call functions of the shim keep that interface.
// These take the LIST: an array of `const Shape *`, one transformation and one output section per shape -- and run SURVEY.md 8(d)'s whole end-to-end path
// as one pipeline: the list is cut into chunks, and while the device renders chunk k (kernels incl. msdfErrorCorrection) the library's host threads
// flatten the Shape objects of chunks k+1 / k+2 straight into pinned staging (core/Shape.h:24-27: contours[i].edges[j], one heap object per edge,
// core/EdgeHolder.cpp:12), chunk k+1's upload + digest are queued behind, and chunk k-1's tiles travel back into the caller's bitmaps.
//
// Results: bit-identical to calling msdfgen::generate*() of the shim once per shape (tests/test_gpu_shim.py), i.e. to the reference within the documented
// 1e-5 (in practice 0 differing texels). The byte overloads convert with the reference's pixelFloatToByte (core/pixel-conversion.hpp:8-10) on the device,
// so an 8-bit atlas costs a quarter of the copy back.
//
// Include after <msdfgen.h>; link libmsdfgen_hip_shim.so (+ libmsdfgen_hip.so). Failures throw std::runtime_error (or set the shim's status in no-throw mode,
// msdfgen_hip_shim_set_nothrow), like the per-call functions. ErrorCorrectionConfig::buffer is not used by the batch functions (one buffer cannot serve a list).
#pragma once

#include "msdfgen.h"

namespace msdfgen_hip {

/// outputs[i] <- generateSDF(shapes[i], transformations[i]); sections of equal size are rendered together, any placement / row stride / orientation per section
void generateSDFBatch(const msdfgen::BitmapSection<float, 1> *outputs, const msdfgen::Shape *const *shapes, const msdfgen::SDFTransformation *transformations, int count,
                      const msdfgen::GeneratorConfig &config = msdfgen::GeneratorConfig());
void generatePSDFBatch(const msdfgen::BitmapSection<float, 1> *outputs, const msdfgen::Shape *const *shapes, const msdfgen::SDFTransformation *transformations, int count,
                       const msdfgen::GeneratorConfig &config = msdfgen::GeneratorConfig());
/// incl. msdfErrorCorrection as configured (core/msdfgen.cpp:92-98)
void generateMSDFBatch(const msdfgen::BitmapSection<float, 3> *outputs, const msdfgen::Shape *const *shapes, const msdfgen::SDFTransformation *transformations, int count,
                       const msdfgen::MSDFGeneratorConfig &config = msdfgen::MSDFGeneratorConfig());
void generateMTSDFBatch(const msdfgen::BitmapSection<float, 4> *outputs, const msdfgen::Shape *const *shapes, const msdfgen::SDFTransformation *transformations, int count,
                        const msdfgen::MSDFGeneratorConfig &config = msdfgen::MSDFGeneratorConfig());

/// The same into 8-bit bitmaps, e.g. outputs[i] = atlas.getSection(x, y, x+w, y+h) of one BitmapRef<byte, N> (core/BitmapRef.hpp:36-43): every texel is
/// pixelFloatToByte of what the float overload writes.
void generateSDFBatch(const msdfgen::BitmapSection<msdfgen::byte, 1> *outputs, const msdfgen::Shape *const *shapes, const msdfgen::SDFTransformation *transformations, int count,
                      const msdfgen::GeneratorConfig &config = msdfgen::GeneratorConfig());
void generatePSDFBatch(const msdfgen::BitmapSection<msdfgen::byte, 1> *outputs, const msdfgen::Shape *const *shapes, const msdfgen::SDFTransformation *transformations, int count,
                       const msdfgen::GeneratorConfig &config = msdfgen::GeneratorConfig());
void generateMSDFBatch(const msdfgen::BitmapSection<msdfgen::byte, 3> *outputs, const msdfgen::Shape *const *shapes, const msdfgen::SDFTransformation *transformations, int count,
                       const msdfgen::MSDFGeneratorConfig &config = msdfgen::MSDFGeneratorConfig());
void generateMTSDFBatch(const msdfgen::BitmapSection<msdfgen::byte, 4> *outputs, const msdfgen::Shape *const *shapes, const msdfgen::SDFTransformation *transformations, int count,
                        const msdfgen::MSDFGeneratorConfig &config = msdfgen::MSDFGeneratorConfig());

}
