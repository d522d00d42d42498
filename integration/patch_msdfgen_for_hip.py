#!/usr/bin/env python3
"""The source change INTEGRATION.md section 2 asks a msdfgen maintainer to make, as a script: wraps the definitions that the HIP shim
(msdfgen_amd/shim/msdfgen_shim.cpp) provides in `#ifndef MSDFGEN_USE_HIP ... #endif`, in COPIES of two files of a msdfgen 1.13 checkout:

    core/msdfgen.cpp                 DistancePixelConversion / generateDistanceField / generateSDF, PSDF, MSDF, MTSDF and all their overloads
                                     (lines 11-162); the _legacy generators after them stay
    core/msdf-error-correction.cpp   msdfErrorCorrectionInner / Shapeless, msdfErrorCorrection x4, msdfFastDistance/EdgeErrorCorrection x12
                                     (lines 12-113); detectClash / msdfErrorCorrection_legacy stay

core/rasterization.cpp and core/render-sdf.cpp are dropped from the build instead (every function in them is in the shim).

    python integration/patch_msdfgen_for_hip.py <msdfgen checkout> <output dir>

The anchors are the function signatures, not line numbers; the script fails loudly if the checkout does not look like 1.13.
Nothing is written into the checkout."""
import os
import sys


def guard(text, first_anchor, end_anchor, what):
    """Inserts #ifndef before the line containing first_anchor and #endif before the line containing end_anchor."""
    a = text.find(first_anchor)
    b = text.find(end_anchor)
    if a < 0 or b < 0 or b <= a:
        raise SystemExit("patch_msdfgen_for_hip: anchors of %s not found -- not a msdfgen 1.13 source?" % what)
    a = text.rfind("\n", 0, a)+1
    b = text.rfind("\n", 0, b)+1
    return text[:a]+"#ifndef MSDFGEN_USE_HIP // provided by libmsdfgen_hip_shim (MI355X)\n"+text[a:b]+"#endif // MSDFGEN_USE_HIP\n\n"+text[b:]


def main():
    src, out = sys.argv[1], sys.argv[2]
    os.makedirs(out, exist_ok=True)
    t = open(os.path.join(src, "core", "msdfgen.cpp")).read()
    t = guard(t, "template <typename DistanceType>\nclass DistancePixelConversion;", "// Legacy version", "core/msdfgen.cpp")
    open(os.path.join(out, "msdfgen.cpp"), "w").write(t)
    t = open(os.path.join(src, "core", "msdf-error-correction.cpp")).read()
    t = guard(t, "template <int N>\nstatic void msdfErrorCorrectionInner(", "// Legacy version", "core/msdf-error-correction.cpp")
    open(os.path.join(out, "msdf-error-correction.cpp"), "w").write(t)
    print("patched copies in", out)


if __name__ == "__main__":
    main()
