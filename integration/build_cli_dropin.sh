#!/bin/bash
# Builds the reference's UNMODIFIED main.cpp twice:
#   tests/cli/msdfgen_cpu   all of msdfgen 1.13 core, as shipped (the all-CPU binary)
#   tests/cli/msdfgen_hip   the same CLI with INTEGRATION.md section 2 applied: core/msdfgen.cpp and core/msdf-error-correction.cpp patched by
#                           integration/patch_msdfgen_for_hip.py (-DMSDFGEN_USE_HIP), core/rasterization.cpp and core/render-sdf.cpp dropped,
#                           msdfgen_shim linked in -- every generate* / msdfErrorCorrection / distanceSignCorrection / renderSDF of the CLI runs on
#                           the MI355X.
# The patched copies and all objects live in a scratch directory; only the two binaries land in the tree (git-ignored, shipped to the GPU box).
set -e
REF=${1:-/root/reference}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SCRATCH=${TMPDIR:-/tmp}/msdfgen_cli_dropin
DEFS="-DMSDFGEN_PUBLIC= -DMSDFGEN_USE_CPP11 -DMSDFGEN_STANDALONE -DMSDFGEN_VERSION=1.13.0 -DMSDFGEN_COPYRIGHT_YEAR=2025"
rm -rf "$SCRATCH"; mkdir -p "$SCRATCH/cpu" "$SCRATCH/hip" "$ROOT/tests/cli"
python3 "$ROOT/integration/patch_msdfgen_for_hip.py" "$REF" "$SCRATCH/patched" > /dev/null
for f in "$REF"/core/*.cpp; do
    b=$(basename "$f" .cpp)
    g++ -O2 -std=c++11 $DEFS -I"$REF" -c "$f" -o "$SCRATCH/cpu/$b.o" &
    case $b in
        rasterization|render-sdf) ;;                                   # dropped: the shim provides every function of these files
        msdfgen|msdf-error-correction) g++ -O2 -std=c++11 $DEFS -DMSDFGEN_USE_HIP -I"$REF" -I"$REF/core" -c "$SCRATCH/patched/$b.cpp" -o "$SCRATCH/hip/$b.o" & ;;
        *) ln -sf "$SCRATCH/cpu/$b.o" "$SCRATCH/hip/$b.o" ;;
    esac
done
g++ -O2 -std=c++11 $DEFS -I"$REF" -c "$REF/main.cpp" -o "$SCRATCH/main.o" &
wait
g++ -o "$ROOT/tests/cli/msdfgen_cpu" "$SCRATCH/main.o" "$SCRATCH"/cpu/*.o -lpthread
g++ -o "$ROOT/tests/cli/msdfgen_hip" "$SCRATCH/main.o" "$SCRATCH"/hip/*.o -L"$ROOT/msdfgen_amd/lib" -lmsdfgen_hip_shim -lmsdfgen_hip \
    -Wl,-rpath,'$ORIGIN/../../msdfgen_amd/lib' -lpthread
ls -la "$ROOT/tests/cli/"
