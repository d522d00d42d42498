# Builds variants/<name>.so from the current sources with extra compiler flags (A/B of kernel variants: MSDFGEN_HIP_LIB=variants/<name>.so, tools/r06_call.sh ab:<name>).
# Usage: bash tools/build_variant.sh <name> [-DMSDF_... ...]
NAME=$1; shift
mkdir -p variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -shared -Wall -Wno-unused-value "$@" msdfgen_amd/csrc/msdf_capi.hip -o variants/$NAME.so && echo "built variants/$NAME.so ($*)"
