#!/bin/bash
# Builds libmsdfgen_hip's HOST code with sanitizers (device code unchanged) and runs tests/sanitize/san_driver.cpp on the GPU box:
#   asan   AddressSanitizer + UndefinedBehaviorSanitizer (leak check on; the HIP runtime's own one-time allocations are suppressed)
#   tsan   ThreadSanitizer over the group-commit micro-batcher, the arena pool, the pipeline slots and the sharded generator
# Usage (from the repository root, needs a GPU): bash tests/sanitize/run.sh [asan|tsan|both]   -> gpurun_out/sanitize_*.log
#        (no GPU needed)                          bash tests/sanitize/run.sh build [asan|tsan|both]   builds only; the run then finds them up to date
set -u
MODE=${1:-both}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT $ROOT/tests/sanitize/build
B=$ROOT/tests/sanitize/build
COMMON="--offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -Wno-unused-value"
stale() {   # target: true when missing or older than any source it is built from
    [ -f "$1" ] || return 0
    for f in $ROOT/msdfgen_amd/csrc/* $ROOT/include/msdfgen_hip.h $ROOT/tests/sanitize/san_driver.cpp; do [ "$f" -nt "$1" ] && return 0; done
    return 1
}
build() {   # name, sanitizer flags  (hipcc cross-compiles: `run.sh build both` in the authoring container saves the GPU box two 70 s compiles; tests/sanitize/build/ travels with gpurun)
    local name=$1 flags=$2
    if stale $B/libmsdfgen_hip_$name.so; then
        /opt/rocm/bin/hipcc $COMMON -shared -Xarch_host "$flags" -Xarch_host -fno-omit-frame-pointer $ROOT/msdfgen_amd/csrc/msdf_capi.hip -o $B/libmsdfgen_hip_$name.so $flags > $OUT/sanitize_${name}_build.log 2>&1 || { echo "$name: library build failed"; tail -5 $OUT/sanitize_${name}_build.log; return 1; }
    fi
    if stale $B/san_driver_$name; then
        /opt/rocm/llvm/bin/clang++ -O1 -g -std=c++17 $flags -fno-omit-frame-pointer -I $ROOT/include $ROOT/tests/sanitize/san_driver.cpp -o $B/san_driver_$name \
            -L $B -l:libmsdfgen_hip_$name.so -Wl,-rpath,'$ORIGIN' -pthread >> $OUT/sanitize_${name}_build.log 2>&1 || { echo "$name: driver build failed"; tail -5 $OUT/sanitize_${name}_build.log; return 1; }
    fi
}
run() {   # name, sanitizer flags, env
    local name=$1 flags=$2
    build "$name" "$flags" || return 1
    shift 2
    env "$@" timeout 600 $B/san_driver_$name 3 > $OUT/sanitize_$name.log 2>&1
    echo "$name: exit $? -- $(tail -1 $OUT/sanitize_$name.log)"
    grep -c "ERROR: AddressSanitizer\|runtime error:\|WARNING: ThreadSanitizer" $OUT/sanitize_$name.log | sed "s/^/$name: sanitizer reports: /"
}
if [ "$MODE" = build ]; then
    WHAT=${2:-both}
    { [ "$WHAT" = asan ] || [ "$WHAT" = both ]; } && build asan "-fsanitize=address,undefined"
    { [ "$WHAT" = tsan ] || [ "$WHAT" = both ]; } && build tsan "-fsanitize=thread"
    ls -la $B
    exit 0
fi
if [ "$MODE" = asan ] || [ "$MODE" = both ]; then
    run asan "-fsanitize=address,undefined" ASAN_OPTIONS=detect_leaks=1:protect_shadow_gap=0:suppressions=$ROOT/tests/sanitize/asan.supp LSAN_OPTIONS=suppressions=$ROOT/tests/sanitize/lsan.supp UBSAN_OPTIONS=print_stacktrace=1
fi
if [ "$MODE" = tsan ] || [ "$MODE" = both ]; then
    run tsan "-fsanitize=thread" TSAN_OPTIONS=suppressions=$ROOT/tests/sanitize/tsan.supp:history_size=4
fi
