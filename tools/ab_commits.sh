# Same-box A/B of library VERSIONS (boxes differ by up to 10 % on the latency-bound figures; only a same-box comparison separates code from box).
#   bash tools/ab_commits.sh build <commit>...     here (no GPU): builds variants/<commit>.so from that commit's msdfgen_amd/csrc + include
#   bash tools/ab_commits.sh run "<command>" <commit>...   on the GPU box: the command with each variant and with the current library, interleaved
# e.g. gpurun -- 'bash tools/ab_commits.sh run "python tools/host_call_latency.py --threads 64 --leaders 4" 548e27d b4c23f0'
MODE=$1; shift
if [ "$MODE" = build ]; then
  mkdir -p variants
  for c in "$@"; do
    rm -rf /tmp/ab_$c; mkdir -p /tmp/ab_$c
    git archive $c msdfgen_amd/csrc include | tar -x -C /tmp/ab_$c
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm -fPIC -shared -Wno-unused-value \
      /tmp/ab_$c/msdfgen_amd/csrc/msdf_capi.hip -o variants/$c.so 2>&1 | grep -E "error" ; ls -la variants/$c.so
  done
else
  CMD=$1; shift
  for v in "$@" current "$@" current; do
    if [ $v = current ]; then unset MSDFGEN_HIP_LIB; else export MSDFGEN_HIP_LIB=$PWD/variants/$v.so; fi
    $CMD 2>/dev/null | cut -c1-300 | sed "s/^/$v /"
  done
fi
