# duration of the correction kernels per library variant (kernel trace of the bench step).   VARIANTS="main a b" bash tools/r06_call.sh <tag> r06_variants_kt.sh
TAG=$1; REPO=$PWD; export TMPDIR=/tmp
for v in ${VARIANTS:-main}; do
  if [ $v = main ]; then L=; else L=$REPO/variants/$v.so; fi
  (cd /tmp && rm -rf /tmp/kt_${TAG}_$v && MSDFGEN_HIP_LIB=$L timeout 150 rocprofv3 --kernel-trace --stats -d /tmp/kt_${TAG}_$v -o kt -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1)
  echo "== $v"; python tools/rocpd_summary.py $(find /tmp/kt_${TAG}_$v -name "*.db") 2>/dev/null | grep -E "${KERNELS:-k_ec_query|k_ec_scan|k_ec_fast}" | cut -c1-60,73-140
done
