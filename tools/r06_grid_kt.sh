# k_ec_query's own duration per MSDFHIP_QUERY_GRID setting (kernel trace of the bench step).   bash tools/r06_call.sh <tag> r06_grid_kt.sh
TAG=$1; REPO=$PWD; export TMPDIR=/tmp
for gsteps in ${GRIDS:-0 8 16 32}; do
  (cd /tmp && MSDFHIP_QUERY_GRID=$gsteps rocprofv3 --kernel-trace --stats -d /tmp/kt_${TAG}_$gsteps -o kt -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1)
  echo "== MSDFHIP_QUERY_GRID=$gsteps"; python tools/rocpd_summary.py $(find /tmp/kt_${TAG}_$gsteps -name "*.db") 2>/dev/null | grep -E "k_ec_query|k_ec_fast|k_ec_scan" | cut -c1-60,73-140
done
