# Grid form of the distance checks (k_ec_query): parity tests, then the knob MSDFHIP_QUERY_GRID (0 = off) on the configs.   bash tools/r06_call.sh <tag> r06_grid_ab.sh
TAG=$1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -3
for gsteps in ${GRIDS:-0 8 4 16 0 8}; do
  echo "== MSDFHIP_QUERY_GRID=$gsteps"
  MSDFHIP_QUERY_GRID=$gsteps timeout 300 python tools/bench_configs.py --reps 8 --only "${ONLY:-headline,bench workload,cfg4: 8192 CJK,cfg4 real,cfg5}" 2>/dev/null | tee gpurun_out/${TAG}_grid_$gsteps.jsonl | python tools/ab_show.py /dev/stdin | grep -v "^=="
done
