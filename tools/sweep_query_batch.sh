# k_ec_query: cooperative items per ticket (MSDFHIP_QUERY_BATCH) on the configs whose correction pass has distance checks
for k in 1 2 4 8 16; do
  echo "== MSDFHIP_QUERY_BATCH=$k"
  MSDFHIP_QUERY_BATCH=$k timeout 300 python tools/bench_configs.py --reps 6 --only "${ONLY:-headline,bench workload,cfg4: 8192 CJK,cfg4 real,cfg5}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms_per_step' in d: print('  %-70s %8.3f ms %s'%(d['config'][:70], d['ms_per_step'], d['kernel_ms_distance_and_post']))
"
done
