# k_ec_query: cost of an edge in a lane-per-candidate chunk relative to a cooperative round (first number of MSDFHIP_QUERY_POLICY), and the edge bounds
show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms_per_step' in d: print('  %-70s %8.3f ms %s'%(d['config'][:70], d['ms_per_step'], d['kernel_ms_distance_and_post']))
"; }
ONLY=${ONLY:-headline,bench workload,cfg4: 8192 CJK,cfg4 real}
for pol in ${POLICIES:-"340,48,2147483647,128,4e8" "200,48,2147483647,128,4e8" "120,48,2147483647,128,4e8" "60,48,2147483647,128,4e8" "120,64,2147483647,128,4e8" "120,32,2147483647,128,4e8"}; do
  echo "== MSDFHIP_QUERY_POLICY=$pol"; MSDFHIP_QUERY_POLICY=$pol timeout 300 python tools/bench_configs.py --reps 6 --only "$ONLY" 2>/dev/null | show
done
