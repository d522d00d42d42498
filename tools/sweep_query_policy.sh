# k_ec_query: which glyphs take the lane-per-candidate chunks (<= maxEdges edges; above a launch load of wideLoad instructions: <= wideMaxEdges)
show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms_per_step' in d: print('  %-70s %8.3f ms %s'%(d['config'][:70], d['ms_per_step'], d['kernel_ms_distance_and_post']))
"; }
ONLY=${ONLY:-headline,bench workload,cfg4: 8192 CJK,cfg4 real,cfg5}
for pol in "340,48,2147483647,128,4e8" "340,0,2147483647,0,4e8" "340,12,2147483647,12,4e8" "340,24,2147483647,24,4e8" "340,24,2147483647,128,4e8" "340,32,2147483647,64,4e8" "340,12,2147483647,128,4e8"; do
  echo "== MSDFHIP_QUERY_POLICY=$pol"; MSDFHIP_QUERY_POLICY=$pol timeout 300 python tools/bench_configs.py --reps 6 --only "$ONLY" 2>/dev/null | show
done
