show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms_per_step' in d: print('  %-70s %8.3f ms %s'%(d['config'][:70], d['ms_per_step'], d['kernel_ms_distance_and_post']))
"; }
ONLY="bench workload,cfg4: 8192 CJK,cfg4 real,cfg5"
for q in 160,24 600,24 1000,24 1000,8; do echo "== MSDFHIP_QUERY_LDS=$q"; MSDFHIP_QUERY_LDS=$q timeout 300 python tools/bench_configs.py --reps 6 --only "$ONLY" 2>/dev/null | show; done
