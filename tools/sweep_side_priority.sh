show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms_per_step' in d: print('  %-70s %8.3f ms %s'%(d['config'][:70], d['ms_per_step'], d['kernel_ms_distance_and_post']))
"; }
for p in ${PRIOS:-none low one rest low one rest}; do echo "== MSDFHIP_SIDE_PRIORITY=$p"; export MSDFHIP_SIDE_PRIORITY=$p; timeout 300 python tools/bench_configs.py --reps 6 --only "${ONLY:-headline,bench workload,cfg4 real,cfg4: 8192 CJK,cfg3}" 2>/dev/null | show; done
