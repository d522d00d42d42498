show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms_per_step' in d: print('  %-70s %8.3f ms %s'%(d['config'][:70], d['ms_per_step'], d['kernel_ms_distance_and_post']))
"; }
ONLY="bench workload,cfg4 real"
for kv in "A=1" "MSDFHIP_RES_LDS_BUDGET=9216" "MSDFHIP_RES_LDS_BUDGET=10752" "MSDFHIP_RES_LDS_BUDGET=16384" "MSDFHIP_RES_LDS_BUDGET=20480" "MSDFHIP_PERSISTENT_ROUNDS=3" "A=1"; do echo "== $kv"; env $kv timeout 300 python tools/bench_configs.py --reps 6 --only "$ONLY" 2>/dev/null | show; done
