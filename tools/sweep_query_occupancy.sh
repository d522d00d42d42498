# k_ec_query: wavefronts per SIMD (build variants q3 / q4 = launch bound 3 / 4)
show() { python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms_per_step' in d: print('  %-70s %8.3f ms %s'%(d['config'][:70], d['ms_per_step'], d['kernel_ms_distance_and_post']))
"; }
ONLY=${ONLY:-headline,bench workload,cfg4: 8192 CJK,cfg4 real,cfg5}
echo "== main (bound 2)"; timeout 300 python tools/bench_configs.py --reps 6 --only "$ONLY" 2>/dev/null | show
echo "== q3"; MSDFGEN_HIP_LIB=$PWD/variants/q3.so timeout 300 python tools/bench_configs.py --reps 6 --only "$ONLY" 2>/dev/null | show
echo "== q4, LDS 160,19"; MSDFGEN_HIP_LIB=$PWD/variants/q4.so MSDFHIP_QUERY_LDS=160,19 timeout 300 python tools/bench_configs.py --reps 6 --only "$ONLY" 2>/dev/null | show
