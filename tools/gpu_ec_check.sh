# quick look at the correction pass: profile of k_ec_query (variants/profquery.so) + the configs with distance checks + the GPU tests
for w in bench cjk logo; do MSDFGEN_HIP_LIB=$PWD/variants/profquery.so timeout 200 python tools/profile_query.py $w 2>&1 | tail -1 ; done | tee gpurun_out/${1:-r03}_profquery.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['workload'], 'coop item', round(d['cycles_per_cooperative_item']), 'chunk item', round(d['cycles_per_chunk_item']), json.dumps(d['cooperative_query']))
"
timeout 300 python tools/bench_configs.py --reps 6 --only "${ONLY:-headline,bench workload,cfg4: 8192 CJK,cfg4 real,cfg5}" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'ms_per_step' in d: print('  %-70s %8.3f ms %s'%(d['config'][:70], d['ms_per_step'], d['kernel_ms_distance_and_post']))
"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
