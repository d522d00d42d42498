for g in 0 3072 1536 1024 768 512; do
  MSDFHIP_PERSISTENT_ROUNDS=1 MSDFHIP_PERSISTENT_GRID=$g timeout 200 python tools/bench_configs.py --reps 8 --only "bench workload,cfg4 real,cfg4: 8192 CJK" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('grid $g', d['config'][:50], d['ms_per_step'], d['kernel_ms_distance_and_post'])"
done
