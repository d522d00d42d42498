# Round 5 (session 2), call 11: knobs tuned at three wavefronts per SIMD, re-checked at four (share of the persistent grid, side-stream priority, short-launch rounds).
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8
one() {
  env "$@" python tools/bench_configs.py --reps 8 --only "bench workload,cfg4 real" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-50s %.3f ms  %s' % (d['config'][:50], d['ms_per_step'], d.get('kernel_ms_distance_and_post')))"
}
(
for v in "A=1" "MSDFHIP_SHARE_GRID=0.6" "MSDFHIP_SHARE_GRID=0.8" "MSDFHIP_SHARE_GRID=1.25" "MSDFHIP_SHARE_GRID=1.6" "MSDFHIP_SHARE_GRID=0" "MSDFHIP_SIDE_PRIORITY=none" "MSDFHIP_SIDE_PRIORITY=high" "MSDFHIP_SMALL_MAX_EDGES=96" "MSDFHIP_SMALL_MAX_EDGES=160" "A=2"; do echo "== $v"; one $v; done
) > gpurun_out/r05_knobs_w4.txt 2>&1
cat gpurun_out/r05_knobs_w4.txt
