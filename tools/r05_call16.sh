# Round 5 (session 2), call 16: combiner state without constant initialisation (no per-tile scratch stores of nine doubles) + streaming tile stores: GPU suite, traffic, step time
# (main = both; variants/nt.so = streaming stores only).
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
REPO=$PWD
timeout 400 python -m pytest tests -x -q -m gpu > gpurun_out/r05_gputests_h.log 2>&1; tail -2 gpurun_out/r05_gputests_h.log
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
cd /tmp
for v in main prev; do
  if [ $v = main ]; then export MSDFGEN_HIP_LIB=$REPO/msdfgen_amd/lib/libmsdfgen_hip.so; else export MSDFGEN_HIP_LIB=$REPO/variants/$v.so; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/tr_$v/fetch -o f -- $BENCH > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/tr_$v/write -o w -- $BENCH > /dev/null 2>&1
  python $REPO/tools/traffic_ab.py /tmp/tr_$v
  for i in 1 2; do python $REPO/tools/bench_configs.py --reps 8 --only "bench workload,cfg4: 8192 CJK,cfg4 real,cfg5" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-50s %.3f ms  %s' % (d['config'][:50], d['ms_per_step'], d.get('kernel_ms_distance_and_post')))"; done
done > $REPO/gpurun_out/r05_parked.txt 2>&1
cat $REPO/gpurun_out/r05_parked.txt
