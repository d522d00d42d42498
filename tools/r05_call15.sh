# Round 5 (session 2), call 15: do streaming (nontemporal) stores of the distance tiles keep the 128-VGPR kernels' scratch lines in L2?  FETCH_SIZE / WRITE_SIZE per kernel
# (MB per step, raw counter x 1 KiB) and the step time, main library vs variants/nt.so (-DMSDF_NT_TILE_STORES).
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
REPO=$PWD
BENCH="python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
cd /tmp
for v in main nt; do
  if [ $v = main ]; then export MSDFGEN_HIP_LIB=$REPO/msdfgen_amd/lib/libmsdfgen_hip.so; else export MSDFGEN_HIP_LIB=$REPO/variants/$v.so; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/tr_$v/fetch -o f -- $BENCH > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/tr_$v/write -o w -- $BENCH > /dev/null 2>&1
  python $REPO/tools/traffic_ab.py /tmp/tr_$v
  for i in 1 2; do python $REPO/tools/bench_configs.py --reps 8 --only "bench workload,cfg4: 8192 CJK" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   %-50s %.3f ms  %s' % (d['config'][:50], d['ms_per_step'], d.get('kernel_ms_distance_and_post')))"; done
done > $REPO/gpurun_out/r05_nt_stores.txt 2>&1
cat $REPO/gpurun_out/r05_nt_stores.txt
