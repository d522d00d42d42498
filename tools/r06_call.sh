# One gpurun call of round 6 (a parameterised replacement of the per-call scripts of round 5): bash tools/r06_call.sh <tag> <steps...>
#   steps: tests | bench | smoke | configs[:only-list] | ab:<variant>[,<variant>...] | traffic:<variant>[,...] | hostcalls | san | prof | any other word = a script under tools/ run with the tag
#   (traffic: HBM FETCH_SIZE / WRITE_SIZE of the distance pass per library variant, tools/traffic_ab.py -- the A/B that round 5 ran from one-off scripts)
TAG=$1; shift
mkdir -p gpurun_out
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
for STEP in "$@"; do
  case "$STEP" in
    tests) timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File \|dist-packages" | tail -40 > gpurun_out/${TAG}_gputests.log; tail -4 gpurun_out/${TAG}_gputests.log ;;
    bench) timeout 300 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; cut -c1-600 gpurun_out/${TAG}_bench.json ;;
    smoke) timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log ;;
    configs*) ONLY=${STEP#configs}; ONLY=${ONLY#:}
       timeout 500 python tools/bench_configs.py --reps 8 ${ONLY:+--only "$ONLY"} > gpurun_out/${TAG}_configs.jsonl 2> gpurun_out/${TAG}_configs.err; wc -l gpurun_out/${TAG}_configs.jsonl ;;
    ab:*) ONLY=${AB_ONLY:-"headline,bench workload,cfg4: 8192 CJK,cfg4 real,cfg5"}
       for v in main $(echo ${STEP#ab:} | tr , ' '); do
         if [ $v = main ]; then L=; else L=$PWD/variants/$v.so; fi
         MSDFGEN_HIP_LIB=$L timeout 300 python tools/bench_configs.py --reps ${AB_REPS:-6} --only "$ONLY" > gpurun_out/${TAG}_ab_$v.jsonl 2> gpurun_out/${TAG}_ab_$v.err
       done
       python tools/ab_show.py gpurun_out/${TAG}_ab_*.jsonl 2>/dev/null || true ;;
    traffic:*) export TMPDIR=/tmp; REPO=$PWD
       for v in main $(echo ${STEP#traffic:} | tr , ' '); do
         if [ $v = main ]; then L=; else L=$REPO/variants/$v.so; fi
         for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
           (cd /tmp && MSDFGEN_HIP_LIB=$L rocprofv3 --kernel-trace --pmc ${c#*:} -d /tmp/tr_${TAG}_$v/${c%%:*} -o t -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1)
         done
         python tools/traffic_ab.py /tmp/tr_${TAG}_$v | tee -a gpurun_out/${TAG}_traffic_ab.txt
       done ;;
    hostcalls) timeout 200 python tools/host_call_latency.py --threads 1,4,64 --leaders 4 > gpurun_out/${TAG}_host_calls.jsonl 2> gpurun_out/${TAG}_host_calls.err; cut -c1-420 gpurun_out/${TAG}_host_calls.jsonl ;;
    san) bash tests/sanitize/run.sh both > gpurun_out/${TAG}_sanitizers.txt 2>&1; tail -5 gpurun_out/${TAG}_sanitizers.txt ;;
    prof) WITH_CONFIGS=${WITH_CONFIGS:-0} timeout 400 bash tools/profile_round.sh $TAG $(cat .commit 2>/dev/null || echo "?") > gpurun_out/${TAG}_profile_round.log 2>&1; tail -2 gpurun_out/${TAG}_profile_round.log | cut -c1-200 ;;
    *) bash tools/$STEP $TAG ;;
  esac
done
