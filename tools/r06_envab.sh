# A/B of an environment knob of the library: kernel trace of the bench step + the configs per setting.   ENVVAR=MSDFHIP_X VALUES="0 1" bash tools/r06_call.sh <tag> r06_envab.sh
TAG=$1; REPO=$PWD; export TMPDIR=/tmp
for val in ${VALUES:-0 1}; do
  (cd /tmp && rm -rf /tmp/kt_${TAG}_$val && env $ENVVAR=$val rocprofv3 --kernel-trace --stats -d /tmp/kt_${TAG}_$val -o kt -- python $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2>&1)
  echo "== $ENVVAR=$val"; python tools/rocpd_summary.py $(find /tmp/kt_${TAG}_$val -name "*.db") 2>/dev/null | grep -E "${KERNELS:-k_ec_query|k_ec_scan}" | cut -c1-60,73-140
  env $ENVVAR=$val timeout 300 python tools/bench_configs.py --reps 8 --only "${ONLY:-headline,bench workload,cfg4: 8192 CJK,cfg4 real,cfg5}" 2>/dev/null | python tools/ab_show.py /dev/stdin | grep -v "^=="
done
