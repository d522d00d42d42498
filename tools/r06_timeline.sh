# Kernel timeline of one bench step under several environment settings: VARIANTS="name:ENV=1,ENV2=x name2:..." bash tools/r06_call.sh <tag> r06_timeline.sh
#   (a variant "default:" has no settings; comma-separated assignments)
TAG=$1; REPO=$PWD; export TMPDIR=/tmp
for spec in ${VARIANTS:-default:}; do
  v=${spec%%:*}; envs=$(echo "${spec#*:}" | tr , ' ')
  (cd /tmp && env $envs rocprofv3 --kernel-trace -d $REPO/gpurun_out/tl_${TAG}_$v -o tl -- python $REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $REPO/gpurun_out/tl_${TAG}_$v.log 2>&1)
  echo "== $v ($envs)"; python tools/step_timeline.py $(find gpurun_out/tl_${TAG}_$v -name "*.db" | head -1) | tee gpurun_out/${TAG}_timeline_$v.txt
  find gpurun_out/tl_${TAG}_$v -name "*.db" -delete
  env $envs timeout 120 python tools/bench_configs.py --reps 6 --only "bench workload,headline" 2>/dev/null | python tools/ab_show.py /dev/stdin
done
