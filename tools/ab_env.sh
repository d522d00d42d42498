# A/B of an environment knob on the BASELINE configs: bash tools/ab_env.sh VAR value1 value2 ...
VAR=$1; shift
ONLY="headline,bench workload,cfg4: 8192 CJK,cfg5"
for v in "$@"; do
  tag=$(echo $v | tr ',' '_')
  env $VAR=$v python tools/bench_configs.py --reps 5 --only "$ONLY" > gpurun_out/ab_env_$tag.jsonl 2> gpurun_out/ab_env_$tag.err
done
