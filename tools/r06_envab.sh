# Same-box A/B of an environment knob on the BASELINE configs: ENVAB="MSDFHIP_X=1" [AB_ONLY=...] [AB_REPS=6] [AB_ROUNDS=2] bash tools/r06_call.sh <tag> r06_envab.sh
#   -> gpurun_out/<tag>_envab_{default,knob}<round>.jsonl, table on stdout (default = knob unset)
TAG=$1
ONLY=${AB_ONLY:-"headline,bench workload,cfg4: 8192 CJK,cfg4 real,cfg5"}
for r in $(seq 1 ${AB_ROUNDS:-2}); do
  for v in default knob; do
    if [ $v = knob ]; then P="env $ENVAB"; else P=""; fi
    $P timeout 300 python tools/bench_configs.py --reps ${AB_REPS:-6} --only "$ONLY" > gpurun_out/${TAG}_envab_$v$r.jsonl 2> gpurun_out/${TAG}_envab_$v$r.err
  done
done
python tools/ab_show.py gpurun_out/${TAG}_envab_*.jsonl 2>/dev/null || true
