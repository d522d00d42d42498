# bench step under values of one env knob: bash tools/knob_sweep.sh <tag> <ENVVAR> v1 v2 ... (uses tools/bench_configs.py on $ONLY)
TAG=$1; VAR=$2; shift 2
ONLY=${ONLY:-"bench workload,cfg4 real,cfg4: 8192 CJK,headline"}
for t in "$@"; do
  env $VAR=$t timeout 300 python tools/bench_configs.py --reps 8 --only "$ONLY" > gpurun_out/${TAG}_$t.jsonl 2> gpurun_out/${TAG}_$t.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_*.jsonl")):
    for l in open(f):
        try: d=json.loads(l)
        except Exception: continue
        print("%-12s %-62s %8.3f ms  %s" % (f.split("${TAG}_")[1][:-6], d["config"][:62], d["ms_per_step"], d["kernel_ms_distance_and_post"]))
PY
