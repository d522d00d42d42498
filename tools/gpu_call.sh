# One gpurun call of round 3: parity tests, A/B of build variants (variants/<name>.so) on the BASELINE configs, optional profile script.
# Usage: bash tools/gpu_call.sh <tag> "<variants>" [profile-script args...]
TAG=$1; VARIANTS=$2; shift 2
timeout 600 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File \|dist-packages" | tail -60 > gpurun_out/${TAG}_gputests.log
tail -4 gpurun_out/${TAG}_gputests.log
ONLY=${ONLY:-"headline,bench workload,cfg4: 8192 CJK,cfg4 real,cfg5,DejaVu glyphs msdf 64x64, simple"}
timeout 300 python tools/bench_configs.py --reps 6 --only "$ONLY" > gpurun_out/${TAG}_ab_main.jsonl 2> gpurun_out/${TAG}_ab_main.err
for v in $VARIANTS; do
  MSDFGEN_HIP_LIB=$PWD/variants/$v.so timeout 300 python tools/bench_configs.py --reps 6 --only "$ONLY" > gpurun_out/${TAG}_ab_$v.jsonl 2> gpurun_out/${TAG}_ab_$v.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_ab_*.jsonl")):
    for l in open(f):
        try: d=json.loads(l)
        except Exception: continue
        print("%-28s %-62s %8.3f ms  %s" % (f.split("_ab_")[1][:-6], d["config"][:62], d["ms_per_step"], d["kernel_ms_distance_and_post"]))
PY
if [ -n "$1" ]; then bash "$@"; fi
