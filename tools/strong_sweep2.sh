# Strong-scaling rehearsal under a knob: bash tools/strong_sweep2.sh <tag> <ENVVAR> v1 v2 ...
TAG=$1; VAR=$2; shift 2
for t in "$@"; do
  for set in dejavu cjk_like; do
    env $VAR=$t python bench.py --strong --strong-set $set --steps 12 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_strong_${set}_$t.json 2> gpurun_out/${TAG}_strong_${set}_$t.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_strong_*.json")):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, "unreadable", e); continue
    for name,row in d.get("strong_scaling",{}).items():
        if not isinstance(row,dict): continue
        print(f.split("_strong_")[1][:-5], name, "whole %.3f ms" % row["ms_whole_set"], " ".join("%s: eff %.3f max %.3f" % (k, v["efficiency"], max(v["ms_per_shard"])) for k,v in row.items() if k.startswith("x")))
PY
