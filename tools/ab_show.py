import json, sys
for f in sys.argv[1:]:
    print("==", f)
    for l in open(f):
        d = json.loads(l)
        if "ms_per_step" in d:
            print("%-72s %8.3f ms  dist/post %s" % (d["config"][:72], d["ms_per_step"], d["kernel_ms_distance_and_post"]))
