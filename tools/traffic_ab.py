"""HBM-side traffic of the distance pass for a library variant: sums the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the k_distance launches of a bench step.
    python tools/traffic_ab.py <dir with fetch/ and write/ rocprofv3 outputs>      (bash tools/r06_call.sh <tag> traffic:<variant> makes them and calls this)"""
import glob
import os
import sqlite3
import sys


def per_step(d, counter):
    hits = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    cur = sqlite3.connect(hits[0]).cursor()
    rows = cur.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    steps = max(n for name, _, n in rows if "k_ec_fast" in name)
    return {("distance" if "k_distance" in name else name.split("(")[0].replace("void msdfhip::", "")[:24]): 0 for name, _, _ in rows}, rows, steps


def main():
    base = sys.argv[1]
    out = {}
    for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        _, rows, steps = per_step(os.path.join(base, sub), counter)
        agg = {}
        for name, v, _ in rows:
            key = "k_distance (all classes)" if "k_distance" in name else name.split("(")[0].replace("void msdfhip::", "")[:28]
            agg[key] = agg.get(key, 0)+v/steps*1024/1e6
        out[counter] = {k: round(v, 1) for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:5]}
    print(base, out)


if __name__ == "__main__":
    main()
