"""Turns the two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE) of `bench.py` into profiles/pmc_traffic.json:
HBM bytes per launch of the dominant kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes:
FETCH_SIZE / WRITE_SIZE are in units of 1024 B, and on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read
(so the read side is doubled; for this kernel reads are <5 % of the traffic either way).

    python tools/pmc_traffic.py <fetch.db> <write.db> <glyphs_per_gpu> <tile> [kernel substring]
"""
import json
import os
import sqlite3
import sys


def per_dispatch(db, counter, kernel):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, avg(value) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    for name, avg in rows:
        if kernel in name:
            return name, float(avg)
    raise SystemExit("kernel %r not found in %s" % (kernel, db))


def main():
    fetch_db, write_db, glyphs, tile = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    kernel = sys.argv[5] if len(sys.argv) > 5 else "k_distance"
    name, fetch = per_dispatch(fetch_db, "FETCH_SIZE", kernel)
    _, write = per_dispatch(write_db, "WRITE_SIZE", kernel)
    out = {"kernel": name.split("(")[0], "glyphs_per_gpu": glyphs, "tile": [tile, tile], "FETCH_SIZE_raw": fetch, "WRITE_SIZE_raw": write,
           "hbm_read_bytes": 2*fetch*1024, "hbm_write_bytes": write*1024, "hbm_bytes_per_launch": 2*fetch*1024+write*1024,
           "correction": "x1024 B per counter unit; FETCH_SIZE x2 (gfx950 half-count of wide coalesced reads, MI355X_MICROARCH.md)"}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
