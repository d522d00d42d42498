"""Where a kernel's scratch (spill) traffic executes: dynamic dwords of scratch_store / scratch_load per source function, from the basic-block counts of
tools/isa_bbcount.py (variants/<name>_map.json + gpurun_out/<tag>_bbcount_raw.json).  MB = dwords x 64 lanes x 4 B per launch.
    python tools/scratch_by_source.py <name> <tag> [kernel-substring ...]"""
import collections
import json
import sys


def main(name, tag, subs):
    m = json.load(open("variants/%s_map.json" % name))
    cnt = json.load(open("gpurun_out/%s_bbcount_raw.json" % tag))["counts"]
    for k in m["kernels"]:
        if subs and not any(s in k["symbol"] for s in subs):
            continue
        per = {"store": collections.Counter(), "load": collections.Counter()}
        blocks = {"store": collections.Counter(), "load": collections.Counter()}
        for i, b in enumerate(k["block_list"]):
            c = cnt[k["base"]+i]
            for mn, _cls, stack in b["ins"]:
                if not mn.startswith("scratch_"):
                    continue
                kind = "store" if "store" in mn else "load"
                width = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}.get(mn.split("_")[-1], 1)
                where = " < ".join("%s:%s" % (s[0].split("<")[0], s[2]) for s in stack[:3])
                per[kind][where] += c*width
                blocks[kind][(b["id"], c)] += width
        print("== %s" % k["symbol"][:60])
        for kind in ("store", "load"):
            total = sum(per[kind].values())
            print("  scratch %ss: %.1f MB per launch" % (kind, total*256/1e6))
            for where, v in per[kind].most_common(10):
                print("     %8.1f MB  %s" % (v*256/1e6, where[:170]))
            print("     blocks (id, executions): dwords  " + ", ".join("%s x%d: %d" % (bid, c, w) for (bid, c), w in sorted(blocks[kind].items(), key=lambda t: -t[0][1]*t[1])[:8]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
