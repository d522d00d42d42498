"""Static instruction histogram of one kernel in hipcc's --save-temps assembly (tools only): per basic block the VALU / other
instruction counts and the most frequent opcodes -- where a kernel's per-edge instruction budget goes.

    hipcc <flags of msdfgen_amd/build.py> --save-temps=obj ... -o /tmp/isa/lib.so
    python tools/isa_blocks.py /tmp/isa/*gfx950.s k_distanceILi3ELb0ELb0E [min_valu_per_block]
"""
import re
import sys
from collections import Counter


def main():
    text = open(sys.argv[1]).read()
    key = sys.argv[2]
    floor = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    m = re.search(r"^(_ZN\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(key), text, re.S | re.M)
    if not m:
        raise SystemExit("kernel not found")
    blocks, cur = [], ["entry", 0, 0, Counter()]
    blocks.append(cur)
    for ln in m.group(2).split("\n"):
        t = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            cur = [t.split(":")[0], 0, 0, Counter()]
            blocks.append(cur)
            continue
        if not t or t[0] in ";." :
            continue
        op = t.split()[0]
        cur[1 if op.startswith("v_") else 2] += 1
        cur[3][op] += 1
    print("%s: VALU %d, other %d, blocks %d" % (m.group(1)[:60], sum(b[1] for b in blocks), sum(b[2] for b in blocks), len(blocks)))
    tot = Counter()
    for b in blocks:
        tot.update(b[3])
        if b[1] >= floor:
            print("%-12s valu %4d other %4d  %s" % (b[0], b[1], b[2], " ".join("%s:%d" % kv for kv in b[3].most_common(7))))
    print("top opcodes:", " ".join("%s:%d" % kv for kv in tot.most_common(30)))


if __name__ == "__main__":
    main()
