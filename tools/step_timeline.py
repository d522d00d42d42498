"""Timeline of ONE bench step from a rocprofv3 --kernel-trace result (rocpd sqlite): start / end of every kernel relative to the step's first
launch (k_prep_records), for the last complete step of the run.
    python tools/step_timeline.py gpurun_out/kt_<tag>/.../*_results.db"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    firsts = [i for i, r in enumerate(rows) if "k_prep_records" in r[0]]
    if len(firsts) < 2:
        print("no complete step in", path)
        return
    a, b = firsts[-2], firsts[-1]                                # the last step that is followed by another one's digest
    t0 = rows[a][1]
    print("%-44s %9s %9s %9s" % ("kernel", "start ms", "end ms", "ms"))
    for name, s, e in rows[a:b]:
        print("%-44s %9.3f %9.3f %9.3f" % (name.replace("void msdfhip::", "")[:44], (s-t0)/1e6, (e-t0)/1e6, (e-s)/1e6))
    print("step: %.3f ms (first start to last end), next step starts at %.3f" % ((max(r[2] for r in rows[a:b])-t0)/1e6, (rows[b][1]-t0)/1e6))


if __name__ == "__main__":
    main(sys.argv[1])
