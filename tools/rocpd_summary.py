"""Summarises a rocprofv3 (rocpd sqlite) result: per-kernel time statistics and, if present, PMC counter sums per kernel.
    python tools/rocpd_summary.py gpurun_out/prof_stats/r01_results.db [...]  > profiles/rNN_*.txt"""
import sqlite3
import sys


def main(paths):
    for path in paths:
        con = sqlite3.connect(path)
        cur = con.cursor()
        print("== %s" % path)
        rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
        total = sum(r[2] for r in rows) or 1
        print("%-72s %6s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "avg_ms", "min_ms", "max_ms", "pct"))
        for name, n, tot, avg, mn, mx in rows:
            print("%-72s %6d %12.4f %12.4f %12.4f %12.4f %6.2f" % (name[:72], n, tot/1e6, avg/1e6, mn/1e6, mx/1e6, 100.*tot/total))
        try:
            cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            if cols:
                q = "select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, counter_name order by 1, 2"
                rows = cur.execute(q).fetchall()
                if rows:
                    print("%-72s %-24s %6s %18s %18s" % ("kernel", "counter", "disp", "sum", "avg/dispatch"))
                    for k, c, n, s, a in rows:
                        print("%-72s %-24s %6d %18.1f %18.1f" % (k[:72], c, n, s, a))
        except sqlite3.Error as e:
            print("(no counters: %s)" % e)
        con.close()


if __name__ == "__main__":
    main(sys.argv[1:])
