"""Per-HIP-API-call time summary from a rocprofv3 --hip-trace rocpd database (run on the GPU box; the database is too big to ship).
    python tools/hip_api_summary.py <results.db>"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = next((n for n in names if n == "regions"), None) or next((n for n in names if n.startswith("regions")), None)
if view is None:
    print("tables:", names)
    sys.exit(0)
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), max(end-start) from %s group by name order by 3 desc limit 25" % view).fetchall()
print("%-44s %8s %12s %10s %10s" % ("api", "calls", "total_ms", "avg_us", "max_us"))
for n, c, t, a, m in rows:
    print("%-44s %8d %12.3f %10.2f %10.2f" % (str(n)[:44], c, t/1e6, a/1e3, m/1e3))
