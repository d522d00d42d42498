# kernel timeline of the single-shape host-pointer call (one thread, 300 calls): where do the ~150 us go?
REPO=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/single_stats -o single -- python $REPO/tools/host_call_latency.py --threads 1 --calls 300 --leaders 2 > $REPO/gpurun_out/single_stats.log 2>&1
cd $REPO
python tools/rocpd_summary.py $(find gpurun_out/single_stats -name "*.db") > gpurun_out/single_kernel_stats.txt
head -16 gpurun_out/single_kernel_stats.txt
python - <<'PY'
import sqlite3, glob
db = glob.glob("gpurun_out/single_stats/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
# group into calls: a call starts at k_prep_records
calls, curc = [], []
for n, s, e in rows:
    if "k_prep_records" in n and curc:
        calls.append(curc); curc = []
    curc.append((n.split("(")[0][-40:], s, e))
calls.append(curc)
mid = calls[len(calls)//2: len(calls)//2+3]
for c in mid:
    t0 = c[0][1]
    print(" | ".join("%s +%.1f..%.1f" % (n[-22:], (s-t0)/1e3, (e-t0)/1e3) for n, s, e in c))
import statistics
spans = [(c[-1][2]-c[0][1])/1e3 for c in calls[10:] if len(c) >= 5]
busy = [sum(e-s for _, s, e in c)/1e3 for c in calls[10:] if len(c) >= 5]
print("kernels per call:", statistics.median(len(c) for c in calls[10:]), "first-kernel-start to last-kernel-end us: median %.1f; sum of kernel durations: median %.1f" % (statistics.median(spans), statistics.median(busy)))
PY
