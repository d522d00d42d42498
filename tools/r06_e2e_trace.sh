# Kernel + copy timeline of the last streamed end-to-end call (8-bit atlas): bash tools/r06_call.sh <tag> r06_e2e_trace.sh
TAG=$1; REPO=$PWD; export TMPDIR=/tmp; export GPU_MAX_HW_QUEUES=8
(cd /tmp && rm -rf /tmp/e2e_$TAG && timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/e2e_$TAG -o e2e -- python $REPO/tools/e2e_stream.py 2 > /dev/null 2>&1)
python - <<'PY' $(find /tmp/e2e_$1 -name "*.db" | head -1) | tee gpurun_out/${1}_e2e_timeline.txt
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
ev = [(s, e, n.replace("void msdfhip::", "").replace("msdfhip::", "")[:46]) for n, s, e in cur.execute("select name, start, end from kernels")]
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
for t in tables:
    if "memory_cop" in t.lower():
        cols = [c[1] for c in cur.execute("pragma table_info(%s)" % t)]
        if "start" in cols and "end" in cols:
            namecol = "name" if "name" in cols else cols[0]
            sizecol = "size" if "size" in cols else None
            for r in cur.execute("select start, end, %s%s from %s" % (namecol, ", "+sizecol if sizecol else "", t)):
                ev.append((r[0], r[1], "COPY %s %s" % (str(r[2])[:24], r[3] if sizecol else "")))
            break
ev.sort()
# the last call: events after the last gap of more than 2 ms
cut = 0
for i in range(1, len(ev)):
    if ev[i][0]-max(e[1] for e in ev[max(0, i-40):i]) > 0.6e6:
        cut = i
ev = ev[cut:]
t0 = ev[0][0]
for s, e, n in ev:
    print("%8.3f %8.3f %7.3f  %s" % ((s-t0)/1e6, (e-t0)/1e6, (e-s)/1e6, n))
print("span %.3f ms, %d events" % ((max(e for _, e, _ in ev)-t0)/1e6, len(ev)))
PY
