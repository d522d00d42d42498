"""Timeline of ONE msdfhip_batch_generate_host call (8 192 distinct glyphs, float tiles) from a rocprofv3 kernel + memory-copy trace: per
chunk, when its kernels and its copy back ran.   rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o pt -- python tools/pipeline_timeline.py run
                                                 python tools/pipeline_timeline.py report DIR"""
import glob
import os
import sqlite3
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import msdfgen_amd as M
    from bench import load_dejavu
    M.init(0)
    batch, xfs, _ = load_dejavu()
    tiles = M.host_alloc((batch.n_glyphs, 64, 64, 3))
    hb = M.HostBatch(batch)
    mode = sys.argv[2] if len(sys.argv) > 2 else "float"
    for _ in range(3):
        if mode == "float":
            hb.generate_host(M.MODE_MSDF, 64, 64, xfs, out=tiles)
        else:
            atlas = M.host_alloc((64*64, 128*64, 3), np.uint8)
            offs = np.array([((g//128)*64*128*64+(g % 128)*64)*3 for g in range(batch.n_glyphs)], np.int64)
            hb.generate_bytes_host(M.MODE_MSDF, 64, 64, xfs, atlas, offs, 128*64*3)
    hb.close()


def report(d):
    db = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    ks = cur.execute("select name, start, end from kernels order by start").fetchall()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    cp = []
    for t in tables:
        if "memory_cop" in t.lower():
            cols = [r[1] for r in cur.execute("pragma table_info(%s)" % t)]
            if "start" in cols and "end" in cols:
                size = "size" if "size" in cols else "0"
                cp = cur.execute("select start, end, %s from %s order by start" % (size, t)).fetchall()
                break
    # the last call: after the last big gap between kernels
    starts = [k[1] for k in ks]
    cut = max(range(1, len(ks)), key=lambda i: starts[i]-ks[i-1][2] if i > len(ks)*2//3 else -1)
    ks = ks[cut:]
    t0 = ks[0][1]
    cp = [c for c in cp if c[0] >= t0-2_000_000]
    print("kernels of the last call: %d, span %.2f ms; copies: %d" % (len(ks), (ks[-1][2]-t0)/1e6, len(cp)))
    for n, s, e in ks:
        name = n.replace("void msdfhip::", "").split("(")[0][:34]
        if (e-s) > 30000:
            print("  %-36s %8.3f .. %8.3f ms  (%.3f)" % (name, (s-t0)/1e6, (e-t0)/1e6, (e-s)/1e6))
    for s, e, size in cp:
        if (e-s) > 100000:
            print("  copy %-30s %8.3f .. %8.3f ms  (%.3f ms, %.1f MB)" % ("", (s-t0)/1e6, (e-t0)/1e6, (e-s)/1e6, (size or 0)/1e6))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
