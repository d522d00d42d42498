import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
from bench import load_latin, tile_batch
from oracle.pyoracle import Ref, Oracle
impl = Ref() if Ref.available() else Oracle()
latin, xf64 = load_latin()
b, x = tile_batch(latin, xf64, 94*8)
shapes = b.shapes()
for th in (1, 8, 32, 64, 128, 256):
    n = min(len(shapes), max(94, th*12))
    idx = [i % len(shapes) for i in range(n)]
    _, secs = impl.generate_batch_timed([shapes[i] for i in idx], 3, 64, 64, x[idx], threads=th)
    print("threads %3d: %6.0f glyphs/s (%d glyphs, %.2f s) -> %.1f per thread" % (th, n/secs, n, secs, n/secs/th))
