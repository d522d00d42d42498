"""Latency / throughput of the literal drop-in call (single shape, host pointers in and out, PCIe both ways) from one host thread
and from a pool of threads -- the way msdf-atlas-gen's workers call generateMSDF.  Reported in DESIGN.md (it is never bench.py's value)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import msdfgen_amd as M  # noqa: E402
from msdfgen_amd.shape import ShapeBatch  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "latin.npz"))
batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                   z["colors"].astype(np.int32), z["inverse_y"], [str(n) for n in z["names"]])
M.init(0)
shapes = batch.shapes()
ts = [M.SDFTransformation.from_xf(x) for x in z["xf64"]]


def run(reps):
    out = np.zeros((64, 64, 3), np.float32)
    for rep in range(reps):
        for g in range(94):
            M.generate_msdf(out, shapes[g], ts[g])


run(1)
t0 = time.perf_counter()
run(5)
dt = time.perf_counter()-t0
print("1 host thread : %.1f us per generateMSDF(64x64) call, %.0f glyphs/s (PCIe both ways + ctypes overhead included)" % (1e6*dt/470, 470/dt))
for nt in (4, 16):
    threads = [threading.Thread(target=run, args=(5,)) for _ in range(nt)]
    t0 = time.perf_counter()
    [t.start() for t in threads]
    [t.join() for t in threads]
    dt = time.perf_counter()-t0
    print("%d host threads: %.0f glyphs/s" % (nt, nt*470/dt))
