"""End-to-end host pipeline (bench.py's end_to_end) as a function of the chunk size (glyphs per chunk; 0 = automatic)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import msdfgen_amd as M  # noqa: E402
from msdfgen_amd import lib as L  # noqa: E402
from bench import load_dejavu  # noqa: E402

M.init(0)
batch, xfs, _ = load_dejavu()
n, w, h = batch.n_glyphs, 64, 64
tiles = M.host_alloc((n, h, w, 3))
cols = 128
atlas = M.host_alloc(((n+cols-1)//cols*h, cols*w, 3), np.uint8)
offs = np.array([((g//cols)*h*cols*w+(g % cols)*w)*3 for g in range(n)], np.int64)
hb = M.HostBatch(batch)
for chunk in [int(a) for a in sys.argv[1:]] or [0, 512, 1024, 2048, 4096, 8192]:
    L.load().msdfhip_set_pipeline_chunk(chunk)
    tf, tb = [], []
    for rep in range(4):
        t0 = time.perf_counter()
        hb.generate_host(M.MODE_MSDF, w, h, xfs, out=tiles)
        t1 = time.perf_counter()
        hb.generate_bytes_host(M.MODE_MSDF, w, h, xfs, atlas, offs, cols*w*3)
        t2 = time.perf_counter()
        if rep:
            tf.append(t1-t0), tb.append(t2-t1)
    print(json.dumps({"chunk": chunk, "float_ms": round(1e3*float(np.median(tf)), 3), "bytes_ms": round(1e3*float(np.median(tb)), 3)}), flush=True)
hb.close()
