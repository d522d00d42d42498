"""Kernel-level timing of SMALL batches (what the micro-batcher produces): run under rocprofv3 --kernel-trace.
    python tools/small_batch_probe.py [n_glyphs]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import msdfgen_amd as M  # noqa: E402
from bench import load_latin, tile_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
M.init(0)
latin, xf64 = load_latin()
b, x = tile_batch(latin, xf64, n, offset=11)
gb = M.GlyphBatch(b)
for _ in range(20):
    gb.digest()
    out = gb.generate(3, 64, 64, x)
    torch.cuda.synchronize()
