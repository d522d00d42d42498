"""Stress (tools only): many back-to-back renders of the CJK-like set, alternating the persistent and the direct launch mapping and
the digest, every result compared bitwise with the first one -- run as the FIRST process on a fresh box (cold code objects, fresh memory)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import msdfgen_amd as M
from msdfgen_amd import synth
from msdfgen_amd.shape import ShapeBatch, autoframe

M.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
base = [synth.cjk_like_shape(20000+i) for i in range(512)]
cj = ShapeBatch.from_shapes([base[i % 512] for i in range(n)])
cx = np.stack([autoframe(s.bounds(), 48, 48, 4) for s in base])[np.arange(n) % 512]
gb = M.GlyphBatch(cj)
out = torch.empty((n, 48, 48, 3), dtype=torch.float32, device="cuda")
desc = gb.descriptors(cx, 48, 48, 3)
ref = None
bad = 0
for i in range(iters):
    os.environ["MSDFHIP_PERSISTENT_ROUNDS"] = (sys.argv[3] if len(sys.argv) > 3 else "0,8").split(",")[i % len((sys.argv[3] if len(sys.argv) > 3 else "0,8").split(","))]
    for _ in range(3):                                   # several steps in flight, as bench_configs does
        gb.digest()
        gb.generate(3, 48, 48, descriptors=desc, out=out)
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint32)
    if ref is None:
        ref = got.copy()
    else:
        d = int((got != ref).sum())
        bad += d
        if d:
            print("iteration %d (%s): %d values differ" % (i, os.environ["MSDFHIP_PERSISTENT_ROUNDS"], d), flush=True)
print("stress: %d iterations, %d differing values" % (iters, bad), flush=True)
