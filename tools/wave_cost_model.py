"""CPU cost model of k_distance's phase 2 (tools only): wave-level (tile, edge) evaluations with the product's cull + per-texel wave vote,
through tests/hostemu's emu_wave_cost.  order: 0 visit order, 1 fully sorted nearest-first, 2 nearest moved to the front,
16 / 64 sorted within phase-1 chunks of that many edges; + 256: STUDY of a tighter relevance bound for quadratic edges (chord-to-apex slab, hostemu.cpp: relevantWithSlab).

    python tools/wave_cost_model.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from emu import Emu  # noqa: E402
from msdfgen_amd import synth  # noqa: E402
from msdfgen_amd.shape import ShapeBatch, autoframe, distance_mapping  # noqa: E402


def cost(e, shapes, xfs, size, overlap, order):
    tot = np.zeros(16, np.int64)
    for s, xf in zip(shapes, xfs):
        ms, mt = distance_mapping(xf[4], xf[5])
        x6 = np.array([xf[0], xf[1], xf[2], xf[3], ms, mt])
        keep, args = e._shape(s)
        out = np.zeros(16, np.int64)
        e.lib.emu_wave_cost(size, size, *args, x6.ctypes.data_as(C.POINTER(C.c_double)), overlap, order, out.ctypes.data_as(C.POINTER(C.c_long)))
        tot += out
    return tot


def main():
    e = Emu()
    z = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), np.zeros(8192, bool), [str(n) for n in z["names"]])
    pick = list(range(0, 8192, 41))
    dj = [batch.shape(g) for g in pick]
    djx = [z["xf64"][g] for g in pick]
    cj = [synth.cjk_like_shape(20000+i) for i in range(24)]
    cjx = [autoframe(s.bounds(), 48, 48, 4) for s in cj]
    orders = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 16, 64]
    for name, sh, xf, size in (("dejavu64", dj, djx, 64), ("cjk48", cj, cjx, 48)):
        for ov in (0, 1):
            for order in orders:
                t = cost(e, sh, xf, size, ov, order)
                print("%s overlap=%d order=%2d: evals/tile %.2f, survivors/tile %.2f, walks/tile %.2f, second-walk evals/tile %.2f (%.2f passes/tile)"
                      % (name, ov, order, t[0]/t[2], t[1]/t[2], t[3]/t[2], t[4]/t[2], t[5]/t[2]), flush=True)
                print("    evaluations that change no lane's state: %.1f %% (linear %.1f %% of %d, quadratic %.1f %% of %d, cubic %.1f %% of %d); only perpendicular minima changed: %.1f %%"
                      % (100.*t[6]/t[0], 100.*t[11]/max(t[8], 1), t[8], 100.*t[12]/max(t[9], 1), t[9], 100.*t[13]/max(t[10], 1), t[10], 100.*t[14]/t[0]), flush=True)


if __name__ == "__main__":
    main()
