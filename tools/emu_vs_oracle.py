"""Host emulation of the device headers (tests/hostemu) against the oracle on nested / overlapping / many-contour shapes (tools only;
a broader sweep than tests/test_device_logic_host.py, used while restructuring the combiner).  Prints the number of differing values."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from emu import Emu  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402
from msdfgen_amd import synth  # noqa: E402
from msdfgen_amd.shape import ShapeBatch, autoframe  # noqa: E402


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def main():
    e, o = Emu(), Oracle()
    total = 0
    s = synth.logo_shape(5)
    for mode in (1, 2, 3, 4):
        xf = autoframe(s.bounds(), 64, 64, 4)
        n = int((bits(e.generate(s, mode, 64, 64, xf)) != bits(o.generate(s, mode, 64, 64, xf))).sum())
        total += n
        print("logo mode", mode, n)
    z = np.load(os.path.join(ROOT, "tests", "golden", "dejavu8192.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), np.zeros(8192, bool), [str(n) for n in z["names"]])
    nc = np.diff(batch.glyph_contour_offsets)
    pick = list(np.argsort(nc)[-12:])+list(range(0, 8192, 257))
    bad = 0
    for g in pick:
        sh = batch.shape(int(g))
        for ov in (True, False):
            n = int((bits(e.generate(sh, 3, 48, 48, z["xf48"][g], overlap=ov)) != bits(o.generate(sh, 3, 48, 48, z["xf48"][g], overlap=ov))).sum())
            bad += n
            if n:
                print("glyph", g, batch.names[g], nc[g], ov, n)
    total += bad
    print("dejavu sample", len(pick), "glyphs, differing values:", bad)
    for seed in range(6):
        sh = synth.random_shape(100+seed, n_contours=6, spread=.25)   # heavily overlapping blobs
        xf = autoframe(sh.bounds(), 40, 40, 4)
        for mode in (1, 2, 3, 4):
            for ecd in (1, 2):
                n = int((bits(e.generate(sh, mode, 40, 40, xf, ec_dist=ecd)) != bits(o.generate(sh, mode, 40, 40, xf, ec_dist=ecd))).sum())
                total += n
                if n:
                    print("overlap seed", seed, mode, n)
    for seed in range(4):
        sh = synth.cjk_like_shape(30000+seed)
        xf = autoframe(sh.bounds(), 48, 48, 4)
        n = int((bits(e.generate(sh, 3, 48, 48, xf)) != bits(o.generate(sh, 3, 48, 48, xf))).sum())
        total += n
        if n:
            print("cjk", seed, n)
    print("TOTAL differing values:", total)


if __name__ == "__main__":
    main()
