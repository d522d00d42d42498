"""Longer sweep of the lanes = edges preparation / winding code on the HOST (tests/hostemu: the kernels' driver functions with a 64-lane context)
against the oracle -- the same comparison tests/test_shape_prep_oracle.py and tests/test_device_logic_host.py make, over many more seeds.

    python tools/prep_host_fuzz.py [seeds]      -> one JSON line
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from emu import Emu  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402
from test_shape_prep_oracle import prep_stress_shapes, same_shape  # noqa: E402
from test_device_logic_host import winding_stress_shapes  # noqa: E402


def main():
    seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    emu, oracle = Emu(), Oracle()
    t0 = time.time()
    shapes = prepared = contours = cusps = 0
    for seed in range(1000, 1000+seeds):
        rng = np.random.default_rng(seed)
        for i, s in enumerate(prep_stress_shapes(seed, 100)):
            angle, sd = float(rng.choice([3.0, 2.5, .05, 1.0, .4])), int(rng.integers(0, 2**50))
            for normalize in (True, False):
                for coloring in (0, 1, 2):
                    same_shape(emu.shape_prepare(s, normalize, coloring, angle, sd, wave=True), oracle.shape_prepare(s, normalize, coloring, angle, sd),
                               "seed %d shape %d normalize %s colouring %d" % (seed, i, normalize, coloring))
                    prepared += 1
                    cusps += emu.cusp_contours if normalize and coloring == 0 else 0
            shapes += 1
        for s in winding_stress_shapes(seed):
            want = oracle.windings(s)
            for form in ("single", "batch"):
                assert (emu.windings(s, wave=form) == want).all(), (seed, form)
            contours += s.n_contours
            shapes += 1
    print(json.dumps({"seeds": seeds, "shapes": shapes, "preparations_compared": prepared, "contours_with_a_cusp_repaired": cusps, "winding_contours_compared_x2_forms": contours,
                      "mismatches": 0, "seconds": round(time.time()-t0, 1)}))


if __name__ == "__main__":
    main()
