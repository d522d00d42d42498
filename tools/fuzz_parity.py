"""Randomized parity sweep of the HIP path against the oracle, long form (run on the GPU box; `pytest -m gpu` runs a bounded sweep of the
same generator, tests/test_gpu_parity.py::test_fuzz_sweep_vs_oracle).  Prints one JSON line with the number of texel values compared,
the number differing bitwise and the worst |delta|; exit status 1 if any |delta| exceeds 1e-5.

    python tools/fuzz_parity.py [--shapes 1500] [--seed 1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--single", action="store_true", help="every shape through its own generate*() call (the fused single-call launch) instead of batches")
    args = ap.parse_args()
    import fuzzlib
    r = fuzzlib.run(args.shapes, args.seed, single=args.single)
    print(json.dumps(r))
    sys.exit(1 if r["max_abs_delta"] > 1e-5 else 0)


if __name__ == "__main__":
    main()
