"""Round-6 experiment: does the device have slack that BATCHES IN FLIGHT TOGETHER can use?  The bench step (digest -> distance -> correction of 8 192 glyphs) ends in a
latency chain (k_ec_scan -> k_ec_query, 0.4 ms at 23 % VALU busy) and starts with one (digest 0.12 ms).  Forms measured, same box, wall clock over K steps:
  serial        one batch object, one stream (the bench line)
  two_batches   two batch objects with outputs of their own on two streams, steps alternate (step k+1's digest + distance start under step k's tail)
  halves        ONE step = the 8 192 glyphs as two 4 096-glyph batch objects on two streams (the same overlap inside a step)
  quarters      ... as four 2 048-glyph batch objects on four streams
    python tools/r06_overlap.py [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--prio", type=int, default=0)
    args = ap.parse_args()
    import torch
    import msdfgen_amd as M
    from bench import load_dejavu
    M.init(0)
    dev = torch.device("cuda", 0)
    batch, xfs, _ = load_dejavu()
    cfg = M.MSDFGeneratorConfig()
    w = h = 64

    def make(idx):
        sub = batch.select(idx)
        gb = M.GlyphBatch(sub, dev)
        return gb, gb.descriptors(xfs[idx], w, h, 3), torch.empty((len(idx), h, w, 3), dtype=torch.float32, device=dev)

    def run(parts, streams, steps, per_step_all):
        """parts: list of (gb, desc, out). per_step_all: every step runs ALL parts (each on its stream); else step k runs part k % len(parts)."""
        def step(k):
            todo = range(len(parts)) if per_step_all else [k % len(parts)]
            for i in todo:
                gb, desc, out = parts[i]
                s = streams[i % len(streams)]
                gb.digest(s)
                gb.generate(M.MODE_MSDF, w, h, descriptors=desc, out=out, stream=s, config=cfg)
        for k in range(4):
            step(k)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(steps):
            step(k)
        torch.cuda.synchronize(dev)
        return 1e3*(time.perf_counter()-t0)/steps

    n = batch.n_glyphs
    allg = np.arange(n)
    streams = [torch.cuda.Stream(dev, priority=args.prio) for _ in range(4)]
    res = {}
    one = make(allg)
    two = make(allg)
    for rep in range(2):
        res.setdefault("serial", []).append(round(run([one], streams[:1], args.steps, True), 4))
        res.setdefault("two_batches", []).append(round(run([one, two], streams[:2], args.steps, False), 4))
    # halves / quarters: contiguous and dealt cuts
    for name, cuts in (("halves_contiguous", [allg[:n//2], allg[n//2:]]), ("halves_dealt", [allg[0::2], allg[1::2]]),
                       ("quarters_dealt", [allg[i::4] for i in range(4)]), ("thirds_dealt", [allg[i::3] for i in range(3)])):
        parts = [make(ix) for ix in cuts]
        for rep in range(2):
            res.setdefault(name, []).append(round(run(parts, streams[:len(parts)], args.steps, True), 4))
        ref = one[2]
        # the pieces' tiles equal the whole batch's (same kernels, other launch shapes)
        same = all(bool((p[2] == ref[torch.as_tensor(ix, device=dev)]).all()) for p, ix in zip(parts, cuts))
        res[name+"_identical"] = same
        for p in parts:
            p[0].close()
    print(json.dumps({"ms_per_step_of_8192_glyphs": res, "steps": args.steps}))


if __name__ == "__main__":
    main()
