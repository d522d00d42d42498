"""How much of the bench workload's distance pass is each glyph class's tail? Times the step with one class REMOVED from the glyph set
(the classes run concurrently: removing one shows what the others cost without it).   python tools/class_ablation.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import msdfgen_amd as M
    from bench import load_dejavu, step_ms
    M.init(0)
    lib = M.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev)
    batch, xf64, _ = load_dejavu()
    gco, co = batch.glyph_contour_offsets, batch.contour_offsets
    e = co[gco[1:]]-co[gco[:-1]]
    c = gco[1:]-gco[:-1]
    cls = np.where(c <= 1, 0, np.where((c <= 7) & (e <= 128), 1, 2))
    cfg = M.MSDFGeneratorConfig()
    sets = {"all": np.arange(batch.n_glyphs), "without one-contour": np.nonzero(cls != 0)[0], "without LDS class": np.nonzero(cls != 1)[0],
            "without global class": np.nonzero(cls != 2)[0], "only LDS class": np.nonzero(cls == 1)[0], "only global class": np.nonzero(cls == 2)[0],
            "only one-contour": np.nonzero(cls == 0)[0],
            "without the 16 heaviest (E*C)": np.argsort((e*np.maximum(c, 1)))[:-16]}
    for name, idx in sets.items():
        idx = np.sort(idx)
        sub = batch.select([int(g) for g in idx])
        ms, kd, kc = step_ms(M, torch, lib, dev, stream, sub, xf64[idx], 64, 64, cfg, 8)
        print(json.dumps({"set": name, "glyphs": int(len(idx)), "ms_per_step": round(ms, 3), "distance_ms": round(kd, 3), "correction_ms": round(kc, 3)}), flush=True)


if __name__ == "__main__":
    main()
