"""Kernel timeline of ONE shard of BASELINE config 4 (strong scaling): python tools/shard_trace.py <set: dejavu|cjk_like> <parts> <rank> [steps]
Run under rocprofv3 --kernel-trace (tools/shard_trace.sh) to see which kernels the shard's step consists of and how long each lasts."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name, parts, rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    import torch
    import msdfgen_amd as M
    from bench import config4_sets, step_ms
    from msdfgen_amd.shard import partition_contiguous, glyph_costs
    M.init(0)
    lib = M.load()
    dev = torch.device("cuda", 0)
    batch, xfs = config4_sets()[name]
    b = partition_contiguous(glyph_costs(batch, 48, 48), parts)
    lo, hi = int(b[rank]), int(b[rank+1])
    sub = batch.select(range(lo, hi))
    gco, co = sub.glyph_contour_offsets, sub.contour_offsets
    e = co[gco[1:]]-co[gco[:-1]]
    c = gco[1:]-gco[:-1]
    ms, kd, kc = step_ms(M, torch, lib, dev, torch.cuda.current_stream(dev), sub, xfs[lo:hi], 48, 48, M.MSDFGeneratorConfig(), steps)
    print("shard %d/%d of %s: glyphs [%d, %d) max edges %d max contours %d (heaviest E*C %d) -> %.3f ms/step (distance %.3f, correction %.3f)" % (
        rank, parts, name, lo, hi, e.max(), c.max(), (e*c).max(), ms, kd, kc))


if __name__ == "__main__":
    main()
