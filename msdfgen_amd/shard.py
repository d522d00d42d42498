"""Glyph-sharded multi-GPU execution: one process per GPU, a static split of the glyph list (contiguous ranges of equal modelled cost; for
strong scaling of ONE atlas optionally dealt out by cost: partition_dealt), no data-path collective.

Glyphs are independent units, so rank r simply renders glyphs [bounds[r], bounds[r+1]) on its own GPU.  The split is
balanced by a per-glyph cost model fitted to measured kernel times (COST_MODEL below).  Outputs are byte-identical for any world size because no
arithmetic crosses glyphs.  The only optional exchange is assembling the final atlas (`gather_tiles`, one all_gather over RCCL/xGMI).
"""
from typing import Sequence

import numpy as np

from .shape import ShapeBatch


# Microseconds per glyph at 64x64 (msdf, library-default config) = a + b*E + c*C + d*E*C with one coefficient set per KERNEL CLASS of the glyph
# (E edges, C contours; classes as msdf_capi.hip: ensureBuckets sorts them): least squares over (contours, edges) bins of the 8 192 distinct
# DejaVu glyphs, each bin timed on an MI355X -- tools/fit_cost_model.py. Refitted in round 6 for the kernels of rounds 5 / 6 (four wavefronts per SIMD, the
# LDS class up to 5 contours, the combiner's split contour loop): profiles/r06_cost_model.json, rms error of the fit 4.3 percent (round 3's table priced the
# global-workspace class at twice what it costs now). The same numbers live in msdf_capi.hip: glyphCost.
# What the model does NOT say: a shard of ~1 000 glyphs lasts max(sum of these, ~17x the cost of its heaviest glyph -- that glyph's 64 tiles are one
# serial walk each) + ~0.3 ms of launch chain; no cut of the list changes the second term.
COST_MODEL = {
    "one_contour": [0.25625, 0.01190, 0.0, 0.0],        # <= 1 contour: simple-combiner kernel
    "lds": [0.38398, 0.00846, -0.02755, 0.003446],      # 2..5 contours and <= 128 edges: per-contour distances in LDS
    "global": [1.11611, 0.015776, -0.05039, 0.000681],  # the rest: per-contour distances in the global workspace
}


LDS_MAX_CONTOURS, LDS_MAX_EDGES = 5, 128       # msdf_capi.hip: COST_LDS_MAX_CONTOURS, SMALL_MAX_EDGES (the one place these live on this side)


def glyph_class(contours, edges):
    return np.where(contours <= 1, 0, np.where((contours <= LDS_MAX_CONTOURS) & (edges <= LDS_MAX_EDGES), 1, 2))


def glyph_costs(batch: ShapeBatch, width: int, height: int) -> np.ndarray:
    gco, co = batch.glyph_contour_offsets, batch.contour_offsets
    edges = (co[gco[1:]]-co[gco[:-1]]).astype(np.float64)
    contours = (gco[1:]-gco[:-1]).astype(np.float64)
    coef = np.array([COST_MODEL["one_contour"], COST_MODEL["lds"], COST_MODEL["global"]])[glyph_class(contours, edges)]
    per_glyph = coef[:, 0]+coef[:, 1]*edges+coef[:, 2]*contours+coef[:, 3]*edges*contours
    # never below the class's intercept: the fit's negative contour terms are local to the measured range (a 30-contour, 30-edge glyph of the
    # global class would otherwise count as free while taking the most expensive kernel)
    return float(width*height)/4096.*np.maximum(per_glyph, coef[:, 0])


def partition_contiguous(costs: Sequence[float], parts: int) -> np.ndarray:
    """Boundaries b[0..parts] of contiguous ranges whose cost sums are as even as a prefix-sum cut allows (deterministic)."""
    costs = np.asarray(costs, np.float64)
    n = len(costs)
    prefix = np.concatenate([[0.], np.cumsum(costs)])
    total = prefix[-1]
    bounds = [0]
    for r in range(1, parts):
        target = total*r/parts
        i = int(np.searchsorted(prefix, target, side="left"))
        if i > 0 and abs(prefix[i-1]-target) <= abs(prefix[min(i, n)]-target):
            i -= 1
        bounds.append(min(max(i, bounds[-1]), n))
    bounds.append(n)
    return np.array(bounds, np.int64)


def partition_dealt(costs: Sequence[float], parts: int):
    """`parts` index lists (each ascending) that are statistically THE SAME shard: the glyphs in order of modelled cost, dealt out like cards in a
    snake (0 .. N-1, N-1 .. 0, ...). An alternative to the contiguous cut for STRONG scaling of one atlas (bench.py --strong-cut dealt): every shard
    gets every N-th glyph of every weight, whatever the font's blocks look like. Measured on one MI355X (8-way cut of the 8 192-glyph DejaVu set,
    every shard alone): 0.90-1.22 ms per shard against 0.83-1.23 ms for the contiguous cut -- the spread between 1 000-glyph shards is NOT their
    content, so the contiguous cut stays the default. The output is a permutation of the atlas order: `gather_tiles_indexed` puts
    the tiles back. Deterministic (stable sort)."""
    costs = np.asarray(costs, np.float64)
    order = np.argsort(-costs, kind="stable")
    k = np.arange(len(order))
    pos, rnd = k % parts, k//parts
    part = np.where(rnd % 2 == 0, pos, parts-1-pos)
    return [np.sort(order[part == r]) for r in range(parts)]


def shard_indices(batch: ShapeBatch, world: int, width: int, height: int, cut: str = "contiguous"):
    """The glyph indices of every rank for `cut` = "contiguous" (ranges of equal modelled cost) or "dealt" (partition_dealt)."""
    costs = glyph_costs(batch, width, height)
    if cut == "dealt":
        return partition_dealt(costs, world)
    if cut != "contiguous":
        raise ValueError("cut must be 'contiguous' or 'dealt', not %r" % (cut,))
    b = partition_contiguous(costs, world)
    return [np.arange(int(b[r]), int(b[r+1])) for r in range(world)]


def shard(batch: ShapeBatch, xfs: np.ndarray, rank: int, world: int, width: int, height: int):
    """Returns (sub-batch, xfs slice, (begin, end)) owned by `rank`."""
    b = partition_contiguous(glyph_costs(batch, width, height), world)
    lo, hi = int(b[rank]), int(b[rank+1])
    return batch.select(range(lo, hi)), np.asarray(xfs)[lo:hi], (lo, hi)


def gather_tiles(local_tiles, bounds, group=None):
    """Optional final-atlas assembly: every rank contributes its (n_r, H, W, N) tile tensor; returns the (G, H, W, N) tensor on
    every rank.  One padded all_gather (RCCL over xGMI on GPUs, gloo on CPU in the tests)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = [int(bounds[r+1]-bounds[r]) for r in range(world)]
    cap = max(counts) if counts else 0
    pad = torch.zeros((cap,)+tuple(local_tiles.shape[1:]), dtype=local_tiles.dtype, device=local_tiles.device)
    pad[:local_tiles.shape[0]] = local_tiles
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][:counts[r]] for r in range(world)], dim=0)


def gather_tiles_indexed(local_tiles, index_lists, group=None):
    """gather_tiles for shards that are index lists (shard_indices, any cut): returns the (G, H, W, N) tensor in atlas order on every rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    counts = [len(ix) for ix in index_lists]
    cap = max(counts) if counts else 0
    pad = torch.zeros((cap,)+tuple(local_tiles.shape[1:]), dtype=local_tiles.dtype, device=local_tiles.device)
    pad[:local_tiles.shape[0]] = local_tiles
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = torch.empty((sum(counts),)+tuple(local_tiles.shape[1:]), dtype=local_tiles.dtype, device=local_tiles.device)
    for r in range(world):
        out[torch.as_tensor(np.asarray(index_lists[r], np.int64), device=local_tiles.device)] = parts[r][:counts[r]]
    return out
