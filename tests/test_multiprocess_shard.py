"""N>1 path on CPU: two gloo ranks shard a glyph list exactly like bench.py --gpus N does, agree on the partition, and assemble
the final atlas with the optional gather (all_gather).  No GPU: the per-rank "render" is a stand-in that tags tiles."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from msdfgen_amd.shape import ShapeBatch
    from msdfgen_amd.shard import shard, partition_contiguous, glyph_costs, gather_tiles
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(GOLDEN, "latin.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), z["inverse_y"], [str(n) for n in z["names"]])
    sub, xfs, (lo, hi) = shard(batch, z["xf64"], rank, world, 64, 64)
    bounds = partition_contiguous(glyph_costs(batch, 64, 64), world)
    # stand-in tiles: glyph index in channel 0, edge count in channel 1
    tiles = torch.zeros((hi-lo, 4, 4, 3))
    for i in range(hi-lo):
        tiles[i, ..., 0] = lo+i
        tiles[i, ..., 1] = sub.shape(i).n_edges
    atlas = gather_tiles(tiles, bounds)
    t = torch.tensor([float(hi-lo)])
    dist.all_reduce(t)
    q.put((rank, lo, hi, int(t.item()), atlas[:, 0, 0, 0].tolist(), atlas[:, 0, 0, 1].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_glyph_sharding_and_gather():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    z = np.load(os.path.join(GOLDEN, "latin.npz"))
    co, gco = z["contour_offsets"], z["glyph_contour_offsets"]
    edges = (co[gco[1:]]-co[gco[:-1]]).tolist()
    (r0, lo0, hi0, tot0, ids0, e0), (r1, lo1, hi1, tot1, ids1, e1) = results
    assert (lo0, hi1) == (0, 94) and hi0 == lo1 and 0 < hi0 < 94
    assert tot0 == tot1 == 94
    assert ids0 == ids1 == list(range(94)) and e0 == e1 == edges  # both ranks hold the same, correctly ordered atlas
    work = [sum(edges[lo0:hi0]), sum(edges[lo1:hi1])]
    assert abs(work[0]-work[1]) <= max(edges)+1


def _worker_dealt(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from msdfgen_amd.shape import ShapeBatch
    from msdfgen_amd.shard import shard_indices, gather_tiles_indexed
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(os.path.join(GOLDEN, "latin.npz"))
    batch = ShapeBatch(z["glyph_contour_offsets"].astype(np.int32), z["contour_offsets"].astype(np.int32), z["points"], z["types"].astype(np.int32),
                       z["colors"].astype(np.int32), z["inverse_y"], [str(n) for n in z["names"]])
    lists = shard_indices(batch, world, 64, 64, "dealt")
    mine = lists[rank]
    sub = batch.select(mine)
    tiles = torch.zeros((len(mine), 4, 4, 3))
    for i, g in enumerate(mine):
        tiles[i, ..., 0] = int(g)
        tiles[i, ..., 1] = sub.shape(i).n_edges
    atlas = gather_tiles_indexed(tiles, lists)
    q.put((rank, [int(g) for g in mine], atlas[:, 0, 0, 0].tolist(), atlas[:, 0, 0, 1].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_dealt_sharding_and_indexed_gather():
    """The strong-scaling cut (msdfgen_amd.shard.partition_dealt): the ranks' index lists partition the atlas, carry the same modelled cost to within
    one glyph, and gather_tiles_indexed puts the tiles back in atlas order on every rank (three gloo ranks: uneven list lengths)."""
    from msdfgen_amd.shard import partition_dealt
    world = 3
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_dealt, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    z = np.load(os.path.join(GOLDEN, "latin.npz"))
    co, gco = z["contour_offsets"], z["glyph_contour_offsets"]
    edges = (co[gco[1:]]-co[gco[:-1]]).tolist()
    owned = sorted(g for r in results for g in r[1])
    assert owned == list(range(94)) and {len(r[1]) for r in results} <= {31, 32}
    for r in results:
        assert r[1] == sorted(r[1]) and r[2] == list(range(94)) and r[3] == edges
    rng = np.random.default_rng(3)
    costs = rng.lognormal(0, 1.2, 8192)
    lists = partition_dealt(costs, 8)
    sums = np.array([costs[ix].sum() for ix in lists])
    assert sorted(np.concatenate(lists).tolist()) == list(range(8192)) and sums.max()-sums.min() <= costs.max()
    assert [ix.tolist() for ix in partition_dealt(costs, 8)] == [ix.tolist() for ix in lists]                 # deterministic
    assert [len(ix) for ix in partition_dealt(costs[:5], 8)] == [1, 1, 1, 1, 1, 0, 0, 0]                        # fewer glyphs than ranks


def test_bench_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` outside torchrun must launch two ranks itself (--gpus was parsed and ignored). --mock runs the
    N > 1 control path on CPU: gloo ranks, the real shard computation on the bench workload, barrier + gather, rank 0 prints one line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mock", "--glyphs", "2048"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                                          # rank 0 only
    r = lines[0]
    assert r["n_gpus"] == 2 and r["bounds"][0] == 0 and r["bounds"][-1] == 4096
    assert sum(r["glyphs_per_rank"]) == 4096 and all(g > 0 for g in r["glyphs_per_rank"])
    e = r["edges_per_rank"]
    assert abs(e[0]-e[1]) < 0.02*sum(e), e                                    # balanced by cost, not by count
    # a launch with a mismatching torchrun world is refused rather than silently benchmarking one rank
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mock"], capture_output=True, text=True, timeout=120, env=env2)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr+p.stdout)


def test_bench_strong_scaling_mode_cuts_one_set():
    """`bench.py --strong --gpus N` = BASELINE config 4 as stated: ONE 8 192-glyph 48x48 set cut into N shards (dealt by modelled cost, or contiguous ranges of equal modelled cost
    (total work fixed; the default mode gives every rank a full set). Control path on CPU (--mock, gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for name in ("dejavu", "cjk_like"):
        for cut in ("dealt", "contiguous"):
            p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mock", "--strong", "--strong-set", name, "--strong-cut", cut],
                               capture_output=True, text=True, timeout=300, env=env)
            assert p.returncode == 0, p.stderr[-2000:]
            r = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")][0]
            assert r["scaling"] == "strong" and r["bounds"][0] == 0 and r["bounds"][-1] == 8192 and sum(r["glyphs_per_rank"]) == 8192
            assert all(g > 2000 for g in r["glyphs_per_rank"]), r
            if cut == "dealt":                                                    # every shard the same mix: glyph and edge counts agree closely
                assert abs(r["glyphs_per_rank"][0]-r["glyphs_per_rank"][1]) <= 1 and abs(r["edges_per_rank"][0]-r["edges_per_rank"][1]) < .02*sum(r["edges_per_rank"]), r
