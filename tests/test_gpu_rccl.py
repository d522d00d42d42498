"""RCCL under this code: the process-group calls bench.py makes for N > 1 -- init with device_id, barrier, MAX all-reduce,
all_gather of the per-rank rows, gather_tiles' all_gather of tiles -- on the real backend ("nccl" is RCCL on ROCm). World size 1 runs on any GPU box;
the two-rank test needs two devices and is skipped otherwise (RCCL refuses two ranks on one device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(local); dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
t = torch.tensor([1.5+rank], dtype=torch.float64, device=dev)
dist.barrier(); dist.all_reduce(t, op=dist.ReduceOp.MAX); torch.cuda.synchronize()
from msdfgen_amd.shard import gather_tiles
bounds = [3*r for r in range(world+1)]
tiles = torch.full((3, 4, 4, 3), float(rank), device=dev)
atlas = gather_tiles(tiles, bounds)
ok = all(float(atlas[3*r:3*r+3].min()) == r == float(atlas[3*r:3*r+3].max()) for r in range(world))
rows = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
dist.all_gather(rows, torch.tensor([rank, local], dtype=torch.float64, device=dev))
if rank == 0:
    print(json.dumps({"backend": dist.get_backend(), "world": world, "max": float(t.item()), "gather_ok": bool(ok), "ranks": [int(r[0].item()) for r in rows]}))
dist.barrier(); dist.destroy_process_group()
"""


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(extra):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra)
    return env


def test_rccl_world_of_one_runs_the_bench_collectives(tmp_path):
    script = tmp_path/"w.py"
    script.write_text(WORKER % {"root": ROOT})
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300,
                       env=_env({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_PORT": str(_free_port())}))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])   # (RCCL prints a "Librccl path" line of its own on stdout)
    assert d == {"backend": "nccl", "world": 1, "max": 1.5, "gather_ok": True, "ranks": [0]}


def _devices():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def test_rccl_two_ranks_over_two_devices(tmp_path):
    if _devices() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    script = tmp_path/"w.py"
    script.write_text(WORKER % {"root": ROOT})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                        str(script)], capture_output=True, text=True, timeout=600, env=_env({}))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d == {"backend": "nccl", "world": 2, "max": 2.5, "gather_ok": True, "ranks": [0, 1]}


@pytest.mark.parametrize("strong", [False, True])
def test_bench_two_gpus_weak_and_strong_lines(strong):
    """The driver's command for N = 2, both modes: one JSON line from rank 0 that names both ranks, their devices and their shards."""
    if _devices() < 2:
        pytest.skip("needs two GPUs")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"]+(["--strong"] if strong else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=_env({}), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["scaling"] == ("strong" if strong else "weak")
    pr = d["per_rank"]
    assert pr["backend"] == "nccl" and pr["ranks_seen"] == [0, 1] and pr["device_index"] == [0, 1]
    assert sum(pr["glyphs"]) == (8192 if strong else 2*8192)
