"""The strong-scaling rehearsal of bench.py alone (config 4 as stated, ONE 8 192-glyph 48x48 atlas cut into N shards, every shard timed alone on one GPU): python tools/r06_strong.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import msdfgen_amd as M
from msdfgen_amd import lib as L
import bench
M.init(0)
lib = L.load()
cfg = L.default_config()
dev = torch.device("cuda:0")
r = bench.strong_scaling_one_gpu(M, torch, lib, dev, torch.cuda.current_stream(), cfg, steps=6)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("ms_full_set", "x2", "x4", "x8", "dealt_x8")}) for k, v in r.items()}))
