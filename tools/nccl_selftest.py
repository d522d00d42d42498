"""Self-test of the torch.distributed / RCCL calls bench.py makes for N > 1, on one GPU (world size 1)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.barrier(); dist.all_reduce(t, op=dist.ReduceOp.MAX); torch.cuda.synchronize()
print("nccl world 1 ok", float(t.item()))
dist.barrier(); dist.destroy_process_group()
