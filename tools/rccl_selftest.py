"""Rehearsal of the RCCL calls bench.py / the trainer issue for N > 1 (flat fp32 gradient all-reduce, 3-double advantage
statistics all-reduce, weight broadcast, MAX all-reduce of the elapsed time, barrier), on however many ranks torchrun
starts (1 is enough to prove the backend initialises and every dtype / op is supported on this image)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

rank, ws, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(local % torch.cuda.device_count())
dist.init_process_group(backend="nccl", rank=rank, world_size=ws)
dev = torch.device("cuda", local % torch.cuda.device_count())
g = torch.full((65 * 2500 + 9123,), float(rank + 1), device=dev)
dist.all_reduce(g, op=dist.ReduceOp.SUM)
assert float(g[0]) == ws * (ws + 1) / 2
s = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=dev)
dist.all_reduce(s, op=dist.ReduceOp.SUM)
assert s.tolist() == [ws * 1.0, ws * 2.0, ws * 3.0]
w = torch.arange(1000, dtype=torch.float32, device=dev) * (1 if rank == 0 else 0)
dist.broadcast(w, src=0)
assert float(w[999]) == 999.0
m = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
dist.all_reduce(m, op=dist.ReduceOp.MAX)
assert float(m) == ws - 1
dist.barrier()
torch.cuda.synchronize()
if rank == 0:
    print(f"RCCL self-test ok on {ws} rank(s): all_reduce fp32 / f64 SUM, MAX, broadcast, barrier", flush=True)
dist.destroy_process_group()
