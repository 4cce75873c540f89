"""Why is the update slower once collectives run?  One process, RCCL world of one: the one-launch optimiser step (latency-bound, 64 workgroups reading the
712 kB gradient bucket) timed before any collective, after all-reducing ANOTHER tensor, and after all-reducing the gradient bucket itself."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29520")
import torch, torch.distributed as dist
from booster_gym_amd.utils.runner import FlatAdam
dev = "cuda:0"
torch.cuda.set_device(0)
ps = [torch.nn.Parameter(torch.randn(177945, device=dev))]
fa = FlatAdam(ps, lr=1e-3)
stats = torch.zeros(5, dtype=torch.float64, device=dev); acc = torch.zeros_like(stats); last = torch.zeros_like(stats)
other = torch.ones(177945, device=dev)
def opt():
    fa.step_fused(stats, acc, last, 4, 1000.0, 0.01)
def bench(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fa.grad.normal_()
print(f"optimiser step before init:                          {bench(opt):6.1f} us", flush=True)
dist.init_process_group("nccl", rank=0, world_size=1)
print(f"after init_process_group:                            {bench(opt):6.1f} us", flush=True)
for _ in range(20): dist.all_reduce(other)
torch.cuda.synchronize()
print(f"after 20 all_reduces of ANOTHER tensor:              {bench(opt):6.1f} us", flush=True)
for _ in range(20): dist.all_reduce(fa.grad, op=dist.ReduceOp.AVG)
torch.cuda.synchronize()
print(f"after 20 all_reduces of the gradient bucket itself:  {bench(opt):6.1f} us", flush=True)
def both():
    dist.all_reduce(fa.grad, op=dist.ReduceOp.AVG); opt()
print(f"all_reduce(grad) + optimiser step, back to back:     {bench(both):6.1f} us", flush=True)
side = torch.cuda.Stream()
def on_side():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dist.all_reduce(stats)
    torch.cuda.current_stream().wait_stream(side)
    opt()
print(f"all_reduce(stats) on a side stream + join + optimiser step: {bench(on_side):6.1f} us", flush=True)
dist.broadcast(fa.flat, src=0); torch.cuda.synchronize()
print(f"after a broadcast of the parameter buffer:           {bench(opt):6.1f} us", flush=True)
box = [123]
dist.broadcast_object_list(box, src=0); torch.cuda.synchronize()
print(f"after broadcast_object_list:                         {bench(opt):6.1f} us", flush=True)
dist.barrier(device_ids=[0]); torch.cuda.synchronize()
print(f"after barrier(device_ids):                           {bench(opt):6.1f} us", flush=True)
print(f"optimiser step at the end:                           {bench(opt):6.1f} us", flush=True)
dist.destroy_process_group()
