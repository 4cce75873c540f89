"""Can two RCCL ranks share the one GPU of a gpurun box?  (The only multi-rank RCCL execution this pool could offer.)

python tools/probe/two_ranks_one_gpu.py  -> one JSON line per rank: the error text if communicator setup refuses the duplicate device.
"""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def rank_main(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    out = {"rank": rank}
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        t = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        out["sum"] = float(t[0])
    except Exception as e:  # noqa: BLE001 (the text of the refusal is the result)
        out["error"] = str(e).splitlines()[0][:300]
        out["detail"] = [ln for ln in str(e).splitlines() if "uplicate" in ln or "invalid" in ln.lower()][:3]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.start_processes(rank_main, args=(2, port), nprocs=2, start_method="spawn", join=True)
