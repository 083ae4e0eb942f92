#!/usr/bin/env python
"""Can two ranks share ONE GPU under RCCL?  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 scripts/rccl_two_ranks_one_gpu.py
Both ranks use cuda:0 and try init_process_group('nccl') + broadcast + all_reduce.  Prints one JSON line per rank; exit code 0 either way
(the answer is data, not a failure)."""
import json
import os
import sys

import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    out = dict(rank=rank, world=world, device="cuda:0")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        t = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.broadcast(t, src=0)
        out["broadcast_ok"] = bool((t == 1.0).all().item())
        u = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.all_reduce(u)
        out["all_reduce_ok"] = bool((u == float(sum(range(1, world + 1)))).all().item())
        torch.cuda.synchronize()
        dist.destroy_process_group()
    except BaseException as e:      # RCCL refuses duplicate devices in one communicator
        out["error"] = f"{type(e).__name__}: {str(e)[:300]}"
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    sys.exit(0)
