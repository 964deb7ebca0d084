#!/usr/bin/env python3
"""Two (or more) ranks on GPU 0 over gloo, nothing but torch: repeated all_gather_into_tensor on device tensors,
asynchronous with one wait per step, as raytracer_amd.tiles.StripPipeline issues them.  Tells whether a stall of
that pattern on a one-GPU box comes from gloo's staging of device tensors or from this package.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/gloo_device_gather.py [steps]"""
import faulthandler
import os
import sys

import torch
import torch.distributed as dist

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
faulthandler.dump_traceback_later(40, exit=True)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
mine = torch.full((256, 512, 3), float(rank), device=dev)
out = torch.zeros((world * 256, 512, 3), device=dev)
work = None
for k in range(steps):
    if work is not None:
        work.wait()
    mine.add_(1.0)                       # stands in for the strip render
    work = dist.all_gather_into_tensor(out, mine, async_op=True)
    print(f"[rank {rank}] step {k} issued", file=sys.stderr, flush=True)
work.wait()
torch.cuda.synchronize()
print(f"[rank {rank}] {steps} gathers done, out[0,0,0] = {out[0, 0, 0].item()}", file=sys.stderr, flush=True)
dist.barrier()
dist.destroy_process_group()
