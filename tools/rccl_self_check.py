#!/usr/bin/env python3
"""Does a large all-to-all deliver every byte? Each rank sends n int64 items to EVERY rank (itself included) and checks
what it received, for the list form and for all_to_all_single.

    python tools/rccl_self_check.py [n ...]                                  one rank, backend nccl (sends to itself)
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/rccl_self_check.py [n ...]
    QH_DIST_BACKEND=gloo ... (CPU tensors: a dry run of the script itself)

Seen on this image (RCCL 2.26.6, one MI355X, one rank): up to 1 GiB per message every byte arrives; 1.6 GB and more
arrive with the second half missing and no error (profiles/r02_rccl_self_check*.log). Whether messages between two GPUs
behave alike is what the two-rank form is for; quorum_amd/dist_build.py caps messages at 512 MiB meanwhile."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0"))
backend = os.environ.get("QH_DIST_BACKEND", "nccl")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29555")
if backend == "nccl":
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dev = torch.device("cpu")
    dist.init_process_group(backend, rank=rank, world_size=world)


def payload(src, dst, n):
    """what rank src sends to rank dst: a pattern both sides can compute"""
    return torch.arange(n, dtype=torch.int64, device=dev) * 2654435761 + (src * 1000003 + dst * 7919 + 12345)


out = []
for n in [int(x) for x in sys.argv[1:]] or [1 << 20, 200_000_000, 300_000_000, 456_483_337]:
    for form in ("list", "single"):
        send = [payload(rank, d, n) for d in range(world)]
        recv = torch.zeros(world * n, dtype=torch.int64, device=dev)
        if form == "list" and backend == "nccl":
            w = dist.all_to_all(list(recv.split(n)), send, async_op=True)
        else:
            w = dist.all_to_all_single(recv, torch.cat(send), output_split_sizes=[n] * world, input_split_sizes=[n] * world,
                                       async_op=True)
        w.wait()
        if backend == "nccl":
            torch.cuda.synchronize()
        for s in range(world):
            want = payload(s, rank, n)
            got = recv[s * n:(s + 1) * n]
            bad = int((got != want).sum().item())
            first = int(torch.nonzero(got != want)[0].item()) if bad else -1
            out.append({"n": n, "bytes": n * 8, "form": form, "from": s, "to": rank, "mismatches": bad, "first": first})
        del send, recv
print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
