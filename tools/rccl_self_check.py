#!/usr/bin/env python3
"""One rank, backend nccl: does a large all-to-all to itself deliver every byte? (sizes in int64 items)"""
import os, sys, json
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
out = []
for n in [int(x) for x in sys.argv[1:]] or [1 << 20, 200_000_000, 300_000_000, 456_483_337]:
    src = torch.arange(n, dtype=torch.int64, device=dev) * 2654435761 + 12345
    for form in ("list", "single"):
        dst = torch.zeros(n + 1000, dtype=torch.int64, device=dev)
        if form == "list":
            w = dist.all_to_all([dst[:n]], [src], async_op=True)
        else:
            w = dist.all_to_all_single(dst[:n], src, output_split_sizes=[n], input_split_sizes=[n], async_op=True)
        w.wait()
        torch.cuda.synchronize()
        bad = int((dst[:n] != src).sum().item())
        first = int(torch.nonzero(dst[:n] != src)[0].item()) if bad else -1
        out.append({"n": n, "bytes": n * 8, "form": form, "mismatches": bad, "first": first})
        del dst
print(json.dumps(out))
dist.destroy_process_group()
