"""Does all_to_all_single deliver large uneven splits over RCCL? World size 1 (one GPU): sizes from 1 GiB to 9 GiB, content checked.
Round 3: the 54 GB exchange of a full cfg5 share came back without its bytes; this probe finds the size at which that starts."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
for gib in (0.5, 1.5, 2.5, 3.5, 4.5, 9.0):
    n = int(gib * (1 << 30)) + 12345
    src = (torch.arange(n, device="cuda", dtype=torch.int64) % 251).to(torch.uint8)
    dst = torch.zeros(n, dtype=torch.uint8, device="cuda")
    dist.all_to_all_single(dst, src, [n], [n])
    torch.cuda.synchronize()
    ok = bool(torch.equal(dst, src))
    print(f"all_to_all_single {gib} GiB: {'ok' if ok else 'WRONG'}; first wrong byte at", "-" if ok else int((dst != src).nonzero()[0]), flush=True)
    # the same through explicit send / recv pairs to self in 1 GiB chunks is what shard.exchange_lines does now: nothing to test at world 1
    del src, dst
dist.destroy_process_group()
