"""RCCL sanity on one GPU (world_size 1): the exact distributed calls bench.py / run_group_pnp_edit.py make at N > 1 --
init_process_group("nccl", device_id=...), barrier, all_gather of the edited latents, all_reduce(MAX) of the timing.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_smoke.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402

local_rank = int(os.environ.get("LOCAL_RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(local_rank)
device = torch.device("cuda", local_rank)
dist.init_process_group("nccl", device_id=device)
dist.barrier()
lat = torch.randn(1, 4, 16, 64, 64, device=device).half()
dt, gathered = bench.finish_distributed(dist, 1.25, lat, world, device)
assert dt == 1.25 and len(gathered) == world and torch.equal(gathered[0], lat)
dist.barrier()
dist.destroy_process_group()
print("rccl smoke ok: world", world)
