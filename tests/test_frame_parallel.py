"""Frame-parallel clip over a world_size-2 process group.

CPU (`not gpu`): gloo + the TEST-ONLY op emulation -- validates the re-sharding (frames <-> pixels all-to-all), the
sharded GroupNorm contract and the wiring.  GPU (`gpu`): two processes share the one MI355X of the test box, real HIP
kernels, gloo with host-staged collectives (RCCL refuses two ranks on one device; on a node the same code runs over
RCCL / xGMI with backend "nccl")."""
import os
import socket
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, device):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), ANYV2V_NO_GRAPH="1")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import gpu_checks as gc
    from anyv2v_amd.parallel import init_distributed
    torch.set_grad_enabled(False)
    if device == "cpu":
        import cpu_ops_emulation as emu
        emu.install()
        torch.set_num_threads(2)
    gc.DEV = device
    init_distributed("gloo")
    res = gc.check_frame_parallel()
    torch.save(res, os.path.join(out_dir, f"fp{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, device):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), device), nprocs=2, join=True)
    bad = []
    for r in range(2):
        res = torch.load(tmp_path / f"fp{r}.pt")
        assert len(res) >= 5
        for x in res:
            print(f"{'ok  ' if x['ok'] else 'FAIL'} rank {r}: {x['name']}: {x['err']:.3e} (tol {x['tol']:.1e})")
        bad += [f"{x['name']}: {x['err']:.3e} > {x['tol']:.1e}" for x in res if not x["ok"]]
    assert not bad, "\n".join(bad)


@pytest.mark.timeout(900)
def test_frame_parallel_unet_world2_gloo_cpu(tmp_path):
    _run(tmp_path, "cpu")


@pytest.mark.gpu
def test_frame_parallel_unet_two_ranks_on_one_gpu(tmp_path):
    _run(tmp_path, "cuda")


def test_sharded_groupnorm_contract_single_rank():
    """partial + apply(shards=1) is the one-call GroupNorm; with an 'all-reduce' that doubles the sums and shards=2 the
    result is unchanged (two identical shards)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_ops_emulation as emu
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2 * 24, 64, generator=g).half()
    ga, be = torch.randn(64, generator=g).half(), torch.randn(64, generator=g).half()
    st = torch.empty(4096, dtype=torch.float32)
    ref = emu.groupnorm(x, ga, be, st, 24, groups=8, silu=True)
    one = emu.groupnorm(x, ga, be, st, 24, groups=8, silu=True, shard=(1, lambda t: t))
    two = emu.groupnorm(x, ga, be, st, 24, groups=8, silu=True, shard=(2, lambda t: t.mul_(2)))
    assert (one.float() - ref.float()).abs().max() < 2e-3 and (two.float() - ref.float()).abs().max() < 2e-3
