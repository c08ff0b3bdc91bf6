"""Full-size frame-parallel check (GPU): the 1.42 B-parameter UNet at the bench workload (16 f x 512 x 512, B=3 PnP step
with all injections on), frames sharded over 2 ranks that SHARE the one GPU of the test box (gloo, host-staged
collectives -- RCCL refuses two ranks on one device; timing is therefore meaningless here, this checks numerics and
reports the all-to-all payload).  Each rank compares its sharded forward with its own unsharded forward.

    python tools/frame_parallel_probe.py          -> gpurun_out/frame_parallel_probe.txt
"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def worker(rank, world, port):
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    from anyv2v_amd import pnp_utils
    from anyv2v_amd.parallel import FrameParallel, init_distributed
    from anyv2v_amd.pipeline import I2VGenXLPipeline
    from anyv2v_amd.schedulers import DDIMScheduler
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    init_distributed("gloo")
    fp = FrameParallel()
    pipe = I2VGenXLPipeline.from_pretrained("ali-vilab/i2vgen-xl", torch_dtype=torch.float16, variant="fp16", random_init_seed=0)
    pipe.to(dev)
    lat, ehs, ie, il = bench.synthetic_clip(dev, 8888)
    fwd = DDIMScheduler()
    fwd.set_timesteps(50)
    pnp_utils.register_conv_injection(pipe, fwd.timesteps)
    pnp_utils.register_spatial_attention_pnp(pipe, fwd.timesteps)
    pnp_utils.register_temp_attention_pnp(pipe, fwd.timesteps)
    lines = []
    for B in (3, 1):
        if B == 3:
            pnp_utils.register_time(pipe, int(fwd.timesteps[0]))
            sample = lat.repeat(3, 1, 1, 1, 1).contiguous()
            cond = dict(encoder_hidden_states=ehs, fps=torch.tensor([8, 8, 8], device=dev), image_latents=il, image_embeddings=ie)
        else:
            pnp_utils.clear_time(pipe)
            sample = lat.clone()
            cond = dict(encoder_hidden_states=ehs[:1].contiguous(), fps=torch.tensor([8], device=dev),
                        image_latents=il[:1].contiguous(), image_embeddings=ie[:1].contiguous())
        t = int(fwd.timesteps[0])
        res = []
        for use in (None, fp):
            pipe.unet.set_frame_parallel(use)
            fp.bytes_moved = 0
            res.append(pipe.unet(sample, t, **cond)[0].float())
            torch.cuda.synchronize()
        err = (res[1] - res[0]).abs().max().item() / res[0].abs().max().item()
        lines.append(f"rank {rank}/{world} B={B} 16f x 64x64 latents: frame-parallel vs unsharded max|diff|/max|ref| = {err:.2e}; "
                     f"finite={bool(torch.isfinite(res[1]).all())}; all-to-all payload sent by this rank {fp.bytes_moved / 2**20:.0f} MiB "
                     f"per forward")
    import torch.distributed as dist
    gathered = [None] * world
    dist.all_gather_object(gathered, lines)
    if rank == 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        txt = "\n".join(l for ls in gathered for l in ls)
        open(os.path.join(ROOT, "gpurun_out", "frame_parallel_probe.txt"), "w").write(txt + "\n")
        print(txt)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
