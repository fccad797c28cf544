"""Developer check (one GPU, two processes, gloo on CUDA tensors): overlapped gradient sync through the native grad hook
gives the same averaged gradients as the plain post-backward bucketed all-reduce."""
import os, sys, torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from eegldm import distributed as D
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import ldm_train_step
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = UNetModel(image_size=64, in_channels=1, out_channels=1, model_channels=32, num_res_blocks=1, attention_resolutions=[2],
                    channel_mult=[1, 2], resblock_updown=True, dtype="float32")
    sd = net.state_dict()
    net.load_state_dict({k: torch.randn(v.shape) * 0.05 for k, v in sd.items()})
    D.broadcast_flat(net.flat); net.sync_weights()
    sched = DDPMScheduler(num_train_timesteps=1000, beta_schedule="scaled_linear", beta_start=0.0015, beta_end=0.0195)
    g = torch.Generator().manual_seed(100 + rank)
    dev = net.device
    lat = torch.randn(4, 1, 64, generator=g).to(dev); noise = torch.randn(4, 1, 64, generator=g).to(dev)
    t = torch.randint(0, 1000, (4,), generator=g).to(dev)
    net.zero_grad(); ldm_train_step(net, sched, lat, noise, t); D.allreduce_mean_flat(net.flat_grad); ref = net.flat_grad.clone()
    gs = D.OverlappedGradSync(net.flat_grad, bucket_elems=50000)
    net.zero_grad(); ldm_train_step(net, sched, lat, noise, t, grad_sync=gs); gs.wait()
    torch.cuda.synchronize()
    err = float((net.flat_grad - ref).abs().max() / ref.abs().max())
    print(f"rank {rank}: overlapped vs plain all-reduce rel err {err:.2e}, launched ranges {gs.done}", flush=True)
    assert err < 1e-4
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(worker, args=(2, 29533), nprocs=2, join=True)
    print("ddp gloo check ok")
