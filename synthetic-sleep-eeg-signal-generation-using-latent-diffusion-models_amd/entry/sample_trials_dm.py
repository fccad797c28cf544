"""DDIM sampling from the pixel-space diffusion model -- counterpart of /root/reference/src/sample_trials_ddpm.py:
UNet with in/out channels 1 on (1,1,3072) noise, DDIMScheduler(scaled_linear_beta 0.0015-0.0205, v_prediction, clip_sample
False) (:83-92), no autoencoder, crop [36:-36] -> sample_{i}.npy (1,1,3000) (:104-105).  Seeds are batched and sharded over
ranks (no collective); --num_inference_steps is honoured (the reference passes it as num_train_timesteps and hard-codes 200
inference steps, :84,:91); PSD plots (mne) are out of scope."""
import argparse
import os

import numpy as np
import torch

from .. import distributed as D
from ..models import UNetModel
from ..sampling import make_sampling_scheduler, sample_seeds
from .common import load_config


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--output_dir", required=True); p.add_argument("--config_file", required=True)
    p.add_argument("--diffusion_path", required=True)
    p.add_argument("--start_seed", type=int, default=0); p.add_argument("--stop_seed", type=int, default=1000)
    p.add_argument("--num_inference_steps", type=int, default=200); p.add_argument("--spe", default="no-spectral")
    p.add_argument("--dataset", default="edfx"); p.add_argument("--prediction_type", default="v_prediction")
    p.add_argument("--batch", type=int, default=64); p.add_argument("--dtype", default="float32")
    return p.parse_args(argv)


def main(args):
    rank, local, world = D.init_from_env()
    torch.cuda.set_device(local)
    out = os.path.join(args.output_dir, f"samples_dm_{args.spe}_{args.dataset}")
    os.makedirs(out, exist_ok=True)
    up = dict(load_config(args.config_file)["model"]["params"]["unet_config"]["params"])
    up["in_channels"] = up["out_channels"] = 1                                   # sample_trials_ddpm.py:71-73
    unet = UNetModel(**up, dtype=args.dtype, device=local)
    unet.load_state_dict(torch.load(os.path.join(args.diffusion_path, "best_model.pth"), map_location="cpu"))
    sched = make_sampling_scheduler(args.num_inference_steps, prediction_type=args.prediction_type, device=local)
    lo, hi = D.shard_range(args.stop_seed - args.start_seed, rank, world)
    seeds = list(range(args.start_seed + lo, args.start_seed + hi))
    for k in range(0, len(seeds), args.batch):
        chunk = seeds[k:k + args.batch]
        windows, _ = sample_seeds(unet, None, sched, chunk, latent_len=3072)
        arr = windows.cpu().numpy()
        for j, sd in enumerate(chunk):
            np.save(os.path.join(out, f"sample_{sd}.npy"), arr[j:j + 1])
    return out


if __name__ == "__main__":
    main(parse_args())
