"""DDIM sampling of synthetic EEG windows -- counterpart of /root/reference/src/sample_trials.py: loads the stage-1
autoencoder + UNet checkpoints, reads scale_factor from checkpoint.pth (:130-132), samples one window per seed
(batched; seeds sharded over ranks, no collective), writes sample_{i}.npy of shape (1,1,3000) (:169-170).
--num_inference_steps is honoured (the reference hard-codes 200, :144); PSD plots (mne) are out of scope."""
import argparse
import os

import numpy as np
import torch

from .. import distributed as D
from ..models import AutoencoderKL, UNetModel
from ..sampling import make_sampling_scheduler, sample_seeds
from .common import load_config


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--output_dir", required=True); p.add_argument("--best_model_path", required=True)
    p.add_argument("--diffusion_path", required=True); p.add_argument("--autoencoderkl_config_file_path", required=True)
    p.add_argument("--ldm_config_file_path", required=True)
    p.add_argument("--start_seed", type=int, default=0); p.add_argument("--stop_seed", type=int, default=1000)
    p.add_argument("--guidance_scale", type=float, default=7.0); p.add_argument("--num_inference_steps", type=int, default=200)
    p.add_argument("--path_pre_processed", default=None); p.add_argument("--spe", default="no-spectral")
    p.add_argument("--latent_channels", type=int, default=1); p.add_argument("--type_dataset", default="edfx")
    p.add_argument("--prediction_type", default="v_prediction", help="sample_trials.py:141 uses v_prediction")
    p.add_argument("--batch", type=int, default=256); p.add_argument("--dtype", default="float32")
    return p.parse_args(argv)


def main(args):
    rank, local, world = D.init_from_env()
    torch.cuda.set_device(local)
    out = os.path.join(args.output_dir, f"samples_ldm_{args.latent_channels}_{args.spe}_{args.type_dataset}")
    os.makedirs(out, exist_ok=True)
    ae_cfg = dict(load_config(args.autoencoderkl_config_file_path).autoencoderkl.params)
    ae_cfg.setdefault("num_channels", [32, 32, 64]); ae_cfg["latent_channels"] = args.latent_channels     # sample_trials.py:97-98
    stage1 = AutoencoderKL(**ae_cfg, dtype=args.dtype, device=local)
    stage1.load_state_dict(torch.load(os.path.join(args.best_model_path, "best_model.pth"), map_location="cpu"))
    up = dict(load_config(args.ldm_config_file_path)["model"]["params"]["unet_config"]["params"])
    up["in_channels"] = up["out_channels"] = args.latent_channels
    unet = UNetModel(**up, dtype=args.dtype, device=local)
    unet.load_state_dict(torch.load(os.path.join(args.diffusion_path, "best_model.pth"), map_location="cpu"))
    scale_factor = float(torch.load(os.path.join(args.diffusion_path, "checkpoint.pth"), map_location="cpu")["scale_factor"])
    sched = make_sampling_scheduler(args.num_inference_steps, prediction_type=args.prediction_type, device=local)
    lo, hi = D.shard_range(args.stop_seed - args.start_seed, rank, world)
    seeds = list(range(args.start_seed + lo, args.start_seed + hi))
    for k in range(0, len(seeds), args.batch):
        chunk = seeds[k:k + args.batch]
        windows, _ = sample_seeds(unet, stage1, sched, chunk, latent_len=up.get("image_size", 768), scale_factor=scale_factor)
        arr = windows.cpu().numpy()
        for j, sd in enumerate(chunk):
            np.save(os.path.join(out, f"sample_{sd}.npy"), arr[j:j + 1])
    return out


if __name__ == "__main__":
    main(parse_args())
