"""-m gpu: the three entry scripts run end to end on synthetic windows with reference-style yaml configs:
train_autoencoderkl -> best_model.pth, train_ldm -> checkpoint.pth (scale_factor), sample_trials -> sample_{i}.npy (1,1,3000)."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

AEKL_YAML = {
    "train": {"seed": 2, "batch_size": 8, "n_epochs": 2, "val_interval": 1, "num_workers": 0, "drop_last": False,
              "output_dir": "OUT", "run_dir": "aekl_eeg", "experiment": "AEKL"},
    "models": {"optimizer_g_lr": 0.005, "optimizer_d_lr": 0.0005, "adv_weight": 0.01, "kl_weight": 1e-9, "spectral_weight": 1e4},
    "autoencoderkl": {"params": {"spatial_dims": 1, "in_channels": 1, "out_channels": 1, "num_res_blocks": 2, "norm_num_groups": 1,
                                 "attention_levels": [False, False, False], "with_encoder_nonlocal_attn": False,
                                 "with_decoder_nonlocal_attn": False, "num_channels": [8, 8, 16], "latent_channels": 1}},
    "patchdiscriminator": {"params": {"spatial_dims": 1, "num_layers_d": 3, "num_channels": 16, "in_channels": 1, "out_channels": 1,
                                      "kernel_size": 3, "norm": "BATCH", "bias": False, "padding": 1}},
}
LDM_YAML = {
    "train": {"seed": 2, "batch_size": 8, "n_epochs": 1, "eval_freq": 1, "num_workers": 0, "experiment": "DDPM", "output_dir": "OUT",
              "run_dir": "ldm_eeg", "drop_last": False, "base_lr": 0.0001},
    "model": {"params": {"timesteps": 1000, "unet_config": {"params": {
        "image_size": 768, "in_channels": 3, "out_channels": 1, "model_channels": 32, "attention_resolutions": [8, 4], "num_res_blocks": 1,
        "channel_mult": [1, 2, 4], "dropout": 0.0, "conv_resample": True, "num_heads": 1, "use_scale_shift_norm": False, "resblock_updown": True}}}},
}


def test_train_aekl_train_ldm_sample(tmp_path):
    from eegldm.entry import train_autoencoderkl as TA, train_ldm as TL, sample_trials as ST
    out = str(tmp_path)
    a_yaml, l_yaml = os.path.join(out, "aekl.yaml"), os.path.join(out, "ldm.yaml")
    a = dict(AEKL_YAML); a["train"] = dict(a["train"], output_dir=out)
    l = dict(LDM_YAML); l["train"] = dict(l["train"], output_dir=out)
    yaml.safe_dump(a, open(a_yaml, "w")); yaml.safe_dump(l, open(l_yaml, "w"))
    run_a = TA.main(TA.parse_args(["--config_file", a_yaml, "--spe", "spectral", "--synthetic_windows", "16", "--latent_channels", "1"]))
    for f in ("best_model.pth", "checkpoint.pth", "final_model.pth"):
        assert os.path.exists(os.path.join(run_a, f))
    ck = torch.load(os.path.join(run_a, "checkpoint.pth"))
    assert set(ck) >= {"epoch", "state_dict", "discriminator", "optimizer_g", "optimizer_d", "best_loss", "init_batch"}      # train_autoencoderkl.py:320-328
    assert tuple(ck["init_batch"].shape[1:]) == (1, 3000)      # first(train_loader)['eeg'][:, :, 36:-36] (:188); the reference's resume reads it unconditionally (:182)
    # resume path
    TA.main(TA.parse_args(["--config_file", a_yaml, "--spe", "spectral", "--synthetic_windows", "16", "--latent_channels", "1"]))
    run_l = TL.main(TL.parse_args(["--config_file", l_yaml, "--autoencoderkl_config_file_path", a_yaml, "--best_model_path", run_a,
                                   "--synthetic_windows", "16", "--latent_channels", "1", "--max_steps", "2", "--grad_scaler"]))   # scaler path as in training.py:441-443
    ck = torch.load(os.path.join(run_l, "checkpoint.pth"))
    assert set(ck) >= {"epoch", "diffusion", "optimizer", "best_loss", "scale_factor"} and float(ck["scale_factor"]) > 0   # training.py:381-387
    # a second invocation finds checkpoint.pth and continues from it (epoch budget already used: nothing left to train, same scale factor)
    run_l2 = TL.main(TL.parse_args(["--config_file", l_yaml, "--autoencoderkl_config_file_path", a_yaml, "--best_model_path", run_a,
                                    "--synthetic_windows", "16", "--latent_channels", "1", "--max_steps", "2", "--grad_scaler"]))
    ck2 = torch.load(os.path.join(run_l2, "checkpoint.pth"))
    assert run_l2 == run_l and ck2["epoch"] == ck["epoch"] and int(ck2["steps"]) == int(ck["steps"]) == 2
    assert float(ck2["scale_factor"]) == float(ck["scale_factor"])
    sdir = ST.main(ST.parse_args(["--output_dir", out, "--best_model_path", run_a, "--diffusion_path", run_l,
                                  "--autoencoderkl_config_file_path", a_yaml, "--ldm_config_file_path", l_yaml,
                                  "--start_seed", "3", "--stop_seed", "6", "--num_inference_steps", "5", "--latent_channels", "1"]))
    for i in (3, 4, 5):
        s = np.load(os.path.join(sdir, f"sample_{i}.npy"))
        assert s.shape == (1, 1, 3000) and np.isfinite(s).all()
    assert not os.path.exists(os.path.join(sdir, "sample_6.npy"))


def test_train_pixel_dm(tmp_path):
    """train_pure_ldm.py counterpart: UNet directly on (B,1,3072) windows, in/out channels forced to 1, optional spectral term."""
    from eegldm.entry import train_dm as TD
    out = str(tmp_path)
    d_yaml = os.path.join(out, "dm.yaml")
    d = dict(LDM_YAML); d["train"] = dict(d["train"], output_dir=out, run_dir="dm_eeg", batch_size=4)
    yaml.safe_dump(d, open(d_yaml, "w"))
    run_d = TD.main(TD.parse_args(["--config_file", d_yaml, "--spe", "spectral", "--synthetic_windows", "8", "--max_steps", "2"]))
    ck = torch.load(os.path.join(run_d, "checkpoint.pth"))
    assert set(ck) >= {"epoch", "diffusion", "optimizer", "best_loss"} and np.isfinite(ck["best_loss"])
    assert ck["diffusion"]["input_blocks.0.0.weight"].shape == (32, 1, 3)          # in_channels forced to 1 (train_pure_ldm.py:113-115)
    from eegldm.entry import sample_trials_dm as SD
    sdir = SD.main(SD.parse_args(["--output_dir", out, "--config_file", d_yaml, "--diffusion_path", run_d, "--start_seed", "1", "--stop_seed", "3",
                                  "--num_inference_steps", "4"]))
    for i in (1, 2):
        s = np.load(os.path.join(sdir, f"sample_{i}.npy"))
        assert s.shape == (1, 1, 3000) and np.isfinite(s).all()                    # sample_trials_ddpm.py:104-105
