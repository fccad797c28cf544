"""Developer tool: DDIM-50 + decode of ONE window (how sample_trials.py:149-163 runs), for a rocprofv3 kernel trace of the launch chain."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eegldm
from eegldm.models import UNetModel, AutoencoderKL
from eegldm.training import randn
from eegldm.sampling import ddim_sample, make_sampling_scheduler
ctx = eegldm.default_context(0); L = 768
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
u = UNetModel(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4],
              resblock_updown=True, dtype="bfloat16")
g = torch.Generator().manual_seed(42); sd = u.state_dict()
u.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in sd.items()})
ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2, norm_num_groups=1,
                   attention_levels=[False] * 3, dtype="bfloat16")
sch = make_sampling_scheduler(50)
nz = randn(ctx, (B, 1, L), seed=4242)
for r in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    ddim_sample(u, ae, sch, nz); torch.cuda.synchronize()
    print(f"DDIM-50 + decode B={B}: {(time.time() - t0) * 1e3:.1f} ms", flush=True)
