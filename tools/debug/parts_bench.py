"""Developer tool: DDIM-50 sampling + decode (B=256 and B=1) and the pixel-space DM train step alone (for rocprofv3 traces)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import eegldm
from eegldm.models import UNetModel, AutoencoderKL
from eegldm.schedulers import DDPMScheduler
from eegldm.training import Adam, dm_train_step, randint, randn
from eegldm.sampling import ddim_sample, make_sampling_scheduler
from param_gen import eeg_windows
ctx = eegldm.default_context(0); dev = torch.device("cuda", 0); L = 768
def mk(Lx):
    u = UNetModel(image_size=Lx, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4],
                  resblock_updown=True, dtype="bfloat16")
    g = torch.Generator().manual_seed(42); sd = u.state_dict()
    u.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in sd.items()})
    return u
unet = mk(L)
ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2, norm_num_groups=1,
                   attention_levels=[False] * 3, dtype="bfloat16")
sch = make_sampling_scheduler(50)
nz = randn(ctx, (256, 1, L), seed=4242)
for B in (256, 1):
    ddim_sample(unet, ae, sch, nz[:B]); torch.cuda.synchronize(); t0 = time.time()
    ddim_sample(unet, ae, sch, nz[:B]); torch.cuda.synchronize(); dt = time.time() - t0
    print(f"DDIM-50 + decode B={B}: {dt*1e3:.1f} ms  {B/dt:.1f} windows/s")
udm = mk(4 * L); sdm = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
opt = Adam(udm, lr=1e-4); x = torch.from_numpy(eeg_windows(64, seed=77)).to(dev); loss = torch.zeros(1, device=dev)
def step(i):
    t = randint(ctx, 64, 1000, seed=31, offset=i * 64); n = randn(ctx, (64, 1, 4 * L), seed=32, offset=i * 64 * 4 * L)
    opt.zero_grad(); dm_train_step(udm, sdm, x, n, t, spectral_weight=1e-6, spectral_loss=True, loss_out=loss); opt.step()
for i in range(2): step(i)
torch.cuda.synchronize(); t0 = time.time()
for i in range(4): step(2 + i)
torch.cuda.synchronize(); dt = (time.time() - t0) / 4
print(f"pixel DM step B=64 L=3072: {dt*1e3:.2f} ms  {64/dt:.0f} windows/s")
