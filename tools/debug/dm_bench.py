"""Developer tool: pixel-space DM train step alone (config_dm.yaml UNet on (64,1,3072) windows) for rocprofv3 traces."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import eegldm
from eegldm.models import UNetModel
from eegldm.schedulers import DDPMScheduler
from eegldm.training import Adam, dm_train_step, randint, randn
from param_gen import eeg_windows
ctx = eegldm.default_context(0); dev = torch.device("cuda", 0); L = 3072; B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
u = UNetModel(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4],
              resblock_updown=True, dtype="bfloat16")
g = torch.Generator().manual_seed(42); sd = u.state_dict()
u.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in sd.items()})
sdm = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
opt = Adam(u, lr=1e-4); x = torch.from_numpy(eeg_windows(B, seed=77)).to(dev); loss = torch.zeros(1, device=dev)
def step(i):
    t = randint(ctx, B, 1000, seed=31, offset=i * B); n = randn(ctx, (B, 1, L), seed=32, offset=i * B * L)
    opt.zero_grad(); dm_train_step(u, sdm, x, n, t, spectral_weight=1e-6, spectral_loss=True, loss_out=loss); opt.step()
for i in range(2): step(i)
torch.cuda.synchronize(); t0 = time.time()
for i in range(4): step(2 + i)
torch.cuda.synchronize(); dt = (time.time() - t0) / 4
print(f"pixel DM step B={B} L={L}: {dt*1e3:.2f} ms  {B/dt:.0f} windows/s")
