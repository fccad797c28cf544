"""Developer tool: the pixel-space DM train step alone (B = 64, L = 3072, bf16) -- A/B under environment switches."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import eegldm
from eegldm.models import UNetModel
from eegldm.schedulers import DDPMScheduler
from eegldm.training import Adam, dm_train_step, randint, randn
from param_gen import eeg_windows
ctx = eegldm.default_context(0); dev = torch.device("cuda", 0); L = 3072
torch.manual_seed(0)
udm = UNetModel(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4],
                resblock_updown=True, dtype="bfloat16")
g = torch.Generator().manual_seed(42); sd = udm.state_dict()
udm.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in sd.items()})
sdm = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
opt = Adam(udm, lr=1e-4); x = torch.from_numpy(eeg_windows(64, seed=77)).to(dev); loss = torch.zeros(1, device=dev)
def step(i):
    t = randint(ctx, 64, 1000, seed=31, offset=i * 64); n = randn(ctx, (64, 1, L), seed=32, offset=i * 64 * L)
    opt.zero_grad(); dm_train_step(udm, sdm, x, n, t, spectral_weight=1e-6, spectral_loss=True, loss_out=loss); opt.step()
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.time()
n = 8
for i in range(n): step(3 + i)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f"pixel DM step B=64 L=3072 [{os.environ.get('EEGLDM_ATTN_NO_LONG', 'long-attention kernel')}]: {dt*1e3:.2f} ms  {64/dt:.0f} windows/s  loss {float(loss):.5f}")
