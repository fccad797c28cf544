"""Developer tool: time the AutoencoderKL [2,2,4] + PatchDiscriminator GAN step (config_aekl_eeg_2_2_4_spec.yaml) alone."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm.models import AutoencoderKL, PatchDiscriminator
from eegldm.training import Adam, aekl_train_step, randn
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = 3072
dtype = sys.argv[2] if len(sys.argv) > 2 else "bfloat16"
ctx = eegldm.default_context(0)
ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[2, 2, 4], latent_channels=1, num_res_blocks=2,
                   norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=0)
disc = PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3,
                          norm="BATCH", bias=False, padding=1, dtype=dtype, device=0)
og, od = Adam(ae, lr=5e-3), Adam(disc, lr=5e-4)
x = torch.randn(B, 1, L, device="cuda")
lo = torch.zeros(6, device="cuda")
def step(i):
    eps = randn(ctx, (B, 1, L // 4), seed=5, offset=i * B * L)
    ae.zero_grad(); disc.zero_grad()
    aekl_train_step(ae, disc, x, eps, 0.01, 1e-9, 1e4, True, losses_out=lo)
    og.step(); od.step()
for i in range(3): step(i)
torch.cuda.synchronize(); t0 = time.time()
n = 5
for i in range(n): step(3 + i)
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f"AEKL GAN step B={B} {dtype}: {dt*1e3:.2f} ms  {B/dt:.0f} windows/s  losses {[round(float(v), 4) for v in lo.cpu()]}")
