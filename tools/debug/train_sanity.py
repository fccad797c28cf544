"""Developer check: the bf16 engine actually optimises -- 60 Adam steps of the config_ldm UNet on a fixed synthetic batch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm.models import UNetModel
from eegldm.schedulers import DDPMScheduler
from eegldm.training import Adam, ldm_train_step
B, L = 64, 768
net = UNetModel(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
                channel_mult=[1, 2, 4], resblock_updown=True, dtype=sys.argv[1] if len(sys.argv) > 1 else "bfloat16")
sched = DDPMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)
opt = Adam(net, lr=1e-4)
g = torch.Generator().manual_seed(0)
lat = torch.randn(B, 1, L, generator=g).cuda(); loss = torch.zeros(1, device="cuda")
hist = []
for i in range(60):
    noise = torch.randn(B, 1, L, generator=g).cuda(); t = torch.randint(0, 1000, (B,), generator=g).cuda()
    net.zero_grad(); ldm_train_step(net, sched, lat, noise, t, loss_out=loss); opt.step()
    hist.append(float(loss))
print("loss every 10 steps:", [round(v, 4) for v in hist[::10]], "last", round(hist[-1], 4))
assert all(v == v and v < 10 for v in hist), "NaN / divergence"
assert sum(hist[-10:]) / 10 < 0.6 * sum(hist[:5]) / 5, "loss did not go down"
print("train sanity ok")
