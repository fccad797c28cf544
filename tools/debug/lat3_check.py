import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm.models import UNetModel
from eegldm.schedulers import DDPMScheduler
from eegldm.training import Adam, ldm_train_step
for dt in ["float32", "bfloat16"]:
    net = UNetModel(image_size=768, in_channels=3, out_channels=3, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
                    channel_mult=[1, 2, 4], resblock_updown=True, dtype=dt)
    sched = DDPMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)
    opt = Adam(net, lr=1e-4); g = torch.Generator().manual_seed(0)
    lat = torch.randn(16, 3, 768, generator=g).cuda(); loss = torch.zeros(1, device="cuda"); h = []
    for i in range(12):
        noise = torch.randn(16, 3, 768, generator=g).cuda(); t = torch.randint(0, 1000, (16,), generator=g).cuda()
        net.zero_grad(); ldm_train_step(net, sched, lat, noise, t, loss_out=loss); opt.step(); h.append(round(float(loss), 4))
    print(dt, "latent_channels=3 params", net.n_flat, "loss", h[0], "->", h[-1])
