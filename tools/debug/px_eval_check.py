import sys, torch
sys.path.insert(0, '/root/repo')
from eegldm.models import UNetModel
torch.manual_seed(0)
cfg = dict(image_size=3072, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True)
net = UNetModel(**cfg, dtype="bfloat16")
g = torch.Generator().manual_seed(0); sd = net.state_dict()
net.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu()) for k, v in sd.items()})
nf = UNetModel(**cfg, dtype="float32"); nf.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()}); nf.eval()
for B in (1, 2):
    x = torch.randn(B, 1, 3072, generator=g); t = torch.randint(0, 1000, (B,), generator=g)
    yf = nf(x, timesteps=t).float().cpu()
    net.train(); yt = net(x, timesteps=t).float().cpu().clone()
    net.eval(); ye = net(x, timesteps=t).float().cpu().clone()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print("L=3072 B=%d eval-vs-train %.3e  eval-vs-fp32 %.3e  train-vs-fp32 %.3e finite %s" % (B, rel(ye, yt), rel(ye, yf), rel(yt, yf), bool(torch.isfinite(ye).all())))
