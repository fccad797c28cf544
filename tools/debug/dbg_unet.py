import sys, os, torch, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
from make_golden_cases import UNET_CASES
from param_gen import gen_param, normal, timesteps
from oracle import unet as U
from eegldm.models import UNetModel
name = sys.argv[1] if len(sys.argv) > 1 else "tiny_l64"
dtype = sys.argv[2] if len(sys.argv) > 2 else "float32"
cfg, B, L = UNET_CASES[name]
shapes = U.unet_param_shapes(cfg)
sd = {k: torch.from_numpy(gen_param(42, k, s)).requires_grad_(True) for k, s in shapes.items()}
x = torch.from_numpy(normal((B, cfg["in_channels"], L), seed=600)).requires_grad_(True)
t = torch.from_numpy(timesteps(B, seed=700))
y = U.unet_forward(sd, cfg, x, t)
dy = torch.from_numpy(normal(tuple(y.shape), seed=800))
y.backward(dy)
net = UNetModel(**cfg, dtype=dtype)
net.load_state_dict({k: v.detach() for k, v in sd.items()})
yd = net(x.detach(), timesteps=t)
print("y maxerr", float((yd.cpu() - y.detach()).abs().max()), "scale", float(y.abs().max()))
net.zero_grad()
dx = net.backward(dy, need_dx=True)
print("dx maxerr", float((dx.cpu() - x.grad).abs().max()), "scale", float(x.grad.abs().max()))
g = net.grad_dict()
rows = []
for k in shapes:
    a, b = g[k].cpu(), sd[k].grad
    err = float((a - b).abs().max()); sc = float(b.abs().max())
    rows.append((err / (sc + 1e-9), k, err, sc, float(a.norm()), float(b.norm())))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.3e %-50s err %.3e scale %.3e |got| %.4e |want| %.4e" % r)
