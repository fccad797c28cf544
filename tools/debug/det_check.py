"""Developer tool: is the full-size bf16 backward deterministic, and linear in dy?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm.models import UNetModel
B, L = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 768
net = UNetModel(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True, dtype=sys.argv[2] if len(sys.argv) > 2 else "bfloat16")
g = torch.Generator().manual_seed(0); sd = net.state_dict()
net.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu()) for k, v in sd.items()})
x = torch.randn(B, 1, L, generator=g); t = torch.randint(0, 1000, (B,), generator=g); dy = torch.randn(B, 1, L, generator=g)
def rel(a, b): a = a.double().cpu().reshape(-1); b = b.double().cpu().reshape(-1); return float((a - b).norm() / (b.norm() + 1e-30))
net.train()
def run(s):
    y = net(x, timesteps=t).float().cpu(); net.zero_grad(); dx = net.backward(s * dy, need_dx=True).float().cpu(); return y, dx, net.flat_grad.clone()
y1, dx1, g1 = run(1.0); y2, dx2, g2 = run(1.0); y3, dx3, g3 = run(2.0)
print("fwd repeat", rel(y2, y1), " dx repeat", rel(dx2, dx1), " grads repeat", rel(g2, g1))
print("dx(2dy) vs 2dx", rel(dx3, 2 * dx1), " grads", rel(g3, 2 * g1))
ks = list(net.entries.items())
worst = sorted(((rel(g2[o:o+n], g1[o:o+n]), k) for k, (o, n, _s) in ks), reverse=True)[:6]
print("worst repeat params", worst)
import collections
cls = collections.Counter(); tot = collections.Counter()
for k, (o, n, _s) in ks:
    kind = ("norm" if (".in_layers.0." in k or ".out_layers.0." in k or k.startswith("out.0") or ".norm." in k) else "emb" if ("emb" in k) else "conv/linear") + (".bias" if k.endswith("bias") else ".weight")
    tot[kind] += 1
    if not torch.equal(g1[o:o+n], g2[o:o+n]): cls[kind] += 1
print("parameters whose gradient differs between two identical runs, by kind:", {k: f"{cls[k]}/{tot[k]}" for k in tot})
print("names:", [k for k, (o, n, _s) in ks if not torch.equal(g1[o:o+n], g2[o:o+n])][:60])
