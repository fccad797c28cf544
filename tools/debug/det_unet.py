import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eegldm.models import UNetModel
CFG = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True)
B, L = 256, 768
nb = UNetModel(**CFG, dtype="bfloat16")
g = torch.Generator().manual_seed(0); sd = nb.state_dict()
nb.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu()) for k, v in sd.items()})
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 1, L, generator=g); t = torch.randint(0, 1000, (B,), generator=g)
dy = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(5))
nb.train()
outs = []
for _ in range(4):
    y = nb(x, timesteps=t).clone(); nb.zero_grad(); dx = nb.backward(dy, need_dx=True).clone(); outs.append((y, dx, nb.flat_grad.clone()))
for i in range(1, 4):
    d = (outs[0][1].float() - outs[i][1].float())
    nd = int((d != 0).sum()); bad_samples = torch.unique((d != 0).nonzero()[:, 0]).tolist()[:10] if nd else []
    print(f"run {i}: y equal {torch.equal(outs[0][0], outs[i][0])}; dx differing {nd} of {d.numel()}, max abs {float(d.abs().max()):.3e} (dx max {float(outs[0][1].abs().max()):.3e}), samples {bad_samples}; grad rel {float((outs[0][2]-outs[i][2]).norm()/outs[0][2].norm()):.2e}")
