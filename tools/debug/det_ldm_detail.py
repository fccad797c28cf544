"""Developer tool: ONE LDM step (forward + backward, no optimiser) twice from identical state: which outputs are bit-reproducible?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eegldm
from eegldm.models import UNetModel
from eegldm.schedulers import DDPMScheduler
from eegldm.training import ldm_train_step
dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
L = 768
net = UNetModel(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True, dtype=dtype)
g = torch.Generator().manual_seed(0); sd = net.state_dict()
net.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu().clone()) for k, v in sd.items()})
sched = DDPMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)
lat = torch.randn(B, 1, L, generator=g).cuda(); nz = torch.randn(B, 1, L, generator=g).cuda(); t = torch.randint(0, 1000, (B,), generator=g).cuda()
runs = []
for r in range(3):
    net.zero_grad(); loss = float(ldm_train_step(net, sched, lat, nz, t)); torch.cuda.synchronize()
    runs.append((loss, net.flat_grad.clone()))
print(f"{dtype} B={B}: losses {[x[0] for x in runs]}")
for a, b in ((0, 1), (1, 2)):
    diff = []; same = []
    for k, (o, n, _s) in net.entries.items():
        ga, gb = runs[a][1][o:o + n], runs[b][1][o:o + n]
        if torch.equal(ga, gb): same.append(k)
        else: diff.append((k, float((ga - gb).abs().max() / ga.abs().max().clamp_min(1e-30))))
    print(f"runs {a} vs {b}: {len(diff)} of {len(net.entries)} gradients differ; largest relative differences: {sorted(diff, key=lambda kv: -kv[1])[:6]}")
    kinds = {}
    for k, _ in diff:
        kind = k.split(".")[-2] + "." + k.split(".")[-1] if k.count(".") >= 2 else k
        kinds[kind] = kinds.get(kind, 0) + 1
    print("   differing by kind:", dict(sorted(kinds.items(), key=lambda kv: -kv[1])))
    print("   reproducible:", same[:40])
