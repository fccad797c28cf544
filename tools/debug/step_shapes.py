import os, sys, torch, collections, csv
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
from eegldm.models import UNetModel
from eegldm.schedulers import DDPMScheduler
from eegldm.training import ldm_train_step
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 768
net = UNetModel(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True, dtype="bfloat16")
sd = net.state_dict(); g = torch.Generator().manual_seed(0)
net.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if v.abs().sum() == 0 else v) for k, v in sd.items()})
sched = DDPMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)
lat = torch.randn(B, 1, L, device="cuda"); nz = torch.randn(B, 1, L, device="cuda"); t = torch.randint(0, 1000, (B,), device="cuda")
for _ in range(2): net.zero_grad(); ldm_train_step(net, sched, lat, nz, t)
net.ctx.prof_enable(True)
net.zero_grad(); ldm_train_step(net, sched, lat, nz, t)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "shapes.csv")
check(lib.eegldm_prof_dump(net.ctx.h, out.encode()))
net.ctx.prof_enable(False)
rows = list(csv.DictReader(open(out)))
agg = collections.OrderedDict()
names = ["convF", "convD", "convW", "NT", "NN", "TN"]
for r in rows:
    k = (names[int(r["class"])], int(r["M"]), int(r["N"]), int(r["K"]), int(r["taps"]), int(r["splitk"]))
    a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["gflop"])
tot = sum(v[1] for v in agg.values())
print(f"total gemm ms {tot:.2f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k[0]:6s} M={k[1]:7d} N={k[2]:5d} K={k[3]:7d} taps={k[4]} splitk={k[5]:3d}  x{v[0]:2d}  {v[1]:7.3f} ms  {v[2]/v[1]:7.1f} TF/s")
