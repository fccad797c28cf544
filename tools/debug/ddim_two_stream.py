"""Experiment: DDIM-50 over B=256 as one batch vs two half-batches on two HIP streams (two UNet executors sharing weights)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import Context
from eegldm.models import UNetModel
from eegldm.sampling import make_sampling_scheduler
B, L = 256, 768
kw = dict(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
          channel_mult=[1, 2, 4], resblock_updown=True, dtype="bfloat16")
netA = UNetModel(**kw)
g = torch.Generator().manual_seed(0)
sd = {k: torch.randn(v.shape, generator=g) * 0.02 for k, v in netA.state_dict().items()}
netA.load_state_dict(sd)
sB = torch.cuda.Stream()
with torch.cuda.stream(sB):
    ctxB = Context(0)
    netB = UNetModel(**kw, ctx=ctxB)
    netB.load_state_dict(sd)
torch.cuda.synchronize()
netA.eval(); netB.eval()
def run_single(n):
    sch = make_sampling_scheduler(50)
    x = torch.randn(n, 1, L, device="cuda"); tt = torch.empty(n, device="cuda", dtype=torch.int64)
    torch.cuda.synchronize(); t0 = time.time()
    with torch.no_grad():
        for t in sch.timesteps:
            tt.fill_(int(t)); out = netA(x, timesteps=tt); x, _ = sch.step(out, int(t), x)
    torch.cuda.synchronize(); return time.time() - t0
def run_two(n):
    h = n // 2
    schA, schB = make_sampling_scheduler(50), make_sampling_scheduler(50)
    xA = torch.randn(h, 1, L, device="cuda"); ttA = torch.empty(h, device="cuda", dtype=torch.int64)
    with torch.cuda.stream(sB):
        xB = torch.randn(h, 1, L, device="cuda"); ttB = torch.empty(h, device="cuda", dtype=torch.int64)
    torch.cuda.synchronize(); t0 = time.time()
    with torch.no_grad():
        for t in schA.timesteps:
            ttA.fill_(int(t)); outA = netA(xA, timesteps=ttA); xA, _ = schA.step(outA, int(t), xA)
            with torch.cuda.stream(sB):
                ttB.fill_(int(t)); outB = netB(xB, timesteps=ttB); xB, _ = schB.step(outB, int(t), xB)
    torch.cuda.synchronize(); return time.time() - t0
run_single(8); run_two(16)
a = min(run_single(B) for _ in range(2)); b = min(run_two(B) for _ in range(2)); c = min(run_single(B // 2) for _ in range(2))
print(f"DDIM-50 B={B}: one stream {a*1e3:.1f} ms ({B/a:.0f} win/s) | two streams x {B//2}: {b*1e3:.1f} ms ({B/b:.0f} win/s) | single B={B//2}: {c*1e3:.1f} ms")
