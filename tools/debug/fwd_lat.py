"""Developer tool: eval-mode UNet forward latency at small batch (latent model L = 768 and pixel-space model L = 3072)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import eegldm
from eegldm.models import UNetModel
for L in (768, 3072):
    u = UNetModel(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4],
                  resblock_updown=True, dtype="bfloat16")
    g = torch.Generator().manual_seed(42); sd = u.state_dict()
    u.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in sd.items()})
    u.eval()
    for B in (1, 2, 4, 8, 16):
        x = torch.randn(B, 1, L, generator=g).cuda(); t = torch.randint(0, 1000, (B,), generator=g).cuda()
        for _ in range(3): u(x, timesteps=t)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): u(x, timesteps=t)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 20
        print(f"L={L} B={B}: forward {dt*1e3:.3f} ms  ({B/dt:.0f} windows/s)", flush=True)
    del u
