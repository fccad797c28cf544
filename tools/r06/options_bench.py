"""Round-6 developer tool: what the optional UNet constructor branches cost at the production size -- LDM train step (config_ldm.yaml UNet,
B = 256, L = 768, bf16, frozen-encoder leg left out) with one option changed at a time.  The options run at layer granularity (none of the
fusions of the default configuration is specialised for them), so this is the price list, not a tuning target.

    python tools/r06/options_bench.py
"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm.models import UNetModel
from eegldm.schedulers import DDPMScheduler
from eegldm.training import Adam, ldm_train_step

BASE = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4],
            resblock_updown=True)
CASES = {
    "default (config_ldm.yaml)": {},
    "num_heads=2 (head width 256: fused attention on column views)": dict(num_heads=2),
    "num_head_channels=64 (8 heads of 64: GEMM + softmax composition per head)": dict(num_head_channels=64),
    "use_scale_shift_norm=True": dict(use_scale_shift_norm=True),
    "resblock_updown=False, conv_resample=True": dict(resblock_updown=False, conv_resample=True),
    "dropout=0.1": dict(dropout=0.1),
}
B, L = 256, 768
dev = torch.device("cuda", 0)
sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
lat = torch.randn(B, 1, L, device=dev); nz = torch.randn(B, 1, L, device=dev); t = torch.randint(0, 1000, (B,), device=dev)
for name, kw in CASES.items():
    net = UNetModel(**dict(BASE, **kw), dtype="bfloat16")
    g = torch.Generator().manual_seed(1)
    net.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in net.state_dict().items()})
    opt = Adam(net, lr=1e-4)
    def step():
        net.zero_grad(); ldm_train_step(net, sched, lat, nz, t); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): step()
    torch.cuda.synchronize()
    print(f"{name:80s} {(time.time() - t0) / 10 * 1e3:7.2f} ms/step   {net.n_flat / 1e6:6.2f} M parameters", flush=True)
    del net, opt
